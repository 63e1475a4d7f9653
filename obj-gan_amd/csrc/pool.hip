// Pooling layers of the frozen Inception-v3 encoder (reference image_generation/model.py:203-287 calls
// F.max_pool2d(k=3, s=2), F.avg_pool2d(k=3, s=1, p=1) and F.avg_pool2d(k=8) around the torchvision blocks).
// HBM-bound streaming kernels: one thread per output element, planes = N*C contiguous NCHW planes.
//   max  : window k x k, stride s, no padding; the arg-max (first maximum in row-major scan order, NaN
//          propagating -- torch's rule) is kept as a plane-local index for the backward pass, which is a
//          deterministic GATHER over the <= ceil(k/s)^2 windows that contain an input pixel (no atomics).
//   avg  : window k x k, stride s, zero padding p, divisor k*k (count_include_pad=True, torch's default);
//          its backward is the transposed box filter.
#include "common.h"

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int* __restrict__ idx, long total, int H, int W,
                                                          int OH, int OW, int k, int s) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % OW);
        const long r = e / OW;
        const int oh = (int)(r % OH);
        const long plane = r / OH;
        const float* xp = x + plane * (long)H * W;
        const int h0 = oh * s, w0 = ow * s;
        float best = xp[h0 * W + w0];
        int bi = h0 * W + w0;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                const int h = h0 + i, w = w0 + j;
                if (h < H && w < W) {
                    const float v = xp[h * W + w];
                    if (v > best || v != v) { best = v; bi = h * W + w; }
                }
            }
        y[e] = best;
        if (idx) idx[e] = bi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                          float* __restrict__ dx, long total, int H, int W,
                                                          int OH, int OW, int k, int s) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W);
        const long r = e / W;
        const int h = (int)(r % H);
        const long plane = r / H;
        const int me = h * W + w;
        // windows containing (h, w): oh*s <= h <= oh*s + k - 1
        int oh_lo = (h - k + s) / s; if (h - k + 1 <= 0) oh_lo = 0;
        int ow_lo = (w - k + s) / s; if (w - k + 1 <= 0) ow_lo = 0;
        const int oh_hi = min(h / s, OH - 1), ow_hi = min(w / s, OW - 1);
        const long ob = plane * (long)OH * OW;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow)
                if (idx[ob + oh * OW + ow] == me) acc += dy[ob + oh * OW + ow];
        dx[e] = acc;
    }
}

// y[oh, ow] = (1 / k^2) * sum_{i,j} x[oh*s - p + i, ow*s - p + j]   (zeros outside)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long total, int H, int W, int OH, int OW, int k,
                                                          int s, int p, float inv) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % OW);
        const long r = e / OW;
        const int oh = (int)(r % OH);
        const long plane = r / OH;
        const float* xp = x + plane * (long)H * W;
        float acc = 0.f;
        for (int i = 0; i < k; ++i) {
            const int h = oh * s - p + i;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int j = 0; j < k; ++j) {
                const int w = ow * s - p + j;
                if ((unsigned)w < (unsigned)W) acc += xp[h * W + w];
            }
        }
        y[e] = acc * inv;
    }
}

// dx[h, w] = (1 / k^2) * sum over windows (oh, ow) containing (h, w) of dy[oh, ow]
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                          long total, int H, int W, int OH, int OW, int k,
                                                          int s, int p, float inv) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W);
        const long r = e / W;
        const int h = (int)(r % H);
        const long plane = r / H;
        const float* gp = dy + plane * (long)OH * OW;
        const int hp = h + p, wp = w + p;            // position in the padded frame
        int oh_lo = hp - k + 1 <= 0 ? 0 : (hp - k + s) / s;
        int ow_lo = wp - k + 1 <= 0 ? 0 : (wp - k + s) / s;
        const int oh_hi = min(hp / s, OH - 1), ow_hi = min(wp / s, OW - 1);
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) acc += gp[oh * OW + ow];
        dx[e] = acc * inv;
    }
}

extern "C" {

// mode 0: max (idx: plane-local arg-max per output, may be null when no backward follows), 1: average.
int objgan_pool2d_forward(const float* x, float* y, int* idx, long planes, int H, int W, int OH, int OW,
                          int k, int s, int p, int mode, void* stream) {
    OG_ENTRY();
    if (k < 1 || s < 1 || p < 0 || (mode != 0 && mode != 1) || (mode == 0 && p != 0)) return OG_BAD_ARGS;
    if (OH != (H + 2 * p - k) / s + 1 || OW != (W + 2 * p - k) / s + 1) return OG_BAD_ARGS;
    const long total = planes * OH * OW;
    if (total <= 0) return OG_OK;
    if (mode == 0)
        hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           x, y, idx, total, H, W, OH, OW, k, s);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           x, y, total, H, W, OH, OW, k, s, p, 1.0f / (float)(k * k));
    return og_launch_status();
}

int objgan_pool2d_backward(const float* dy, const int* idx, float* dx, long planes, int H, int W, int OH, int OW,
                           int k, int s, int p, int mode, void* stream) {
    OG_ENTRY();
    if (k < 1 || s < 1 || p < 0 || (mode != 0 && mode != 1) || (mode == 0 && (p != 0 || !idx))) return OG_BAD_ARGS;
    const long total = planes * H * W;
    if (total <= 0) return OG_OK;
    if (mode == 0)
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           dy, idx, dx, total, H, W, OH, OW, k, s);
    else
        hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           dy, dx, total, H, W, OH, OW, k, s, p, 1.0f / (float)(k * k));
    return og_launch_status();
}

}  // extern "C"
