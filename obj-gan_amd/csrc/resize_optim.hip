// Bilinear resize (align_corners=True) forward/backward, fused flat-arena Adam, EMA.
//
//  * resize: F.interpolate(x, size, mode='bilinear', align_corners=True) as used by the
//    object discriminators to lift the 256x256 image and the 80-channel layout map to
//    512x512 (reference image_generation/model.py:1217-1218, 1283-1284).  The backward is a
//    deterministic gather (every input pixel collects from the <= 4x4 output pixels whose
//    taps touch it), not an atomic scatter.
//  * Adam / EMA: the reference runs 9 torch.optim.Adam instances tensor by tensor and a python
//    loop for the generator EMA (reference image_generation/trainer.py:197-224, 461-462).
//    Here every network keeps its parameters, gradients and both moments in four flat fp32
//    arenas, so one streaming launch updates a whole network (28 B/parameter of HBM traffic,
//    the algorithmic minimum) and the EMA is one more streaming launch.
#include "common.h"

__device__ __forceinline__ void bilin_src(int o, float scale, int in_size, int& i0, int& i1,
                                          float& l0, float& l1) {
    // PyTorch area_pixel_compute_source_index(align_corners=True): src = scale * dst
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, long planes,
                                                           int IH, int IW, int OH, int OW,
                                                           float sh, float sw) {
    const long total = planes * OH * OW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % OW);
        const int oh = (int)((e / OW) % OH);
        const long p = e / ((long)OW * OH);
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bilin_src(oh, sh, IH, h0, h1, lh0, lh1);
        bilin_src(ow, sw, IW, w0, w1, lw0, lw1);
        const float* xp = x + p * IH * IW;
        y[e] = lh0 * (lw0 * xp[h0 * IW + w0] + lw1 * xp[h0 * IW + w1]) +
               lh1 * (lw0 * xp[h1 * IW + w0] + lw1 * xp[h1 * IW + w1]);
    }
}

// weight with which output index o reads input index i (0 if it does not)
__device__ __forceinline__ float bilin_weight(int o, int i, float scale, int in_size) {
    int i0, i1;
    float l0, l1;
    bilin_src(o, scale, in_size, i0, i1, l0, l1);
    float w = 0.f;
    if (i == i0) w += l0;
    if (i == i1) w += l1;
    return w;
}

__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy,
                                                           float* __restrict__ dx, long planes,
                                                           int IH, int IW, int OH, int OW,
                                                           float sh, float sw) {
    const long total = planes * IH * IW;
    const float inv_sh = sh > 0.f ? 1.0f / sh : 0.f;
    const float inv_sw = sw > 0.f ? 1.0f / sw : 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int iw = (int)(e % IW);
        const int ih = (int)((e / IW) % IH);
        const long p = e / ((long)IW * IH);
        // candidate outputs: scale*o in (i-1, i+1), widened by one on each side for rounding
        int oh_lo = sh > 0.f ? (int)floorf(((float)ih - 1.f) * inv_sh) - 1 : 0;
        int oh_hi = sh > 0.f ? (int)ceilf(((float)ih + 1.f) * inv_sh) + 1 : OH - 1;
        int ow_lo = sw > 0.f ? (int)floorf(((float)iw - 1.f) * inv_sw) - 1 : 0;
        int ow_hi = sw > 0.f ? (int)ceilf(((float)iw + 1.f) * inv_sw) + 1 : OW - 1;
        oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
        oh_hi = min(oh_hi, OH - 1); ow_hi = min(ow_hi, OW - 1);
        const float* gp = dy + p * OH * OW;
        float acc = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            const float wh = bilin_weight(oh, ih, sh, IH);
            if (wh == 0.f) continue;
            float row = 0.f;
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const float ww = bilin_weight(ow, iw, sw, IW);
                if (ww != 0.f) row = fmaf(ww, gp[oh * OW + ow], row);
            }
            acc = fmaf(wh, row, acc);
        }
        dx[e] = acc;
    }
}

// ---- nearest x2 upsample backward helper: dx[h, w] = sum of the 2x2 block of dy ------------
__global__ __launch_bounds__(256) void sum2x2_kernel(const float* __restrict__ dy,
                                                     float* __restrict__ dx, long planes, int H,
                                                     int W) {
    const long total = planes * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W);
        const int h = (int)((e / W) % H);
        const long p = e / ((long)W * H);
        const float* g = dy + (p * 2 * H + 2 * h) * 2 * W + 2 * w;
        dx[e] = (g[0] + g[1]) + (g[2 * W] + g[2 * W + 1]);
    }
}

// ---- ReflectionPad2d(1) backward fold: dxp [P, H+2, W+2] -> dx [P, H, W] -------------------
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dxp,
                                                           float* __restrict__ dx, long planes,
                                                           int H, int W) {
    const long total = planes * H * W;
    const int PW = W + 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W);
        const int h = (int)((e / W) % H);
        const long p = e / ((long)W * H);
        const float* g = dxp + p * (H + 2) * PW;
        // padded row r maps to source row |r-1| (top) / 2(H-1)-(r-1) (bottom); source row h
        // therefore receives padded rows h+1, plus 0 if h == 1, plus H+1 if h == H-2.
        int rows[3], cols[3];
        int nr = 0, nc = 0;
        rows[nr++] = h + 1;
        if (h == 1) rows[nr++] = 0;
        if (h == H - 2) rows[nr++] = H + 1;
        cols[nc++] = w + 1;
        if (w == 1) cols[nc++] = 0;
        if (w == W - 2) cols[nc++] = W + 1;
        float acc = 0.f;
        for (int a = 0; a < nr; ++a)
            for (int b = 0; b < nc; ++b) acc += g[rows[a] * PW + cols[b]];
        dx[e] = acc;
    }
}

// ---- Adam (torch.optim.Adam semantics, no weight decay, no amsgrad) --------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   long n, float lr, float beta1, float beta2,
                                                   float omb1, float omb2,
                                                   float eps, float step_size, float bc2_sqrt,
                                                   float grad_scale) {
    // omb = (float)(1.0 - beta) and step_size = (float)(lr / (1 - beta1^t)) are evaluated in double by the
    // host, as torch does with its python-float hyper-parameters, and rounded once
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;   // 1/world_size folds the DDP average in here
        const float mi = m[i] + (gi - m[i]) * omb1;                 // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * beta2 + omb2 * gi * gi;             // mul_(beta2).addcmul_
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// Gated Adam: the update is taken only when a DEVICE flag is positive, and the step counter lives on the
// device as well.  This is what keeps the conditional object-discriminator update (reference
// trainer.py:429,440: `if float(err) > 0`) consistent across data-parallel ranks without a host sync:
// every rank adds 1 to a flag slot behind its gradient arena when its batch produced a loss, the slot
// rides along in the gradient all-reduce, and all ranks then take -- or skip -- the same update.
//   state[0] = steps taken, state[1] = beta1^steps, state[2] = beta2^steps  (doubles, device)
//   coef[0] = take the update (0/1), coef[1] = lr / (1 - beta1^t), coef[2] = sqrt(1 - beta2^t)
__global__ void adam_gate_kernel(double* __restrict__ state, const float* __restrict__ flag,
                                 float* __restrict__ coef, double lr, double beta1, double beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool on = flag[0] > 0.f;
    if (on) {
        state[0] += 1.0;
        state[1] *= beta1;
        state[2] *= beta2;
    }
    coef[0] = on ? 1.f : 0.f;
    coef[1] = (float)(lr / (1.0 - state[1]));
    coef[2] = (float)sqrt(1.0 - state[2]);
}

__global__ __launch_bounds__(256) void adam_gated_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         long n, float lr, float beta1, float beta2,
                                                         float omb1, float omb2,
                                                         float eps, const float* __restrict__ coef,
                                                         const float* __restrict__ flag, float grad_scale) {
    if (coef[0] <= 0.f) return;
    const float step_size = coef[1];
    const float bc2_sqrt = coef[2];
    // grad_scale < 0: divide by the flag itself -- under data parallelism the all-reduced flag counts the ranks that
    // contributed a gradient, and the mean is taken over THOSE (a rank without boxes of the wanted scale adds zeros)
    const float gs = grad_scale < 0.f ? 1.0f / flag[0] : grad_scale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (gi - m[i]) * omb1;
        const float vi = v[i] * beta2 + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// avg = avg * decay + (1 - decay) * p     (avg_p.mul_(0.999).add_(0.001, p.data))
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ avg, const float* __restrict__ p,
                                                  long n, float decay, float one_minus) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x)
        avg[i] = avg[i] * decay + one_minus * p[i];
}

extern "C" {

int objgan_bilinear_forward(const float* x, float* y, long planes, int ih, int iw, int oh, int ow,
                            void* stream) {
    OG_ENTRY();
    if (planes <= 0 || oh <= 0 || ow <= 0) return OG_OK;
    const float sh = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f;
    const float sw = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    const long total = planes * oh * ow;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, planes, ih, iw, oh, ow, sh, sw);
    return og_launch_status();
}

int objgan_bilinear_backward(const float* dy, float* dx, long planes, int ih, int iw, int oh,
                             int ow, void* stream) {
    OG_ENTRY();
    if (planes <= 0 || ih <= 0 || iw <= 0) return OG_OK;
    const float sh = oh > 1 ? (float)(ih - 1) / (float)(oh - 1) : 0.f;
    const float sw = ow > 1 ? (float)(iw - 1) / (float)(ow - 1) : 0.f;
    const long total = planes * ih * iw;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, planes, ih, iw, oh, ow, sh, sw);
    return og_launch_status();
}

// dy [planes, 2H, 2W] -> dx [planes, H, W]
int objgan_sum2x2(const float* dy, float* dx, long planes, int h, int w, void* stream) {
    OG_ENTRY();
    const long total = planes * h * w;
    if (total <= 0) return OG_OK;
    hipLaunchKernelGGL(sum2x2_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, planes, h, w);
    return og_launch_status();
}

// dxp [planes, H+2, W+2] -> dx [planes, H, W]   (requires H, W >= 3)
int objgan_reflect_fold(const float* dxp, float* dx, long planes, int h, int w, void* stream) {
    OG_ENTRY();
    if (h < 3 || w < 3) return OG_BAD_ARGS;
    const long total = planes * h * w;
    if (total <= 0) return OG_OK;
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dxp, dx, planes, h, w);
    return og_launch_status();
}

int objgan_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                     double beta2, double eps, int step, float grad_scale, void* stream) {
    OG_ENTRY();
    if (n <= 0) return OG_OK;
    if (step < 1) return OG_BAD_ARGS;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(og_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       p, g, m, v, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2), grad_scale);
    return og_launch_status();
}

// Adam step gated by the device flag `flag[0] > 0`; `state` = 3 doubles on the device initialised to
// {0, 1, 1}, `coef` = 3 floats of device scratch (see adam_gate_kernel).  Nothing is read back.
// grad_scale < 0: the gradient is divided by flag[0] (the number of ranks that contributed, after the all-reduce).
int objgan_adam_step_gated(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                           double beta2, double eps, double* state, const float* flag, float* coef,
                           float grad_scale, void* stream) {
    OG_ENTRY();
    if (n <= 0) return OG_OK;
    if (!state || !flag || !coef) return OG_BAD_ARGS;
    hipLaunchKernelGGL(adam_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, flag, coef,
                       lr, beta1, beta2);
    hipLaunchKernelGGL(adam_gated_kernel, dim3(og_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       p, g, m, v, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)eps, coef, flag, grad_scale);
    return og_launch_status();
}

int objgan_ema_update(float* avg, const float* p, long n, float decay, float one_minus_decay,
                      void* stream) {
    OG_ENTRY();
    if (n <= 0) return OG_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(og_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       avg, p, n, decay, one_minus_decay);
    return og_launch_status();
}

}  // extern "C"
