// ROIAlign forward / backward for gfx950 (MI355X), plus the 2x2/s1 average pool
// that RoIAlignAvg applies on top of it.
//
// Replaces, behind the same C-ABI shape, the reference native op
//   image_generation/models/roi_align/src/roi_align_kernel.cu:15-70   (ROIAlignForward)
//   image_generation/models/roi_align/src/roi_align_kernel.cu:94-143  (ROIAlignBackward)
//   image_generation/models/roi_align/src/roi_align_cuda.c:7-76       (THC glue, return codes)
// and follows the CPU loop image_generation/models/roi_align/src/roi_align.c:80-136
// bit-for-bit on the forward pass.
//
// Design (not a translation of the CUDA launch shape): the reference launches one thread
// per output element and every thread re-derives the ROI geometry.  Here one workgroup
// owns one ROI (x a channel slab): the AH*AW sample positions -- integer corner indices,
// validity and the four bilinear weights -- are derived ONCE per ROI by the first AH*AW
// lanes and parked in LDS; the workgroup then streams (channel, sample) pairs so that the
// stores to the (n, c, ph, pw) output are fully coalesced and the only per-element work is
// four gathers + four multiplies.
//
// Bit-exactness contract (trap 13).  This file is compiled with
// -ffp-contract=off.  The C reference mixes float variables with double literals
// (`+ 1.`, `/ (aligned_height - 1.)`, `(1. - h_ratio)`); those sub-expressions are
// evaluated in double and rounded to float on assignment.  `h = (float)ph * bin + start`
// is an un-fused float multiply followed by a float add.  All of that is reproduced below
// so that floor(h)/floor(w) -- the integer index math -- and the interpolated value are
// identical to the gcc-compiled reference loop.
#include "common.h"

struct RoiSample {
    int   off;      // (hstart * W + wstart), offset inside one channel plane
    int   valid;    // 0 -> output is 0 / no gradient
    float h_ratio, w_ratio;   // fractional offsets (float, as the reference stores them)
};

__device__ __forceinline__ void roi_geometry(const float* __restrict__ roi, float spatial_scale,
                                             int H, int W, int AH, int AW, int s,
                                             RoiSample& out) {
    const int pw = s % AW;
    const int ph = s / AW;
    const float roi_start_w = roi[1] * spatial_scale;
    const float roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w   = roi[3] * spatial_scale;
    const float roi_end_h   = roi[4] * spatial_scale;
    // fmaxf(<double expr>, 0.) : the double expression is converted to float at the call.
    const float roi_width  = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.0), 0.0f);
    const float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.0), 0.0f);
    const float bin_size_h = (float)((double)roi_height / ((double)AH - 1.0));
    const float bin_size_w = (float)((double)roi_width  / ((double)AW - 1.0));
    // un-fused float multiply + add (file is built with -ffp-contract=off)
    const float hm = (float)ph * bin_size_h;
    const float wm = (float)pw * bin_size_w;
    const float h = hm + roi_start_h;
    const float w = wm + roi_start_w;
    // int hstart = fminf(floor(h), height - 2): double floor -> float -> int
    const int hstart = (int)fminf((float)floor((double)h), (float)(H - 2));
    const int wstart = (int)fminf((float)floor((double)w), (float)(W - 2));
    const bool outside = (h < 0.0f) || (h >= (float)H) || (w < 0.0f) || (w >= (float)W);
    const float h_ratio = h - (float)hstart;
    const float w_ratio = w - (float)wstart;
    out.off = hstart * W + wstart;
    out.valid = outside ? 0 : 1;
    out.h_ratio = h_ratio;
    out.w_ratio = w_ratio;
}

#define ROI_MAX_SAMPLES 256   // AH*AW <= 256 (the hot path uses 6*6 = 36)

// grid = (num_rois, channel_slabs), block = 256
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int AH, int AW, float spatial_scale, int c_per_block) {
    __shared__ RoiSample smp[ROI_MAX_SAMPLES];
    const int n = blockIdx.x;
    const int S = AH * AW;
    const float* roi = rois + (size_t)n * 5;
    for (int s = threadIdx.x; s < S; s += blockDim.x)
        roi_geometry(roi, spatial_scale, H, W, AH, AW, s, smp[s]);
    __syncthreads();
    // int img_start = roi_batch_ind * channels * height * width  (float chain -> int)
    const float roi_batch_ind = roi[0];
    const int img_start = (int)(((roi_batch_ind * (float)C) * (float)H) * (float)W);
    const int c0 = blockIdx.y * c_per_block;
    const int c1 = min(C, c0 + c_per_block);
    const int total = (c1 - c0) * S;
    const size_t out_base = ((size_t)n * C + c0) * S;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c = c0 + i / S;
        const int s = i - (i / S) * S;
        const RoiSample g = smp[s];
        float v = 0.0f;
        if (g.valid) {
            const float* p = feat + (size_t)img_start + (size_t)c * H * W + g.off;
            const double ul = (double)p[0], ur = (double)p[1];
            // reference: d_ul*(1.-hr)*(1.-wr) + d_ur*(1.-hr)*wr + d_dl*hr*(1.-wr) + d_dr*hr*wr
            // with the double literal `1.` promoting every factor to double (roi_align.c:131-134)
            // C typing, term by term: ul*(1.-hr)*(1.-wr) and ur*(1.-hr)*wr are double products;
            // dl*hr is a FLOAT product that then meets the double (1.-wr); dr*hr*wr is all float.
            const double omh = 1.0 - (double)g.h_ratio, omw = 1.0 - (double)g.w_ratio;
            const double t1 = (ul * omh) * omw;
            const double t2 = (ur * omh) * (double)g.w_ratio;
            const float f3 = p[W] * g.h_ratio;
            const double t3 = (double)f3 * omw;
            const float f4 = (p[W + 1] * g.h_ratio) * g.w_ratio;
            const double acc = ((t1 + t2) + t3) + (double)f4;
            v = (float)acc;
        }
        out[out_base + i] = v;
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_kernel(
    const float* __restrict__ top_grad, const float* __restrict__ rois,
    float* __restrict__ bottom_grad,
    int C, int H, int W, int AH, int AW, float spatial_scale, int c_per_block) {
    __shared__ RoiSample smp[ROI_MAX_SAMPLES];
    const int n = blockIdx.x;
    const int S = AH * AW;
    const float* roi = rois + (size_t)n * 5;
    for (int s = threadIdx.x; s < S; s += blockDim.x)
        roi_geometry(roi, spatial_scale, H, W, AH, AW, s, smp[s]);
    __syncthreads();
    const float roi_batch_ind = roi[0];
    const int img_start = (int)(((roi_batch_ind * (float)C) * (float)H) * (float)W);
    const int c0 = blockIdx.y * c_per_block;
    const int c1 = min(C, c0 + c_per_block);
    const int total = (c1 - c0) * S;
    const size_t top_base = ((size_t)n * C + c0) * S;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        const int c = c0 + i / S;
        const int s = i - (i / S) * S;
        const RoiSample g = smp[s];
        if (!g.valid) continue;
        const float d = top_grad[top_base + i];
        float* p = bottom_grad + (size_t)img_start + (size_t)c * H * W + g.off;
        // reference kernel.cu:137-140: `(1. - h_ratio)` is double, `(1 - w_ratio)` is FLOAT (int
        // literal); the two h_ratio terms are all-float products.
        const double omh = 1.0 - (double)g.h_ratio;
        const float omw = 1.0f - g.w_ratio;
        const float dh_ = d * g.h_ratio;
        atomicAdd(p,         (float)(((double)d * omh) * (double)omw));
        atomicAdd(p + 1,     (float)(((double)d * omh) * (double)g.w_ratio));
        atomicAdd(p + W,     dh_ * omw);
        atomicAdd(p + W + 1, dh_ * g.w_ratio);
    }
}

// Backward, privatised and ORDERED: one workgroup owns one (image, channel) plane.  The ROIs of its image are taken in
// index order, ROI_BATCH at a time: their sample geometry and incoming gradients go to LDS, the bounding box of all
// their taps is found (integer LDS min / max), and every pixel of the box GATHERS -- in sample order -- the taps that
// land on it into the LDS copy of the plane (one writer per pixel).  Same per-term arithmetic as roi_align_bwd_kernel
// (and the reference's CUDA kernel); the reference leaves the order of its atomicAdds unspecified, this kernel fixes
// it: bit-reproducible.  r02's scatter form kept 36 of 256 threads busy (146 us per launch).
#define ROI_BATCH 8
#define ROI_LIST_MAX 256          // rois per image this kernel takes (the hot path has 10)
__global__ __launch_bounds__(256) void roi_align_bwd_plane_kernel(
    const float* __restrict__ top_grad, const float* __restrict__ rois,
    float* __restrict__ bottom_grad, int num_rois,
    int C, int H, int W, int AH, int AW, float spatial_scale) {
    extern __shared__ __attribute__((aligned(16))) float plane[];      // H*W floats, then samples, then gradients
    const int HW = H * W;
    const int S = AH * AW;
    RoiSample* smp = reinterpret_cast<RoiSample*>(plane + HW);          // [ROI_BATCH * S]
    float* sd = reinterpret_cast<float*>(smp + ROI_BATCH * S);          // [ROI_BATCH * S]
    __shared__ int s_list[ROI_LIST_MAX];                                // rois of this image, in index order
    __shared__ int s_wave[4], s_box[4];                                 // box: rmin, rmax, cmin, cmax
    const int c = blockIdx.x;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) plane[i] = 0.f;
    // ordered compaction of the rois that belong to image b (ballot + prefix count per wave, waves in order)
    int nlist = 0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int base = 0; base < num_rois; base += 256) {
        const int r = base + threadIdx.x;
        bool mine = false;
        if (r < num_rois) {
            // reference: img_start = (int)(roi_batch_ind * C * H * W) -- the plane of image roi[0]
            const int img_start = (int)(((rois[(size_t)r * 5] * (float)C) * (float)H) * (float)W);
            mine = img_start == b * C * HW;
        }
        const unsigned long long m = __ballot(mine);
        if (lane == 0) s_wave[wid] = __popcll(m);
        __syncthreads();
        int pos = nlist + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wid; ++w) pos += s_wave[w];
        if (mine && pos < ROI_LIST_MAX) s_list[pos] = r;
        nlist += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    nlist = min(nlist, ROI_LIST_MAX);
    if (nlist == 0) return;                            // uniform: nothing lands in this plane
    for (int k0 = 0; k0 < nlist; k0 += ROI_BATCH) {
        const int n = min(ROI_BATCH, nlist - k0);
        if (threadIdx.x == 0) { s_box[0] = H; s_box[1] = -1; s_box[2] = W; s_box[3] = -1; }
        __syncthreads();
        for (int i = threadIdx.x; i < n * S; i += blockDim.x) {
            const int k = i / S, s = i - k * S;
            const int r = s_list[k0 + k];
            RoiSample g;
            roi_geometry(rois + (size_t)r * 5, spatial_scale, H, W, AH, AW, s, g);
            smp[i] = g;
            sd[i] = top_grad[((size_t)r * C + c) * S + s];
            if (g.valid) {
                const int row = g.off / W, col = g.off - row * W;
                atomicMin(&s_box[0], row); atomicMax(&s_box[1], row + 1);
                atomicMin(&s_box[2], col); atomicMax(&s_box[3], col + 1);
            }
        }
        __syncthreads();
        const int r0 = s_box[0], r1 = min(s_box[1], H - 1), c0 = s_box[2], c1 = min(s_box[3], W - 1);
        const int bw = c1 - c0 + 1, bh = r1 - r0 + 1;
        if (bw > 0 && bh > 0) {
            for (int i = threadIdx.x; i < bw * bh; i += blockDim.x) {
                const int py = r0 + i / bw, px = c0 + i % bw;
                const int pix = py * W + px;
                float acc = 0.f;
                for (int j = 0; j < n * S; ++j) {
                    const RoiSample g = smp[j];
                    if (!g.valid) continue;
                    const int t = pix - g.off;          // 0, 1, W, W + 1: the four taps of the sample
                    if (t != 0 && t != 1 && t != W && t != W + 1) continue;
                    const float d = sd[j];
                    // reference kernel.cu:137-140: `(1. - h_ratio)` is double, `(1 - w_ratio)` is FLOAT (int
                    // literal); the two h_ratio terms are all-float products.
                    const double omh = 1.0 - (double)g.h_ratio;
                    const float omw = 1.0f - g.w_ratio;
                    const float dh_ = d * g.h_ratio;
                    float v;
                    if (t == 0) v = (float)(((double)d * omh) * (double)omw);
                    else if (t == 1) v = (float)(((double)d * omh) * (double)g.w_ratio);
                    else if (t == W) v = dh_ * omw;
                    else v = dh_ * g.w_ratio;
                    acc += v;
                }
                plane[pix] += acc;
            }
        }
        __syncthreads();
    }
    float* dst = bottom_grad + ((size_t)b * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float v = plane[i];
        if (v != 0.f) dst[i] += v;
    }
}

// ---- 2x2 stride-1 average pool over the last two dims (RoIAlignAvg tail) ------------
// in: [P, IH, IW] -> out: [P, IH-1, IW-1]
__global__ __launch_bounds__(256) void avgpool2s1_fwd_kernel(
    const float* __restrict__ in, float* __restrict__ out, long total, int IH, int IW) {
    const int OH = IH - 1, OW = IW - 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(i % OW);
        const int oh = (int)((i / OW) % OH);
        const long p = i / ((long)OW * OH);
        const float* q = in + (p * IH + oh) * IW + ow;
        out[i] = (q[0] + q[1] + q[IW] + q[IW + 1]) * 0.25f;
    }
}
__global__ __launch_bounds__(256) void avgpool2s1_bwd_kernel(
    const float* __restrict__ gout, float* __restrict__ gin, long total, int IH, int IW) {
    const int OH = IH - 1, OW = IW - 1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int iw = (int)(i % IW);
        const int ih = (int)((i / IW) % IH);
        const long p = i / ((long)IW * IH);
        const float* g = gout + p * OH * OW;
        float a = 0.f;
        if (ih < OH && iw < OW)   a += g[ih * OW + iw];
        if (ih < OH && iw > 0)    a += g[ih * OW + iw - 1];
        if (ih > 0 && iw < OW)    a += g[(ih - 1) * OW + iw];
        if (ih > 0 && iw > 0)     a += g[(ih - 1) * OW + iw - 1];
        gin[i] = a * 0.25f;
    }
}

static inline int roi_c_per_block(int C, int S) {
    // ~4K output elements per workgroup: 16 elements per lane, grid >> 256 CUs at the
    // hot-path sizes (160 ROIs x 384/768 channels).
    int cpb = 4096 / (S > 0 ? S : 1);
    if (cpb < 1) cpb = 1;
    if (cpb > C) cpb = C;
    return cpb;
}

extern "C" {

int objgan_roi_align_forward(const float* features, const float* rois, float* output,
                             int num_rois, int roi_cols, int channels, int height, int width,
                             int aligned_height, int aligned_width, float spatial_scale,
                             void* stream) {
    OG_ENTRY();
    if (roi_cols != 5) return OG_BAD_ARGS;                       // roi_align_cuda.c:18-22
    if (aligned_height * aligned_width > ROI_MAX_SAMPLES || aligned_height < 2 || aligned_width < 2)
        return OG_BAD_ARGS;
    if (num_rois <= 0 || channels <= 0) return OG_OK;
    const int S = aligned_height * aligned_width;
    const int cpb = roi_c_per_block(channels, S);
    dim3 grid(num_rois, og_cdiv(channels, cpb));
    hipLaunchKernelGGL(roi_align_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       features, rois, output, channels, height, width,
                       aligned_height, aligned_width, spatial_scale, cpb);
    return og_launch_status();
}

int objgan_roi_align_backward(const float* top_grad, const float* rois, float* bottom_grad,
                              int batch_size, int num_rois, int roi_cols, int channels,
                              int height, int width, int aligned_height, int aligned_width,
                              float spatial_scale, void* stream) {
    OG_ENTRY();
    if (roi_cols != 5) return OG_BAD_ARGS;
    if (aligned_height * aligned_width > ROI_MAX_SAMPLES || aligned_height < 2 || aligned_width < 2)
        return OG_BAD_ARGS;
    if (num_rois <= 0 || channels <= 0) return OG_OK;
    const int S = aligned_height * aligned_width;
    const size_t lds = (size_t)height * width * sizeof(float) + (size_t)ROI_BATCH * S * (sizeof(RoiSample) + sizeof(float));
    if (batch_size > 0 && lds <= 64 * 1024 && (double)batch_size * channels * height * width < 2.0e9) {
        dim3 grid(channels, batch_size);
        hipLaunchKernelGGL(roi_align_bwd_plane_kernel, grid, dim3(256), lds, (hipStream_t)stream,
                           top_grad, rois, bottom_grad, num_rois, channels, height, width,
                           aligned_height, aligned_width, spatial_scale);
        return og_launch_status();
    }
    const int cpb = roi_c_per_block(channels, S);
    dim3 grid(num_rois, og_cdiv(channels, cpb));
    hipLaunchKernelGGL(roi_align_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       top_grad, rois, bottom_grad, channels, height, width,
                       aligned_height, aligned_width, spatial_scale, cpb);
    return og_launch_status();
}

int objgan_avgpool2s1_forward(const float* in, float* out, long planes, int ih, int iw,
                              void* stream) {
    OG_ENTRY();
    if (ih < 2 || iw < 2) return OG_BAD_ARGS;
    const long total = planes * (ih - 1) * (iw - 1);
    if (total <= 0) return OG_OK;
    hipLaunchKernelGGL(avgpool2s1_fwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, out, total, ih, iw);
    return og_launch_status();
}

int objgan_avgpool2s1_backward(const float* grad_out, float* grad_in, long planes, int ih,
                               int iw, void* stream) {
    OG_ENTRY();
    if (ih < 2 || iw < 2) return OG_BAD_ARGS;
    const long total = planes * ih * iw;
    if (total <= 0) return OG_OK;
    hipLaunchKernelGGL(avgpool2s1_bwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, grad_out, grad_in, total, ih, iw);
    return og_launch_status();
}

}  // extern "C"
