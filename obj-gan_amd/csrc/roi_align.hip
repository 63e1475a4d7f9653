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

// Backward, ORDERED: the taps of an image's ROI samples are the same for all C channel planes, so the samples are sorted
// ONCE per image by the pixel of their first tap (stable: roi, then sample order -- roi_tap_table_kernel, one workgroup
// per image), and every (touched pixel, channel) then GATHERS what lands on it in a fixed order: tap 0 of the samples
// anchored at p, tap 1 of those anchored at p - 1, tap 2 of p - W, tap 3 of p - W - 1 (roi_align_bwd_gather_kernel).
// One writer per element, no atomics on floats, bit-reproducible.  Same per-term arithmetic as roi_align_bwd_kernel
// (and the reference's CUDA kernel, which leaves the order of its atomicAdds unspecified).  r02's privatised scatter
// took 146 us per launch, an ordered gather without the table 609 us.
#define ROI_TAB_MAX 512           // rois per call the table path takes (the hot path has 160: 16 images x 10; 320 at B = 32)
struct RoiTabDims { int maxs, maxp, o_start, o_sorted, o_smp, o_rlist, stride; };
static inline RoiTabDims roi_tab_dims(int num_rois, int HW, int S) {
    RoiTabDims d;
    d.maxs = num_rois * S;                                   // samples of one image (all rois may belong to it)
    d.maxp = HW < 4 * d.maxs ? HW : 4 * d.maxs;              // touched pixels
    // ints per image: [hdr 4][pix maxp][start HW + 1][sorted maxs][smp 4 x maxs][rois of the image], every part 16-byte aligned
    d.o_start = 4 + ((d.maxp + 3) & ~3);
    d.o_sorted = d.o_start + ((HW + 1 + 3) & ~3);
    d.o_smp = d.o_sorted + ((d.maxs + 3) & ~3);
    d.o_rlist = d.o_smp + 4 * d.maxs;
    d.stride = d.o_rlist + ((num_rois + 3) & ~3);
    return d;
}

__global__ __launch_bounds__(256) void roi_tap_table_kernel(
    const float* __restrict__ rois, int* __restrict__ ws, int num_rois,
    int C, int H, int W, int AH, int AW, float spatial_scale, RoiTabDims d) {
    extern __shared__ int tab_lds[];                                    // cnt[HW], st[HW], soff[maxs] (16 bit)
    const int HW = H * W;
    const int S = AH * AW;
    int* cnt = tab_lds;
    int* st = tab_lds + HW;
    unsigned short* soff = reinterpret_cast<unsigned short*>(tab_lds + 2 * HW);     // anchor pixel, 0xffff: no gradient
    __shared__ int s_list[ROI_TAB_MAX];                                 // rois of this image, in index order
    __shared__ int s_wave[4];
    __shared__ int s_scan[2 * 256];
    const int b = blockIdx.x;
    int* hdr = ws + (size_t)b * d.stride;
    int* pix = hdr + 4;
    int* start = hdr + d.o_start;
    int* sorted = hdr + d.o_sorted;
    int* smp = hdr + d.o_smp;
    int* rlist = hdr + d.o_rlist;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) cnt[i] = 0;
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
        if (mine) s_list[pos] = r;
        nlist += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int n_smp = nlist * S;
    for (int i = threadIdx.x; i < nlist; i += blockDim.x) rlist[i] = s_list[i];
    for (int i = threadIdx.x; i < n_smp; i += blockDim.x) {
        const int k = i / S, sidx = i - k * S;
        const int r = s_list[k];
        RoiSample g;
        roi_geometry(rois + (size_t)r * 5, spatial_scale, H, W, AH, AW, sidx, g);
        const int off = g.valid ? g.off : -1;
        soff[i] = (unsigned short)(off >= 0 ? off : 0xffff);
        smp[4 * i + 0] = k;                            // the roi's position in this image's list (rlist[k] = r)
        smp[4 * i + 1] = __float_as_int(g.h_ratio);
        smp[4 * i + 2] = __float_as_int(g.w_ratio);
        smp[4 * i + 3] = (r << 8) | sidx;
        if (off >= 0) atomicAdd(&cnt[off], 1);       // anchors have row <= H - 2 and column <= W - 2
    }
    __syncthreads();
    // exclusive scans over the pixels (samples anchored before a pixel; touched pixels before it): 256 contiguous slices
    const int per = (HW + 255) / 256;
    const int p0 = min(HW, (int)threadIdx.x * per), p1 = min(HW, p0 + per);
    auto touched = [&](int p) -> int {
        return (cnt[p] | (p >= 1 ? cnt[p - 1] : 0) | (p >= W ? cnt[p - W] : 0) | (p >= W + 1 ? cnt[p - W - 1] : 0)) != 0;
    };
    int e_sum = 0, n_sum = 0;
    for (int p = p0; p < p1; ++p) { e_sum += cnt[p]; n_sum += touched(p); }
    s_scan[2 * threadIdx.x] = e_sum; s_scan[2 * threadIdx.x + 1] = n_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int e = 0, n = 0;
        for (int t = 0; t < 256; ++t) {
            const int a = s_scan[2 * t], c = s_scan[2 * t + 1];
            s_scan[2 * t] = e; s_scan[2 * t + 1] = n;
            e += a; n += c;
        }
        hdr[0] = n; hdr[1] = e; hdr[2] = n_smp; hdr[3] = nlist;
        start[HW] = e;
    }
    __syncthreads();
    {
        int e = s_scan[2 * threadIdx.x], n = s_scan[2 * threadIdx.x + 1];
        for (int p = p0; p < p1; ++p) {
            if (touched(p)) pix[n++] = p;
            st[p] = e; start[p] = e;
            e += cnt[p];
        }
    }
    __syncthreads();
    // stable placement: a sample's slot = start of its anchor + the number of earlier samples with the same anchor
    for (int i = threadIdx.x; i < n_smp; i += blockDim.x) {
        const unsigned short off = soff[i];
        if (off == 0xffff) continue;
        int rank = 0;
        for (int j = 0; j < i; ++j) rank += soff[j] == off;
        sorted[st[off] + rank] = i;
    }
}

// grid (channel slabs, images): 16 lanes per (touched pixel, channel) element.  The element's tap list (the four anchor
// ranges, concatenated) is strided over the 16 lanes and their partial sums meet in a fixed butterfly: with the 1/16
// spatial-scale of the hot path most of an image's 360 samples share a handful of anchors, and one lane walking 300
// dependent loads took 284 us per launch.
// Round 6: the per-tap chain sorted[k] -> smp[..] -> top_grad[..] was three dependent global loads, repeated by every channel
// of the slab (111 us per launch, 20 MB moved: latency, not bytes).  A workgroup now stages what its elements read ONCE:
// the image's samples in anchor order (16 bytes each) and the top gradients of the image's rois for the slab's channels
// ([roi of the image][channel][sample]: one contiguous, fully coalesced piece per roi) -- every tap is then two LDS reads.
// Same taps, same per-lane order, same butterfly: the sums are bit-identical to the global-memory form, which stays as the
// path for an image whose table or gradient slab does not fit (ROI_LDS_SMP samples / ROI_LDS_G floats).
#define ROI_SUB 16
#define ROI_LDS_SMP 1024
#define ROI_LDS_G 4096
__device__ __forceinline__ float roi_tap_value(float dv, float h_ratio, float w_ratio, int t) {
    // reference kernel.cu:137-140: `(1. - h_ratio)` is double, `(1 - w_ratio)` is FLOAT (int
    // literal); the two h_ratio terms are all-float products.
    const double omh = 1.0 - (double)h_ratio;
    const float omw = 1.0f - w_ratio;
    const float dh_ = dv * h_ratio;
    if (t == 0) return (float)(((double)dv * omh) * (double)omw);
    if (t == 1) return (float)(((double)dv * omh) * (double)w_ratio);
    if (t == 2) return dh_ * omw;
    return dh_ * w_ratio;
}

#define ROI_LDS_START 4608          // anchor-range starts of the map in LDS (HW + 1 <= this: the 64 x 64 maps of the hot path)
#define ROI_SEQ 24                  // an element with at most this many taps is summed by ONE lane
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(
    const float* __restrict__ top_grad, const int* __restrict__ ws, float* __restrict__ bottom_grad,
    int C, int HW, int W, int S, RoiTabDims d, int cpb) {
    __shared__ int4 s_smp[ROI_LDS_SMP];
    __shared__ float s_g[ROI_LDS_G];
    __shared__ int s_start[ROI_LDS_START];
    const int b = blockIdx.y;
    const int* hdr = ws + (size_t)b * d.stride;
    const int* pix = hdr + 4;
    const int* start = hdr + d.o_start;
    const int* sorted = hdr + d.o_sorted;
    const int4* smp = reinterpret_cast<const int4*>(hdr + d.o_smp);
    const int* rlist = hdr + d.o_rlist;
    const int n_pix = hdr[0], n_e = hdr[1], nlist = hdr[3];
    const int c0 = blockIdx.x * cpb;
    const int nc = min(cpb, C - c0);
    const bool staged = n_e <= ROI_LDS_SMP && nlist * nc * S <= ROI_LDS_G;          // (uniform over the workgroup)
    const bool st_lds = HW + 1 <= ROI_LDS_START;
    if (staged) {
        for (int k = threadIdx.x; k < n_e; k += 256) s_smp[k] = smp[sorted[k]];
        const int per = nc * S;                                                     // floats of one roi's slab
        for (int i = threadIdx.x; i < nlist * per; i += 256) {
            const int kl = i / per, j = i - kl * per;
            s_g[i] = top_grad[((size_t)rlist[kl] * C + c0) * S + j];
        }
    }
    if (st_lds)
        for (int i = threadIdx.x; i <= HW; i += 256) s_start[i] = start[i];
    __syncthreads();
    auto ranges = [&](int p, int (&k0)[4], int (&lim)[4]) -> int {                  // the four anchor ranges whose taps land on p
        int total = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int q = p - (t & 1) - (t >> 1) * W;                               // anchor whose tap t lands on p
            const int s0 = q >= 0 ? (st_lds ? s_start[q] : start[q]) : 0;
            const int s1 = q >= 0 ? (st_lds ? s_start[q + 1] : start[q + 1]) : 0;
            k0[t] = s0;
            total += s1 - s0;
            lim[t] = total;
        }
        return total;
    };
    auto tap = [&](int m, const int (&k0)[4], const int (&lim)[4], int ci, int c) -> float {
        const int t = (m >= lim[0]) + (m >= lim[1]) + (m >= lim[2]);
        const int k = t == 0 ? k0[0] + m : t == 1 ? k0[1] + m - lim[0] : t == 2 ? k0[2] + m - lim[1] : k0[3] + m - lim[2];
        float dv, h_ratio, w_ratio;
        if (staged) {
            const int4 g = s_smp[k];
            h_ratio = __int_as_float(g.y); w_ratio = __int_as_float(g.z);
            dv = s_g[(g.x * nc + ci) * S + (g.w & 255)];
        } else {
            const int4 g = smp[sorted[k]];
            h_ratio = __int_as_float(g.y); w_ratio = __int_as_float(g.z);
            dv = top_grad[((size_t)(g.w >> 8) * C + c) * S + (g.w & 255)];
        }
        return roi_tap_value(dv, h_ratio, w_ratio, t);
    };
    // Two regimes, decided per element by its tap count (a function of the rois alone: the same path in every run).
    // (1) Few taps -- boxes that spread their 7 x 7 samples over many pixels: ONE lane per (pixel, channel) element sums its
    //     taps in order; 256 read-modify-writes of bottom_grad in flight per workgroup instead of 16 (round 5's 16-lane
    //     groups were latency-bound there: 111 us per launch for 20 MB).
    for (int e = threadIdx.x; e < n_pix * nc; e += 256) {
        const int ci = e / n_pix, ip = e - ci * n_pix;
        int k0[4], lim[4];
        const int p = pix[ip];
        const int total = ranges(p, k0, lim);
        if (total > ROI_SEQ) continue;
        float acc = 0.f;
        for (int m = 0; m < total; ++m) acc += tap(m, k0, lim, ci, c0 + ci);
        bottom_grad[((size_t)b * C + c0 + ci) * HW + p] += acc;
    }
    // (2) Many taps -- small boxes whose samples share a handful of anchors: 16 lanes stride the tap list, fixed butterfly.
    const int sub = threadIdx.x & (ROI_SUB - 1), grp = threadIdx.x / ROI_SUB;
    for (int ip = grp; ip < n_pix; ip += 256 / ROI_SUB) {
        int k0[4], lim[4];
        const int p = pix[ip];
        const int total = ranges(p, k0, lim);
        if (total <= ROI_SEQ) continue;
        for (int ci = 0; ci < nc; ++ci) {
            float acc = 0.f;
            for (int m = sub; m < total; m += ROI_SUB) acc += tap(m, k0, lim, ci, c0 + ci);
#pragma unroll
            for (int o = ROI_SUB / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (sub == 0) bottom_grad[((size_t)b * C + c0 + ci) * HW + p] += acc;
        }
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
    (void)batch_size;
    if (roi_cols != 5) return OG_BAD_ARGS;
    if (aligned_height * aligned_width > ROI_MAX_SAMPLES || aligned_height < 2 || aligned_width < 2)
        return OG_BAD_ARGS;
    if (num_rois <= 0 || channels <= 0) return OG_OK;
    const int S = aligned_height * aligned_width;
    const int cpb = roi_c_per_block(channels, S);
    dim3 grid(num_rois, og_cdiv(channels, cpb));
    hipLaunchKernelGGL(roi_align_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       top_grad, rois, bottom_grad, channels, height, width,
                       aligned_height, aligned_width, spatial_scale, cpb);
    return og_launch_status();
}

// ints (4 bytes, = floats) of workspace for objgan_roi_align_backward_ordered; 0 = shape outside the table path
long objgan_roi_align_backward_ws_floats(int batch_size, int num_rois, int channels, int height, int width,
                                         int aligned_height, int aligned_width) {
    const int S = aligned_height * aligned_width;
    const long HW = (long)height * width;
    if (batch_size <= 0 || num_rois <= 0 || num_rois > ROI_TAB_MAX || S > ROI_MAX_SAMPLES) return 0;
    // table kernel's LDS: dynamic (two HW-int histograms + 16-bit anchors) + its static arrays (s_list, s_scan, s_wave);
    // 16-bit anchors; and its stable rank is quadratic in the samples of ONE image -- beyond ~64 rois x 36 samples of a
    // single image (the hot path has 10) the unordered scatter of objgan_roi_align_backward is the faster kernel
    const long static_lds = (long)(ROI_TAB_MAX + 4 + 2 * 256) * 4;
    if (2 * HW * 4 + (long)num_rois * S * 2 + static_lds > 64 * 1024 || HW >= 65535) return 0;
    if ((long)og_cdiv(num_rois, batch_size) * S > 64 * 36 && (long)num_rois * S > 8 * 64 * 36) return 0;
    if ((double)batch_size * channels * (double)HW >= 2.0e9) return 0;
    return (long)batch_size * roi_tab_dims(num_rois, (int)HW, S).stride;
}

// Same adjoint, fixed summation order (see roi_tap_table_kernel): bit-reproducible.  ws: the floats the query above
// returns; shapes it returns 0 for (or ws == NULL) take objgan_roi_align_backward's unordered atomics.
int objgan_roi_align_backward_ordered(const float* top_grad, const float* rois, float* bottom_grad,
                                      int batch_size, int num_rois, int roi_cols, int channels,
                                      int height, int width, int aligned_height, int aligned_width,
                                      float spatial_scale, float* ws, long ws_floats, void* stream) {
    OG_ENTRY();
    if (roi_cols != 5) return OG_BAD_ARGS;
    if (aligned_height * aligned_width > ROI_MAX_SAMPLES || aligned_height < 2 || aligned_width < 2)
        return OG_BAD_ARGS;
    if (num_rois <= 0 || channels <= 0) return OG_OK;
    const long need = objgan_roi_align_backward_ws_floats(batch_size, num_rois, channels, height, width,
                                                          aligned_height, aligned_width);
    if (need == 0 || !ws)
        return objgan_roi_align_backward(top_grad, rois, bottom_grad, batch_size, num_rois, roi_cols, channels,
                                         height, width, aligned_height, aligned_width, spatial_scale, stream);
    if (ws_floats < need) return OG_BAD_ARGS;
    const int S = aligned_height * aligned_width;
    const int HW = height * width;
    const RoiTabDims d = roi_tab_dims(num_rois, HW, S);
    const size_t lds = 2 * (size_t)HW * sizeof(int) + (((size_t)num_rois * S * 2 + 3) & ~(size_t)3);
    hipLaunchKernelGGL(roi_tap_table_kernel, dim3(batch_size), dim3(256), lds, (hipStream_t)stream,
                       rois, reinterpret_cast<int*>(ws), num_rois, channels, height, width,
                       aligned_height, aligned_width, spatial_scale, d);
    // (round 6 also built a dense form -- the slab's gradient planes in LDS, added to bottom_grad as whole planes with 16-byte
    //  accesses: 292 us per launch against 101 for the gather kernel, one workgroup per CU and 200 MB of plane traffic;
    //  removed, profiles/r06_ab_variants.txt (f))
    const int cpb = channels >= 64 ? 8 : 1;
    hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3(og_cdiv(channels, cpb), batch_size), dim3(256), 0,
                       (hipStream_t)stream, top_grad, reinterpret_cast<const int*>(ws), bottom_grad,
                       channels, HW, width, S, d, cpb);
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
