// Training images on the device: the reference resizes every decoded image to the branch sizes on the host,
//     transforms.Resize((s, s))(img)  ->  ToTensor()  ->  Normalize((.5,.5,.5), (.5,.5,.5))
// (reference image_generation/miscc/load.py:141-150, trainDataset.py:60-66), i.e. Pillow's antialiased bilinear
// Image.resize once per branch and image.  Here the decoded 8-bit image crosses PCIe once and the three sizes are
// produced on the GPU, BIT FOR BIT what Pillow computes (ImagingResample, src/libImaging/Resample.c; restated in
// oracle/pil_resize.py and pinned there against the installed Pillow):
//   * coefficients: triangle filter stretched by max(scale, 1), evaluated and normalised in double in Pillow's
//     operation order (this file is compiled with -ffp-contract=off: no fused multiply-adds), rounded to 22-bit
//     fixed point;
//   * horizontal pass first, int32 accumulation with the rounding offset, clip to uint8; vertical pass on the
//     uint8 intermediate; then u8 / 255 -> (v - 0.5) / 0.5 in fp32 like torch.
// A pass whose size does not change is the identity in this arithmetic (weight 2^22 on one tap), so Pillow's
// "skip the pass" needs no special case.  Images of a batch have different sizes: they arrive back to back in one
// byte buffer with an offset / height / width table.  All three kernels are tiny HBM-bound streaming kernels
// (a batch of sixteen 640x480 images is 15 MB).
#include "common.h"

#define PIL_PRECISION_BITS (32 - 8 - 2)

struct PilGeom {
    const unsigned char* src;
    const long* offs;
    const int* hs;
    const int* ws;
    int B, Hmax, kmax, S;
};

// coef[((b*2 + axis)*S + xx) * (kmax + 2)] = {first source index, tap count, kmax fixed-point weights}
__global__ __launch_bounds__(256) void pil_coeff_kernel(PilGeom g, int* __restrict__ coef) {
    const int b = blockIdx.y, axis = blockIdx.z;
    const int in_size = axis == 0 ? g.ws[b] : g.hs[b];
    const double scale = (double)(float)in_size / g.S;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int xx = blockIdx.x * blockDim.x + threadIdx.x; xx < g.S; xx += gridDim.x * blockDim.x) {
        int* row = coef + ((long)(b * 2 + axis) * g.S + xx) * (g.kmax + 2);
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        int n = xmax - xmin;
        if (n > g.kmax) n = g.kmax;                    // cannot happen when kmax covers the largest image
        double ww = 0.0;
        for (int x = 0; x < n; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            ww += a < 1.0 ? 1.0 - a : 0.0;
        }
        row[0] = xmin;
        row[1] = n;
        for (int x = 0; x < g.kmax; ++x) {
            double w = 0.0;
            if (x < n) {
                double a = (x + xmin - center + 0.5) * ss;
                if (a < 0.0) a = -a;
                w = a < 1.0 ? 1.0 - a : 0.0;
                if (ww != 0.0) w /= ww;
            }
            row[2 + x] = (int)(0.5 + w * (double)(1 << PIL_PRECISION_BITS));
        }
    }
}

__device__ __forceinline__ unsigned char pil_clip8(int v) {
    v >>= PIL_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[b][y][xx][c] (u8, [B][Hmax][S][3]) = horizontal pass of image b, row y
__global__ __launch_bounds__(256) void pil_horizontal_kernel(PilGeom g, const int* __restrict__ coef,
                                                            unsigned char* __restrict__ tmp) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int H = g.hs[b], W = g.ws[b];
    if (y >= H) return;
    const unsigned char* line = g.src + g.offs[b] + (long)y * W * 3;
    unsigned char* out = tmp + ((long)b * g.Hmax + y) * g.S * 3;
    for (int xx = blockIdx.x * blockDim.x + threadIdx.x; xx < g.S; xx += gridDim.x * blockDim.x) {
        const int* row = coef + ((long)(b * 2 + 0) * g.S + xx) * (g.kmax + 2);
        const int x0 = row[0], n = row[1];
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int k = row[2 + x];
            const unsigned char* p = line + (long)(x0 + x) * 3;
            s0 += p[0] * k;
            s1 += p[1] * k;
            s2 += p[2] * k;
        }
        out[xx * 3 + 0] = pil_clip8(s0);
        out[xx * 3 + 1] = pil_clip8(s1);
        out[xx * 3 + 2] = pil_clip8(s2);
    }
}

// out[b][c][yy][xx] (float, [B][3][S][S]) = normalise(vertical pass of tmp)
__global__ __launch_bounds__(256) void pil_vertical_kernel(PilGeom g, const int* __restrict__ coef,
                                                          const unsigned char* __restrict__ tmp,
                                                          float* __restrict__ out) {
    const int b = blockIdx.z, yy = blockIdx.y;
    const int* row = coef + ((long)(b * 2 + 1) * g.S + yy) * (g.kmax + 2);
    const int y0 = row[0], n = row[1];
    const unsigned char* base = tmp + ((long)b * g.Hmax + y0) * g.S * 3;
    for (int xx = blockIdx.x * blockDim.x + threadIdx.x; xx < g.S; xx += gridDim.x * blockDim.x) {
        int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const int k = row[2 + y];
            const unsigned char* p = base + ((long)y * g.S + xx) * 3;
            s0 += p[0] * k;
            s1 += p[1] * k;
            s2 += p[2] * k;
        }
        const int v[3] = {pil_clip8(s0), pil_clip8(s1), pil_clip8(s2)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = (float)v[c] / 255.0f;
            out[(((long)b * 3 + c) * g.S + yy) * g.S + xx] = (t - 0.5f) / 0.5f;
        }
    }
}

extern "C" {

// Number of taps Pillow allots per output pixel for the largest source side of a batch:
// (int)ceil(max(in / S, 1)) * 2 + 1 (precompute_coeffs, bilinear support 1.0).
int objgan_resize_pil_kmax(int max_in, int S) {
    if (max_in < 1 || S < 1) return 0;
    double scale = (double)(float)max_in / S;
    if (scale < 1.0) scale = 1.0;
    int c = (int)scale;
    if ((double)c < scale) ++c;
    return c * 2 + 1;
}

// src: B RGB images, 8 bits per channel, rows of W*3 bytes, image b at byte offs[b] with hs[b] x ws[b] pixels
// (offs / hs / ws: device arrays).  out: [B, 3, S, S] float = Normalize(ToTensor(resize((S, S), BILINEAR))).
// Hmax >= every height, kmax >= objgan_resize_pil_kmax(largest side, S).
// coef_scratch: B*2*S*(kmax+2) ints; tmp_scratch: B*Hmax*S*3 bytes.
int objgan_resize_pil_rgb8(const unsigned char* src, const long* offs, const int* hs, const int* ws, int B, int Hmax,
                           int kmax, int S, int* coef_scratch, unsigned char* tmp_scratch, float* out, void* stream) {
    OG_ENTRY();
    if (B <= 0) return OG_OK;
    if (!src || !offs || !hs || !ws || !coef_scratch || !tmp_scratch || !out || Hmax < 1 || kmax < 3 || S < 1 ||
        B > 65535 || Hmax > 65535 || S > 65535)
        return OG_BAD_ARGS;
    PilGeom g = {src, offs, hs, ws, B, Hmax, kmax, S};
    hipStream_t s = (hipStream_t)stream;
    const int gx = og_cdiv(S, 256);
    hipLaunchKernelGGL(pil_coeff_kernel, dim3(gx, B, 2), dim3(256), 0, s, g, coef_scratch);
    hipLaunchKernelGGL(pil_horizontal_kernel, dim3(gx, Hmax, B), dim3(256), 0, s, g, (const int*)coef_scratch,
                       tmp_scratch);
    hipLaunchKernelGGL(pil_vertical_kernel, dim3(gx, S, B), dim3(256), 0, s, g, (const int*)coef_scratch,
                       (const unsigned char*)tmp_scratch, out);
    return og_launch_status();
}

}  // extern "C"
