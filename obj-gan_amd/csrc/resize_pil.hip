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

// ---------------------------------------------------------------------------------------------------------------
// Per-box instance masks of the training loader (reference image_generation/miscc/load.py:160-176): every 64 x 64
// mask is resized to the feature scale (32) and to the three branch sizes (64 / 128 / 256) with
// skimage.transform.resize's defaults = scipy.ndimage.gaussian_filter (when shrinking) + scipy.ndimage.zoom(order 1,
// mode 'mirror', grid_mode) + clip.  One workgroup per mask: the source in LDS as float64, every output element
// evaluated in float64 in scipy's operation order (NI_Correlate1D symmetric branch, axis 0 then axis 1; NI_ZoomShift
// order 1: t = 0, t += (x * wy) * wx per neighbour, row-major) -- this file is built with -ffp-contract=off -- so the
// results equal scipy's BIT FOR BIT (tests/test_kernels_gpu.py; oracle/mask_resize.py states the same order on the CPU).
#define MASK_MAX_N 64
#define MASK_MAX_TAPS 17
#define MASK_MAX_SIZES 4
struct MaskResizeArgs {
    const double* src;              // [count][n][n]
    double* out[MASK_MAX_SIZES];    // out[k]: [count][size[k]][size[k]]
    int size[MASK_MAX_SIZES];
    int ntaps[MASK_MAX_SIZES];      // 2 * radius + 1 of the anti-aliasing filter of size k (0: none)
    double taps[MASK_MAX_SIZES][MASK_MAX_TAPS];
    int n, nsizes;
};

__device__ __forceinline__ int mask_mirror(int i, int n) {      // d c b | a b c d | c b a
    if (n == 1) return 0;
    const int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

__device__ __forceinline__ double mask_map_mirror(double c, int n) {      // map_coordinate, NI_EXTEND_MIRROR
    if (n <= 1) return 0.0;
    const int sz2 = 2 * n - 2;
    if (c < 0) {
        c = sz2 * (double)(int)(-c / sz2) + c;
        c = c <= 1 - n ? c + sz2 : -c;
    } else if (c > n - 1) {
        c -= sz2 * (double)(int)(c / sz2);
        if (c >= n) c = sz2 - c;
    }
    return c;
}

__global__ __launch_bounds__(256) void mask_resize_kernel(const MaskResizeArgs a) {
    __shared__ double A[MASK_MAX_N * MASK_MAX_N];
    __shared__ double Bf[MASK_MAX_N * MASK_MAX_N];
    __shared__ double red[2][4];
    const int n = a.n, nn = n * n, tid = threadIdx.x;
    const double* src = a.src + (size_t)blockIdx.x * nn;
    double lo = 1.0 / 0.0, hi = -1.0 / 0.0;
    for (int i = tid; i < nn; i += 256) {
        const double v = src[i];
        A[i] = v;
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = lo; red[1][tid >> 6] = hi; }
    __syncthreads();
    lo = fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3]));
    hi = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
    // sizes that need no filter first (they read the unfiltered source), the shrinking ones after them: each filters
    // A -> Bf (axis 0) -> A (axis 1) from a re-loaded source
    for (int pass = 0; pass < 2; ++pass) {
        for (int k = 0; k < a.nsizes; ++k) {
            const int S = a.size[k];
            const bool shrink = a.ntaps[k] > 0;
            if (shrink != (pass == 1)) continue;
            double* out = a.out[k] + (size_t)blockIdx.x * S * S;
            if (S == n) {
                for (int i = tid; i < nn; i += 256) out[i] = A[i];
                continue;
            }
            if (shrink) {
                __syncthreads();
                for (int i = tid; i < nn; i += 256) A[i] = src[i];
                __syncthreads();
                const int c = a.ntaps[k] / 2;
                for (int ax = 0; ax < 2; ++ax) {
                    const double* X = ax == 0 ? A : Bf;
                    double* Y = ax == 0 ? Bf : A;
                    for (int i = tid; i < nn; i += 256) {
                        const int r = i / n, q = i - r * n;
                        const int l = ax == 0 ? r : q;
                        double tmp = X[i] * a.taps[k][c];
                        for (int jj = -c; jj < 0; ++jj) {
                            const int l0 = mask_mirror(l + jj, n), l1 = mask_mirror(l - jj, n);
                            const double x0 = ax == 0 ? X[l0 * n + q] : X[r * n + l0];
                            const double x1 = ax == 0 ? X[l1 * n + q] : X[r * n + l1];
                            tmp = tmp + (x0 + x1) * a.taps[k][jj + c];
                        }
                        Y[i] = tmp;
                    }
                    __syncthreads();
                }
            }
            const double zoom = (double)n / (double)S;
            for (int e = tid; e < S * S; e += 256) {
                const int oy = e / S, ox = e - oy * S;
                const double cy = mask_map_mirror(((double)oy + 0.5) * zoom - 0.5, n);
                const double cx = mask_map_mirror(((double)ox + 0.5) * zoom - 0.5, n);
                const int iy = (int)floor(cy), ix = (int)floor(cx);
                const double wy1 = cy - iy, wx1 = cx - ix, wy0 = 1.0 - wy1, wx0 = 1.0 - wx1;
                const int i0 = mask_mirror(iy, n), i1 = mask_mirror(iy + 1, n);
                const int j0 = mask_mirror(ix, n), j1 = mask_mirror(ix + 1, n);
                double t = 0.0;
                t += (A[i0 * n + j0] * wy0) * wx0;
                t += (A[i0 * n + j1] * wy0) * wx1;
                t += (A[i1 * n + j0] * wy1) * wx0;
                t += (A[i1 * n + j1] * wy1) * wx1;
                t = t < lo ? lo : t;
                t = t > hi ? hi : t;
                out[e] = t;
            }
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

// count square masks src[count][n][n] (float64, n <= 64) -> nsizes resized copies out[k][count][sizes[k]][sizes[k]]
// (float64, device pointers in a HOST array), skimage.transform.resize defaults as scipy evaluates them (see
// mask_resize_kernel).  taps[k * 17 ..]: the normalised anti-aliasing Gaussian of size k (ntaps[k] = 2 radius + 1 <= 17
// values, host array; 0 taps: not shrinking) -- the caller computes them in double exactly as scipy does.
int objgan_mask_resize(const double* src, int count, int n, int nsizes, const int* sizes, double* const* out,
                       const int* ntaps, const double* taps, void* stream) {
    OG_ENTRY();
    if (count <= 0) return OG_OK;
    if (!src || !sizes || !out || !ntaps || n < 2 || n > MASK_MAX_N || nsizes < 1 || nsizes > MASK_MAX_SIZES)
        return OG_BAD_ARGS;
    MaskResizeArgs a;
    a.src = src; a.n = n; a.nsizes = nsizes;
    for (int k = 0; k < nsizes; ++k) {
        if (!out[k] || sizes[k] < 1 || sizes[k] > 4096 || ntaps[k] < 0 || ntaps[k] > MASK_MAX_TAPS ||
            (ntaps[k] > 0 && (!taps || !(ntaps[k] & 1))) || (sizes[k] < n) != (ntaps[k] > 0))
            return OG_BAD_ARGS;
        a.out[k] = out[k]; a.size[k] = sizes[k]; a.ntaps[k] = ntaps[k];
        for (int j = 0; j < ntaps[k]; ++j) a.taps[k][j] = taps[k * MASK_MAX_TAPS + j];
    }
    hipLaunchKernelGGL(mask_resize_kernel, dim3(count), dim3(256), 0, (hipStream_t)stream, a);
    return og_launch_status();
}

}  // extern "C"
