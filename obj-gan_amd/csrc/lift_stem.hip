// Layout-map stem of the object discriminators WITHOUT the 512x512 lift (reference
// image_generation/model.py:1217-1226, 1283-1292):
//
//     s_code = shp_code(F.interpolate(seg, 512, 'bilinear', align_corners=True))
//     shp_code = ReflectionPad2d(1) -> Conv2d(80, 12, 3) -> InstanceNorm -> LeakyReLU
//
// Lift U, reflection R and the tap shifts are linear and act on pixels, the filter bank acts on channels:
//
//     conv(R U seg)[co] = sum_t  S_t R U ( sum_ci W[co, ci, t] seg[ci] )  =  sum_t (S_t R U) z[t, co]
//
// so the channel contraction runs at the LOW resolution as a 1x1 convolution 80 -> 9*12 on the MFMA kernel
// (4x fewer pixels, no 32-row padding for 12 outputs), and what is left is this file: the separable operator
// A_d = S_d R U per axis (d = tap offset -1, 0, +1), applied along columns, then rows, to 9*Mo / 3*Mo planes --
// HBM-bound streaming kernels.  The 1.3 GB lifted layout map, its bilinear kernel, the thin 80 -> 12
// convolution at 512^2 and its weight gradient (a 12-row MFMA tile: 37 TFLOP/s) disappear; results differ
// from the reference formulation by fp32 re-association only.
//
// The per-axis operators come as tables built by the host (objgan_hip/ops.py: same fp32 source-index
// arithmetic as bilinear_fwd_kernel): forward  y[o] = sum_d l0[d][o] x_d[i0[d][o]] + l1[d][o] x_d[i1[d][o]],
// backward = the CSR transpose of the same table, so the pair is adjoint by construction.
#include "common.h"

struct AxisFwd { const int* i0; const int* i1; const float* l1; };      // each [3][S]
struct AxisBwd { const int* off; const int* idx; const float* wt; };    // off [3][h + 1]

// F[n][dh*Mo + co][qh][ow] = sum_dw A_dw(z[n][(dh*3 + dw)*Mo + co][qh][:])[ow]
__global__ __launch_bounds__(256) void lift_cols_fwd_kernel(const float* __restrict__ z, float* __restrict__ F,
                                                            long total, int Mo, int h, int w, int SW, AxisFwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int qh = (int)(r % h); r /= h;
        const int c3 = (int)(r % (3 * Mo));
        const long n = r / (3 * Mo);
        const int dh = c3 / Mo, co = c3 - dh * Mo;
        float acc = 0.f;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const float* zp = z + ((n * 9 + dh * 3 + dw) * Mo + co) * (long)h * w + (long)qh * w;
            const float l1 = t.l1[dw * SW + ow];
            acc += (1.0f - l1) * zp[t.i0[dw * SW + ow]] + l1 * zp[t.i1[dw * SW + ow]];
        }
        F[e] = acc;
    }
}

// y[n][co][oh][ow] = bias[co] + sum_dh A_dh(F[n][dh*Mo + co][:][ow])[oh]
__global__ __launch_bounds__(256) void lift_rows_fwd_kernel(const float* __restrict__ F, const float* __restrict__ bias,
                                                            float* __restrict__ y, long total, int Mo, int h,
                                                            int SH, int SW, AxisFwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int oh = (int)(r % SH); r /= SH;
        const int co = (int)(r % Mo);
        const long n = r / Mo;
        float acc = bias ? bias[co] : 0.f;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const float* fp = F + ((n * 3 + dh) * Mo + co) * (long)h * SW + ow;
            const float l1 = t.l1[dh * SH + oh];
            acc += (1.0f - l1) * fp[(long)t.i0[dh * SH + oh] * SW] + l1 * fp[(long)t.i1[dh * SH + oh] * SW];
        }
        y[e] = acc;
    }
}

// dF[n][dh*Mo + co][qh][ow] = sum_{e in CSR_dh[qh]} wt_e dy[n][co][idx_e][ow]
__global__ __launch_bounds__(256) void lift_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dF,
                                                            long total, int Mo, int h, int SH, int SW, AxisBwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int qh = (int)(r % h); r /= h;
        const int c3 = (int)(r % (3 * Mo));
        const long n = r / (3 * Mo);
        const int dh = c3 / Mo, co = c3 - dh * Mo;
        const float* gp = dy + (n * Mo + co) * (long)SH * SW + ow;
        const int b = t.off[dh * (h + 1) + qh], en = t.off[dh * (h + 1) + qh + 1];
        float acc = 0.f;
        for (int k = b; k < en; ++k) acc += t.wt[k] * gp[(long)t.idx[k] * SW];
        dF[e] = acc;
    }
}

// dz[n][(dh*3 + dw)*Mo + co][qh][qw] = sum_{e in CSR_dw[qw]} wt_e dF[n][dh*Mo + co][qh][idx_e]
__global__ __launch_bounds__(256) void lift_cols_bwd_kernel(const float* __restrict__ dF, float* __restrict__ dz,
                                                            long total, int Mo, int h, int w, int SW, AxisBwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int qw = (int)(e % w);
        long r = e / w;
        const int qh = (int)(r % h); r /= h;
        const int c9 = (int)(r % (9 * Mo));
        const long n = r / (9 * Mo);
        const int tt = c9 / Mo, co = c9 - tt * Mo;
        const int dh = tt / 3, dw = tt - dh * 3;
        const float* fp = dF + (((n * 3 + dh) * Mo + co) * (long)h + qh) * SW;
        const int b = t.off[dw * (w + 1) + qw], en = t.off[dw * (w + 1) + qw + 1];
        float acc = 0.f;
        for (int k = b; k < en; ++k) acc += t.wt[k] * fp[t.idx[k]];
        dz[e] = acc;
    }
}

extern "C" {

// z [N, 9*Mo, h, w] (channel (dh*3 + dw)*Mo + co: the 1x1 convolution of the layout map with tap (dh, dw) of the
// 3x3 bank) -> y [N, Mo, SH, SW] = bias + sum_taps shift_tap(reflect_pad(lift(z_tap))).  scratch: N*3*Mo*h*SW
// floats.  Tables (device): ci0/ci1/cl1 [3][SW] for the columns, ri0/ri1/rl1 [3][SH] for the rows.
int objgan_lift_taps_forward(const float* z, const float* bias, float* y, float* scratch, int N, int Mo, int h,
                             int w, int SH, int SW, const int* ci0, const int* ci1, const float* cl1,
                             const int* ri0, const int* ri1, const float* rl1, void* stream) {
    OG_ENTRY();
    if (N <= 0 || Mo <= 0) return OG_OK;
    if (h < 1 || w < 1 || SH < 2 || SW < 2 || !scratch) return OG_BAD_ARGS;
    const long t1 = (long)N * 3 * Mo * h * SW, t2 = (long)N * Mo * SH * SW;
    AxisFwd tc = {ci0, ci1, cl1}, trw = {ri0, ri1, rl1};
    hipLaunchKernelGGL(lift_cols_fwd_kernel, dim3(og_stream_grid(t1, 256)), dim3(256), 0, (hipStream_t)stream,
                       z, scratch, t1, Mo, h, w, SW, tc);
    hipLaunchKernelGGL(lift_rows_fwd_kernel, dim3(og_stream_grid(t2, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch, bias, y, t2, Mo, h, SH, SW, trw);
    return og_launch_status();
}

// dy [N, Mo, SH, SW] -> dz [N, 9*Mo, h, w] (the adjoint of the above).  Tables: CSR transposes of the forward
// tables: roff [3][h+1], ridx / rwt for the rows, coff [3][w+1], cidx / cwt for the columns.
int objgan_lift_taps_backward(const float* dy, float* dz, float* scratch, int N, int Mo, int h, int w, int SH,
                              int SW, const int* roff, const int* ridx, const float* rwt, const int* coff,
                              const int* cidx, const float* cwt, void* stream) {
    OG_ENTRY();
    if (N <= 0 || Mo <= 0) return OG_OK;
    if (h < 1 || w < 1 || SH < 2 || SW < 2 || !scratch) return OG_BAD_ARGS;
    const long t1 = (long)N * 3 * Mo * h * SW, t2 = (long)N * 9 * Mo * h * w;
    AxisBwd trw = {roff, ridx, rwt}, tc = {coff, cidx, cwt};
    hipLaunchKernelGGL(lift_rows_bwd_kernel, dim3(og_stream_grid(t1, 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, scratch, t1, Mo, h, SH, SW, trw);
    hipLaunchKernelGGL(lift_cols_bwd_kernel, dim3(og_stream_grid(t2, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch, dz, t2, Mo, h, w, SW, tc);
    return og_launch_status();
}

}  // extern "C"
