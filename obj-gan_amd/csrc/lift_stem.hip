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

// One thread = one pixel position; its table entries are read ONCE and serve all planes of the image (the
// first version had one thread per output element and re-read the tables for each: 15-18 loads per output,
// 1 ms per call for the column adjoint).
#define LIFT_MAXE 8      // a source index is touched by at most 2 * ceil(S / h) + 2 lifted positions (6 for x2)

// F[n][dh*Mo + co][qh][ow] = sum_dw A_dw(z[n][(dh*3 + dw)*Mo + co][qh][:])[ow]
__global__ __launch_bounds__(256) void lift_cols_fwd_kernel(const float* __restrict__ z, float* __restrict__ F,
                                                            long total, int Mo, int h, int w, int SW, AxisFwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int qh = (int)(r % h);
        const long n = r / h;
        int i0[3], i1[3];
        float l1[3];
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) { i0[dw] = t.i0[dw * SW + ow]; i1[dw] = t.i1[dw * SW + ow]; l1[dw] = t.l1[dw * SW + ow]; }
        const long hw = (long)h * w;
        const float* zn = z + n * 9 * Mo * hw + (long)qh * w;
        float* Fn = F + (n * 3 * Mo * h + qh) * (long)SW + ow;
        for (int dh = 0; dh < 3; ++dh)
            for (int co = 0; co < Mo; ++co) {
                float acc = 0.f;
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const float* zp = zn + ((dh * 3 + dw) * Mo + co) * hw;
                    acc += (1.0f - l1[dw]) * zp[i0[dw]] + l1[dw] * zp[i1[dw]];
                }
                Fn[(long)(dh * Mo + co) * h * SW] = acc;
            }
    }
}

// y[n][co][oh][ow] = bias[co] + sum_dh A_dh(F[n][dh*Mo + co][:][ow])[oh]
__global__ __launch_bounds__(256) void lift_rows_fwd_kernel(const float* __restrict__ F, const float* __restrict__ bias,
                                                            float* __restrict__ y, long total, int Mo, int h,
                                                            int SH, int SW, AxisFwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int oh = (int)(r % SH);
        const long n = r / SH;
        long o0[3], o1[3];
        float l1[3];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            o0[dh] = (long)t.i0[dh * SH + oh] * SW; o1[dh] = (long)t.i1[dh * SH + oh] * SW; l1[dh] = t.l1[dh * SH + oh];
        }
        const long plane = (long)h * SW;
        const float* Fn = F + n * 3 * Mo * plane + ow;
        float* yn = y + (n * Mo * SH + oh) * (long)SW + ow;
        for (int co = 0; co < Mo; ++co) {
            float acc = bias ? bias[co] : 0.f;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const float* fp = Fn + (dh * Mo + co) * plane;
                acc += (1.0f - l1[dh]) * fp[o0[dh]] + l1[dh] * fp[o1[dh]];
            }
            yn[(long)co * SH * SW] = acc;
        }
    }
}

// dF[n][dh*Mo + co][qh][ow] = sum_{e in CSR_dh[qh]} wt_e dy[n][co][idx_e][ow]
__global__ __launch_bounds__(256) void lift_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dF,
                                                            long total, int Mo, int h, int SH, int SW, AxisBwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ow = (int)(e % SW);
        long r = e / SW;
        const int qh = (int)(r % h);
        const long n = r / h;
        const float* gn = dy + n * Mo * (long)SH * SW + ow;
        float* dn = dF + (n * 3 * Mo * h + qh) * (long)SW + ow;
        for (int dh = 0; dh < 3; ++dh) {
            const int b = t.off[dh * (h + 1) + qh];
            const int cnt = min(t.off[dh * (h + 1) + qh + 1] - b, LIFT_MAXE);
            long of[LIFT_MAXE];
            float wv[LIFT_MAXE];
#pragma unroll
            for (int k = 0; k < LIFT_MAXE; ++k) {
                const bool on = k < cnt;
                of[k] = on ? (long)t.idx[b + k] * SW : 0;
                wv[k] = on ? t.wt[b + k] : 0.f;
            }
            for (int co = 0; co < Mo; ++co) {
                const float* gp = gn + (long)co * SH * SW;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < LIFT_MAXE; ++k) acc += wv[k] * gp[of[k]];
                dn[(long)(dh * Mo + co) * h * SW] = acc;
            }
        }
    }
}

// dz[n][(dh*3 + dw)*Mo + co][qh][qw] = sum_{e in CSR_dw[qw]} wt_e dF[n][dh*Mo + co][qh][idx_e]
__global__ __launch_bounds__(256) void lift_cols_bwd_kernel(const float* __restrict__ dF, float* __restrict__ dz,
                                                            long total, int Mo, int h, int w, int SW, AxisBwd t) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int qw = (int)(e % w);
        long r = e / w;
        const int qh = (int)(r % h);
        const long n = r / h;
        const long hw = (long)h * w;
        const float* fn = dF + (n * 3 * Mo * h + qh) * (long)SW;
        float* zn = dz + n * 9 * Mo * hw + (long)qh * w + qw;
        for (int dw = 0; dw < 3; ++dw) {
            const int b = t.off[dw * (w + 1) + qw];
            const int cnt = min(t.off[dw * (w + 1) + qw + 1] - b, LIFT_MAXE);
            int of[LIFT_MAXE];
            float wv[LIFT_MAXE];
#pragma unroll
            for (int k = 0; k < LIFT_MAXE; ++k) {
                const bool on = k < cnt;
                of[k] = on ? t.idx[b + k] : 0;
                wv[k] = on ? t.wt[b + k] : 0.f;
            }
            for (int dh = 0; dh < 3; ++dh)
                for (int co = 0; co < Mo; ++co) {
                    const float* fp = fn + (long)(dh * Mo + co) * h * SW;
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < LIFT_MAXE; ++k) acc += wv[k] * fp[of[k]];
                    zn[((dh * 3 + dw) * Mo + co) * hw] = acc;
                }
        }
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
    const long t1 = (long)N * h * SW, t2 = (long)N * SH * SW;           // one thread per pixel position
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
    const long t1 = (long)N * h * SW, t2 = (long)N * h * w;
    AxisBwd trw = {roff, ridx, rwt}, tc = {coff, cidx, cwt};
    hipLaunchKernelGGL(lift_rows_bwd_kernel, dim3(og_stream_grid(t1, 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, scratch, t1, Mo, h, SH, SW, trw);
    hipLaunchKernelGGL(lift_cols_bwd_kernel, dim3(og_stream_grid(t2, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)scratch, dz, t2, Mo, h, w, SW, tc);
    return og_launch_status();
}

}  // extern "C"
