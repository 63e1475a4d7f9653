// fp16x2 on pre-split records: the pixel operand of the implicit-GEMM convolution as fp16 pieces in HBM.
//
// The fp16x2 arithmetic (conv_igemm3.h, math 4) computes fp32 convolutions as three fp16 MFMA products of
// x * 2^s = h + l.  Math 4 gathers the fp32 NCHW source (eight channel-strided dwords per lane and K step) and splits
// it on the VALU inside the loop; the loop is bound by exactly that operand path (profiles/r04_roofline_table.md: MFMA
// pipe 30 % busy, HBM 0.17).  Here the split happens ONCE per tensor: a record
//
//     rec[n][c / 16][piece h | l][pixel][c % 16]   fp16,   x[n][c][pixel] * 2^s = h + l    (channels C..Cp-1: 0)
//
// costs the 4 bytes per element of the fp32 tensor, and the kernel (conv_igemm3_kernel<.., 5, NW, NG>) reads the eight k
// of a lane as two 16-byte loads; the 32 pixels of a wave read 1 KiB of contiguous memory per instruction.  The scale
// exponent s comes from the tensor's 1 024 partial maxima (common.h), the same slots the consuming kernel undoes the
// scale with: a record is a pure function of (x, maxima), and math 5 reproduces math 4 bit for bit.
//
// Records are written by objgan_h2_records (one pass: 4 B read + 4 B written per element) or by the producer of the
// tensor itself (norm.hip apply kernels).  Reference: the convolutions of image_generation/model.py:30-81, 589-617,
// 986-1048, 1184-1312, which the reference hands to cuDNN.
#include "conv_igemm3.h"

// fp32 [N][C][HW] -> records [N][Cp/16][2][HW][16]; 64 channels x 64 pixels per workgroup through LDS
// (256-byte rows in, 1 KiB runs per wave and piece out).
__global__ __launch_bounds__(256) void h2_records_kernel(const float* __restrict__ x, const float* __restrict__ xmax,
                                                         _Float16* __restrict__ out, int C, int HW, int Cp) {
    __shared__ float tile[64][65];
    og_fp16_saturate();
    const float xs = og_pow2(og_h2_exponent(xmax, threadIdx.x & 63));
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* xn = x + (size_t)n * C * HW;
#pragma unroll 4
    for (int cc = ty; cc < 64; cc += 4) {
        const int c = c0 + cc, p = p0 + tx;
        tile[cc][tx] = (c < C && p < HW) ? xn[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    const int cg = threadIdx.x >> 5;                   // 8 channels = one 16-byte store per piece
    if (c0 + cg * 8 >= Cp) return;
    const int chunk = (c0 + cg * 8) >> 4, half = cg & 1;
    _Float16* oh = out + ((size_t)n * (Cp / 16) + chunk) * 2 * (size_t)HW * 16 + half * 8;
    _Float16* ol = oh + (size_t)HW * 16;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pp = (threadIdx.x & 31) + 32 * it;
        const int p = p0 + pp;
        if (p >= HW) continue;
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = tile[cg * 8 + j][pp] * xs;
            h[j] = (_Float16)sv;
            l[j] = (_Float16)og_sub(sv, (float)h[j]);
        }
        *reinterpret_cast<f16x8*>(oh + (size_t)p * 16) = h;
        *reinterpret_cast<f16x8*>(ol + (size_t)p * 16) = l;
    }
}

// launch of the record-reading instances (called by run_igemm2 in conv_igemm.hip)
int og_launch_igemm3_rec(const IgemmArgs& a, int TM, int nw, int ng, dim3 grid, hipStream_t s) {
#define OG_REC(TMv, NGv)                                                                                              \
        if (nw == 8) hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 5, 8, NGv>), grid, dim3(512), 0, s, a);       \
        else hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 5, 4, NGv>), grid, dim3(256), 0, s, a);
    if (ng == 2) {
        switch (TM) {
            case 1: OG_REC(1, 2) break;
            case 2: OG_REC(2, 2) break;
            case 3: OG_REC(3, 2) break;
            case 4: OG_REC(4, 2) break;
            default: return OG_BAD_ARGS;
        }
        return og_launch_status();
    }
    switch (TM) {
        case 1: OG_REC(1, 1) break;
        case 2: OG_REC(2, 1) break;
        case 3: OG_REC(3, 1) break;
        case 4: OG_REC(4, 1) break;
        case 5: OG_REC(5, 1) break;
        case 6: OG_REC(6, 1) break;
        default: OG_REC(7, 1) break;
    }
#undef OG_REC
    return og_launch_status();
}

extern "C" {

// floats (4-byte units) of the record of an [N, C, H*W] tensor
long objgan_h2_records_floats(int N, int C, long HW) {
    return (long)N * (((long)C + 15) / 16 * 16) * HW;
}

// rec <- the fp16x2 record of x [N, C, HW] (16-byte aligned) under the scale of its partial maxima xmax[1024]
// (objgan_absmax_partials or a producer).  rec: objgan_h2_records_floats(N, C, HW) floats, 16-byte aligned.
int objgan_h2_records(const float* x, const float* xmax, void* rec, int N, int C, long HW, void* stream) {
    OG_ENTRY();
    if (!x || !xmax || !rec || ((size_t)rec & 15)) return OG_BAD_ARGS;
    if (N <= 0 || C <= 0 || HW <= 0) return OG_OK;
    const int Cp = (C + 15) / 16 * 16;
    if ((double)N * Cp * (double)HW * 4.0 >= 4.0e9 || N > 65535) return OG_BAD_ARGS;
    hipLaunchKernelGGL(h2_records_kernel, dim3(og_cdiv(HW, 64), og_cdiv(Cp, 64), N), dim3(256), 0, (hipStream_t)stream,
                       x, xmax, reinterpret_cast<_Float16*>(rec), C, (int)HW, Cp);
    return og_launch_status();
}

}  // extern "C"
