// Winograd F(2x2, 3x3) data transforms (EXPERIMENTAL, opt-in from the host side: OBJGAN_WINOGRAD=1).
//
// The residual blocks of the generator (reference image_generation/model.py:63-81: reflect-pad +
// 3x3 conv 194 -> 388, 194 -> 194 at 32^2 / 64^2 / 128^2) are the largest bucket of the step and run
// on the exact fp32 MFMA path, i.e. against the 157 TFLOP/s fp32 matrix roof.  F(2x2, 3x3)
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A
// needs 16 multiplies per 2x2 outputs instead of 36: the 16 element-wise products become 16
// independent [M x C] x [C x tiles] GEMMs, which are 1x1 convolutions of the transformed input
// V[xi] (an NCHW tensor of the tiles) -- they run on the existing implicit-GEMM kernel unchanged.
// This file holds the two HBM-bound ends:
//   wino_input:   V[xi][n][c][tile] = (B^T d B)[xi]   of the 4x4 patch of tile (ty, tx), patches
//                 overlap by two pixels; zero or reflection padding is applied while gathering
//   wino_output:  y[n][m][2ty + a][2tx + b] = (A^T Mt A)[a][b]  from the 16 GEMM outputs Mt[xi]
// fp32 throughout; the transforms only add and subtract (B^T, A^T hold 0 / +-1), so the only
// rounding beyond the direct form is the re-association inside the 16 products (1e-6 relative).
#include "common.h"

// V layout: [16][N][C][TH*TW]   (every xi slice is a contiguous NCHW tensor [N, C, TH, TW])
__global__ __launch_bounds__(256) void wino_input_f23_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                             int N, int C, int H, int W, int TH, int TW,
                                                             int pad, int refl) {
    const long tiles = (long)TH * TW;
    const long total = (long)N * C * tiles;
    const long slice = total;                               // elements of one xi slice
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int tx = (int)(e % TW);
        const int ty = (int)((e / TW) % TH);
        const long plane = e / tiles;                       // n * C + c
        const float* xp = x + plane * (long)H * W;
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ih = 2 * ty + i - pad;
            bool rok = true;
            if (refl) {
                ih = ih < 0 ? -ih : ih;
                ih = ih >= H ? 2 * (H - 1) - ih : ih;
            } else {
                rok = (unsigned)ih < (unsigned)H;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int iw = 2 * tx + j - pad;
                bool ok = rok;
                if (refl) {
                    iw = iw < 0 ? -iw : iw;
                    iw = iw >= W ? 2 * (W - 1) - iw : iw;
                } else {
                    ok = ok && (unsigned)iw < (unsigned)W;
                }
                d[i][j] = ok ? xp[(long)ih * W + iw] : 0.f;
            }
        }
        // t = B^T d   (rows), v = t B   (columns);  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = d[0][j] - d[2][j];
            t[1][j] = d[1][j] + d[2][j];
            t[2][j] = d[2][j] - d[1][j];
            t[3][j] = d[1][j] - d[3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* vo = V + (long)(4 * i) * slice + e;
            vo[0] = t[i][0] - t[i][2];
            vo[slice] = t[i][1] + t[i][2];
            vo[2 * slice] = t[i][2] - t[i][1];
            vo[3 * slice] = t[i][1] - t[i][3];
        }
    }
}

// Mt layout: [16][N][M][TH*TW];  y: [N][M][OH][OW] with OH = 2*TH, OW = 2*TW
__global__ __launch_bounds__(256) void wino_output_f23_kernel(const float* __restrict__ Mt, float* __restrict__ y,
                                                              int N, int M, int TH, int TW) {
    const long tiles = (long)TH * TW;
    const long total = (long)N * M * tiles;
    const long slice = total;
    const int OW = 2 * TW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int tx = (int)(e % TW);
        const int ty = (int)((e / TW) % TH);
        const long plane = e / tiles;                       // n * M + m
        float m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = Mt[(long)(4 * i + j) * slice + e];
        // s = A^T m  (2 x 4),  out = s A  (2 x 2);  A^T = [1 1 1 0; 0 1 -1 -1]
        float s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = m[0][j] + m[1][j] + m[2][j];
            s[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
        float* yp = y + plane * (long)(2 * TH) * OW + (long)(2 * ty) * OW + 2 * tx;
        float2 r0, r1;
        r0.x = s[0][0] + s[0][1] + s[0][2];
        r0.y = s[0][1] - s[0][2] - s[0][3];
        r1.x = s[1][0] + s[1][1] + s[1][2];
        r1.y = s[1][1] - s[1][2] - s[1][3];
        *reinterpret_cast<float2*>(yp) = r0;                // 2*tx is even and OW is even: 8-byte aligned
        *reinterpret_cast<float2*>(yp + OW) = r1;
    }
}

extern "C" {

// V must hold 16 * N * C * TH * TW floats.  The patch of tile (ty, tx) starts at input pixel
// (2*ty - pad, 2*tx - pad); refl = 1 reflects out-of-range pixels (ReflectionPad2d), 0 reads zeros.
int objgan_wino_input_f23(const float* x, float* V, int N, int C, int H, int W, int TH, int TW,
                          int pad, int refl, void* stream) {
    OG_ENTRY();
    if (N <= 0 || C <= 0 || TH <= 0 || TW <= 0) return OG_OK;
    if (pad < 0 || (refl && (pad >= H || pad >= W || 2 * TH + 2 - pad > 2 * H - 1 || 2 * TW + 2 - pad > 2 * W - 1)))
        return OG_BAD_ARGS;
    const long total = (long)N * C * TH * TW;
    hipLaunchKernelGGL(wino_input_f23_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, V, N, C, H, W, TH, TW, pad, refl);
    return og_launch_status();
}

// Mt: 16 * N * M * TH * TW floats;  y: N * M * (2*TH) * (2*TW) floats, fully written.
int objgan_wino_output_f23(const float* Mt, float* y, int N, int M, int TH, int TW, void* stream) {
    OG_ENTRY();
    if (N <= 0 || M <= 0 || TH <= 0 || TW <= 0) return OG_OK;
    const long total = (long)N * M * TH * TW;
    hipLaunchKernelGGL(wino_output_f23_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       Mt, y, N, M, TH, TW);
    return og_launch_status();
}

}  // extern "C"
