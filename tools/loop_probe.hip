// Main-loop laboratory for the bf16x3 implicit-GEMM convolution kernels (development aid, compiles in seconds).
//
// A GEMM with the operand layouts and access pattern of conv_igemm3_kernel<TM, false, 2, NW>, minus the convolution
// geometry: C[M][N] = sum_k A[m][k] * B[k][n], A = pre-split filter bank [M][K/16][h,m,l][16] bf16 (96 bytes per row and
// 16-deep K step), B = fp32 activations [Cb][N] (pixel-contiguous planes like NCHW), K = T * Cb tap-major with a pixel
// shift per tap (so the T taps re-read the same planes through L1 / L2 like a convolution does).  Variants of the
// software pipeline are timed against each other here before the winner goes into csrc/conv_igemm.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/loop_probe tools/loop_probe.hip && tools/loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define OG_BUF_FLAGS 0x00020000
#define OG_OOB 0x7ffffff0u
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

struct Args {
    const float* B; const __bf16* Bs; const __bf16* A; float* C;
    int M, N, Cb, T, W;      // K = T * Cb; tap t shifts the pixel by ((t & 3) - 1) + ((t >> 2) - 1) * W
};

__device__ __forceinline__ float og_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void split_h(const float* v, bf16x8& h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[j];
}
__device__ __forceinline__ void split_ml(const float* v, const bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r1 = og_sub(v[j], (float)h[j]);
        const __bf16 mj = (__bf16)r1;
        m[j] = mj; l[j] = (__bf16)og_sub(r1, (float)mj);
    }
}
template <int PAIRS, int VALU>
__device__ __forceinline__ void interleave() {
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
    }
}
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, j = id >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + j;
}

// ABL (compile-time ablation, timing only): 1 no pixel gathers in the loop, 2 no split VALU (m = l = h), 4 no row-tile
// loads / LDS stores in the loop, 8 no barriers, 16 no LDS fragment reads in the loop, 32 only 3 of the 6 products.
// NG: 32-pixel groups per wave (every row fragment feeds NG MFMAs; the workgroup tile is 32 TM x 32 NW NG).
// BSRC 0: fp32 [Cb][N] planes, eight dword gathers per lane, step and group, split in registers (round 3);
//      1: pre-split channel-blocked copy [Cb/16][N][h16|m16|l16] bf16 (96 bytes per pixel and chunk): three 16-byte loads.
// BK2 1: two 16-deep K steps per LDS row-tile stage and barrier.
template <int TM, int NW, int NG, int BSRC, int ABL = 0>
__global__ __launch_bounds__(64 * NW) void loop_kernel(const Args a) {
    constexpr int NT = 64 * NW, BM = 32 * TM, BN = 32 * NW * NG, LD = 28, PIECES = 6, ABYTES = 96;
    constexpr int NA4 = BM * PIECES, NA_PER = (NA4 + NT - 1) / NT, TILE = BM * LD;
    constexpr int NBR = BSRC ? 12 : 8;                  // registers per pixel fragment
    __shared__ __attribute__((aligned(16))) float lds[3 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tiles_m = a.M / BM, tiles_n = a.N / BN, nwg = tiles_m * tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg % tiles_m, tile_n = wg / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = a.T * a.Cb, Krow = 3 * K;
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        BSRC ? (void*)a.Bs : (void*)a.B, 0, BSRC ? (int)((unsigned)a.Cb / 16 * a.N * 96u) : (int)((unsigned)a.Cb * a.N * 4u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)((unsigned)a.M * Krow * 2u), OG_BUF_FLAGS);
    const int pix = n0 + wid * 32 * NG + lcol;
    const int N4 = a.N * 4;
    int t_ld = 0, cb_ld = 0;
    unsigned bvoff[NG];
    auto tap_geometry = [&](int t) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int p = pix + 32 * g + ((t & 3) - 1) + ((t >> 2) - 1) * a.W;
            if (BSRC) bvoff[g] = ((unsigned)p < (unsigned)a.N) ? (unsigned)p * 96u + (unsigned)(lrow * 16) : OG_OOB;
            else bvoff[g] = ((unsigned)p < (unsigned)a.N) ? (unsigned)(lrow * 8) * (unsigned)N4 + (unsigned)p * 4u : OG_OOB;
        }
    };
    tap_geometry(0);
    auto load_b = [&](float (&rb)[NG][NBR]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (BSRC) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[g] + q * 32u, (cb_ld >> 4) * a.N * 96, 0));
#pragma unroll
                    for (int i = 0; i < 4; ++i) rb[g][q * 4 + i] = v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    rb[g][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, bvoff[g], (cb_ld + i) * N4, 0));
            }
        }
        cb_ld += 16;
        if (cb_ld >= a.Cb) { cb_ld = 0; t_ld += 1; if (t_ld < a.T) tap_geometry(t_ld); }
    };
    unsigned avoff[NA_PER]; int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + NT * i;
        const int row = idx / PIECES, q = idx - row * PIECES;
        const bool on = (NA4 % NT == 0 || idx < NA4);
        avoff[i] = on ? (unsigned)(m0 + row) * (unsigned)Krow * 2u + q * 16u : OG_OOB;
        alds[i] = on ? row * LD + q * 4 : -1;
    }
    f32x4 ra[NA_PER];
    auto load_a = [&](int kt) {
        const int so = __builtin_amdgcn_readfirstlane(kt * ABYTES);
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, avoff[i], so, 0));
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * TILE;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
    };
    const int nk = K / 16;
    f32x16 acc[TM][NG];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.f;
    float rb0[NG][NBR], rb1[NG][NBR], rb2[NG][NBR];
    bf16x8 ah[TM], am[TM], al[TM];
    const int frag_off = lcol * (LD * 4) + lrow * 16;

    auto mma = [&](const float (&rb)[NG][NBR], int cur, auto&& mid) {
        bf16x8 bh[NG], bm[NG], bl[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (BSRC) {
                f32x4 v0, v1, v2;
#pragma unroll
                for (int i = 0; i < 4; ++i) { v0[i] = rb[g][i]; v1[i] = rb[g][4 + i]; v2[i] = rb[g][8 + i]; }
                bh[g] = __builtin_bit_cast(bf16x8, v0); bm[g] = __builtin_bit_cast(bf16x8, v1); bl[g] = __builtin_bit_cast(bf16x8, v2);
            } else {
                split_h(rb[g], bh[g]);
            }
        }
        const char* T = reinterpret_cast<const char*>(lds + cur * TILE) + frag_off;
        if (!(ABL & 16)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                al[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 64);
                am[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 32);
                ah[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(al[i], bh[g], acc[i][g]);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        if (!BSRC) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (ABL & 2) { bm[g] = bh[g]; bl[g] = bh[g]; } else split_ml(rb[g], bh[g], bm[g], bl[g]);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(am[i], bh[g], acc[i][g]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(ah[i], bh[g], acc[i][g]);
        if (!BSRC && !(ABL & 2)) interleave<2 * TM * NG, (40 * NG + 2 * TM * NG - 1) / (2 * TM * NG)>();
        if (ABL & 32) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(am[i], bm[g], acc[i][g]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(ah[i], bm[g], acc[i][g]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < NG; ++g) MFMA(ah[i], bl[g], acc[i][g]);
    };
    load_a(0); store_a(0); load_a(1);
    load_b(rb0); load_b(rb1);
    __syncthreads();
    if (ABL & 16) {
        const char* T = reinterpret_cast<const char*>(lds) + frag_off;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            al[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 64);
            am[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 32);
            ah[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4);
        }
    }
    if (ABL & 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < NBR; ++j) rb2[g][j] = rb0[g][j] + rb1[g][j];
    }
    int ks = 0;
    auto la = [&](int kt) { if (!(ABL & 4)) load_a(kt); };
    auto sa = [&](int b) { if (!(ABL & 4)) store_a(b); };
    auto lb = [&](float (&r)[NG][NBR]) { if (!(ABL & 1)) load_b(r); };
    auto sync = [&]() { if (!(ABL & 8)) __syncthreads(); };
    if (ks + 2 < nk) {
        do {
            mma(rb0, 0, [&]() { sa(1); la(ks + 2); lb(rb2); });
            sync();
            mma(rb1, 1, [&]() { sa(2); la(ks + 3); lb(rb0); });
            sync();
            mma(rb2, 2, [&]() { sa(0); la(ks + 4); lb(rb1); });
            sync();
            ks += 3;
        } while (ks + 2 < nk);
    }
    if (ks < nk) { mma(rb0, 0, [&]() { store_a(1); }); __syncthreads(); }
    if (ks + 1 < nk) mma(rb1, 1, [] {});

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float* cb = a.C + pix + 32 * g;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                cb[(size_t)m * a.N] = acc[i][g][r];
            }
    }
}


// ---- fp16x2: fp32 operands as TWO fp16 pieces (x * 2^s = h + l, |residual| <= 2^-24 |x|), three products hh, hl, lh on
// v_mfma_f32_32x32x16_f16 (same rate as the bf16 MFMA), fp32 accumulation, exact power-of-two scales undone in the epilogue.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMAH(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
struct Args16 { const float* B; const _Float16* A; float* C; int M, N, Cb, T, W; float sb, inv; const _Float16* Bs; };

// BSRC 1: the pixel operand arrives PRE-SPLIT and channel-blocked by 8, [Cb/8][h | l][N][8] fp16 (x * 2^s = h + l done by the
// producer): per lane and K step two 16-byte loads, the 32 pixels of a half-wave read 512 contiguous bytes, no VALU work.
// NG: 32-pixel groups per wave (every row fragment read from LDS feeds NG MFMAs).
template <int TM, int NW, int NG = 1, int BSRC = 0>
__global__ __launch_bounds__(64 * NW) void loop16_kernel(const Args16 a) {
    constexpr int NT = 64 * NW, BM = 32 * TM, BN = 32 * NW * NG, LD = 20, PIECES = 4, ABYTES = 64;
    constexpr int NA4 = BM * PIECES, NA_PER = (NA4 + NT - 1) / NT, TILE = BM * LD;
    __shared__ __attribute__((aligned(16))) float lds[3 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tiles_m = a.M / BM, tiles_n = a.N / BN, nwg = tiles_m * tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg % tiles_m, tile_n = wg / tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = a.T * a.Cb, Krow = 2 * K;
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(BSRC ? (void*)a.Bs : (void*)a.B, 0, (int)((unsigned)a.Cb * a.N * 4u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)((unsigned)a.M * Krow * 2u), OG_BUF_FLAGS);
    const int pix = n0 + wid * 32 * NG + lcol;
    const int N4 = a.N * 4, N16 = a.N * 16;
    int t_ld = 0, cb_ld = 0;
    unsigned bvoff[NG];
    auto tap_geometry = [&](int t) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int p = pix + g * 32 + ((t & 3) - 1) + ((t >> 2) - 1) * a.W;
            if (BSRC) bvoff[g] = ((unsigned)p < (unsigned)a.N) ? (unsigned)(lrow * 2) * (unsigned)N16 + (unsigned)p * 16u : OG_OOB;
            else bvoff[g] = ((unsigned)p < (unsigned)a.N) ? (unsigned)(lrow * 8) * (unsigned)N4 + (unsigned)p * 4u : OG_OOB;
        }
    };
    tap_geometry(0);
    struct BReg { f32x4 v[2]; };          // 8 fp32 values, or the h and l fragments (8 fp16 each)
    auto load_b = [&](BReg (&rb)[NG]) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (BSRC) {
                const int so = __builtin_amdgcn_readfirstlane((cb_ld >> 3) * 2 * N16);
                rb[g].v[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[g], so, 0));
                rb[g].v[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[g], so + N16, 0));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    rb[g].v[i >> 2][i & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, bvoff[g], (cb_ld + i) * N4, 0));
            }
        }
        cb_ld += 16;
        if (cb_ld >= a.Cb) { cb_ld = 0; t_ld += 1; if (t_ld < a.T) tap_geometry(t_ld); }
    };
    unsigned avoff[NA_PER]; int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + NT * i;
        const int row = idx / PIECES, q = idx - row * PIECES;
        const bool on = (NA4 % NT == 0 || idx < NA4);
        avoff[i] = on ? (unsigned)(m0 + row) * (unsigned)Krow * 2u + q * 16u : OG_OOB;
        alds[i] = on ? row * LD + q * 4 : -1;
    }
    f32x4 ra[NA_PER];
    auto load_a = [&](int kt) {
        const int so = __builtin_amdgcn_readfirstlane(kt * ABYTES);
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, avoff[i], so, 0));
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * TILE;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
    };
    const int nk = K / 16;
    f32x16 acc[NG][TM];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][i][r] = 0.f;
    BReg rb0[NG], rb1[NG], rb2[NG];
    f16x8 ah[TM], al[TM];
    const int frag_off = lcol * (LD * 4) + lrow * 16;
    const float sb = a.sb;
    auto mma = [&](const BReg (&rb)[NG], int cur, auto&& mid) {
        f16x8 bh[NG], bl[NG];
        float sc[NG][8];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (BSRC) { bh[g] = __builtin_bit_cast(f16x8, rb[g].v[0]); bl[g] = __builtin_bit_cast(f16x8, rb[g].v[1]); }
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[g][j] = rb[g].v[j >> 2][j & 3] * sb; bh[g][j] = (_Float16)sc[g][j]; }
            }
        }
        const char* T = reinterpret_cast<const char*>(lds + cur * TILE) + frag_off;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            al[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4 + 32);
            ah[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i) MFMAH(al[i], bh[g], acc[g][i]);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        if (!BSRC) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < 8; ++j) bl[g][j] = (_Float16)og_sub(sc[g][j], (float)bh[g][j]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i) MFMAH(ah[i], bh[g], acc[g][i]);
        if (!BSRC) interleave<TM, (16 + TM - 1) / TM>();
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i) MFMAH(ah[i], bl[g], acc[g][i]);
    };
    load_a(0); store_a(0); load_a(1);
    load_b(rb0); load_b(rb1);
    __syncthreads();
    int ks = 0;
    if (ks + 2 < nk) {
        do {
            mma(rb0, 0, [&]() { store_a(1); load_a(ks + 2); load_b(rb2); });
            __syncthreads();
            mma(rb1, 1, [&]() { store_a(2); load_a(ks + 3); load_b(rb0); });
            __syncthreads();
            mma(rb2, 2, [&]() { store_a(0); load_a(ks + 4); load_b(rb1); });
            __syncthreads();
            ks += 3;
        } while (ks + 2 < nk);
    }
    if (ks < nk) { mma(rb0, 0, [&]() { store_a(1); }); __syncthreads(); }
    if (ks + 1 < nk) mma(rb1, 1, [] {});
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float* cb = a.C + pix + g * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                cb[(size_t)m * a.N] = acc[g][i][r] * a.inv;
            }
    }
}

static unsigned g_seed = 12345u;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffffff) / 8388608.0f - 1.0f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int TM, int NW, int NG, int BSRC, int ABL = 0>
static void run(const char* name, const Args& a, const std::vector<float>& hA, const std::vector<float>& hB, int iters) {
    const int blocks = (a.M / (32 * TM)) * (a.N / (32 * NW * NG));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(a.C, 0xff, (size_t)a.M * a.N * 4));
    hipLaunchKernelGGL((loop_kernel<TM, NW, NG, BSRC, ABL>), dim3(blocks), dim3(64 * NW), 0, 0, a);
    CK(hipDeviceSynchronize());
    // check a sample against fp64
    std::vector<float> hC((size_t)a.M * a.N);
    CK(hipMemcpy(hC.data(), a.C, hC.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    const int K = a.T * a.Cb;
    for (int s = 0; s < 200; ++s) {
        const int m = (s * 37 + 5) % a.M, n = (int)(((long)s * 104729 + 77) % a.N);
        double ref = 0;
        for (int t = 0; t < a.T; ++t) {
            const int p = n + ((t & 3) - 1) + ((t >> 2) - 1) * a.W;
            if (p < 0 || p >= a.N) continue;
            for (int c = 0; c < a.Cb; ++c) ref += (double)hA[(size_t)m * K + t * a.Cb + c] * (double)hB[(size_t)c * a.N + p];
        }
        num += (ref - hC[(size_t)m * a.N + n]) * (ref - hC[(size_t)m * a.N + n]); den += ref * ref;
    }
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < iters; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((loop_kernel<TM, NW, NG, BSRC, ABL>), dim3(blocks), dim3(64 * NW), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { best = std::min(best, ms); sum += ms; }
    }
    const double fl = 2.0 * a.M * (double)K * a.N;
    printf("%-34s TM %d NW %d NG %d BSRC %d ABL %2d  avg %7.3f ms %6.1f TF  best %6.1f TF  rel-l2 err %.2e\n", name, TM, NW, NG, BSRC, ABL, sum / (iters - 2),
           fl / (sum / (iters - 2) * 1e-3) * 1e-12, fl / (best * 1e-3) * 1e-12, sqrt(num / den));
    fflush(stdout);
}

template <int TM, int NW, int NG = 1, int BSRC = 0>
static void run16(const char* name, const Args16& a, const std::vector<float>& hA, const std::vector<float>& hB, int iters) {
    const int blocks = (a.M / (32 * TM)) * (a.N / (32 * NW * NG));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(a.C, 0xff, (size_t)a.M * a.N * 4));
    hipLaunchKernelGGL((loop16_kernel<TM, NW, NG, BSRC>), dim3(blocks), dim3(64 * NW), 0, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)a.M * a.N);
    CK(hipMemcpy(hC.data(), a.C, hC.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    const int K = a.T * a.Cb;
    for (int s = 0; s < 200; ++s) {
        const int m = (s * 37 + 5) % a.M, n = (int)(((long)s * 104729 + 77) % a.N);
        double ref = 0;
        for (int t = 0; t < a.T; ++t) {
            const int p = n + ((t & 3) - 1) + ((t >> 2) - 1) * a.W;
            if (p < 0 || p >= a.N) continue;
            for (int c = 0; c < a.Cb; ++c) ref += (double)hA[(size_t)m * K + t * a.Cb + c] * (double)hB[(size_t)c * a.N + p];
        }
        num += (ref - hC[(size_t)m * a.N + n]) * (ref - hC[(size_t)m * a.N + n]); den += ref * ref;
    }
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < iters; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((loop16_kernel<TM, NW, NG, BSRC>), dim3(blocks), dim3(64 * NW), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { best = std::min(best, ms); sum += ms; }
    }
    const double fl = 2.0 * a.M * (double)K * a.N;
    printf("%-34s TM %d NW %d NG %d BSRC %d fp16x2  avg %7.3f ms %6.1f TF  best %6.1f TF  rel-l2 err %.2e\n", name, TM, NW, NG, BSRC, sum / (iters - 2),
           fl / (sum / (iters - 2) * 1e-3) * 1e-12, fl / (best * 1e-3) * 1e-12, sqrt(num / den));
    fflush(stdout);
}

int main(int argc, char** argv) {
    Args a;
    a.M = 192; a.Cb = 192; a.T = 16; a.W = 130; a.N = 16 * 64 * 64;
    const int K = a.T * a.Cb;
    std::vector<float> hA((size_t)a.M * K), hB((size_t)a.Cb * a.N);
    for (auto& v : hA) v = frand() * 0.05f;
    for (auto& v : hB) v = frand();
    std::vector<unsigned short> bank((size_t)a.M * 3 * K);
    for (int m = 0; m < a.M; ++m)
        for (int k = 0; k < K; ++k) {
            const float v = hA[(size_t)m * K + k];
            const unsigned short h = f2bf(v); const float r1 = v - bf2f(h);
            const unsigned short mm = f2bf(r1); const unsigned short l = f2bf(r1 - bf2f(mm));
            unsigned short* o = &bank[(size_t)m * 3 * K + (size_t)(k >> 4) * 48 + (k & 15)];
            o[0] = h; o[16] = mm; o[32] = l;
        }
    std::vector<unsigned short> hBs((size_t)a.Cb / 16 * a.N * 48);
    for (int c = 0; c < a.Cb; ++c)
        for (int n = 0; n < a.N; ++n) {
            const float v = hB[(size_t)c * a.N + n];
            const unsigned short h = f2bf(v); const float r1 = v - bf2f(h);
            const unsigned short mm = f2bf(r1); const unsigned short l = f2bf(r1 - bf2f(mm));
            unsigned short* o = &hBs[((size_t)(c >> 4) * a.N + n) * 48 + (c & 15)];
            o[0] = h; o[16] = mm; o[32] = l;
        }
    __bf16* dBs; CK(hipMalloc(&dBs, hBs.size() * 2)); CK(hipMemcpy(dBs, hBs.data(), hBs.size() * 2, hipMemcpyHostToDevice));
    a.Bs = dBs;
    float* dB; __bf16* dA; float* dC;
    CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dA, bank.size() * 2)); CK(hipMalloc(&dC, (size_t)384 * a.N * 4));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA, bank.data(), bank.size() * 2, hipMemcpyHostToDevice));
    a.A = dA; a.B = dB; a.C = dC;
    // fp16x2 bank: w * 2^18 = h + l in fp16 (max |w| = 0.05 -> 13107), activations scaled by 2^13 in the kernel
    std::vector<_Float16> bank16((size_t)a.M * 2 * K);
    for (int m = 0; m < a.M; ++m)
        for (int k = 0; k < K; ++k) {
            const float v = hA[(size_t)m * K + k] * 262144.0f;
            const _Float16 h = (_Float16)v; const _Float16 l = (_Float16)(v - (float)h);
            _Float16* o = &bank16[(size_t)m * 2 * K + (size_t)(k >> 4) * 32 + (k & 15)];
            o[0] = h; o[16] = l;
        }
    _Float16* dA16; CK(hipMalloc(&dA16, bank16.size() * 2)); CK(hipMemcpy(dA16, bank16.data(), bank16.size() * 2, hipMemcpyHostToDevice));
    Args16 a16; a16.B = dB; a16.A = dA16; a16.C = dC; a16.M = a.M; a16.N = a.N; a16.Cb = a.Cb; a16.T = a.T; a16.W = a.W;
    a16.sb = 8192.0f; a16.inv = 1.0f / (8192.0f * 262144.0f);
    // pre-split pixel operand [Cb/8][h | l][N][8]: x * 2^13 = h + l
    std::vector<_Float16> hBs16((size_t)a.Cb * a.N * 2);
    for (int c = 0; c < a.Cb; ++c)
        for (int n = 0; n < a.N; ++n) {
            const float v = hB[(size_t)c * a.N + n] * 8192.0f;
            const _Float16 h = (_Float16)v; const _Float16 l = (_Float16)(v - (float)h);
            hBs16[(((size_t)(c >> 3) * 2 + 0) * a.N + n) * 8 + (c & 7)] = h;
            hBs16[(((size_t)(c >> 3) * 2 + 1) * a.N + n) * 8 + (c & 7)] = l;
        }
    _Float16* dBs16; CK(hipMalloc(&dBs16, hBs16.size() * 2)); CK(hipMemcpy(dBs16, hBs16.data(), hBs16.size() * 2, hipMemcpyHostToDevice));
    a16.Bs = dBs16;
    const int iters = argc > 1 ? atoi(argv[1]) : 12;
    for (int pass = 0; pass < 2; ++pass) {
        run<6, 8, 1, 0>("bf16x3 (6 MFMAs)", a, hA, hB, iters);
        run16<6, 8>("fp16x2", a16, hA, hB, iters);
        run<3, 4, 1, 0>("bf16x3 (6 MFMAs)", a, hA, hB, iters);
        run16<3, 4>("fp16x2", a16, hA, hB, iters);
        run16<3, 8>("fp16x2", a16, hA, hB, iters);
        run16<6, 4>("fp16x2", a16, hA, hB, iters);
        run16<6, 8, 1, 1>("fp16x2 pre-split B", a16, hA, hB, iters);
        run16<3, 4, 1, 1>("fp16x2 pre-split B", a16, hA, hB, iters);
        run16<3, 8, 1, 1>("fp16x2 pre-split B", a16, hA, hB, iters);
        run16<6, 4, 1, 1>("fp16x2 pre-split B", a16, hA, hB, iters);
        run16<3, 4, 2, 0>("fp16x2 2 pixel groups", a16, hA, hB, iters);
        run16<3, 4, 2, 1>("fp16x2 2 groups, pre-split B", a16, hA, hB, iters);
        run16<3, 8, 2, 1>("fp16x2 2 groups, pre-split B", a16, hA, hB, iters);
        run16<6, 4, 2, 1>("fp16x2 2 groups, pre-split B", a16, hA, hB, iters);
        run16<2, 4, 2, 1>("fp16x2 2 groups, pre-split B", a16, hA, hB, iters);
        run16<1, 4, 2, 1>("fp16x2 2 groups, pre-split B", a16, hA, hB, iters);
        run16<1, 4, 1, 0>("fp16x2", a16, hA, hB, iters);
        run16<1, 4, 1, 1>("fp16x2 pre-split B", a16, hA, hB, iters);
    }
    return 0;
}
