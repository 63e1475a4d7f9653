// Word / object attention kernels for gfx950.
//
//  A1  attn_general_*   GlobalAttentionGeneral.forward   (reference GlobalAttention.py:73-122)
//  A2  attn_bu_*        GlobalBUAttentionGeneral.forward (reference GlobalAttention.py:125-181)
//  A4  masked_max_*     pprocess_bt_attns                (reference miscc/utils.py:401-413)
//      softmax_strided_* the two softmaxes of func_attention (reference GlobalAttention.py:32-70)
//
// All of them are HBM-bound (about 5 flop/B): one lane owns one query pixel, the 48-channel
// query vector lives in VGPRs, the tiny projected context (idf x L, L <= 32 words) lives in
// LDS and is read as wave-wide broadcasts, the softmax is lane-local (no cross-lane traffic),
// and every global access is coalesced along the pixel dimension.  The reference materialises
// transposes, the masked score matrix and (for pprocess_bt_attns) B x R x C x H x W products;
// none of those exist here.  The one real reduction -- d(sourceT) in the A1 backward, a
// [idf x pixels] x [pixels x L] contraction -- runs on the matrix cores (v_mfma_f32_16x16x4_f32)
// out of LDS-staged tiles, with the accumulators carried across all pixel chunks of a wave.
//
// Reference quirk kept on purpose (trap 1): `mask.repeat(queryL, 1)`
// tiles whole-batch blocks while the score rows are ordered (b, q), so the mask row applied
// to pixel (b, q) is mask[(b * queryL + q) % B], not mask[b].
#include "common.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ======================================================================================
// A1 forward
// ======================================================================================
// x [B, IDF, Q], src [B, IDF, L], mask [B, L] (uint8, may be null)
// wc [B, IDF, Q], attn [B, L, Q]
template <int IDF, int LMAX>
__global__ __launch_bounds__(256) void attn_general_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const unsigned char* __restrict__ mask,
    float* __restrict__ wc, float* __restrict__ attn, int B, int Q, int L) {
    __shared__ float s_src[IDF][LMAX];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < IDF * LMAX; i += blockDim.x) {
        const int c = i / LMAX, l = i - c * LMAX;
        s_src[c][l] = l < L ? src[((size_t)b * IDF + c) * L + l] : 0.f;
    }
    __syncthreads();
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const float* xp = x + (size_t)b * IDF * Q + q;
    float sc[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) sc[l] = 0.f;
    // rolled over channels on purpose: a full unroll makes the compiler hoist all IDF*LMAX
    // LDS operands into registers (and spill); 4 channels per trip keep 4 loads in flight.
#pragma unroll 4
    for (int c = 0; c < IDF; ++c) {
        const float xc = xp[(size_t)c * Q];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) sc[l] = fmaf(xc, s_src[c][l], sc[l]);
    }
    // mask row of the reference's mis-tiled repeat, gathered into one bit per word
    unsigned deadbits = (L >= 32) ? 0u : ~((1u << L) - 1u);         // padded words l >= L
    if (mask) {
        const unsigned char* mrow = mask + (size_t)(((long)b * Q + q) % B) * L;
        for (int l = 0; l < L; ++l) deadbits |= (mrow[l] != 0 ? 1u : 0u) << l;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        sc[l] = ((deadbits >> l) & 1u) ? -INFINITY : sc[l];
        mx = fmaxf(mx, sc[l]);
    }
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        sc[l] = expf(sc[l] - mx);   // exp(-inf) = 0 for masked / padded entries
        sum += sc[l];
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        sc[l] *= inv;
        if (l < L) attn[((size_t)b * L + l) * Q + q] = sc[l];
    }
    float* wp = wc + (size_t)b * IDF * Q + q;
#pragma unroll 4
    for (int c = 0; c < IDF; ++c) {
        float o = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) o = fmaf(s_src[c][l], sc[l], o);
        wp[(size_t)c * Q] = o;
    }
}

// ======================================================================================
// A1 backward
// ======================================================================================
// dx [B, IDF, Q] written; every wave stores its partial dsrc tile [IDF][16] into ws[b][wave] and
// attn_dsrc_combine_kernel sums the waves in order into dsrc [B, IDF, L] (no atomics, no zero-fill).
// A wave walks `chunks` groups of 64 pixels; per group: lane-local softmax backward, then
// dsrc += [x | dwc] (IDF x 128) * [ds ; attn] (128 x L) on v_mfma_f32_16x16x4_f32.
template <int IDF, int LMAX>
__global__ __launch_bounds__(256) void attn_general_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const float* __restrict__ attn,
    const float* __restrict__ dwc, const float* __restrict__ dattn,
    float* __restrict__ dx, float* __restrict__ ws, int B, int Q, int L, int chunks) {
    static_assert(IDF % 16 == 0 && LMAX == 16, "tile shape");
    constexpr int TM = IDF / 16;
    constexpr int KQ = 64;           // pixels per contraction pass (x*ds, then dwc*attn)
    constexpr int LDA = KQ + 1;
    __shared__ float s_src[IDF][LMAX];
    __shared__ float s_a[4][IDF][LDA];      // per wave: A[c][k]
    __shared__ float s_b[4][KQ][LMAX + 1];  // per wave: B[k][l]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < IDF * LMAX; i += blockDim.x) {
        const int c = i / LMAX, l = i - c * LMAX;
        s_src[c][l] = l < L ? src[((size_t)b * IDF + c) * L + l] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    f32x4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wave_global = blockIdx.x * 4 + wid;
    for (int ch = 0; ch < chunks; ++ch) {
        const int q0 = (wave_global * chunks + ch) * 64;
        if (q0 >= Q) break;                      // wave-uniform
        const int q = q0 + lane;
        const bool ok = q < Q;
        const int qq = ok ? q : q0;
        float at[LMAX], da[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            at[l] = (ok && l < L) ? attn[((size_t)b * L + l) * Q + qq] : 0.f;
            da[l] = (ok && dattn && l < L) ? dattn[((size_t)b * L + l) * Q + qq] : 0.f;
        }
        // pass A operand 1: dwc -> this wave's LDS slab, and da += src^T dwc on the fly
#pragma unroll 4
        for (int c = 0; c < IDF; ++c) {
            const float g = ok ? dwc[((size_t)b * IDF + c) * Q + qq] : 0.f;
            s_a[wid][c][lane] = g;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) da[l] = fmaf(s_src[c][l], g, da[l]);
        }
        float dot = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) dot = fmaf(at[l], da[l], dot);
        float ds[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) ds[l] = at[l] * (da[l] - dot);
        if (ok) {
#pragma unroll 4
            for (int c = 0; c < IDF; ++c) {
                float o = 0.f;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) o = fmaf(ds[l], s_src[c][l], o);
                dx[((size_t)b * IDF + c) * Q + q] = o;
            }
        }
        // dsrc contraction on the matrix cores, two passes through this wave's private LDS
        // slab: (A = dwc, B = attn) then (A = x, B = ds).  LDS operations of one wave execute
        // in order, so the slab needs no barrier between the passes.
        // 16x16x4 MFMA operands: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]
        const int fi = lane & 15, fk = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
#pragma unroll 4
                for (int c = 0; c < IDF; ++c)
                    s_a[wid][c][lane] = ok ? x[((size_t)b * IDF + c) * Q + qq] : 0.f;
            }
#pragma unroll
            for (int l = 0; l < LMAX; ++l) s_b[wid][lane][l] = pass == 0 ? at[l] : ds[l];
#pragma unroll 4
            for (int kk = 0; kk < KQ / 4; ++kk) {
                const float bv = s_b[wid][4 * kk + fk][fi];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s_a[wid][16 * i + fi][4 * kk + fk];
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15 (= l), row = (lane >> 4) * 4 + r (= c in tile)
    const int l = lane & 15;
    float* slot = ws + ((size_t)b * (gridDim.x * 4) + wave_global) * (IDF * LMAX);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * i + (lane >> 4) * 4 + r;
            slot[c * LMAX + l] = acc[i][r];
        }
}

// dsrc[b, c, l] = sum over the waves of attn_general_bwd_kernel, in wave order
__global__ __launch_bounds__(256) void attn_dsrc_combine_kernel(const float* __restrict__ ws, float* __restrict__ dsrc,
                                                                int B, int IDF, int L, int nwave) {
    const int total = B * IDF * L;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int l = e % L;
        const int c = (e / L) % IDF;
        const int b = e / (L * IDF);
        const float* p = ws + (size_t)b * nwave * IDF * 16 + c * 16 + l;
        float v = 0.f;
        for (int w = 0; w < nwave; ++w) v += p[(size_t)w * IDF * 16];
        dsrc[e] = v;
    }
}

// ======================================================================================
// A2: bottom-up (word -> object) attention.  Tiny: one workgroup per sample.
// ======================================================================================
// tgt [B, D2, R] (label features), ctx1 [B, D2, L] (GloVe words), src [B, IDF, L] (projected
// word embeddings), mask [B, L] or null.  attn [B, L, R], wc [B, IDF, R].
__global__ __launch_bounds__(256) void attn_bu_fwd_kernel(
    const float* __restrict__ tgt, const float* __restrict__ ctx1, const float* __restrict__ src,
    const unsigned char* __restrict__ mask, float* __restrict__ wc, float* __restrict__ attn,
    int B, int D2, int IDF, int R, int L, int normalize, float eps) {
    extern __shared__ float sm[];
    float* s_attn = sm;               // [R][L]
    float* s_nt = s_attn + R * L;     // [R]
    float* s_nc = s_nt + R;           // [L]
    const int b = blockIdx.x;
    const float* tb = tgt + (size_t)b * D2 * R;
    const float* cb = ctx1 + (size_t)b * D2 * L;
    for (int i = threadIdx.x; i < R + L; i += blockDim.x) {
        float s = 0.f;
        if (i < R) { for (int d = 0; d < D2; ++d) { const float v = tb[d * R + i]; s = fmaf(v, v, s); } s_nt[i] = sqrtf(s); }
        else { const int l = i - R; for (int d = 0; d < D2; ++d) { const float v = cb[d * L + l]; s = fmaf(v, v, s); } s_nc[l] = sqrtf(s); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * L; i += blockDim.x) {
        const int r = i / L, l = i - r * L;
        float s = 0.f;
        for (int d = 0; d < D2; ++d) s = fmaf(tb[d * R + r], cb[d * L + l], s);
        if (normalize) s = s / fmaxf(s_nt[r] * s_nc[l], eps);
        if (mask && mask[(size_t)(((long)b * R + r) % B) * L + l]) s = -INFINITY;
        s_attn[i] = s;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float mx = -INFINITY;
        for (int l = 0; l < L; ++l) mx = fmaxf(mx, s_attn[r * L + l]);
        float sum = 0.f;
        for (int l = 0; l < L; ++l) { const float e = expf(s_attn[r * L + l] - mx); s_attn[r * L + l] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int l = 0; l < L; ++l) s_attn[r * L + l] *= inv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * L; i += blockDim.x) {
        const int r = i / L, l = i - r * L;
        attn[((size_t)b * L + l) * R + r] = s_attn[i];
    }
    const float* sb = src + (size_t)b * IDF * L;
    for (int i = threadIdx.x; i < IDF * R; i += blockDim.x) {
        const int c = i / R, r = i - c * R;
        float o = 0.f;
        for (int l = 0; l < L; ++l) o = fmaf(sb[c * L + l], s_attn[r * L + l], o);
        wc[((size_t)b * IDF + c) * R + r] = o;
    }
}

// dsrc[b, c, l] = sum_r dwc[b, c, r] * attn[b, l, r]   (the only differentiable input on the
// hot path: label features and GloVe words are constants there)
__global__ __launch_bounds__(256) void attn_bu_bwd_kernel(
    const float* __restrict__ dwc, const float* __restrict__ attn, float* __restrict__ dsrc,
    int IDF, int R, int L) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < IDF * L; i += blockDim.x) {
        const int c = i / L, l = i - c * L;
        float s = 0.f;
        for (int r = 0; r < R; ++r)
            s = fmaf(dwc[((size_t)b * IDF + c) * R + r], attn[((size_t)b * L + l) * R + r], s);
        dsrc[((size_t)b * IDF + c) * L + l] = s;
    }
}

// ======================================================================================
// A4: masked max over object slots
// ======================================================================================
// f [B, NUM, R], m [B, R, P] with an optional per-NUM stride (0 = one mask shared by all NUM
// feature channels, the hot-path case), out[b, c, p] = max_r f[b, c, r] * m[b, r, (c,) p]
#define MM_RMAX 16
__global__ __launch_bounds__(256) void masked_max_fwd_kernel(
    const float* __restrict__ f, const float* __restrict__ m, float* __restrict__ out,
    int NUM, int R, int P, long m_stride_b, long m_stride_r, long m_stride_c) {
    extern __shared__ float s_f[];   // [NUM][R]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NUM * R; i += blockDim.x) s_f[i] = f[(size_t)b * NUM * R + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float* mb = m + (size_t)b * m_stride_b + p;
    if (m_stride_c == 0) {
        float mv[MM_RMAX];
#pragma unroll
        for (int r = 0; r < MM_RMAX; ++r) mv[r] = r < R ? mb[(size_t)r * m_stride_r] : 0.f;
        for (int c = 0; c < NUM; ++c) {
            float best = -INFINITY;
#pragma unroll
            for (int r = 0; r < MM_RMAX; ++r)
                if (r < R) best = fmaxf(best, s_f[c * R + r] * mv[r]);
            out[((size_t)b * NUM + c) * P + p] = best;
        }
    } else {
        for (int c = 0; c < NUM; ++c) {
            float best = -INFINITY;
            for (int r = 0; r < R; ++r)
                best = fmaxf(best, s_f[c * R + r] * mb[(size_t)r * m_stride_r + (size_t)c * m_stride_c]);
            out[((size_t)b * NUM + c) * P + p] = best;
        }
    }
}

// df[b, c, r] = sum_p dout[b, c, p] * m[b, r, p] * [r == argmax_r' f*m]   (first max wins).  Every wave owns an LDS
// slab [NUM][R] (lane 0 adds the wave's shuffle-reduced contributions: no LDS atomics), the four slabs of a workgroup
// are added in wave order into ws[b][block], masked_max_combine_kernel sums the blocks in order: bit-reproducible.
__global__ __launch_bounds__(256) void masked_max_bwd_kernel(
    const float* __restrict__ f, const float* __restrict__ m, const float* __restrict__ dout,
    float* __restrict__ ws, int NUM, int R, int P, long m_stride_b, long m_stride_r,
    long m_stride_c) {
    extern __shared__ float sm[];
    float* s_f = sm;                  // [NUM][R]
    float* s_all = sm + NUM * R;      // [4 waves][NUM][R]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NUM * R; i += blockDim.x) s_f[i] = f[(size_t)b * NUM * R + i];
    for (int i = threadIdx.x; i < 4 * NUM * R; i += blockDim.x) s_all[i] = 0.f;
    __syncthreads();
    float* s_df = s_all + (threadIdx.x >> 6) * NUM * R;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = p < P;
    const float* mb = m + (size_t)b * m_stride_b + (ok ? p : 0);
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < NUM; ++c) {
        float best = -INFINITY;
        int arg = 0;
        float marg = 0.f;
        for (int r = 0; r < R; ++r) {
            const float mv = mb[(size_t)r * m_stride_r + (size_t)c * m_stride_c];
            const float v = s_f[c * R + r] * mv;
            if (v > best) { best = v; arg = r; marg = mv; }
        }
        const float contrib = ok ? dout[((size_t)b * NUM + c) * P + p] * marg : 0.f;
        // reduce per argmax slot across the wave, one LDS atomic per (wave, slot)
        // (only the slots some lane of the wave points at: neighbouring pixels share their arg-max box, so a wave
        // holds one or two distinct slots -- the ballot skips the other R - 2 shuffle reductions)
        for (int r = 0; r < R; ++r) {
            if (__ballot(arg == r && contrib != 0.f) == 0) continue;
            float v = (arg == r) ? contrib : 0.f;
            v = og_wave_sum(v);
            if (lane == 0) s_df[c * R + r] += v;
        }
    }
    __syncthreads();
    float* slot = ws + ((size_t)b * gridDim.x + blockIdx.x) * NUM * R;
    for (int i = threadIdx.x; i < NUM * R; i += blockDim.x)
        slot[i] = ((s_all[i] + s_all[NUM * R + i]) + s_all[2 * NUM * R + i]) + s_all[3 * NUM * R + i];
}

__global__ __launch_bounds__(256) void masked_max_combine_kernel(const float* __restrict__ ws, float* __restrict__ df,
                                                                 int B, int NR, int nblk) {
    const int total = B * NR;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int b = e / NR, i = e - b * NR;
        const float* p = ws + (size_t)b * nblk * NR + i;
        float v = 0.f;
        for (int k = 0; k < nblk; ++k) v += p[(size_t)k * NR];
        df[e] = v;
    }
}

// ======================================================================================
// Strided softmax (func_attention's two softmaxes), y = softmax(scale * x) along `dim`
// ======================================================================================
// x viewed as [outer, dim, inner].  lens (optional, int32 [nlens]): the softmax of outer row o
// only spans the first lens[o % nlens] entries of dim, the rest of y is 0.  rowvalid (optional,
// uint8 [outer]): rows with rowvalid == 0 produce y = 0.
__global__ __launch_bounds__(256) void softmax_strided_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, long outer, int dim, long inner, float scale,
    const int* __restrict__ lens, int nlens, const unsigned char* __restrict__ rowvalid) {
    const long total = outer * inner;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const long o = e / inner;
        const long i = e - o * inner;
        const float* xp = x + o * dim * inner + i;
        float* yp = y + o * dim * inner + i;
        const int n = lens ? min(dim, lens[o % nlens]) : dim;
        const bool valid = (!rowvalid || rowvalid[o]) && n > 0;
        float mx = -INFINITY;
        for (int d = 0; d < n; ++d) mx = fmaxf(mx, scale * xp[d * inner]);
        float sum = 0.f;
        for (int d = 0; d < n; ++d) sum += expf(scale * xp[d * inner] - mx);
        const float inv = valid ? 1.f / sum : 0.f;
        for (int d = 0; d < dim; ++d)
            yp[d * inner] = (d < n && valid) ? expf(scale * xp[d * inner] - mx) * inv : 0.f;
    }
}

// inner == 1 (softmax along the contiguous axis: the 289 regions of a word's attention row, reference
// GlobalAttention.py:52-57): ONE WAVE PER ROW -- lanes stride along the row (coalesced), the row lives in registers
// (dim <= 64 * OG_SM_PER), maximum and sum are wave shuffles.  The thread-per-row form above reads a row per thread:
// adjacent lanes 4 * dim bytes apart, three passes (r02: 180 us for 4.7 MB = 26 GB/s).
#define OG_SM_PER 16
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(
    const float* __restrict__ x, float* __restrict__ y, long outer, int dim, float scale,
    const int* __restrict__ lens, int nlens, const unsigned char* __restrict__ rowvalid) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= outer) return;
    const float* xp = x + o * dim;
    float* yp = y + o * dim;
    const int n = lens ? min(dim, lens[o % nlens]) : dim;
    const bool valid = (!rowvalid || rowvalid[o]) && n > 0;
    float v[OG_SM_PER];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < OG_SM_PER; ++j) {
        const int d = lane + 64 * j;
        v[j] = d < n ? scale * xp[d] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = og_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < OG_SM_PER; ++j) {
        const int d = lane + 64 * j;
        v[j] = d < n ? expf(v[j] - mx) : 0.f;
        sum += v[j];
    }
    sum = og_wave_sum(sum);
    const float inv = valid ? 1.f / sum : 0.f;
#pragma unroll
    for (int j = 0; j < OG_SM_PER; ++j) {
        const int d = lane + 64 * j;
        if (d < dim) yp[d] = valid ? v[j] * inv : 0.f;
    }
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long outer, int dim, float scale) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= outer) return;
    const long base = o * dim;
    float yv[OG_SM_PER], gv[OG_SM_PER];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < OG_SM_PER; ++j) {
        const int d = lane + 64 * j;
        yv[j] = d < dim ? y[base + d] : 0.f;
        gv[j] = d < dim ? dy[base + d] : 0.f;
        dot = fmaf(yv[j], gv[j], dot);
    }
    dot = og_wave_sum(dot);
#pragma unroll
    for (int j = 0; j < OG_SM_PER; ++j) {
        const int d = lane + 64 * j;
        if (d < dim) dx[base + d] = scale * yv[j] * (gv[j] - dot);
    }
}

// dx = scale * y * (dy - sum_d y * dy)
__global__ __launch_bounds__(256) void softmax_strided_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long outer,
    int dim, long inner, float scale) {
    const long total = outer * inner;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const long o = e / inner;
        const long i = e - o * inner;
        const long base = o * dim * inner + i;
        float dot = 0.f;
        for (int d = 0; d < dim; ++d) dot = fmaf(y[base + d * inner], dy[base + d * inner], dot);
        for (int d = 0; d < dim; ++d)
            dx[base + d * inner] = scale * y[base + d * inner] * (dy[base + d * inner] - dot);
    }
}

// ---- small batched matrix product with arbitrary strides ---------------------------------------------
//   C[b][m][n] = sum_k A[b][m][k] * B[b][k][n]
// The DAMSM word loss forms, per image b, the region-context of every word of every caption:
// weightedContext[b] = context[b] (nef x 289) . attn[b]^T (289 x B*L) (reference GlobalAttention.py:62-68,
// losses.py:108-112) -- B tiny products that the reference issues as B bmm calls inside a python loop.  One
// launch serves the whole batch and, through the strides, both gradients (dA = dC B^T, dB = A^T dC).
// fp32 VALU, 64 x 64 tile, 16-deep K chunks through LDS; the products total < 1 GFLOP per step.
struct BmmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long sab, sam, sak;     // strides of A: batch, m, k
    long sbb, sbk, sbn;     // strides of B: batch, k, n
    long scb, scm, scn;     // strides of C
};

__global__ __launch_bounds__(256) void bmm_strided_kernel(const BmmArgs a) {
    __shared__ float As[16][64 + 1];
    __shared__ float Bs[16][64 + 1];
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;      // this thread's 4 x 4 outputs
    const float* Ab = a.A + (long)b * a.sab;
    const float* Bb = a.B + (long)b * a.sbb;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;                      // 1024 elements per operand tile
            // A tile: consecutive threads along whichever of (m, k) is contiguous in memory
            int mm, kk;
            if (a.sak <= a.sam) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
            const int gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < a.M && gk < a.K) ? Ab[(long)gm * a.sam + (long)gk * a.sak] : 0.f;
            int nn, k2;
            if (a.sbn <= a.sbk) { nn = e & 63; k2 = e >> 6; } else { k2 = e & 15; nn = e >> 4; }
            const int gn = n0 + nn, gk2 = k0 + k2;
            Bs[k2][nn] = (gn < a.N && gk2 < a.K) ? Bb[(long)gk2 * a.sbk + (long)gn * a.sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = As[kk][tm + i]; bv[i] = Bs[kk][tn + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* Cb = a.C + (long)b * a.scb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + tm + i, gn = n0 + tn + j;
            if (gm < a.M && gn < a.N) Cb[(long)gm * a.scm + (long)gn * a.scn] = acc[i][j];
        }
}

// ---- binary cross-entropy against a constant target -------------------------------------------------------
// Every adversarial term of the step is nn.BCELoss()(sigmoid output, ones / zeros) on B x 1 x k x k
// probabilities (reference losses.py:182-204, 230-246, ...: ~60 per step).  torch spends a fill (the label
// tensor), the loss kernel and a mean on each, and three more launches on the way back; here it is one small
// workgroup each way.  Same arithmetic as torch: log clamped at -100, backward (p - t) / max(p (1 - p), 1e-12).
__global__ __launch_bounds__(256) void bce_const_fwd_kernel(const float* __restrict__ p, float* __restrict__ out,
                                                            int n, float t) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = p[i];
        const float l1 = fmaxf(logf(v), -100.f), l0 = fmaxf(logf(1.f - v), -100.f);
        s -= t * l1 + (1.f - t) * l0;
    }
    s = og_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ __launch_bounds__(256) void bce_const_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ dp, int n, float t) {
    const float go = g[0] / (float)n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = p[i];
        dp[i] = go * (v - t) / fmaxf((1.f - v) * v, 1e-12f);
    }
}

extern "C" {

int objgan_attn_general_forward(const float* x, const float* src, const unsigned char* mask,
                                float* wc, float* attn, int B, int idf, int Q, int L,
                                void* stream) {
    OG_ENTRY();
    if (L < 1 || L > 16) return OG_BAD_ARGS;
    if (B <= 0 || Q <= 0) return OG_OK;
    dim3 grid(og_cdiv(Q, 256), B);
    hipStream_t s = (hipStream_t)stream;
    if (idf == 48) hipLaunchKernelGGL((attn_general_fwd_kernel<48, 16>), grid, dim3(256), 0, s, x, src, mask, wc, attn, B, Q, L);
    else if (idf == 32) hipLaunchKernelGGL((attn_general_fwd_kernel<32, 16>), grid, dim3(256), 0, s, x, src, mask, wc, attn, B, Q, L);
    else if (idf == 64) hipLaunchKernelGGL((attn_general_fwd_kernel<64, 16>), grid, dim3(256), 0, s, x, src, mask, wc, attn, B, Q, L);
    else return OG_BAD_ARGS;
    return og_launch_status();
}

// dsrc is fully written (per-wave partial tiles in `ws`, summed in wave order by a second kernel): nothing to pre-zero,
// no atomics; dattn may be null.
static inline int attn_bwd_chunks(int Q) {
    // pixels per wave: enough chunks that a wave's partial tile is amortised, while the grid still covers the chip
    int chunks = Q / (64 * 4 * 16);
    if (chunks < 1) chunks = 1;
    if (chunks > 8) chunks = 8;
    return chunks;
}

// floats of workspace objgan_attn_general_backward needs (per-wave partial dsrc tiles)
long objgan_attn_general_backward_ws_floats(int B, int idf, int Q, int L) {
    if (B <= 0 || Q <= 0) return 0;
    const long nwave = 4L * og_cdiv(Q, 64 * 4 * attn_bwd_chunks(Q));
    return (long)B * nwave * idf * 16;
}

int objgan_attn_general_backward(const float* x, const float* src, const float* attn,
                                 const float* dwc, const float* dattn, float* dx, float* dsrc,
                                 int B, int idf, int Q, int L, float* ws, void* stream) {
    OG_ENTRY();
    if (L < 1 || L > 16) return OG_BAD_ARGS;
    if (B <= 0 || Q <= 0) return OG_OK;
    if (!ws) return OG_BAD_ARGS;
    const int chunks = attn_bwd_chunks(Q);
    dim3 grid(og_cdiv(Q, 64 * 4 * chunks), B);
    hipStream_t s = (hipStream_t)stream;
    if (idf == 48) hipLaunchKernelGGL((attn_general_bwd_kernel<48, 16>), grid, dim3(256), 0, s, x, src, attn, dwc, dattn, dx, ws, B, Q, L, chunks);
    else if (idf == 32) hipLaunchKernelGGL((attn_general_bwd_kernel<32, 16>), grid, dim3(256), 0, s, x, src, attn, dwc, dattn, dx, ws, B, Q, L, chunks);
    else if (idf == 64) hipLaunchKernelGGL((attn_general_bwd_kernel<64, 16>), grid, dim3(256), 0, s, x, src, attn, dwc, dattn, dx, ws, B, Q, L, chunks);
    else return OG_BAD_ARGS;
    hipLaunchKernelGGL(attn_dsrc_combine_kernel, dim3(og_cdiv((long)B * idf * L, 256)), dim3(256), 0, s, ws, dsrc, B, idf, L,
                       (int)grid.x * 4);
    return og_launch_status();
}

int objgan_attn_bu_forward(const float* tgt, const float* ctx1, const float* src,
                           const unsigned char* mask, float* wc, float* attn,
                           int B, int d2, int idf, int R, int L, int normalize, float eps,
                           void* stream) {
    OG_ENTRY();
    if (B <= 0 || R <= 0 || L <= 0) return OG_OK;
    const size_t shm = sizeof(float) * ((size_t)R * L + R + L);
    if (shm > 60000) return OG_BAD_ARGS;
    hipLaunchKernelGGL(attn_bu_fwd_kernel, dim3(B), dim3(256), shm, (hipStream_t)stream, tgt, ctx1,
                       src, mask, wc, attn, B, d2, idf, R, L, normalize, eps);
    return og_launch_status();
}

int objgan_attn_bu_backward(const float* dwc, const float* attn, float* dsrc,
                            int B, int idf, int R, int L, void* stream) {
    OG_ENTRY();
    if (B <= 0 || R <= 0 || L <= 0) return OG_OK;
    hipLaunchKernelGGL(attn_bu_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dwc, attn,
                       dsrc, idf, R, L);
    return og_launch_status();
}

// mask strides in elements: m[b*sb + r*sr + c*sc + p]; sc = 0 when one mask serves all channels
int objgan_masked_max_forward(const float* f, const float* m, float* out, int B, int num, int R,
                              int P, long m_stride_b, long m_stride_r, long m_stride_c,
                              void* stream) {
    OG_ENTRY();
    if (R < 1 || R > MM_RMAX) return OG_BAD_ARGS;
    if (B <= 0 || num <= 0 || P <= 0) return OG_OK;
    const size_t shm = sizeof(float) * (size_t)num * R;
    if (shm > 60000) return OG_BAD_ARGS;
    dim3 grid(og_cdiv(P, 256), B);
    hipLaunchKernelGGL(masked_max_fwd_kernel, grid, dim3(256), shm, (hipStream_t)stream, f, m, out,
                       num, R, P, m_stride_b, m_stride_r, m_stride_c);
    return og_launch_status();
}

// floats of workspace objgan_masked_max_backward needs (per-workgroup partial df)
long objgan_masked_max_backward_ws_floats(int B, int num, int R, int P) {
    if (B <= 0 || num <= 0 || P <= 0) return 0;
    return (long)B * og_cdiv(P, 256) * num * R;
}

// df [B, num, R] is fully written (no zero-fill needed)
int objgan_masked_max_backward(const float* f, const float* m, const float* dout, float* df,
                               int B, int num, int R, int P, long m_stride_b, long m_stride_r,
                               long m_stride_c, float* ws, void* stream) {
    OG_ENTRY();
    if (R < 1 || R > MM_RMAX) return OG_BAD_ARGS;
    if (B <= 0 || num <= 0 || P <= 0) return OG_OK;
    if (!ws) return OG_BAD_ARGS;
    const size_t shm = sizeof(float) * (size_t)num * R * 5;
    if (shm > 60000) return OG_BAD_ARGS;
    dim3 grid(og_cdiv(P, 256), B);
    hipLaunchKernelGGL(masked_max_bwd_kernel, grid, dim3(256), shm, (hipStream_t)stream, f, m, dout,
                       ws, num, R, P, m_stride_b, m_stride_r, m_stride_c);
    hipLaunchKernelGGL(masked_max_combine_kernel, dim3(og_cdiv((long)B * num * R, 256)), dim3(256), 0, (hipStream_t)stream,
                       ws, df, B, num * R, (int)grid.x);
    return og_launch_status();
}

int objgan_softmax_strided_forward(const float* x, float* y, long outer, int dim, long inner,
                                   float scale, const int* lens, int nlens,
                                   const unsigned char* rowvalid, void* stream) {
    OG_ENTRY();
    const long total = outer * inner;
    if (total <= 0 || dim <= 0) return OG_OK;
    if (inner == 1 && dim <= 64 * OG_SM_PER && outer < (1L << 31)) {
        hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)((outer + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           x, y, outer, dim, scale, lens, nlens, rowvalid);
        return og_launch_status();
    }
    hipLaunchKernelGGL(softmax_strided_fwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, y, outer, dim, inner, scale, lens, nlens, rowvalid);
    return og_launch_status();
}

int objgan_softmax_strided_backward(const float* y, const float* dy, float* dx, long outer, int dim,
                                    long inner, float scale, void* stream) {
    OG_ENTRY();
    const long total = outer * inner;
    if (total <= 0 || dim <= 0) return OG_OK;
    if (inner == 1 && dim <= 64 * OG_SM_PER && outer < (1L << 31)) {
        hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((outer + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           y, dy, dx, outer, dim, scale);
        return og_launch_status();
    }
    hipLaunchKernelGGL(softmax_strided_bwd_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, y, dy, dx, outer, dim, inner, scale);
    return og_launch_status();
}

// C[b] = A[b] . B[b] for `batch` small matrices with arbitrary (element) strides; C is overwritten.
int objgan_bmm_strided(const float* A, const float* B, float* C, int batch, int M, int N, int K,
                       long sab, long sam, long sak, long sbb, long sbk, long sbn,
                       long scb, long scm, long scn, void* stream) {
    OG_ENTRY();
    if (batch <= 0 || M <= 0 || N <= 0) return OG_OK;
    if (K < 0 || batch > 65535) return OG_BAD_ARGS;
    BmmArgs a = {A, B, C, M, N, K, sab, sam, sak, sbb, sbk, sbn, scb, scm, scn};
    dim3 grid(og_cdiv(N, 64), og_cdiv(M, 64), batch);
    hipLaunchKernelGGL(bmm_strided_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return og_launch_status();
}

// mean BCE of n probabilities against the constant target t (0 or 1): out[0]; and its gradient g[0] * d/dp.
int objgan_bce_const_forward(const float* p, float* out, int n, float t, void* stream) {
    OG_ENTRY();
    if (n <= 0) return OG_BAD_ARGS;
    hipLaunchKernelGGL(bce_const_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, out, n, t);
    return og_launch_status();
}

int objgan_bce_const_backward(const float* p, const float* g, float* dp, int n, float t, void* stream) {
    OG_ENTRY();
    if (n <= 0) return OG_OK;
    hipLaunchKernelGGL(bce_const_bwd_kernel, dim3(og_cdiv(n, 256) > 64 ? 64 : og_cdiv(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, p, g, dp, n, t);
    return og_launch_status();
}

}  // extern "C"
