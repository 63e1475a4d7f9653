// Implicit-GEMM convolution on the CDNA4 matrix cores (fp32 in / fp32 accumulate,
// v_mfma_f32_32x32x2_f32), forward + data-gradient + weight-gradient.
//
// One kernel family serves every convolution of the Obj-GAN image_generation hot path
// (reference image_generation/model.py:30-81 conv1x1/conv3x3/upBlock/downBlock_G/
// HmapResBlock, :589-617 G_HMAP, :708-719 GET_IMAGE_G, :986-1048 D encoders and heads,
// :1184-1312 object discriminators), which the reference hands to cuDNN:
//
//   y[n, m, oh, ow] = sum_{c, t}  Wp[m][c*T + t] * x[n, c, a*s + dh[t], b*s + dw[t]]
//   (oh, ow) = (a*osh + ooh, b*osw + oow),   (a, b) in a PH x PW grid per image
//
// * forward conv      : m = cout, c = cin, taps t = (kh, kw), dh = kh - pad
// * dgrad, stride 1   : m = cin,  c = cout, flipped taps (host passes dh/dw + tap map)
// * dgrad, stride 2   : one launch per output parity phase, 2x2 (k=4) or <=2x2 (k=3)
//                       taps each, osh = osw = 2 -- no zero-multiplies
// * nearest x2 upsample (upBlock) and ReflectionPad2d are folded into the gather
//   (the up-sampled / padded tensor never exists in HBM)
// * epilogue: + bias, LeakyReLU(0.2) / tanh / sigmoid
//
// GEMM view: M = output channels, N = pixels (n, a, b), K = T*Cp (TAP-MAJOR: k = t*Cp + c with
// Cp = C rounded up to the K step of 16, so that one K step of the main loop touches ONE tap:
// the tap geometry -- bounds / reflect / upsample / offset -- is evaluated once per step and
// lane, the 8-16 gathered elements of the step differ only by a wave-uniform channel offset).  A workgroup of 4 waves
// owns a BM x BN tile (BM in {128, 64, 32}); each wave owns TM x 2 MFMA tiles of 32x32,
// accumulators stay in AGPRs for the whole K loop.  Operands go HBM -> VGPR -> LDS ->
// VGPR -> MFMA, double-buffered in LDS with one barrier per K step (BK = 16); the global
// loads of step k+1 are issued before the MFMAs of step k.  The weight matrix is
// pre-packed K-major ([K][Mpad], zero padded) so its loads are 16-byte and coalesced and
// its LDS image needs no transpose; the activation gather is one dword per lane with
// consecutive lanes on consecutive pixels (coalesced along W).  Workgroup ids are
// remapped so that the M-tiles sharing one pixel tile run on the same XCD (shared L2).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OG_MAX_TAPS 32
#define OG_ACT_NONE 0
#define OG_ACT_LRELU 1
#define OG_ACT_TANH 2
#define OG_ACT_SIGMOID 3
#define OG_ACT_RELU 4

struct IgemmArgs {
    const float* x;      // [N, C, H, W] source activations (or dY for dgrad)
    const float* wt;     // packed weights, K-major: [Kpad][Mpad]
    const float* bias;   // [M] or nullptr
    float* y;            // [N, M, OHf, OWf]
    int N, C, H, W;      // physical source dims
    int LH, LW;          // logical source dims seen by the taps (2H x 2W when upsampling)
    int M, Mpad, K, Kpad;   // K = T*C (algorithmic), Kpad = T*Cp (padded, what the loop walks)
    int T, Cp;              // taps, channels rounded up to 16
    int m_begin, m_end;  // output-channel rows covered by this launch
    int PH, PW;          // GEMM pixel grid per image
    int OHf, OWf;        // physical output dims
    int osh, osw, ooh, oow;
    int stride;
    int pad_mode;        // 0 = zeros outside [0,LH)x[0,LW), 1 = reflect
    int upsample;        // 1 = source index = logical index >> 1
    int act;
    int ksplit_steps;    // > 0: split-K -- blockIdx.y owns this many BK steps, epilogue = atomicAdd
    int tap[OG_MAX_TAPS];   // (dw << 16) | (dh & 0xffff): one scalar load per (uniform) tap
};

__device__ __forceinline__ float og_act(float v, int act) {
    if (act == OG_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == OG_ACT_TANH) return tanhf(v);
    if (act == OG_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == OG_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// bijective XCD-aware remap of a linear workgroup id (dispatcher places id b on XCD b % 8)
__device__ __forceinline__ int og_xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, j = id >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + j;
}

template <int WM, int TM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs a) {
    constexpr int WN = 4 / WM;
    constexpr int TN = 2;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int BK = 16;
    constexpr int BROWS = BK * BN / 256;   // gathered elements per thread per K step
    constexpr int KSTEP = 256 / BN;        // k rows covered by one pass of the workgroup
    constexpr int NA4 = BK * BM / 4;       // float4s in one A tile
    constexpr int NA_PER = (NA4 + 255) / 256;

    __shared__ float As[2][BK][BM];
    __shared__ float Bs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    const int Npix = a.N * a.PH * a.PW;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (Npix + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int wg = og_xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread gather geometry (the pixel of a thread is fixed for the whole K loop)
    const int kr0 = __builtin_amdgcn_readfirstlane(tid / BN);
    const int pix = n0 + (tid % BN);
    const bool pix_ok = pix < Npix;
    int ihb = 0, iwb = 0;
    const float* xb = a.x;
    {
        const int ppi = a.PH * a.PW;
        const int pp = pix_ok ? pix : 0;
        const int n = pp / ppi;
        const int rem = pp - n * ppi;
        const int pa = rem / a.PW;
        const int pb = rem - pa * a.PW;
        ihb = pa * a.stride;
        iwb = pb * a.stride;
        xb = a.x + (size_t)n * a.C * a.H * a.W;
    }
    const int HW = a.H * a.W;

    float rb[BROWS];
    float4 ra[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    // Branch-free gather.  One K step = 16 consecutive channels of ONE tap: the tap geometry is
    // evaluated once per step; every lane always issues its BROWS loads (from a clamped, in-range
    // address) so they are all in flight together; out-of-image / padded-channel elements are
    // zeroed by a bit mask when the tile is written to LDS.
    unsigned okmask = 0;
    const int us = a.upsample ? 1 : 0;
    const int steps_per_tap = a.Cp / BK;
    const bool refl = a.pad_mode == 1;
    auto load_b = [&](int kt) {
        const int t = kt / steps_per_tap;                // wave-uniform
        const int cb = (kt - t * steps_per_tap) * BK;
        const int tp = a.tap[t];
        const int ih = ihb + ((tp << 16) >> 16);
        const int iw = iwb + (tp >> 16);
        int ihr = ih < 0 ? -ih : ih;
        int iwr = iw < 0 ? -iw : iw;
        ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
        iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
        const bool inb = ((unsigned)ih < (unsigned)a.LH) && ((unsigned)iw < (unsigned)a.LW);
        const bool ok = pix_ok && (refl || inb);
        const int ihs = (refl ? ihr : ih) >> us;
        const int iws = (refl ? iwr : iw) >> us;
        const float* src = xb + (ok ? ihs * a.W + iws : 0);
        okmask = 0;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const int c = cb + kr0 + KSTEP * i;          // wave-uniform
            const int cc = min(c, a.C - 1);
            rb[i] = src[(size_t)cc * HW];
            okmask |= ((ok && c < a.C) ? 1u : 0u) << i;
        }
    };
    constexpr bool A_FULL = (NA4 % 256) == 0;   // every thread loads NA_PER float4s
    const bool a_thread = A_FULL || tid < NA4;
    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            const int idx = tid + 256 * i;
            const int k = idx / (BM / 4);
            const int m4 = (idx - k * (BM / 4)) * 4;
            if (a_thread)
                ra[i] = *reinterpret_cast<const float4*>(a.wt + (size_t)(k0 + k) * a.Mpad + m0 + m4);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            const int idx = tid + 256 * i;
            const int k = idx / (BM / 4);
            const int m4 = (idx - k * (BM / 4)) * 4;
            if (a_thread) *reinterpret_cast<float4*>(&As[buf][k][m4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
            Bs[buf][kr0 + KSTEP * i][tid % BN] = ((okmask >> i) & 1u) ? rb[i] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = a.Kpad / BK;
    const int kt0 = a.ksplit_steps > 0 ? blockIdx.y * a.ksplit_steps : 0;
    const int nk = a.ksplit_steps > 0 ? min(nk_all, kt0 + a.ksplit_steps) : nk_all;
    load_a(kt0 * BK);
    load_b(kt0);
    store_tiles(0);
    __syncthreads();

    const int lrow = lane >> 5;          // k sub-index of the 32x32x2 MFMA operand
    const int lcol = lane & 31;
    int cur = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) { load_a((kt + 1) * BK); load_b(kt + 1); }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[cur][2 * kk + lrow][(wm * TM + i) * 32 + lcol];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bs[cur][2 * kk + lrow][(wn * TN + j) * 32 + lcol];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int ppi = a.PH * a.PW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = n0 + (wn * TN + j) * 32 + lcol;
        if (p >= Npix) continue;
        const int n = p / ppi;
        const int rem = p - n * ppi;
        const int pa = rem / a.PW;
        const int pb = rem - pa * a.PW;
        const int oh = pa * a.osh + a.ooh;
        const int ow = pb * a.osw + a.oow;
        float* yb = a.y + (size_t)n * a.M * a.OHf * a.OWf + (size_t)oh * a.OWf + ow;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (m < a.m_end) {
                    float v = acc[i][j][r];
                    if (a.ksplit_steps > 0) {
                        atomicAdd(&yb[(size_t)m * a.OHf * a.OWf], v);   // bias/act: follow-up pass
                    } else {
                        if (a.bias) v += a.bias[m];
                        v = og_act(v, a.act);
                        yb[(size_t)m * a.OHf * a.OWf] = v;
                    }
                }
            }
        }
    }
}

// ---- weight packing ------------------------------------------------------------------
// wt[(t*Cp + ck) * Mpad + cm] = src_tap[t] >= 0 ? w[...] : 0, zero padded to [T*Cp][Mpad].
// w is the PyTorch conv weight [Cout][Cin][Torig].  transpose = 0: cm = cout, ck = cin
// (forward);  transpose = 1: cm = cin, ck = cout (data gradient).
struct PackArgs {
    const float* w;
    float* wt;
    int Cout, Cin, Torig, Tg;
    int M, Mpad, Ck, Cp;
    int transpose;
    signed char src_tap[OG_MAX_TAPS];
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a) {
    const long total = (long)a.Tg * a.Cp * a.Mpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % a.Mpad);
        const int k = (int)(i / a.Mpad);
        const int t = k / a.Cp;
        const int ck = k - t * a.Cp;
        float v = 0.f;
        if (m < a.M && ck < a.Ck) {
            const int st = a.src_tap[t];
            if (st >= 0) {
                const int co = a.transpose ? ck : m;
                const int ci = a.transpose ? m : ck;
                v = a.w[((size_t)co * a.Cin + ci) * a.Torig + st];
            }
        }
        a.wt[i] = v;
    }
}

// y[n, m, i] = act(y[n, m, i] + bias[m]) -- epilogue of the split-K path
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                       long total, int M, int HW, int act) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        float v = y[e];
        if (bias) v += bias[(e / HW) % M];
        y[e] = og_act(v, act);
    }
}

// ---- weight gradient -------------------------------------------------------------------
//   dw[co][ci*T + t] += sum_{n,oh,ow} dy[n,co,oh,ow] * x[n,ci,oh*s - pad + kh, ow*s - pad + kw]
// GEMM: M = cout, N = cin*T columns, K = pixels (split across gridDim.y, fp32 atomics into
// a zero-initialised dw).  Both operands are contiguous along K (pixels) in HBM, so the
// LDS tiles are [row][BK+1] (padded: conflict-free column reads by the MFMA lanes).
struct WgradArgs {
    const float* x;    // [N, Cin, H, W]
    const float* dy;   // [N, Cout, OH, OW]
    float* dw;         // [Cout][Cin*T]
    int N, Cin, H, W, LH, LW;
    int Cout, OH, OW;
    int stride, pad, pad_mode, upsample;
    int m_begin, m_end;
    int ncol;
    int pix_per_split;
};

template <int KS, int WM, int TM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int T = KS * KS;
    constexpr int WN = 4 / WM;
    constexpr int TN = 2;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int BK = 32;
    constexpr int LD = BK + 1;
    constexpr int AR = BM / 8;     // dy elements per thread per K step
    constexpr int BR = BN / 8;     // gathered x elements per thread per K step

    __shared__ float As[2][BM][LD];
    __shared__ float Bs[2][BN][LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;

    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (a.ncol + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int wg = og_xcd_remap(blockIdx.x, nwg);
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int c0 = tile_n * BN;

    const int Npix = a.N * a.OH * a.OW;
    const int p_begin = blockIdx.y * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    if (p_begin >= p_end) return;

    const int kl = tid & 31;       // pixel within the K tile
    const int r0 = tid >> 5;       // first row handled by this thread (rows r0 + 8*i)
    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;

    float ra[AR], rb[BR];

    unsigned amask = 0, bmask = 0;
    const int us = a.upsample ? 1 : 0;
    auto load_tiles = [&](int pk) {
        const int p = pk + kl;
        const bool ok = p < p_end;
        const int pp = ok ? p : p_begin;
        const int n = pp / OHW;
        const int rem = pp - n * OHW;
        const int oh = rem / a.OW;
        const int ow = rem - oh * a.OW;
        const float* dyb = a.dy + (size_t)n * a.Cout * OHW + rem;
        amask = 0; bmask = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + r0 + 8 * i;
            const bool mok = ok && m < a.m_end;
            ra[i] = dyb[(size_t)(mok ? m : m0) * OHW];
            amask |= (mok ? 1u : 0u) << i;
        }
        const float* xb = a.x + (size_t)n * a.Cin * HW;
        const int ihb = oh * a.stride - a.pad;
        const int iwb = ow * a.stride - a.pad;
        const bool refl = a.pad_mode == 1;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int col = c0 + r0 + 8 * i;
            const int cc = min(col, a.ncol - 1);
            const int ci = cc / T;
            const int t = cc - ci * T;
            const int kh = t / KS;
            const int kw = t - kh * KS;
            const int ih = ihb + kh, iw = iwb + kw;
            int ihr = ih < 0 ? -ih : ih;
            int iwr = iw < 0 ? -iw : iw;
            ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
            iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
            const bool inb = ((unsigned)ih < (unsigned)a.LH) && ((unsigned)iw < (unsigned)a.LW);
            const bool cok = ok && (col < a.ncol) && (refl || inb);
            const int ihs = (refl ? ihr : ih) >> us;
            const int iws = (refl ? iwr : iw) >> us;
            const int off = cok ? (ci * HW + ihs * a.W + iws) : 0;
            rb[i] = xb[off];
            bmask |= (cok ? 1u : 0u) << i;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) As[buf][r0 + 8 * i][kl] = ((amask >> i) & 1u) ? ra[i] : 0.f;
#pragma unroll
        for (int i = 0; i < BR; ++i) Bs[buf][r0 + 8 * i][kl] = ((bmask >> i) & 1u) ? rb[i] : 0.f;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p_end - p_begin + BK - 1) / BK;
    load_tiles(p_begin);
    store_tiles(0);
    __syncthreads();

    const int lrow = lane >> 5;
    const int lcol = lane & 31;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) load_tiles(p_begin + (kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = As[cur][(wm * TM + i) * 32 + lcol][2 * kk + lrow];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = Bs[cur][(wn * TN + j) * 32 + lcol][2 * kk + lrow];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = c0 + (wn * TN + j) * 32 + lcol;
        if (col >= a.ncol) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (m < a.m_end) atomicAdd(a.dw + (size_t)m * a.ncol + col, acc[i][j][r]);
            }
        }
    }
}

// ---- optional per-launch timing (bench.py's roofline leg) -------------------------------------
// When enabled, every conv launch is bracketed by hipEvents on its own stream and tagged with a
// category (kind, taps / ksize, tile config) and its ALGORITHMIC flops 2*M*K*Npix.  Off by default;
// the only mutable global state of the library, touched by the host thread only.
#define OG_PROF_CATS 32
#define OG_PROF_MAX 16384
struct ProfRec { hipEvent_t a, b; int cat; double flops; };
static int g_prof_on = 0;
static ProfRec* g_prof = nullptr;
static int g_prof_n = 0;
static int g_prof_made = 0;

static inline int prof_cat(int wgrad, int t_or_k, int cfg) {
    int ti = wgrad ? (t_or_k == 1 ? 0 : (t_or_k == 3 ? 1 : 2)) : (t_or_k <= 1 ? 0 : (t_or_k <= 4 ? 1 : (t_or_k <= 9 ? 2 : 3)));
    return (wgrad ? 12 : 0) + ti * 3 + cfg;
}
static inline ProfRec* prof_begin(int cat, double flops, hipStream_t s) {
    if (!g_prof_on || g_prof_n >= OG_PROF_MAX) return nullptr;
    if (!g_prof) g_prof = (ProfRec*)calloc(OG_PROF_MAX, sizeof(ProfRec));
    ProfRec* r = &g_prof[g_prof_n];
    if (g_prof_n >= g_prof_made) {
        if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) return nullptr;
        g_prof_made = g_prof_n + 1;
    }
    g_prof_n++;
    r->cat = cat; r->flops = flops;
    (void)hipEventRecord(r->a, s);
    return r;
}
static inline void prof_end(ProfRec* r, hipStream_t s) { if (r) (void)hipEventRecord(r->b, s); }

// ---- host side ---------------------------------------------------------------------------
static inline int igemm_tiles(const IgemmArgs& a, int cfg) {
    const int Npix = a.N * a.PH * a.PW;
    const int rows = a.m_end - a.m_begin;
    if (cfg == 0) return og_cdiv(rows, 128) * og_cdiv(Npix, 128);
    if (cfg == 1) return og_cdiv(rows, 64) * og_cdiv(Npix, 256);
    return og_cdiv(rows, 32) * og_cdiv(Npix, 256);
}

static int launch_igemm(const IgemmArgs& a, int cfg, hipStream_t s) {
    const int g = igemm_tiles(a, cfg);
    const int splits = a.ksplit_steps > 0 ? og_cdiv(a.Kpad / 16, a.ksplit_steps) : 1;
    dim3 grid(g, splits);
    if (cfg == 0) hipLaunchKernelGGL((conv_igemm_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else if (cfg == 1) hipLaunchKernelGGL((conv_igemm_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<1, 1>), grid, dim3(256), 0, s, a);
    return og_launch_status();
}

// Rows [0, M) are covered greedily: 128-row tiles, then one 64-row tile, then 32-row tiles
// for the ragged remainder (388 = 3*128 + 4 -> 384 rows of cfg 0 + one 32-row tile;
// 194 -> 128 + 64 + 32; 96 -> 64 + 32), so that padding waste stays below ~15 %.
struct RowPart { int m_begin, m_end, cfg; };
static int og_row_parts(int M, RowPart* parts) {
    int n = 0, m = 0;
    if (M >= 128) { parts[n++] = {0, (M / 128) * 128, 0}; m = (M / 128) * 128; }
    if (M - m >= 64) { parts[n++] = {m, m + 64, 1}; m += 64; }
    if (M - m > 0) { parts[n++] = {m, M, 2}; }
    return n;
}

// Small-grid / long-K launches (discriminator heads at 4x4..16x16, `outlogits` with one output
// channel and K = 12288) would run on a handful of CUs for hundreds of serial K steps: they are
// split along K across gridDim.y, partial tiles are accumulated with fp32 atomics into a zeroed
// output and bias/activation are applied by a follow-up streaming pass.
static int run_igemm(IgemmArgs a, hipStream_t s, int y_prezeroed) {
    RowPart parts[3];
    const int np = og_row_parts(a.M, parts);
    int tiles = 0;
    for (int i = 0; i < np; ++i) {
        a.m_begin = parts[i].m_begin; a.m_end = parts[i].m_end;
        tiles += igemm_tiles(a, parts[i].cfg);
    }
    const int nk = a.Kpad / 16;
    const bool full_cover = (a.osh == 1 && a.osw == 1 && a.PH == a.OHf && a.PW == a.OWf);
    int splits = 1;
    if (tiles < 128 && nk >= 16 && (full_cover || y_prezeroed)) {
        splits = og_cdiv(512, tiles);
        if (splits > nk / 4) splits = nk / 4;
    }
    const float* bias = a.bias;
    const int act = a.act;
    if (splits > 1) {
        a.ksplit_steps = og_cdiv(nk, splits);
        a.bias = nullptr; a.act = OG_ACT_NONE;
        if (!y_prezeroed)
            (void)hipMemsetAsync(a.y, 0, sizeof(float) * (size_t)a.N * a.M * a.OHf * a.OWf, s);
    } else {
        a.ksplit_steps = 0;
    }
    for (int i = 0; i < np; ++i) {
        a.m_begin = parts[i].m_begin; a.m_end = parts[i].m_end;
        const double fl = 2.0 * (a.m_end - a.m_begin) * (double)a.K * ((double)a.N * a.PH * a.PW);
        ProfRec* pr = prof_begin(prof_cat(0, a.T, parts[i].cfg), fl, s);
        int rc = launch_igemm(a, parts[i].cfg, s);
        prof_end(pr, s);
        if (rc != OG_OK) return rc;
    }
    if (splits > 1 && (bias || act != OG_ACT_NONE) && full_cover) {
        const long total = (long)a.N * a.M * a.OHf * a.OWf;
        hipLaunchKernelGGL(bias_act_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, s, a.y, bias,
                           total, a.M, a.OHf * a.OWf, act);
        return og_launch_status();
    }
    return OG_OK;
}

extern "C" {

// Size (in floats) of the packed-weight scratch for an M x K GEMM.
long objgan_conv_packed_floats(int M, int C, int T) {
    const long Mpad = ((long)M + 127) / 128 * 128;
    const long Cp = ((long)C + 15) / 16 * 16;
    return Mpad * Cp * T;
}

// General entry: see the formula at the top of this file.
//   w        PyTorch-layout conv weight [Cout][Cin][Torig]
//   wt       scratch of objgan_conv_packed_floats(M, C*Tg) floats (overwritten)
//   transpose 0: M = Cout, C = Cin ; 1: M = Cin, C = Cout (data gradient)
//   src_tap[t] index of GEMM tap t in the Torig taps of w (or -1 for a zero tap)
int objgan_conv_igemm(const float* x, const float* w, const float* bias, float* y, float* wt,
                      int N, int C, int H, int W, int upsample, int pad_mode,
                      int Cout, int Cin, int Torig, int transpose,
                      int Tg, const int* dh, const int* dw, const int* src_tap,
                      int PH, int PW, int stride,
                      int OHf, int OWf, int osh, int osw, int ooh, int oow,
                      int act, int y_prezeroed, void* stream) {
    if (Tg < 1 || Tg > OG_MAX_TAPS) return OG_BAD_ARGS;
    if (Torig < 1 || Torig > 127) return OG_BAD_ARGS;
    const int M = transpose ? Cin : Cout;
    const int Ck = transpose ? Cout : Cin;
    if (Ck != C) return OG_BAD_ARGS;
    if (N <= 0 || PH <= 0 || PW <= 0 || M <= 0) return OG_OK;
    hipStream_t s = (hipStream_t)stream;
    PackArgs p;
    p.w = w; p.wt = wt; p.Cout = Cout; p.Cin = Cin; p.Torig = Torig; p.Tg = Tg;
    p.M = M; p.Mpad = (M + 127) / 128 * 128; p.Ck = C; p.Cp = (C + 15) / 16 * 16;
    p.transpose = transpose;
    for (int t = 0; t < OG_MAX_TAPS; ++t) p.src_tap[t] = (signed char)(t < Tg ? src_tap[t] : -1);
    const long ptotal = (long)Tg * p.Cp * p.Mpad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(og_stream_grid(ptotal, 256)), dim3(256), 0, s, p);
    int rc = og_launch_status();
    if (rc != OG_OK) return rc;

    IgemmArgs a;
    a.x = x; a.wt = wt; a.bias = bias; a.y = y;
    a.N = N; a.C = C; a.H = H; a.W = W;
    a.LH = upsample ? 2 * H : H; a.LW = upsample ? 2 * W : W;
    a.M = M; a.Mpad = p.Mpad; a.K = C * Tg; a.Kpad = Tg * p.Cp; a.T = Tg; a.Cp = p.Cp;
    a.m_begin = 0; a.m_end = M;
    a.PH = PH; a.PW = PW; a.OHf = OHf; a.OWf = OWf;
    a.osh = osh; a.osw = osw; a.ooh = ooh; a.oow = oow;
    a.stride = stride; a.pad_mode = pad_mode; a.upsample = upsample; a.act = act;
    a.ksplit_steps = 0;
    for (int t = 0; t < OG_MAX_TAPS; ++t) {
        const int h = t < Tg ? dh[t] : 0, w_ = t < Tg ? dw[t] : 0;
        a.tap[t] = (int)(((unsigned)w_ << 16) | ((unsigned)h & 0xffffu));
    }
    if (!(osh == 1 && osw == 1 && PH == OHf && PW == OWf) && (bias || act)) return OG_BAD_ARGS;
    return run_igemm(a, s, y_prezeroed);
}

// dw must be zero-initialised by the caller (or hold a gradient to accumulate into).
int objgan_conv_wgrad(const float* x, const float* dy, float* dw,
                      int N, int Cin, int H, int W, int upsample, int pad_mode,
                      int Cout, int OH, int OW, int ksize, int stride, int pad,
                      void* stream) {
    if (ksize != 1 && ksize != 3 && ksize != 4) return OG_BAD_ARGS;
    if (N <= 0 || Cout <= 0 || Cin <= 0) return OG_OK;
    hipStream_t s = (hipStream_t)stream;
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W;
    a.LH = upsample ? 2 * H : H; a.LW = upsample ? 2 * W : W;
    a.Cout = Cout; a.OH = OH; a.OW = OW;
    a.stride = stride; a.pad = pad; a.pad_mode = pad_mode; a.upsample = upsample;
    a.ncol = Cin * ksize * ksize;
    const int Npix = N * OH * OW;

    RowPart parts[3];
    const int np = og_row_parts(Cout, parts);
    for (int part = 0; part < np; ++part) {
        a.m_begin = parts[part].m_begin; a.m_end = parts[part].m_end;
        const int cfg = parts[part].cfg;
        const int bm = cfg == 0 ? 128 : (cfg == 1 ? 64 : 32);
        const int bn = cfg == 0 ? 128 : 256;
        const int tiles = og_cdiv(a.m_end - a.m_begin, bm) * og_cdiv(a.ncol, bn);
        // split K (pixels) so that the grid covers the 256 CUs a few times over
        int splits = og_cdiv(256 * 4, tiles);
        const int max_splits = og_cdiv(Npix, 256);   // >= 8 K steps per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        int pps = og_cdiv(Npix, splits);
        pps = (pps + 31) / 32 * 32;
        splits = og_cdiv(Npix, pps);
        a.pix_per_split = pps;
        dim3 grid(tiles, splits);
#define OG_WG(KS)                                                                              \
        if (cfg == 0) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 2, 2>), grid, dim3(256), 0, s, a);       \
        else if (cfg == 1) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 2>), grid, dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 1>), grid, dim3(256), 0, s, a);
        ProfRec* pr = prof_begin(prof_cat(1, ksize, cfg),
                                 2.0 * (a.m_end - a.m_begin) * (double)a.ncol * (double)Npix, s);
        if (ksize == 1) { OG_WG(1) } else if (ksize == 3) { OG_WG(3) } else { OG_WG(4) }
        prof_end(pr, s);
#undef OG_WG
        int rc = og_launch_status();
        if (rc != OG_OK) return rc;
    }
    return OG_OK;
}

// ---- profiling control (see the note above the host section) ---------------------------------
int objgan_prof_enable(int on) {
    g_prof_on = on ? 1 : 0;
    if (on) g_prof_n = 0;
    return OG_OK;
}

// Sums the recorded launches per category (the caller must have synchronised the device).
// ms, flops, count: arrays of 32.  Categories: igemm (taps 1/4/9/16) x (tile 128x128, 64x256,
// 32x256) = 0..11, wgrad (ksize 1/3/4) x tile = 12..20.
int objgan_prof_collect(double* ms, double* flops, long* count) {
    for (int i = 0; i < OG_PROF_CATS; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
        ms[g_prof[i].cat] += t; flops[g_prof[i].cat] += g_prof[i].flops; count[g_prof[i].cat] += 1;
    }
    g_prof_n = 0;
    return OG_OK;
}

}  // extern "C"
