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
// the tap geometry -- bounds / reflect / upsample / offset -- is evaluated once per tap and lane).
//
// Kernels in this file, in the order they are chosen:
//   conv_thin3x3_kernel / conv_thin_kernel   M <= 32 outputs: direct fp32 VALU convolution
//   conv_igemm3_kernel<TM>                   everything else: (32*TM) x 128 tile, pixel fragments
//                                            straight from the gather registers, filter rows via LDS
//   conv_wgrad3_kernel<TM<=2> / conv_wgrad2_kernel<TM>   weight gradient on the same tiling
//   conv_igemm_kernel / conv_wgrad_kernel    the first-generation 128x128 / 64x256 / 32x256 kernels:
//                                            kept for tensors beyond the 2 GiB reach of a buffer
//                                            descriptor and for weight gradients of maps narrower
//                                            than 8 pixels
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" long objgan_conv_packed_floats(int M, int C, int T);
static void og_absmax_launch(const float* x, long n, float* out, hipStream_t s);

#include "conv_igemm3.h"

// conv_igemm_rec.hip: the instances of conv_igemm3_kernel that read pre-split fp16 records (math 5)
int og_launch_igemm3_rec(const IgemmArgs& a, int TM, int nw, int ng, dim3 grid, hipStream_t s);
int og_launch_wgrad_rec(const WgradArgs& a, int tm, int nw, dim3 grid, int ksize, int Cp, int dyp, hipStream_t s);
void og_launch_h2_pair(const float* x, const float* xmax, float* out, long n, hipStream_t s);
int og_launch_wgrad_rec2(const WgradArgs& a, int tm, dim3 grid, int ksize, int Cp, hipStream_t s);

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

    const int kt0 = 0;                         // (this kernel is never split along K: no fp32 atomics in the library's convolutions)
    const int nk = a.Kpad / BK;
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
                    if (a.bias) v += a.bias[m];
                    v = og_act(v, a.act);
                    yb[(size_t)m * a.OHf * a.OWf] = v;
                }
            }
        }
    }
}


// =============================================================================================
// Thin outputs (M <= 32 channels: the 80->12 / 80->24 layout-map stems, to-RGB, data gradients
// down to the 3- / 15-channel discriminator inputs).  A 32-row MFMA tile would spend most of its
// rows on padding; the fp32 VALU has the same peak rate as the fp32 MFMA, so these run as a direct
// convolution: one thread = one output pixel with all M accumulators in registers, the filter bank
// (packed [c][t][MT]) read through the scalar cache into SGPR operands of v_fmac, the T taps of a
// channel as T coalesced buffer loads whose per-lane offsets (bounds / reflection / upsample) are
// computed once per thread.  No LDS, no barriers.
template <int MT, int T, int PX>
__global__ __launch_bounds__(256) void conv_thin_kernel(const IgemmArgs a) {
    // PX output pixels per thread (256 apart): every SGPR filter operand feeds PX FMAs, which
    // keeps the scalar cache (shared between CUs) off the critical path.
    const int Npix = a.N * a.PH * a.PW;
    const int HW = a.H * a.W;
    const int ppi = a.PH * a.PW;
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * a.C * HW * 4u), OG_BUF_FLAGS);

    unsigned voff[PX][T];
    bool pix_ok[PX];
    int on[PX], oa[PX], ob[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int pix = (blockIdx.x * PX + j) * 256 + threadIdx.x;
        pix_ok[j] = pix < Npix;
        const int pp = pix_ok[j] ? pix : 0;
        const int n = pp / ppi;
        const int rem = pp - n * ppi;
        const int pa = rem / a.PW;
        const int pb = rem - pa * a.PW;
        on[j] = n; oa[j] = pa; ob[j] = pb;
        const int ihb = pa * a.stride, iwb = pb * a.stride;
        const unsigned img_off = (unsigned)n * (unsigned)a.C * (unsigned)HW;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tp = a.tap[t];
            const int ih = ihb + ((tp << 16) >> 16);
            const int iw = iwb + (tp >> 16);
            int ihr = ih < 0 ? -ih : ih;
            int iwr = iw < 0 ? -iw : iw;
            ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
            iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
            const bool inb = ((unsigned)ih < (unsigned)a.LH) && ((unsigned)iw < (unsigned)a.LW);
            const bool ok = pix_ok[j] && (refl || inb);
            const int ihs = (refl ? ihr : ih) >> us;
            const int iws = (refl ? iwr : iw) >> us;
            voff[j][t] = ok ? (img_off + (unsigned)(ihs * a.W + iws)) * 4u : OG_OOB;
        }
    }

    float acc[PX][MT];
#pragma unroll
    for (int j = 0; j < PX; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = 0.f;
    const float* __restrict__ wp = a.wt;
    // the taps of the next channel are always in flight behind the FMAs of the current one (the
    // bank carries one zero channel of padding, so an odd C needs no branch)
    auto load_taps = [&](float (&xv)[PX][T], int c) {
        const int so = min(c, a.C - 1) * HW * 4;
#pragma unroll
        for (int j = 0; j < PX; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t)
                xv[j][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, voff[j][t], so, 0));
    };
    auto fma_taps = [&](const float (&xv)[PX][T], int c) {
        const float* __restrict__ wc = wp + (size_t)c * (T * MT);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float wv = wc[t * MT + m];
#pragma unroll
                for (int j = 0; j < PX; ++j) acc[j][m] = fmaf(wv, xv[j][t], acc[j][m]);
            }
    };
    float xa[PX][T], xb[PX][T];
    load_taps(xa, 0);
    for (int c = 0; c < a.C; c += 2) {
        load_taps(xb, c + 1);
        fma_taps(xa, c);
        load_taps(xa, c + 2);
        fma_taps(xb, c + 1);
    }

    const size_t plane = (size_t)a.OHf * a.OWf;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        if (!pix_ok[j]) continue;
        const int oh = oa[j] * a.osh + a.ooh;
        const int ow = ob[j] * a.osw + a.oow;
        float* yb = a.y + (size_t)on[j] * a.M * plane + (size_t)oh * a.OWf + ow;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < a.M) {
                float v = acc[j][m];
                if (a.bias) v += a.bias[m];
                if (MT <= 4) v = og_act(v, a.act);
                else v = a.act == OG_ACT_LRELU ? (v > 0.f ? v : 0.2f * v) : (a.act == OG_ACT_RELU ? fmaxf(v, 0.f) : v);
                yb[(size_t)m * plane] = v;
            }
        }
    }
}

// 3x3 / stride 1 / pad 1 specialisation of the thin kernel (layout-map stems, to-RGB): one thread =
// one output COLUMN of R consecutive rows.  The (R+2) x 3 input window of a channel is loaded once
// (lanes = consecutive columns: fully coalesced dwords) and serves all R pixels -- (R+2)*3/R loads per
// pixel and channel instead of 9; the generic kernel is bound by the vector-memory issue rate of
// its nine tap loads, not by the FMAs.
template <int MT, int R>
__global__ __launch_bounds__(256) void conv_thin3x3_kernel(const IgemmArgs a) {
    const int HW = a.H * a.W;
    const int strips = (a.PH + R - 1) / R;
    const int per_img = strips * a.PW;
    const int total = a.N * per_img;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const bool t_ok = gid < total;
    const int g = t_ok ? gid : 0;
    const int n = g / per_img;
    const int rem = g - n * per_img;
    const int sr = rem / a.PW;
    const int pb = rem - sr * a.PW;
    const int pa0 = sr * R;
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * a.C * HW * 4u), OG_BUF_FLAGS);
    const unsigned img_off = (unsigned)n * (unsigned)a.C * (unsigned)HW;

    unsigned voff[R + 2][3];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
        const int ih = pa0 + r - 1;
        int ihr = ih < 0 ? -ih : ih;
        ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
        ihr = ihr < 0 ? 0 : ihr;                      // rows past the last strip row (unused)
        const bool rok = t_ok && (refl ? (ih <= a.LH) : ((unsigned)ih < (unsigned)a.LH));
        const int ihs = (refl ? ihr : ih) >> us;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int iw = pb + c - 1;
            int iwr = iw < 0 ? -iw : iw;
            iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
            const bool ok = rok && (refl || (unsigned)iw < (unsigned)a.LW);
            const int iws = (refl ? iwr : iw) >> us;
            voff[r][c] = ok ? (img_off + (unsigned)(ihs * a.W + iws)) * 4u : OG_OOB;
        }
    }

    float acc[R][MT];
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = 0.f;
    const float* __restrict__ wp = a.wt;
    auto load_win = [&](float (&xv)[R + 2][3], int c) {
        const int so = min(c, a.C - 1) * HW * 4;
#pragma unroll
        for (int r = 0; r < R + 2; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                xv[r][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, voff[r][q], so, 0));
    };
    auto fma_win = [&](const float (&xv)[R + 2][3], int c) {
        const float* __restrict__ wc = wp + (size_t)c * (9 * MT);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float wv = wc[(kh * 3 + kw) * MT + m];
#pragma unroll
                    for (int j = 0; j < R; ++j) acc[j][m] = fmaf(wv, xv[j + kh][kw], acc[j][m]);
                }
    };
    float xa[R + 2][3], xb[R + 2][3];
    load_win(xa, 0);
    for (int c = 0; c < a.C; c += 2) {
        load_win(xb, c + 1);
        fma_win(xa, c);
        load_win(xa, c + 2);
        fma_win(xb, c + 1);
    }

    if (!t_ok) return;
    const size_t plane = (size_t)a.OHf * a.OWf;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (pa0 + j >= a.PH) break;
        float* yb = a.y + (size_t)n * a.M * plane + (size_t)(pa0 + j) * a.OWf + pb;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < a.M) {
                float v = acc[j][m];
                if (a.bias) v += a.bias[m];
                if (MT <= 4) v = og_act(v, a.act);
                else v = a.act == OG_ACT_LRELU ? (v > 0.f ? v : 0.2f * v) : (a.act == OG_ACT_RELU ? fmaxf(v, 0.f) : v);
                yb[(size_t)m * plane] = v;
            }
        }
    }
}

// Data gradient of a 4x4 / stride-2 / pad-1 convolution w.r.t. an input of <= 32 channels (the first convolution of the
// shape / object discriminators: 12 layout-code channels, 3 image channels), all four output parity phases in ONE launch.
// The per-phase form (conv_thin_kernel, one launch per phase) reads dy four times -- every phase walks the whole
// gradient tensor for a quarter of the output pixels: 16 launches and 6.4 GB of reads per step for the 96 -> 12 layers at
// 256 x 256.  Here a thread owns a SOURCE position (n, a, b) of dy and one ROW parity pa = blockIdx.y: it loads the two
// rows a + pa - 1, a + pa of the 3-wide neighbourhood once per channel (6 values) and produces the two column phases of
// output row 2a + pa -- phase (pa, pb) uses both rows and columns b + {0, -1} (pb = 0) or b + {1, 0} (pb = 1).  dy is read
// twice instead of four times, every output element is written once (8-byte stores of the column pair), same fp32 VALU
// arithmetic: filter bank through the scalar cache (2 phases x 4 taps x MT scalars per channel -- all four phases in one
// thread would need 16 MT and spill SGPRs by the hundred), PX source positions per thread share every SGPR operand.
// Banks: the four phase banks of the thin layout, [Cout + 1][4 taps][MT] each (tap t = i * 2 + j: row choice i, column
// choice j; objgan_conv_dgrad_s2_thin packs them).
template <int MT, int PX>
__global__ __launch_bounds__(256) void conv_thin_ph4_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            float* __restrict__ y, int N, int C, int H, int W, int M) {
    const int HW = H * W;
    const int Npos = N * HW;
    const int pa = blockIdx.y;                       // row parity of the output rows this workgroup writes
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)x, 0, (int)((unsigned)N * C * HW * 4u), OG_BUF_FLAGS);
    unsigned voff[PX][6];
    bool pos_ok[PX];
    int pn[PX], pr[PX], pb_[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int pos = (blockIdx.x * PX + j) * 256 + threadIdx.x;
        pos_ok[j] = pos < Npos;
        const int pp = pos_ok[j] ? pos : 0;
        const int n = pp / HW;
        const int rem = pp - n * HW;
        const int a = rem / W;
        const int b = rem - a * W;
        pn[j] = n; pr[j] = a; pb_[j] = b;
        const unsigned img_off = (unsigned)n * (unsigned)C * (unsigned)HW;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ih = a + pa - 1 + r, iw = b + c - 1;
                const bool ok = pos_ok[j] && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                voff[j][r * 3 + c] = ok ? (img_off + (unsigned)(ih * W + iw)) * 4u : OG_OOB;
            }
    }
    float acc[PX][2][MT];
#pragma unroll
    for (int j = 0; j < PX; ++j)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][pb][m] = 0.f;
    const long bank = (long)(C + 1) * 4 * MT;       // floats per phase bank (one zero channel of padding)
    const float* __restrict__ wp = wt + (long)(pa * 2) * bank;
    auto load_nb = [&](float (&xv)[PX][6], int c) {
        const int so = min(c, C - 1) * HW * 4;
#pragma unroll
        for (int j = 0; j < PX; ++j)
#pragma unroll
            for (int q = 0; q < 6; ++q)
                xv[j][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, voff[j][q], so, 0));
    };
    auto fma_nb = [&](const float (&xv)[PX][6], int c) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const float* __restrict__ wc = wp + pb * bank + (size_t)c * (4 * MT);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // tap t = i * 2 + j: row choice i -> local row 1 - i (dh = 0, -1 for pa = 0; 1, 0 for pa = 1);
                // column choice j -> column pb ? 2 - j : 1 - j of the 3-wide neighbourhood
                const int r = 1 - (t >> 1);
                const int q = pb ? 2 - (t & 1) : 1 - (t & 1);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float wv = wc[t * MT + m];
#pragma unroll
                    for (int j = 0; j < PX; ++j) acc[j][pb][m] = fmaf(wv, xv[j][r * 3 + q], acc[j][pb][m]);
                }
            }
        }
    };
    float xa[PX][6], xb[PX][6];
    load_nb(xa, 0);
    for (int c = 0; c < C; c += 2) {                 // (an odd C runs one step into the bank's zero channel)
        load_nb(xb, c + 1);
        fma_nb(xa, c);
        load_nb(xa, c + 2);
        fma_nb(xb, c + 1);
    }
    const int OWf = 2 * W;
    const size_t plane = (size_t)(2 * H) * OWf;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        if (!pos_ok[j]) continue;
        float* yb = y + (size_t)pn[j] * M * plane + (size_t)(2 * pr[j] + pa) * OWf + 2 * pb_[j];
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < M) *reinterpret_cast<float2*>(yb + (size_t)m * plane) = make_float2(acc[j][0][m], acc[j][1][m]);
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
    int m_major;         // 0: wt[K][Mpad] (v1 kernels), 1: wt[M][Kpad] (k contiguous, v2 kernel),
                         // 2: wt[Ck][Tg][Mpad] with Mpad = MT (thin direct kernel)
                         // 3: bf16 wt[M][Krow], Krow = Kpad rounded up to 32 (bf16 MFMA kernels)
                         // 4: bf16x3 split wt[M][Kpad/16][3][16]: every fp32 entry as its exact three-way bf16
                         //    split h + m + l (og_split8), the three pieces of a 16-deep K step back to back
                         // 5: fp16x2 split wt[M][Kpad/16][2][16] fp16: w * 2^10 = h + l
    int kgroup;          // row-major banks (m_major 1 / 3 / 4 / 5): chunks per K group (see og_kstep)
    const float* wmax;   // m_major 5: the OG_AMAX_SLOTS partial maxima of |w| (behind the bank, written by absmax_w_* before the pack)
    int wexp;            // m_major 5: scale exponent derived from them (set inside the pack kernels)
    signed char src_tap[OG_MAX_TAPS];
};

// K walk order of conv_igemm3_kernel and of its banks.  Tap-major (all channels of tap 0, then tap 1, ...) keeps the
// tap geometry out of the inner steps, but every tap re-reads the SAME source pixels one full channel sweep later: at
// 128 x 128 x 194..388 channels a sweep of the workgroups of one XCD is 6-16 MB, the 4 MB L2 has long lost the lines
// and every tap fetches them again from HBM / MALL (r03 PMC: 3.9-4.1x the algorithmic bytes).  Walking the channels in
// GROUPS of G chunks -- all taps of a group back to back -- bounds the reuse distance to T * G steps.
//   step(t, chunk c): g = c / G; steps of the full groups before it + t * (chunks in group g) + (c - g * G)
__host__ __device__ __forceinline__ int og_kstep(int t, int c16, int spt, int T, int G) {
    const int g = c16 / G;
    const int Gg = min(G, spt - g * G);
    return g * T * G + t * Gg + (c16 - g * G);
}

// Work items of a job.  Row-major banks (m_major 1 / 3: wt[m][t*Cp + ck]) are packed per (m, ck) PAIR: a
// thread reads the Torig taps of its pair -- one contiguous 36..64-byte run of w, adjacent pairs adjacent runs
// for the forward banks -- and writes one element per GEMM tap, adjacent threads adjacent addresses.  The first
// version walked the bank element by element: for the transposed (data-gradient) banks adjacent elements are a
// whole filter apart in w, every 4-byte read pulled its own 64-byte line and the line was gone from the L2
// before its neighbours were wanted (PMC: 1.2 GB fetched per launch for 0.1 GB of banks; 2.2 ms per step).
__device__ __forceinline__ long pack_total(const PackArgs& a, int Kpad, int Krow) {
    return a.m_major == 2 ? (long)(a.Ck + 1) * a.Tg * a.Mpad        // + one zero channel
                          : (a.m_major ? (long)a.M * a.Cp : (long)Kpad * a.Mpad);
}

// element i of the small layouts (0: wt[k][Mpad], 2: wt[ck][t][MT])
__device__ __forceinline__ void pack_element(const PackArgs& a, unsigned i, int Kpad) {
    int m, t, ck;
    if (a.m_major == 2) {
        const unsigned r = i / (unsigned)a.Mpad;
        m = (int)(i - r * (unsigned)a.Mpad);
        t = (int)(r % (unsigned)a.Tg);
        ck = (int)(r / (unsigned)a.Tg);
    } else {
        const unsigned k = i / (unsigned)a.Mpad;
        m = (int)(i - k * (unsigned)a.Mpad);
        t = (int)(k / (unsigned)a.Cp);
        ck = (int)k - t * a.Cp;
    }
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

// element k of the bank row starting at `row` (element units of the layout)
__device__ __forceinline__ void pack_store(const PackArgs& a, size_t row, int k, float v) {
    if (a.m_major == 4) {
        __bf16* o = reinterpret_cast<__bf16*>(a.wt) + row + (size_t)(k >> 4) * 48 + (k & 15);
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        o[0] = h; o[16] = m; o[32] = (__bf16)(r1 - (float)m);
    } else if (a.m_major == 5) {         // fp16x2: w * 2^wexp = h + l (max |w| * 2^wexp in [2^14, 2^15))
        _Float16* o = reinterpret_cast<_Float16*>(a.wt) + row + (size_t)(k >> 4) * 32 + (k & 15);
        const float sv = v * og_pow2(a.wexp);
        const _Float16 h = (_Float16)sv;
        o[0] = h; o[16] = (_Float16)(sv - (float)h);
    } else if (a.m_major == 3) {
        reinterpret_cast<__bf16*>(a.wt)[row + k] = (__bf16)v;
    } else {
        a.wt[row + k] = v;
    }
}

// pair i = m * Cp + ck of the row-major layouts
__device__ __forceinline__ void pack_pair(const PackArgs& a, unsigned i, int Kpad, int Krow) {
    const unsigned m = i / (unsigned)a.Cp;
    const int ck = (int)(i - m * (unsigned)a.Cp);
    const bool live = ck < a.Ck;
    const int co = a.transpose ? ck : (int)m;
    const int ci = a.transpose ? (int)m : ck;
    const float* src = a.w + ((size_t)co * a.Cin + ci) * a.Torig;
    const size_t row = (size_t)m * Krow;
    if (a.Torig == 16) {
        // 4x4 filters (most of the bank bytes): the pair's 16 taps are one aligned 64-byte line -- four 16-byte
        // loads instead of sixteen 4-byte ones that each walk 64 different lines per wave
        float r[16];
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f = live ? s4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            r[4 * q] = f.x; r[4 * q + 1] = f.y; r[4 * q + 2] = f.z; r[4 * q + 3] = f.w;
        }
        for (int t = 0; t < a.Tg; ++t) {
            const int st = a.src_tap[t];                 // uniform: a select chain, no register indexing
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) v = st == j ? r[j] : v;
            pack_store(a, row, og_kstep(t, ck >> 4, a.Cp >> 4, a.Tg, a.kgroup) * 16 + (ck & 15), v);
        }
    } else {
        for (int t = 0; t < a.Tg; ++t) {
            const int st = a.src_tap[t];
            const float v = (live && st >= 0) ? src[st] : 0.f;
            pack_store(a, row, og_kstep(t, ck >> 4, a.Cp >> 4, a.Tg, a.kgroup) * 16 + (ck & 15), v);
        }
    }
    if (a.m_major == 3 && ck < Krow - Kpad)               // bf16 rows are padded to a multiple of 32
        reinterpret_cast<__bf16*>(a.wt)[row + Kpad + ck] = (__bf16)0.f;
}

__device__ __forceinline__ void pack_item(const PackArgs& a, long i, int Kpad, int Krow) {
    if (a.m_major == 1 || a.m_major >= 3) pack_pair(a, (unsigned)i, Kpad, Krow);
    else pack_element(a, (unsigned)i, Kpad);
}

// fp32 [N][C][HW] -> bf16 (RNE) channel-blocked [N][Cp/16][HW][16], channels C..Cp-1 zero: the pixel operand of the
// bf16 mode (16 channels of a pixel = 32 contiguous bytes, neighbouring pixels of a chunk contiguous).
// 64 channels x 64 pixels per workgroup through LDS: 256-byte rows in, 1 KiB runs per wave out.
__global__ __launch_bounds__(256) void nchw_to_nhwc_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out,
                                                                int C, int HW, int Cp) {
    __shared__ float tile[64][65];
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
    const int cg = threadIdx.x >> 5;                   // 8 channels = one 16-byte store; a wave = one 16-channel chunk
    if (c0 + cg * 8 >= Cp) return;
    const int chunk = (c0 + cg * 8) >> 4, half = cg & 1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pp = (threadIdx.x & 31) + 32 * it;
        const int p = p0 + pp;
        if (p >= HW) continue;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)tile[cg * 8 + j][pp];
        *reinterpret_cast<bf16x8*>(out + (((size_t)n * (Cp / 16) + chunk) * HW + p) * 16 + half * 8) = v;
    }
}

// Partial maxima of |w| for the fp16x2 banks: 64 workgroups own the OG_AMAX_SLOTS slots (common.h og_amax_own)
__device__ __forceinline__ void absmax_w_block(const float* __restrict__ w, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += 64L * 256) m = fmaxf(m, fabsf(w[i]));
    og_amax_own(og_block_max(m, red), out, blockIdx.x, 64);
}
__global__ __launch_bounds__(256) void absmax_w_kernel(const float* __restrict__ w, long n, float* __restrict__ out) {
    absmax_w_block(w, n, out);
}
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackArgs a_in) {
    PackArgs a = a_in;
    if (a.m_major == 5) { og_fp16_saturate(); a.wexp = og_h2_exponent(a.wmax, threadIdx.x & 63); }
    const int Kpad = a.Tg * a.Cp;
    const int Krow = a.m_major == 3 ? (Kpad + 31) / 32 * 32 : (a.m_major == 4 ? 3 * Kpad : (a.m_major == 5 ? 2 * Kpad : Kpad));
    const long total = pack_total(a, Kpad, Krow);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        pack_item(a, i, Kpad, Krow);
}

// Many banks in one launch: blockIdx.y = job.  After an optimizer step every cached bank of the updated
// network is stale at once -- ~60 banks per network, 499 7-us launches per training step when each is
// re-packed at its next use; the host keeps the jobs of a network in a device table instead and refreshes
// them right behind the Adam kernel (objgan_conv_pack_jobs_run).  Bank sizes span three orders of magnitude
// (a 3-channel to-RGB bank .. 768 x 1024 x 9): a workgroup takes OG_PACK_CHUNK consecutive work items per
// sweep and workgroups beyond a small bank's end leave at once.
#define OG_PACK_BLOCKS 256
#define OG_PACK_CHUNK 512
__global__ __launch_bounds__(256) void absmax_w_jobs_kernel(const PackArgs* __restrict__ jobs) {
    const PackArgs a = jobs[blockIdx.y];
    if (a.m_major != 5) return;
    absmax_w_block(a.w, (long)a.Cout * a.Cin * a.Torig, const_cast<float*>(a.wmax));
}

__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackArgs* __restrict__ jobs) {
    PackArgs a = jobs[blockIdx.y];
    if (a.m_major == 5) { og_fp16_saturate(); a.wexp = og_h2_exponent(a.wmax, threadIdx.x & 63); }
    const int Kpad = a.Tg * a.Cp;
    const int Krow = a.m_major == 3 ? (Kpad + 31) / 32 * 32 : (a.m_major == 4 ? 3 * Kpad : (a.m_major == 5 ? 2 * Kpad : Kpad));
    const long total = pack_total(a, Kpad, Krow);
    for (long base = (long)blockIdx.x * OG_PACK_CHUNK; base < total; base += (long)gridDim.x * OG_PACK_CHUNK) {
#pragma unroll
        for (int u = 0; u < OG_PACK_CHUNK / 256; ++u) {
            const long i = base + u * 256 + threadIdx.x;
            if (i < total) pack_item(a, i, Kpad, Krow);
        }
    }
}

// out[e] = act(bias[(e / HW) % M] + sum_{s < splits} ws[s * ws_stride + seg_off + e]): the second level of split-K
// (conv_igemm3_kernel writes the partial tiles), splits summed in order.
// ymax (may be null; zeroed by the caller): the partial maxima of |out| for an fp16x2 consumer, added with one integer
// atomicMax per workgroup -- round 6: this was a separate pass over the output behind every split launch with a fused
// ReLU / LeakyReLU (the Inception chain's small maps).
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ ws, int splits, long ws_stride,
                                                             long seg_off, float* __restrict__ out, long total,
                                                             const float* __restrict__ bias, int M, int HW, int act,
                                                             float* __restrict__ ymax) {
    __shared__ float red[4];
    float vmax = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const float* p = ws + seg_off + e;
        float v = p[0];
        for (int k = 1; k < splits; ++k) v += p[(size_t)k * ws_stride];
        if (bias) v += bias[(e / HW) % M];
        v = og_act(v, act);
        out[e] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    if (ymax) og_amax_atomic(og_block_max(vmax, red), ymax, blockIdx.x);
}

// dw rows <- sum over the splits of a weight-gradient launch (WgradArgs::ws): local row r of the slot is dw row
// m_begin + r for r < main_rows, extra row xr_begin + (r - main_rows) behind them.
__global__ __launch_bounds__(256) void wgrad_combine_kernel(const float* __restrict__ ws, int splits, long ws_stride,
                                                            float* __restrict__ dw, int ncol, int m_begin, int main_rows,
                                                            int xr_begin, long total, int accumulate) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const float* p = ws + e;
        float v = p[0];
        for (int k = 1; k < splits; ++k) v += p[(size_t)k * ws_stride];
        const long r = e / ncol;
        const int col = (int)(e - r * ncol);
        const long row = r < main_rows ? m_begin + r : xr_begin + (r - main_rows);
        float* o = dw + row * ncol + col;
        *o = accumulate ? *o + v : v;
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

// (WgradArgs / og_wgrad_store: conv_igemm3.h)
__device__ __forceinline__ void og_wgrad_store_xr(const WgradArgs& a, int j, int col, float v, int split) {
    if (a.ws) {
        a.ws[(size_t)split * a.ws_stride + (size_t)(a.m_end - a.m_begin + j) * a.ncol + col] = v;
    } else {
        float* p = a.dw + (size_t)(a.xr_begin + j) * a.ncol + col;
        *p = a.accumulate ? *p + v : v;
    }
}

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
                if (m < a.m_end) og_wgrad_store(a, m, col, acc[i][j][r], blockIdx.y);
            }
        }
    }
}

// ---- weight gradient, v2 -----------------------------------------------------------------------
// Same tiling / LDS layout / buffer-load scheme as conv_igemm3_kernel (with both operands in LDS): tile (32*TM) x 128 columns
// (column = ci*T + t), K = 16 output pixels per step.  Requires OW % 8 == 0 and (OH*OW) % 16 == 0
// (every layer of the hot path above 4x4 maps), so that the eight pixels a thread gathers per step
// lie in one output row and a K step lies in one image: the pixel part of every address is then a
// SCALAR (n, oh, ow0 .. ow0+7), the per-lane part is the column's (ci, kh, kw) -- dy rows are read
// as aligned 16-byte pieces with a constant per-lane offset, x elements as dwords whose validity
// (zero padding) rides on the buffer range check.
template <int TM, int MATH = 0, int XR = 0>
__global__ __launch_bounds__(256) void conv_wgrad2_kernel(const WgradArgs a, const int KS) {
    // MATH as in conv_igemm3_kernel.  SP (bf16x3): both operands are activations, so both are split when they are
    // written to LDS -- by the thread that loaded them, once per element (not once per wave that reads them) --
    // into the row image [h 16 | m 16 | l 16] bf16 + 16 bytes of padding (112-byte pitch).
    constexpr bool BF = MATH == 1, SP = MATH == 2;
    constexpr int BM = 32 * TM;
    constexpr int BN = 128;
    constexpr int BK = 16;
    constexpr int LD = SP ? 28 : BK + 4;
    constexpr int NA4 = BM * 4;
    constexpr int NA_PER = (NA4 + 255) / 256;
    constexpr int AROWS = BM + XR;                     // dy rows in LDS (XR extra rows, see WgradArgs)
    constexpr int TILE = (AROWS + BN) * LD;
    static_assert(XR == 0 || MATH == 0, "extra rows: fp32 only");

    __shared__ __attribute__((aligned(16))) float lds[2 * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int T = KS * KS;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (a.ncol + BN - 1) / BN;
    // XCD placement over both grid dimensions: all tiles of a pixel split on one XCD (they read the same dy / x pixels),
    // an XCD takes a contiguous range of splits (see conv_wgrad_bfb_kernel)
    const int nwg = tiles_m * tiles_n;
    const int vid = og_xcd_remap(blockIdx.x + nwg * blockIdx.y, nwg * gridDim.y);
    const int split = vid / nwg;
    const int wg = vid - split * nwg;
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int c0 = tile_n * BN;

    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;
    const int Npix = a.N * OHW;
    const int p_begin = split * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    if (p_begin >= p_end) return;

    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * a.Cin * HW * 4u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.N * a.Cout * OHW * 4u), OG_BUF_FLAGS);

    // ---- B (gathered x) geometry: thread = (column of the tile, k half)
    const int bc = tid & (BN - 1);
    const int bg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int col = c0 + bc;
    const bool col_ok = col < a.ncol;
    int dh, dw;
    unsigned ci_off;
    {
        const int cc = col_ok ? col : 0;
        const int ci = cc / T;
        const int t = cc - ci * T;
        const int kh = t / KS;
        dh = kh - a.pad;
        dw = (t - kh * KS) - a.pad;
        ci_off = (unsigned)ci * (unsigned)HW;
    }
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;

    // ---- A (dy) geometry: float4 idx -> (row, quarter); constant per-lane offset
    unsigned avoff[NA_PER];
    int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 2, q = idx & 3;
        const bool on = (NA4 % 256 == 0 || idx < NA4) && (m0 + row) < a.m_end;
        avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + q * 4u) * 4u : OG_OOB;
        alds[i] = (NA4 % 256 == 0 || idx < NA4) ? row * LD + q * (SP ? 2 : 4) : -1;   // SP: 8-byte h piece of 4 pixels
    }

    const bool has_x = XR > 0 && tile_m == 0 && a.xr_count > 0;
    const bool x_loader = has_x && tid < XR * 4;
    const unsigned xvoff = (x_loader && (tid >> 2) < a.xr_count)
        ? ((unsigned)(a.xr_begin + (tid >> 2)) * (unsigned)OHW + (tid & 3) * 4u) * 4u : OG_OOB;
    f32x4 ra[NA_PER];
    f32x4 rax = {0.f, 0.f, 0.f, 0.f};
    float rb[8];
    // scalar pixel state of the next K step to load: image n, offset rem in the image, and the
    // (row, first column) of this wave's eight pixels; advanced incrementally (no divisions)
    int n_ld = p_begin / OHW;
    int rem_ld = p_begin - n_ld * OHW;
    int oh_ld = (rem_ld + bg * 8) / a.OW;
    int ow_ld = (rem_ld + bg * 8) - oh_ld * a.OW;
    auto load_step = [&]() {
        const int n = n_ld, oh = oh_ld, ow0 = ow_ld;
        const int asoff = (n * a.Cout * OHW + rem_ld) * 4;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, avoff[i], asoff, 0));
        if (XR > 0 && has_x)
            rax = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, xvoff, asoff, 0));
        rem_ld += BK;
        ow_ld += BK;
        while (ow_ld >= a.OW) { ow_ld -= a.OW; oh_ld += 1; }
        if (rem_ld >= OHW) {                         // next step starts a new image
            rem_ld = 0; n_ld += 1;
            oh_ld = (bg * 8) / a.OW;
            ow_ld = (bg * 8) - oh_ld * a.OW;
        }
        const int ih = oh * a.stride + dh;
        int ihr = ih < 0 ? -ih : ih;
        ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
        const bool row_ok = col_ok && (refl || (unsigned)ih < (unsigned)a.LH);
        const unsigned rbase = (unsigned)n * (unsigned)a.Cin * (unsigned)HW + ci_off
                             + (unsigned)(((refl ? ihr : ih) >> us) * a.W);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int iw = (ow0 + i) * a.stride + dw;
            int iwr = iw < 0 ? -iw : iw;
            iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
            const bool ok = row_ok && (refl || (unsigned)iw < (unsigned)a.LW);
            const unsigned vo = ok ? (rbase + (unsigned)((refl ? iwr : iw) >> us)) * 4u : OG_OOB;
            rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, vo, 0, 0));
        }
    };
    auto store_step = [&](int buf) {
        float* As = lds + buf * TILE;
        float* Bs = As + AROWS * LD;
        if (SP) {
#pragma unroll
            for (int i = 0; i < NA_PER; ++i) {
                bf16x4 h, m, l;
                og_split4(ra[i], h, m, l);
                if (NA4 % 256 == 0 || alds[i] >= 0) {
                    *reinterpret_cast<bf16x4*>(As + alds[i]) = h;
                    *reinterpret_cast<bf16x4*>(As + alds[i] + 8) = m;
                    *reinterpret_cast<bf16x4*>(As + alds[i] + 16) = l;
                }
            }
            bf16x8 h, m, l;
            og_split8(rb, h, m, l);
            *reinterpret_cast<bf16x8*>(Bs + bc * LD + bg * 4) = h;
            *reinterpret_cast<bf16x8*>(Bs + bc * LD + bg * 4 + 8) = m;
            *reinterpret_cast<bf16x8*>(Bs + bc * LD + bg * 4 + 16) = l;
            return;
        }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % 256 == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
        if (XR > 0 && x_loader) *reinterpret_cast<f32x4*>(As + (BM + (tid >> 2)) * LD + (tid & 3) * 4) = rax;
        f32x4 v0 = {rb[0], rb[1], rb[2], rb[3]}, v1 = {rb[4], rb[5], rb[6], rb[7]};
        *reinterpret_cast<f32x4*>(Bs + bc * LD + bg * 8) = v0;
        *reinterpret_cast<f32x4*>(Bs + bc * LD + bg * 8 + 4) = v1;
    };

    const int nk = (p_end - p_begin) / BK;
    load_step();
    store_step(0);
    if (nk > 1) load_step();
    __syncthreads();

    const int lrow = lane >> 5;
    const int lcol = lane & 31;
    const int a_rd = lcol * LD + lrow * (SP ? 4 : 8);
    const int b_rd = AROWS * LD + (wid * 32 + lcol) * LD + lrow * (SP ? 4 : 8);

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float accx[XR > 0 ? XR : 1];
#pragma unroll
    for (int j = 0; j < (XR > 0 ? XR : 1); ++j) accx[j] = 0.f;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const float* Tl = lds + cur * TILE;
        if (SP) {           // bf16x3: fragments are the pre-split LDS rows; refill behind the first TM MFMAs
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Tl + b_rd);
            const bf16x8 bm = *reinterpret_cast<const bf16x8*>(Tl + b_rd + 8);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Tl + b_rd + 16);
            bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                al[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD + 16);
                ah[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD);
                am[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD + 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(al[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bl, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            if ((kt + 1) < nk) store_step(cur ^ 1);      // (splits the tile loaded one step ago: ~110 VALU)
            if ((kt + 2) < nk) load_step();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bh, acc[i]);
        } else if (BF) {    // bf16 inputs (RNE of the fp32 tiles), one 32x32x16 MFMA per row group and K step
            if ((kt + 1) < nk) store_step(cur ^ 1);      // two-deep register -> LDS pipeline
            if ((kt + 2) < nk) load_step();
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Tl + b_rd);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Tl + b_rd + 4);
            bf16x8 bq;
#pragma unroll
            for (int j = 0; j < 4; ++j) { bq[j] = (__bf16)b0[j]; bq[4 + j] = (__bf16)b1[j]; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD + 4);
                bf16x8 aq;
#pragma unroll
                for (int j = 0; j < 4; ++j) { aq[j] = (__bf16)x0[j]; aq[4 + j] = (__bf16)x1[j]; }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bq, acc[i], 0, 0, 0);
            }
        } else {
            // fp32: the step's 8*TM MFMAs in two halves with the refill of the pipeline BETWEEN them --
            // LDS stores of the next tile, ~130 VALU instructions of gather addressing, 11 global loads.
            // Issue is in order: placed in front of the MFMAs (as the first version had it) that work
            // is exposed every step (98 TFLOP/s); behind the first half it runs in their shadow.
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Tl + b_rd);
            f32x4 a0[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][kk], b0[kk], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((kt + 1) < nk) store_step(cur ^ 1);      // two-deep register -> LDS pipeline
            if ((kt + 2) < nk) load_step();
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Tl + b_rd + 4);
            f32x4 a1[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD + 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][kk], b1[kk], acc[i], 0, 0, 0);
            if (XR > 0 && has_x) {
#pragma unroll
                for (int j = 0; j < XR; ++j) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(Tl + (BM + j) * LD + lrow * 8);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(Tl + (BM + j) * LD + lrow * 8 + 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accx[j] = fmaf(x0[kk], b0[kk], accx[j]);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accx[j] = fmaf(x1[kk], b1[kk], accx[j]);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    if (XR > 0 && has_x) {
#pragma unroll
        for (int j = 0; j < XR; ++j) accx[j] += __shfl_xor(accx[j], 32, 64);
    }
    const int ocol = c0 + wid * 32 + lcol;
    if (ocol >= a.ncol) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
            if (m < a.m_end) og_wgrad_store(a, m, ocol, acc[i][r], split);
        }
    }
    if (XR > 0 && has_x && lrow == 0) {
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (j < a.xr_count) og_wgrad_store_xr(a, j, ocol, accx[j], split);
    }
}


// Weight gradient on the v3 scheme: the gathered-x fragment goes straight to registers (lane =
// (column l & 31 of the wave, pixel half l >> 5)), dy rows through LDS (TM > 1) or direct (TM = 1).
// Requires OW % 8 == 0 and (OH*OW) % 16 == 0 like v2.
// B128: on the stride-1 interior fast path the eight consecutive pixels of a lane are fetched as two
// 16-byte loads (4-byte aligned) instead of eight dwords -- the lanes of a wave sit on different
// (channel, tap) planes, so every gather instruction touches ~20 cache lines.
template <int TM, int MATH = 0, bool B128 = false, int XR = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_wgrad3_kernel(const WgradArgs a, const int KS) {
    constexpr int NT = 64 * NW;                     // NW waves = NW 32-column groups sharing one dy row tile
    // MATH as in conv_igemm3_kernel.  SP (bf16x3): the dy rows are split by their loader thread on the way into
    // LDS (row image [h 16 | m 16 | l 16] bf16, 112-byte pitch), the gathered x fragment in registers; the gather
    // runs two steps ahead (three fragment sets), see conv_igemm3_kernel.
    // H2 (fp16x2, math 4): both operands scaled by their tensors' power-of-two scales, dy rows split into two fp16 pieces
    // on the way into LDS (row image [h 16 | l 16], 80-byte pitch), x in registers; three MFMAs per row group and step.
    constexpr bool BF = MATH == 1, SP = MATH == 2, H2 = MATH == 4, P3 = SP || H2;
    constexpr int BM = 32 * TM;
    constexpr int BN = 32 * NW;
    constexpr int BK = 16;
    constexpr int LD = SP ? 28 : BK + 4;
    constexpr int NA4 = BM * 4;
    constexpr int NA_PER = (NA4 + NT - 1) / NT;
    constexpr int TILE = (BM + XR) * LD;
    constexpr bool ALDS = TM > 1;
    static_assert(XR == 0 || (MATH == 0 && TM > 1), "extra rows: fp32 LDS form only");

    __shared__ __attribute__((aligned(16))) float lds[ALDS ? (P3 ? 3 : 2) * TILE : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5;
    const int lcol = lane & 31;

    const int T = KS * KS;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (a.ncol + BN - 1) / BN;
    // XCD placement over both grid dimensions: all tiles of a pixel split on one XCD (they read the same dy / x pixels),
    // an XCD takes a contiguous range of splits (see conv_wgrad_bfb_kernel)
    const int nwg = tiles_m * tiles_n;
    const int vid = og_xcd_remap(blockIdx.x + nwg * blockIdx.y, nwg * gridDim.y);
    const int split = vid / nwg;
    const int wg = vid - split * nwg;
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int c0 = tile_n * BN;

    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;
    const int Npix = a.N * OHW;
    const int p_begin = split * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    if (p_begin >= p_end) return;

    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * a.Cin * HW * 4u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.N * a.Cout * OHW * 4u), OG_BUF_FLAGS);

    float h2_xs = 1.f, h2_dys = 1.f, h2_inv = 1.f;
    if (H2) {
        const int sx = og_h2_exponent(a.xmax, lane), sd = og_h2_exponent(a.dymax, lane);
        og_fp16_saturate();
        h2_xs = og_pow2(sx); h2_dys = og_pow2(sd); h2_inv = og_pow2_sum(-sx, -sd);
    }

    // ---- column of this lane
    const int col = c0 + wid * 32 + lcol;
    const bool col_ok = col < a.ncol;
    int dh, dw;
    unsigned ci_off;
    {
        const int cc = col_ok ? col : 0;
        const int ci = cc / T;
        const int t = cc - ci * T;
        const int kh = t / KS;
        dh = kh - a.pad;
        dw = (t - kh * KS) - a.pad;
        ci_off = (unsigned)ci * (unsigned)HW;
    }
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;

    // per-lane pixel state: the eight pixels (one output row) of this lane's k half in the next step
    int pn, poh, pow_;
    {
        const int p = p_begin + lrow * 8;
        pn = p / OHW;
        const int r = p - pn * OHW;
        poh = r / a.OW;
        pow_ = r - poh * a.OW;
    }
    auto load_b = [&](float (&rb)[8]) {
        const int ih = poh * a.stride + dh;
        int ihr = ih < 0 ? -ih : ih;
        ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
        const bool row_ok = col_ok && (refl || (unsigned)ih < (unsigned)a.LH);
        const unsigned rbase = (unsigned)pn * (unsigned)a.Cin * (unsigned)HW + ci_off
                             + (unsigned)(((refl ? ihr : ih) >> us) * a.W);
        // Fast path (no upsampling; stride 1 or 2): when the eight taps of every lane of the wave are
        // interior -- or the whole row is padding -- they sit at a constant byte stride from the
        // first one, which folds into the instruction's immediate offset: no per-element address
        // arithmetic.  Spans touching the left / right border (2 of OW/8 per row) take the general path.
        const int iw0 = pow_ * a.stride + dw;
        const bool interior = iw0 >= 0 && iw0 + 7 * a.stride < a.LW;
        if (!us && (a.stride == 1 || a.stride == 2) && __all(interior || !row_ok)) {
            const unsigned vo = row_ok ? (rbase + (unsigned)iw0) * 4u : OG_OOB;
            if (B128 && a.stride == 1) {
                const f32x4 q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vo, 0, 0));
                const f32x4 q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, vo + 16u, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) { rb[i] = q0[i]; rb[4 + i] = q1[i]; }
            } else if (a.stride == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, vo + 4u * i, 0, 0));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, vo + 8u * i, 0, 0));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int iw = (pow_ + i) * a.stride + dw;
                int iwr = iw < 0 ? -iw : iw;
                iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
                const bool ok = row_ok && (refl || (unsigned)iw < (unsigned)a.LW);
                const unsigned vo = ok ? (rbase + (unsigned)((refl ? iwr : iw) >> us)) * 4u : OG_OOB;
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, vo, 0, 0));
            }
        }
        pow_ += BK;
        while (pow_ >= a.OW) { pow_ -= a.OW; poh += 1; }
        while (poh >= a.OH) { poh -= a.OH; pn += 1; }
    };

    // ---- dy rows
    unsigned avoff[NA_PER];
    int alds[NA_PER];
    if (ALDS) {
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            const int idx = tid + NT * i;
            const int row = idx >> 2, q = idx & 3;
            const bool on = (NA4 % NT == 0 || idx < NA4) && (m0 + row) < a.m_end;
            avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + q * 4u) * 4u : OG_OOB;
            alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * LD + q * (P3 ? 2 : 4) : -1;
        }
    }
    const unsigned adir = (m0 + lcol) < a.m_end ? ((unsigned)(m0 + lcol) * (unsigned)OHW + lrow * 8u) * 4u : OG_OOB;
    int n_ld = p_begin / OHW;                        // scalar (image, offset) of the next dy step
    int rem_ld = p_begin - n_ld * OHW;
    auto a_soff = [&]() {
        const int so = (n_ld * a.Cout * OHW + rem_ld) * 4;
        rem_ld += BK;
        if (rem_ld >= OHW) { rem_ld = 0; n_ld += 1; }
        return so;
    };
    const bool has_x = XR > 0 && tile_m == 0 && a.xr_count > 0;
    const bool x_loader = has_x && tid < XR * 4;
    const unsigned xvoff = (x_loader && (tid >> 2) < a.xr_count)
        ? ((unsigned)(a.xr_begin + (tid >> 2)) * (unsigned)OHW + (tid & 3) * 4u) * 4u : OG_OOB;
    f32x4 ra[NA_PER];
    f32x4 rax = {0.f, 0.f, 0.f, 0.f};
    auto load_a = [&]() {
        const int so = a_soff();
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, avoff[i], so, 0));
        if (XR > 0 && has_x)
            rax = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, xvoff, so, 0));
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * TILE;
        if (H2) {
#pragma unroll
            for (int i = 0; i < NA_PER; ++i) {
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f16x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sv = ra[i][j] * h2_dys;
                    h[j] = (_Float16)sv;
                    l[j] = (_Float16)og_sub(sv, (float)h[j]);
                }
                if (NA4 % NT == 0 || alds[i] >= 0) {
                    *reinterpret_cast<f16x4*>(As + alds[i]) = h;
                    *reinterpret_cast<f16x4*>(As + alds[i] + 8) = l;
                }
            }
            return;
        }
        if (SP) {
#pragma unroll
            for (int i = 0; i < NA_PER; ++i) {
                bf16x4 h, m, l;
                og_split4(ra[i], h, m, l);
                if (NA4 % NT == 0 || alds[i] >= 0) {
                    *reinterpret_cast<bf16x4*>(As + alds[i]) = h;
                    *reinterpret_cast<bf16x4*>(As + alds[i] + 8) = m;
                    *reinterpret_cast<bf16x4*>(As + alds[i] + 16) = l;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
        if (XR > 0 && x_loader) *reinterpret_cast<f32x4*>(As + (BM + (tid >> 2)) * LD + (tid & 3) * 4) = rax;
    };
    auto load_adir = [&](f32x4 (&ad)[2]) {
        const int so = a_soff();
        ad[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, adir, so, 0));
        ad[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, adir + 16u, so, 0));
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int a_rd = lcol * LD + lrow * (P3 ? 4 : 8);
    float accx[XR > 0 ? XR : 1];
#pragma unroll
    for (int j = 0; j < (XR > 0 ? XR : 1); ++j) accx[j] = 0.f;
    float rb0[8], rb1[8];
    f32x4 ad0[2], ad1[2];
    auto mma = [&](const float (&rb)[8], const f32x4 (&ad)[2], int cur, auto&& mid) {      // mid: see conv_igemm3_kernel
        if (BF || !ALDS) mid();
        if (H2) {                                       // order and pinning as SP
            f16x8 bh, bl;
            float sc[8];
            og_h2_split_h(rb, h2_xs, sc, bh);
            f16x8 ah[TM], al[TM];
            if (ALDS) {
                const float* Tl = lds + cur * TILE;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    al[i] = *reinterpret_cast<const f16x8*>(Tl + a_rd + i * 32 * LD + 8);
                    ah[i] = *reinterpret_cast<const f16x8*>(Tl + a_rd + i * 32 * LD);
                }
            } else {
                const float d[8] = {ad[0][0], ad[0][1], ad[0][2], ad[0][3], ad[1][0], ad[1][1], ad[1][2], ad[1][3]};
                float dsc[8];
                og_h2_split_h(d, h2_dys, dsc, ah[0]);
                og_h2_split_l(dsc, ah[0], al[0]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(al[i], bh, acc[i]);
            if (ALDS) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            og_h2_split_l(sc, bh, bl);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bh, acc[i]);
            if (ALDS) og_interleave<TM, (16 + TM - 1) / TM>();
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bl, acc[i]);
            return;
        }
        if (SP) {                                       // order and pinning: see conv_igemm3_kernel
            bf16x8 bh, bm, bl;
            og_split8_h(rb, bh);
            bf16x8 ah[TM], am[TM], al[TM];
            if (ALDS) {
                const float* Tl = lds + cur * TILE;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    al[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD + 16);
                    am[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD + 8);
                    ah[i] = *reinterpret_cast<const bf16x8*>(Tl + a_rd + i * 32 * LD);
                }
            } else {
                const float d[8] = {ad[0][0], ad[0][1], ad[0][2], ad[0][3], ad[1][0], ad[1][1], ad[1][2], ad[1][3]};
                og_split8(d, ah[0], am[0], al[0]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(al[i], bh, acc[i]);
            if (ALDS) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            og_split8_ml(rb, bh, bm, bl);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bh, acc[i]);
            if (ALDS) og_interleave<2 * TM, (40 + 2 * TM - 1) / (2 * TM)>();
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bl, acc[i]);
            return;
        }
        if (BF) {
            bf16x8 bq;
#pragma unroll
            for (int j = 0; j < 8; ++j) bq[j] = (__bf16)rb[j];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4 x0, x1;
                if (ALDS) {
                    const float* Tl = lds + cur * TILE;
                    x0 = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD);
                    x1 = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD + 4);
                } else {
                    x0 = ad[0]; x1 = ad[1];
                }
                bf16x8 aq;
#pragma unroll
                for (int j = 0; j < 4; ++j) { aq[j] = (__bf16)x0[j]; aq[4 + j] = (__bf16)x1[j]; }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bq, acc[i], 0, 0, 0);
            }
            return;
        }
        if (ALDS) {
            const float* Tl = lds + cur * TILE;
            f32x4 a0[TM], a1[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i] = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD);
                a1[i] = *reinterpret_cast<const f32x4*>(Tl + a_rd + i * 32 * LD + 4);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][0], rb[0], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            mid();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][kk], rb[kk], acc[i], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][kk], rb[4 + kk], acc[i], 0, 0, 0);
            if (XR > 0 && has_x) {
#pragma unroll
                for (int j = 0; j < XR; ++j) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(Tl + (BM + j) * LD + lrow * 8);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(Tl + (BM + j) * LD + lrow * 8 + 4);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accx[j] = fmaf(x0[kk], rb[kk], accx[j]);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accx[j] = fmaf(x1[kk], rb[4 + kk], accx[j]);
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[0][kk], rb[kk], acc[0], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[1][kk], rb[4 + kk], acc[0], 0, 0, 0);
        }
    };

    const int nk = (p_end - p_begin) / BK;
    if (ALDS) {
        load_a();
        store_a(0);
        if (P3 || nk > 1) load_a();
    } else {
        load_adir(ad0);
    }
    load_b(rb0);
    if (P3 && ALDS) load_b(rb1);
    if (ALDS) __syncthreads();
    int cur = 0;
    int kt = 0;                                       // two steps per trip, see conv_igemm3_kernel
    if (P3 && ALDS) {               // three fragment sets / three LDS tiles, gather two steps ahead (conv_igemm3_kernel)
        float rb2[8];
        int ks = 0;
        if (ks + 2 < nk) {
            do {
                mma(rb0, ad0, 0, [&]() { store_a(1); load_a(); load_b(rb2); });
                __syncthreads();
                mma(rb1, ad0, 1, [&]() { store_a(2); load_a(); load_b(rb0); });
                __syncthreads();
                mma(rb2, ad0, 2, [&]() { store_a(0); load_a(); load_b(rb1); });
                __syncthreads();
                ks += 3;
            } while (ks + 2 < nk);
        }
        if (ks < nk) {
            mma(rb0, ad0, 0, [&]() { store_a(1); });
            __syncthreads();
        }
        if (ks + 1 < nk) mma(rb1, ad0, 1, [] {});
        kt = nk;
    }
    for (; kt + 1 < nk; kt += 2) {
        mma(rb0, ad0, cur, [&]() {
            if (ALDS) store_a(cur ^ 1); else load_adir(ad1);
            load_b(rb1);
            if (ALDS && kt + 2 < nk) load_a();
        });
        if (ALDS) __syncthreads();
        cur ^= 1;
        mma(rb1, ad1, cur, [&]() {
            if (kt + 2 < nk) {
                if (ALDS) store_a(cur ^ 1); else load_adir(ad0);
                load_b(rb0);
            }
            if (ALDS && kt + 3 < nk) load_a();
        });
        if (ALDS) __syncthreads();
        cur ^= 1;
    }
    if (kt < nk) mma(rb0, ad0, cur, [] {});

    if (XR > 0 && has_x) {
#pragma unroll
        for (int j = 0; j < XR; ++j) accx[j] += __shfl_xor(accx[j], 32, 64);
    }
    if (!col_ok) return;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
            if (m < a.m_end) og_wgrad_store(a, m, col, H2 ? acc[i][r] * h2_inv : acc[i][r], split);
        }
    }
    if (XR > 0 && has_x && lrow == 0) {
#pragma unroll
        for (int j = 0; j < XR; ++j)
            if (j < a.xr_count) og_wgrad_store_xr(a, j, col, accx[j], split);
    }
}

// Adds the ring of a padded-grid data gradient (IgemmArgs::ring) onto the unpadded gradient: padded index p mirrors to
// 1 (p = 0) / H - 2 (p = H + 1), interior p to p - 1.  Gather form: one thread per DESTINATION element of rows 1 / H-2
// and columns 1 / W-2 adds its (up to four) ring entries in a fixed order -- top row, bottom row, left column, right
// column -- to y: single writer, no atomics, bit-reproducible.
__global__ __launch_bounds__(256) void reflect_ring_fold_kernel(const float* __restrict__ ring, float* __restrict__ y,
                                                                long planes, int H, int W) {
    const int PH = H + 2, PW = W + 2, R = 2 * PW + 2 * PH;
    const int D = 2 * W + 2 * H;                      // candidate destinations per plane (segments below)
    const long total = planes * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long p = e / D;
        const int d = (int)(e - p * D);
        int oh, ow;
        if (d < W) { oh = 1; ow = d; }
        else if (d < 2 * W) { oh = H - 2; ow = d - W; if (oh == 1) continue; }
        else if (d < 2 * W + H) { oh = d - 2 * W; ow = 1; if (oh == 1 || oh == H - 2) continue; }
        else { oh = d - 2 * W - H; ow = W - 2; if (oh == 1 || oh == H - 2 || ow == 1) continue; }
        const float* rg = ring + p * R;
        const float* top = rg, *bot = rg + PW, *lef = rg + 2 * PW, *rig = rg + 2 * PW + PH;
        float v = 0.f;
        if (oh == 1) {                                   // padded row 0: columns pb with mirror(pb) == ow
            v += top[ow + 1];
            if (ow == 1) v += top[0];
            if (ow == W - 2) v += top[PW - 1];
        }
        if (oh == H - 2) {                               // padded row PH - 1
            v += bot[ow + 1];
            if (ow == 1) v += bot[0];
            if (ow == W - 2) v += bot[PW - 1];
        }
        if (ow == 1) v += lef[oh + 1];                   // padded column 0 (corners belong to the row arrays)
        if (ow == W - 2) v += rig[oh + 1];               // padded column PW - 1
        y[p * (long)H * W + (long)oh * W + ow] += v;
    }
}

// fp32 -> bf16 (RNE), same layout: the dy operand of conv_wgrad_bfb_kernel.  n4 = elements / 4.
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        bf16x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (__bf16)v[j];
        *reinterpret_cast<bf16x4*>(out + 4 * i) = h;
    }
}

// ---- weight gradient of the bf16 mode: bf16 operands, x read from its channel-blocked copy -----------------------------
// dW[m][c][t] = sum over output pixels p = (n, oh, ow) of dy[n][m][p] * x[n][c][p @ tap t]: GEMM rows = output channels,
// K = pixels, columns = (tap, channel).  Both MFMA operands want eight consecutive K = pixels per lane; x arrives
// pixel-major -- its channel-blocked bf16 copy [N][Cp/16][H*W][16] (nchw_to_nhwc_bf16_kernel) holds 16 channels of a pixel
// as one 32-byte record -- so a wave copies the records of its 32 pixels x 32 channels (ONE tap, two chunks) into a
// wave-private LDS image [32 pixels][2 chunks][16 channels] (lane-linear 16-byte stores) and reads them back with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read: lane (i = l & 15 of a 16-lane group) passes the address of row
// i >> 2, columns 4 (i & 3) .. +3 and receives column i of the 4 x 16 block -- four consecutive PIXELS of one channel.
// Two such reads are the lane's operand of one 32x32x16 MFMA (measured on the box: tools/tr_probe.hip).  No fp32
// gathers, no conversions of x, every geometry (zero / reflect padding, stride, nearest-x2 upsampling) is just the
// record address of the lane's pixel; records are 32 bytes, so every load is 16-byte aligned whatever the tap shift.
// dy rows: a bf16 copy of dy in its own NCHW layout (f32_to_bf16_kernel; pixels are already contiguous there), staged
// into LDS as 16-byte pieces (80-byte pitch: conflict-free ds_read_b128), shared by the NW waves of the workgroup --
// NW column groups (tap, 32 channels) per row tile.  (The first version read fp32 dy and rounded it on the way into
// LDS: the kernel runs against the L2 -> L1 fill rate -- 40 KB per workgroup and iteration at 353 TFLOP/s = 4.5 TB/s --
// and the fp32 rows were 24 of those 40 KB.)
// One iteration = 32 pixels = two MFMAs per row group.  Requires (OH * OW) % 32 == 0 and OH, OW <= 256.
// The record address of a lane's pixel costs VALU work every iteration (K = pixels: nothing is constant across the
// loop): the tap geometry -- stride, padding, reflection, upsampling, bounds -- sits in two small LDS tables per
// workgroup (source row offset per (kh, oh), source column per (kw, ow); 0xffff = outside), so an address is two
// 16-bit LDS reads and ~8 VALU instructions; the first version evaluated the geometry per record (~108 VALU per wave and
// iteration next to 12 MFMAs: VALU bound, 344 TFLOP/s).
template <int TM, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_bfb_kernel(const WgradArgs a, const __bf16* __restrict__ xb,
                                                                 const __bf16* __restrict__ dyb, const int KS, const int Cp) {
    constexpr int NT = 64 * NW;
    constexpr int BM = 32 * TM;
    constexpr int BK = 32;
    constexpr int ALD = 20;                          // floats per dy row in LDS: 64 bytes of bf16 + 16 (odd multiple of 16)
    constexpr int ATILE = BM * ALD;
    constexpr int NA4 = BM * 4;                      // 16-byte bf16 pieces (8 pixels) of a row tile per iteration
    constexpr int NA_PER = (NA4 + NT - 1) / NT;
    constexpr int BTILE = 512;                       // floats: 32 pixels x 64 bytes per wave (wave-private, ONE buffer:
                                                     // a wave's LDS instructions execute in order, the store of the next
                                                     // image is issued behind the last transposing read of this one)
    constexpr int TAB = 256;                         // table pitch: OH, OW <= 256
    static_assert((2 * ATILE + NW * BTILE) * 4 + 2 * 4 * TAB * 2 <= 64 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float ldsA[2 * ATILE];
    __shared__ __attribute__((aligned(16))) float ldsB[NW * BTILE];
    __shared__ unsigned short rtab[4 * TAB], ctab[4 * TAB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5;
    const int lcol = lane & 31;

    const int T = KS * KS;
    const int Cc = Cp >> 4;                          // 16-channel chunks
    const int CG = (Cc + 1) >> 1;                    // 32-channel column groups per tap
    const int ngroups = T * CG;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (ngroups + NW - 1) / NW;
    // XCD placement over BOTH grid dimensions: the workgroups of one pixel split read the same dy / x records (every
    // (tap, channel group) walks the same pixels), so all tiles of a split go to ONE XCD -- one L2 -- and an XCD takes
    // a contiguous range of splits.  (With the remap over blockIdx.x alone the tiles of a split were spread over the
    // eight L2s: 39 % L2 misses, 2.2 GB from HBM / MALL per launch on objd_l3, profiles/r03_bf16_pmc_objd_l3.txt.)
    const int nwg = tiles_m * tiles_n;                // = gridDim.x
    const int vid = og_xcd_remap(blockIdx.x + nwg * blockIdx.y, nwg * gridDim.y);
    const int split = vid / nwg;
    const int wg = vid - split * nwg;
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int group = tile_n * NW + wid;
    const bool grp_ok = group < ngroups;
    const int t = grp_ok ? group / CG : 0;
    const int cg = grp_ok ? group - t * CG : 0;
    const int kh = t / KS;
    const int dh = kh - a.pad, dw = (t - kh * KS) - a.pad;

    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;
    const int Npix = a.N * OHW;
    const int p_begin = split * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    const int nk = (p_end - p_begin + BK - 1) / BK;

    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)xb, 0, (int)((unsigned)a.N * (unsigned)Cc * (unsigned)HW * 32u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)dyb, 0, (int)((unsigned)a.N * a.Cout * OHW * 2u), OG_BUF_FLAGS);

    // ---- x records: lane = (pixel j = l >> 2 of a 16-pixel half, chunk (l >> 1) & 1, 16-byte half l & 1)
    const int chunk = cg * 2 + ((lane >> 1) & 1);
    const bool rec_ok = grp_ok && chunk < Cc;
    const unsigned rec_lane = (unsigned)chunk * (unsigned)HW;     // + n * Cc * HW + ih * W + iw, x 32 bytes + half
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    for (int i = tid; i < KS * (a.OH + a.OW); i += NT) {             // tap geometry tables (see the header comment)
        const bool is_row = i < KS * a.OH;
        const int e = is_row ? i : i - KS * a.OH;
        const int L = is_row ? a.OH : a.OW, LL = is_row ? a.LH : a.LW;
        const int kk = e / L, o = e - kk * L;
        const int iv = o * a.stride + kk - a.pad;
        int ivr = iv < 0 ? -iv : iv;
        ivr = ivr >= LL ? 2 * (LL - 1) - ivr : ivr;
        const bool ok = refl || ((unsigned)iv < (unsigned)LL);
        const int src = (refl ? ivr : iv) >> us;
        const unsigned short v = ok ? (unsigned short)(is_row ? src * a.W : src) : (unsigned short)0xffffu;
        if (is_row) rtab[kk * TAB + o] = v; else ctab[kk * TAB + o] = v;
    }
    const unsigned short* rt = rtab + kh * TAB;
    const unsigned short* ct = ctab + (t - kh * KS) * TAB;
    // pixel steps without divisions in the loop: 16 and 32 pixels = (rows, columns) of the output map
    const int rows16 = 16 / a.OW, cols16 = 16 - rows16 * a.OW;
    const int rows32 = 32 / a.OW, cols32 = 32 - rows32 * a.OW;
    int pn, poh, pow_;                               // output pixel of this lane in the first half of the next iteration
    {
        const int p = p_begin + (lane >> 2);
        pn = p / OHW;
        const int r = p - pn * OHW;
        poh = r / a.OW;
        pow_ = r - poh * a.OW;
    }
    int p_ld = p_begin + (lane >> 2);                // pixel index of (pn, poh, pow_)
    const unsigned half16 = (unsigned)((lane & 1) * 16);
    auto load_b = [&](f32x4 (&rb)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int n = pn, oh = poh, ow = pow_;
            if (h == 1) {                            // second half: 16 pixels further
                ow += cols16;
                const int c = ow >= a.OW ? 1 : 0;
                ow -= c ? a.OW : 0;
                oh += rows16 + c;
                while (oh >= a.OH) { oh -= a.OH; n += 1; }
            }
            const unsigned r = rt[oh], c = ct[ow];
            const bool ok = rec_ok && (p_ld + 16 * h < p_end) && r != 0xffffu && c != 0xffffu;
            const unsigned off = ok ? ((unsigned)n * (unsigned)Cc * (unsigned)HW + rec_lane + r + c) * 32u + half16 : OG_OOB;
            rb[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
        }
        p_ld += BK;
        pow_ += cols32;
        const int c = pow_ >= a.OW ? 1 : 0;
        pow_ -= c ? a.OW : 0;
        poh += rows32 + c;
        while (poh >= a.OH) { poh -= a.OH; pn += 1; }
    };
    auto store_b = [&](const f32x4 (&rb)[2]) {
        float* Bs = ldsB + wid * BTILE;
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(Bs + h * 256 + lane * 4) = rb[h];
    };
    // transposing read: 16-lane group g = l >> 4: chunk g & 1, pixel half-octet g >> 1; see the header comment
    const int b_rd = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;   // bytes

    // ---- dy rows
    unsigned avoff[NA_PER];
    int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + NT * i;
        const int row = idx >> 2, q = idx & 3;
        const bool on = (NA4 % NT == 0 || idx < NA4) && (m0 + row) < a.m_end;
        avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + q * 8u) * 2u : OG_OOB;
        alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * ALD + q * 4 : -1;
    }
    int n_ld = p_begin / OHW;                        // scalar (image, offset) of the next dy iteration
    int rem_ld = p_begin - n_ld * OHW;
    f32x4 ra[NA_PER];
    auto load_a = [&]() {
        const int so = (n_ld * a.Cout * OHW + rem_ld) * 2;
        rem_ld += BK;
        if (rem_ld >= OHW) { rem_ld = 0; n_ld += 1; }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, avoff[i], so, 0));
    };
    auto store_a = [&](int buf) {
        float* As = ldsA + buf * ATILE;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
    };

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
    auto mma = [&](int buf, auto&& mid) {
        const char* Bs = reinterpret_cast<const char*>(ldsB + wid * BTILE) + b_rd;
        const float* As = ldsA + buf * ATILE + lcol * ALD + lrow * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024 + 256));
            // (whole-vector casts: an element-wise short -> bf16 copy of the two halves came out of hipcc as
            // {b0.lo, b0.lo, b1.lo, b1.lo} -- found with tools/dbg_wgrad.py)
            const bf16x8 bq = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            bf16x8 aq[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                aq[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * ALD + h * 8);
            if (h == 1) {                 // the refill behind the first TM MFMAs of the iteration
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[i], bq, acc[i], 0, 0, 0);
        }
    };

    // prologue: iteration 0 in LDS buffer 0, iteration 1 in registers
    f32x4 rb[2];
    __syncthreads();                                  // tables
    load_a(); load_b(rb);
    store_a(0); store_b(rb);
    load_a(); load_b(rb);
    __syncthreads();
    // two iterations per trip (literal buffer indices); the loads of iteration k + 2 are issued in iteration k and stored to
    // LDS in iteration k + 1 (unconditionally: past the end they hit the range check or unused records)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        mma(0, [&]() { store_a(1); store_b(rb); load_a(); load_b(rb); });
        __syncthreads();
        mma(1, [&]() { store_a(0); store_b(rb); load_a(); load_b(rb); });
        __syncthreads();
    }
    if (kt < nk) mma(0, [] {});

    // ---- epilogue: column = channel ci of tap t -> dw[m][ci * T + t]
    const int ci = cg * 32 + lcol;
    if (!grp_ok || ci >= a.Cin) return;
    const int ocol = ci * T + t;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
            if (m < a.m_end) {
                if (a.ws) a.ws[(size_t)split * a.ws_stride + (size_t)(m - a.m_begin) * a.ncol + ocol] = acc[i][r];
                else og_wgrad_store(a, m, ocol, acc[i][r], split);  // one split
            }
        }
    }
}


// ---- optional per-launch timing (bench.py's roofline leg) -------------------------------------
// When enabled, every conv launch is bracketed by hipEvents on its own stream and tagged with a
// category (kind, taps / ksize, tile config) and its ALGORITHMIC flops 2*M*K*Npix.  Off by default;
// the only mutable global state of the library, touched by the host thread only.
#define OG_PROF_CATS 192         // 0..47: see OG_CAT_*; 48..95: the same kernel families in their fp16x2 instances;
                                 // 96..143 / 144..191: conv_igemm3_kernel on records (math 5), one / two pixel groups per wave
#define OG_PROF_MAX 65536
// meta: {kind (0 forward / data-gradient GEMM, 1 weight gradient, 2 thin VALU), tile height TM, rows M,
//        K channels C, taps T, images N, pixel-grid rows, pixel-grid columns, stride, grid.y splits}
struct ProfRec { hipEvent_t a, b; int cat; double flops; int meta[10]; };
static int g_prof_on = 0;
static ProfRec* g_prof = nullptr;
static int g_prof_n = 0;
static int g_prof_made = 0;

// categories = kernel instances, so that they line up with the kernel names rocprofv3 reports:
//   0..6  conv_igemm3_kernel<1..7>     7..13 conv_wgrad2_kernel<1..7, false>
//   14 conv_thin_kernel<*>   15 conv_thin3x3_kernel<*>   16 conv_igemm_kernel<*> (v1)   17 conv_wgrad_kernel<*> (v1)
//   18 conv_igemm3_kernel<1, true> (LDS-free form for thin outputs)   19..25 conv_wgrad3_kernel<1..7>
#define OG_CAT_IGEMM2(tm) (((tm) == 1 && a.M <= 32) ? 18 : ((tm) - 1))
#define OG_CAT_IGEMM2_NW8(tm) (25 + (tm))          // 26..32: conv_igemm3_kernel<1..7, false, 2, 8>
#define OG_CAT_WGRAD3_NW8(tm) (32 + (tm))          // 33..39: conv_wgrad3_kernel<1..7, 2, *, 0, 8>
#define OG_CAT_WGRAD2(tm) (6 + (tm))
#define OG_CAT_WGRAD3(tm) (18 + (tm))
#define OG_CAT_THIN 14
#define OG_CAT_THIN3 15
#define OG_CAT_IGEMM1 16
#define OG_CAT_WGRAD1 17
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
    for (int i = 0; i < 10; ++i) r->meta[i] = 0;
    (void)hipEventRecord(r->a, s);
    return r;
}
static inline void prof_meta(ProfRec* r, int kind, int tm, int M, int C, int T, int N, int ph, int pw, int stride,
                             int splits) {
    if (!r) return;
    const int v[10] = {kind, tm, M, C, T, N, ph, pw, stride, splits};
    for (int i = 0; i < 10; ++i) r->meta[i] = v[i];
}
static inline void prof_end(ProfRec* r, hipStream_t s) { if (r) (void)hipEventRecord(r->b, s); }

// ---- development switches (OG_KNOB: common.h -- constants in the shipped library) -----------------
OG_KNOB(og_igemm_v1, "OG_IGEMM_V1", 0)             // 1: first-generation kernels everywhere
OG_KNOB(og_igemm_tmmax_raw, "OG_IGEMM_TMMAX", 8)   // tallest forward / data-gradient tile
OG_KNOB(og_nothin, "OG_NO_THIN", 0)                // 1: no direct VALU kernels for thin outputs
OG_KNOB(og_trace, "OG_TRACE", 0)                   // 1: print every launch plan to stderr
OG_KNOB(og_wgrad3_maxtm, "OG_WGRAD3_MAXTM", 2)     // register-fragment weight-gradient form up to this tile height
OG_KNOB(og_wgrad_oldsplit, "OG_WGRAD_OLDSPLIT", 0)
OG_KNOB(og_wgrad_nob128, "OG_WGRAD_NOB128", 0)     // 1: dword gathers on wide stride-1 maps
OG_KNOB(og_split_target, "OG_SPLIT_TARGET", 1024)
OG_KNOB(og_no_xrows, "OG_NO_XROWS", 0)
OG_KNOB(og_nw8_min, "OG_NW8_MIN", 512)             // bf16x3: 8-wave workgroups from this many workgroups on (0: never)
OG_KNOB(og_ablate, "OG_ABLATE", 0)                 // development builds: IgemmArgs::ablate
OG_KNOB(og_kgroup_s1, "OG_KGROUP_S1", 4)           // chunks per K group (og_kstep), stride-1 multi-tap launches (0: tap-major)
OG_KNOB(og_kgroup_s2, "OG_KGROUP_S2", 0)           // ... stride-2 forward launches
OG_KNOB(og_kgroup_ph, "OG_KGROUP_PH", 4)           // ... the four-phase stride-2 data gradient / up-convolution
OG_KNOB(og_h2_nw8_tm, "OG_H2_NW8_TM", 4)           // fp16x2: 8-wave workgroups from this block-row height on
OG_KNOB(og_h2_pen_pct, "OG_H2_PEN_PCT", 100)        // fp16x2: re-read penalty of short block rows in og_row_plan, % of the table
OG_KNOB(og_x3_wgrad3_maxtm, "OG_X3_WGRAD3_MAXTM", 2)   // bf16x3: register-fragment weight gradient up to this tile height
OG_KNOB(og_rec_ng2_maxtm, "OG_REC_NG2_MAXTM", 3)    // fp16x2 on records: two pixel groups per wave up to this block-row height (0: never)
OG_KNOB(og_rec_ng2_min, "OG_REC_NG2_MIN", 1024)     // ... while the grid keeps this many workgroups (r5c_tileplans: 256 loses on 32x32 maps, 1024 >= 512)
OG_KNOB(og_rec_nw8_tm, "OG_REC_NW8_TM", 4)          // fp16x2 on records: 8-wave workgroups from this block-row height on
OG_KNOB(og_rec_tmmax, "OG_REC_TMMAX", 7)            // ... tallest block row
OG_KNOB(og_rec_ng2_nw8, "OG_REC_NG2_NW8", 0)        // ... 1: two pixel groups per wave also in 8-wave workgroups
OG_KNOB(og_wgrad_rec_tmmax, "OG_WGRAD_REC_TMMAX", 6)  // weight gradient on records: tallest block row (7: one wave per SIMD)
OG_KNOB(og_wgrad_rec_nw8, "OG_WGRAD_REC_NW8", 1)      // ... 8-wave workgroups for block rows <= 6 on >= 16384 pixels
static int og_igemm_tmmax() { const int v = og_igemm_tmmax_raw(); return (v < 1 || v > 8) ? 8 : v; }

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
    dim3 grid(g, 1);
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

// The first-generation kernel serves tensors beyond the 2 GiB reach of a buffer descriptor: grids of thousands of
// workgroups, never split along K.  (Rounds 1-4 kept an fp32-atomics split-K form of it for small grids; the training step
// never took it -- small grids run conv_igemm3_kernel with the ordered workspace reduction -- and it is gone: no fp32
// atomic is left in the library's convolutions, `-munsafe-fp-atomics` is no longer a build flag of this file.)
static int run_igemm(IgemmArgs a, hipStream_t s, int y_prezeroed) {
    (void)y_prezeroed;
    RowPart parts[3];
    const int np = og_row_parts(a.M, parts);
    a.ksplit_steps = 0;
    for (int i = 0; i < np; ++i) {
        a.m_begin = parts[i].m_begin; a.m_end = parts[i].m_end;
        const double fl = 2.0 * (a.m_end - a.m_begin) * (double)a.K * ((double)a.N * a.PH * a.PW);
        ProfRec* pr = prof_begin(OG_CAT_IGEMM1, fl, s);
        int rc = launch_igemm(a, parts[i].cfg, s);
        prof_end(pr, s);
        if (rc != OG_OK) return rc;
    }
    return OG_OK;
}

// ---- v2 launch plan: block rows of TM 32-row groups (TM <= 8) ---------------------------------

// Block-row plan for M = `groups` 32-row groups over `tiles_n` column tiles: block rows of height TM
// (<= 7: two workgroups per CU) plus one lower block row for the remainder in its own launch
// (388 rows = 13 groups -> 7 + 6, 194 -> 7, 768 -> 4 x 6).  The MFMA time of a launch is
// ~ (workgroups per CU) x TM: whole rounds while the grid is small (quantisation is what decides
// there), fractional once it spans many rounds; short tiles re-read the pixel operand more (pen).
// The weight gradient always takes the tallest tiles: its gather is the expensive part.
static double og_rounds(long blocks) {
    if (blocks <= 256) return 1.25;                 // one workgroup per CU: nothing to overlap with
    if (blocks < 1024) return (double)og_cdiv(blocks, 256);
    return (double)blocks / 256.0;
}
static void og_row_plan(int groups, int tiles_n, int tall, int* TM_out, int* full_rows_out, int* rest_out,
                        int pen_pct = 100, int tm_cap = 7) {
    int tmmax = og_igemm_tmmax();
    if (tmmax > 7) tmmax = 7;
    if (tmmax > tm_cap && tm_cap >= 1) tmmax = tm_cap;
    int bt = 1;
    if (tall) {
        const int brows = og_cdiv(groups, tmmax);
        bt = og_cdiv(groups, brows);
    } else {
        // TM = 1 re-reads the pixel operand once per 32 rows and runs against the L2 (~7 TB/s of
        // fills, 86 TFLOP/s at best -- profiles/r01_tm1_l2_bound.txt); TM = 3 reaches ~114, TM >= 4 ~120
        static const double pen[8] = {0, 1.40, 1.15, 1.06, 1.02, 1.0, 1.0, 1.0};
        double best = -1;
        for (int tm = 1; tm <= tmmax && tm <= groups; ++tm) {
            const int full = groups / tm, rest = groups - full * tm;
            const double pt = 1.0 + (pen[tm] - 1.0) * pen_pct / 100.0;
            const double pr = rest ? 1.0 + (pen[rest] - 1.0) * pen_pct / 100.0 : 0.0;
            double cost = og_rounds((long)full * tiles_n) * tm * pt
                        + (rest ? og_rounds(tiles_n) * rest * pr + 0.3 : 0.0);
            if (best < 0 || cost < best - 1e-9 || (cost < best + 1e-9 && tm > bt)) { best = cost; bt = tm; }
        }
    }
    *TM_out = bt; *full_rows_out = groups / bt; *rest_out = groups - (groups / bt) * bt;
}


static int run_thin(IgemmArgs a, int MT, hipStream_t s) {
    const long Npix = (long)a.N * a.PH * a.PW;
    if (og_trace()) fprintf(stderr, "OGTRACE thin MT=%d M=%d C=%d T=%d Npix=%ld\n", MT, a.M, a.C, a.T, Npix);
    bool canon = a.T == 9 && a.stride == 1 && a.osh == 1 && a.osw == 1 && a.ooh == 0 && a.oow == 0
                 && a.PH == a.OHf && a.PW == a.OWf && a.PH == a.LH && a.PW == a.LW;
    for (int t = 0; canon && t < 9; ++t)
        canon = a.tap[t] == (int)((((unsigned)(t % 3 - 1)) << 16) | ((unsigned)(t / 3 - 1) & 0xffffu));
    if (canon) {
        const int R = MT <= 16 ? 4 : 2;
        const long threads = (long)a.N * og_cdiv(a.PH, R) * a.PW;
        dim3 g3(og_cdiv(threads, 256));
        a.m_begin = 0; a.m_end = a.M; a.ksplit_steps = 0;
        ProfRec* pr = prof_begin(OG_CAT_THIN3, 2.0 * a.M * (double)a.K * (double)Npix, s);
        prof_meta(pr, 2, MT, a.M, a.C, a.T, a.N, a.PH, a.PW, a.stride, 1);
        switch (MT) {
            case 4: hipLaunchKernelGGL((conv_thin3x3_kernel<4, 4>), g3, dim3(256), 0, s, a); break;
            case 12: hipLaunchKernelGGL((conv_thin3x3_kernel<12, 4>), g3, dim3(256), 0, s, a); break;
            case 16: hipLaunchKernelGGL((conv_thin3x3_kernel<16, 4>), g3, dim3(256), 0, s, a); break;
            case 24: hipLaunchKernelGGL((conv_thin3x3_kernel<24, 2>), g3, dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL((conv_thin3x3_kernel<32, 2>), g3, dim3(256), 0, s, a); break;
        }
        prof_end(pr, s);
        return og_launch_status();
    }
    const int PX = MT <= 16 ? 2 : 1;
    dim3 grid(og_cdiv(Npix, 256 * PX));
    a.m_begin = 0; a.m_end = a.M; a.ksplit_steps = 0;
    ProfRec* pr = prof_begin(OG_CAT_THIN, 2.0 * a.M * (double)a.K * (double)Npix, s);
    prof_meta(pr, 2, MT, a.M, a.C, a.T, a.N, a.PH, a.PW, a.stride * (a.osh > 1 ? -1 : 1), 1);
#define OG_THIN(MTv, Tv) hipLaunchKernelGGL((conv_thin_kernel<MTv, Tv, (MTv <= 16 ? 2 : 1)>), grid, dim3(256), 0, s, a)
    if (a.T == 9) {
        switch (MT) { case 4: OG_THIN(4, 9); break; case 12: OG_THIN(12, 9); break; case 16: OG_THIN(16, 9); break;
                      case 24: OG_THIN(24, 9); break; default: OG_THIN(32, 9); break; }
    } else {
        switch (MT) { case 4: OG_THIN(4, 4); break; case 12: OG_THIN(12, 4); break; case 16: OG_THIN(16, 4); break;
                      case 24: OG_THIN(24, 4); break; default: OG_THIN(32, 4); break; }
    }
#undef OG_THIN
    prof_end(pr, s);
    return og_launch_status();
}


static int launch_igemm2(const IgemmArgs& a, int TM, dim3 grid, hipStream_t s, int nw = 4, int ng = 1) {
    if (a.math == 5) {
        if (og_trace())
            fprintf(stderr, "OGTRACE igemm-rec TM=%d NW=%d NG=%d M=%d rows=%d C=%d T=%d Npix=%d grid=%u,%u H=%d W=%d stride=%d\n", TM, nw, ng,
                    a.M, a.m_end - a.m_begin, a.C, a.T, a.N * a.PH * a.PW, grid.x, grid.y, a.H, a.W, a.stride);
        return og_launch_igemm3_rec(a, TM, nw, ng, grid, s);
    }
    if (og_trace())
        fprintf(stderr, "OGTRACE igemm TM=%d NW=%d M=%d rows=%d C=%d T=%d Npix=%d grid=%u,%u,%u H=%d W=%d stride=%d\n", TM, nw, a.M,
                a.m_end - a.m_begin, a.C, a.T, a.N * a.PH * a.PW, grid.x, grid.y, grid.z, a.H, a.W, a.stride);
    if (a.math == 1 && a.nhwc) {           // bf16, channel-blocked pixel operand
#define OG_IGNH(TMv)                                                                                                  \
        if (nw == 8) hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 3, 8>), grid, dim3(512), 0, s, a);            \
        else hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 3, 4>), grid, dim3(256), 0, s, a);
        switch (TM) {
            case 1: OG_IGNH(1) break;
            case 2: OG_IGNH(2) break;
            case 3: OG_IGNH(3) break;
            case 4: OG_IGNH(4) break;
            case 5: OG_IGNH(5) break;
            case 6: OG_IGNH(6) break;
            default: OG_IGNH(7) break;
        }
#undef OG_IGNH
        return og_launch_status();
    }
    if (a.math == 4) {                     // fp16x2
#define OG_IGH2(TMv)                                                                                                  \
        if (nw == 8) hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 4, 8>), grid, dim3(512), 0, s, a);            \
        else hipLaunchKernelGGL((conv_igemm3_kernel<TMv, false, 4, 4>), grid, dim3(256), 0, s, a);
        switch (TM) {
            case 1: if (a.M <= 32) hipLaunchKernelGGL((conv_igemm3_kernel<1, true, 4, 4>), grid, dim3(256), 0, s, a);
                    else { OG_IGH2(1) }
                    break;
            case 2: OG_IGH2(2) break;
            case 3: OG_IGH2(3) break;
            case 4: OG_IGH2(4) break;
            case 5: OG_IGH2(5) break;
            case 6: OG_IGH2(6) break;
            default: OG_IGH2(7) break;
        }
#undef OG_IGH2
        return og_launch_status();
    }
    if (a.math == 2 && nw == 8) {          // 8-wave workgroups: 32 * TM rows x 256 pixels
        switch (TM) {
            case 1: hipLaunchKernelGGL((conv_igemm3_kernel<1, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv_igemm3_kernel<2, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            case 3: hipLaunchKernelGGL((conv_igemm3_kernel<3, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            case 4: hipLaunchKernelGGL((conv_igemm3_kernel<4, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            case 5: hipLaunchKernelGGL((conv_igemm3_kernel<5, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            case 6: hipLaunchKernelGGL((conv_igemm3_kernel<6, false, 2, 8>), grid, dim3(512), 0, s, a); break;
            default: hipLaunchKernelGGL((conv_igemm3_kernel<7, false, 2, 8>), grid, dim3(512), 0, s, a); break;
        }
        return og_launch_status();
    }
#define OG_IG3(MATHv)                                                                                             \
        switch (TM) {                                                                                                 \
            case 1: if (a.M <= 32) hipLaunchKernelGGL((conv_igemm3_kernel<1, true, MATHv>), grid, dim3(256), 0, s, a); \
                    else hipLaunchKernelGGL((conv_igemm3_kernel<1, false, MATHv>), grid, dim3(256), 0, s, a);         \
                    break;                                                                                            \
            case 2: hipLaunchKernelGGL((conv_igemm3_kernel<2, false, MATHv>), grid, dim3(256), 0, s, a); break;       \
            case 3: hipLaunchKernelGGL((conv_igemm3_kernel<3, false, MATHv>), grid, dim3(256), 0, s, a); break;       \
            case 4: hipLaunchKernelGGL((conv_igemm3_kernel<4, false, MATHv>), grid, dim3(256), 0, s, a); break;       \
            case 5: hipLaunchKernelGGL((conv_igemm3_kernel<5, false, MATHv>), grid, dim3(256), 0, s, a); break;       \
            case 6: hipLaunchKernelGGL((conv_igemm3_kernel<6, false, MATHv>), grid, dim3(256), 0, s, a); break;       \
            default: hipLaunchKernelGGL((conv_igemm3_kernel<7, false, MATHv>), grid, dim3(256), 0, s, a); break;      \
        }
    if (a.math == 1) { OG_IG3(1) return og_launch_status(); }
    if (a.math == 2) { OG_IG3(2) return og_launch_status(); }
#undef OG_IG3
    switch (TM) {
        // TM = 1: the LDS-free form reads the filter rows directly, which only pays while the bank is tiny
        case 1: if (a.M <= 32) hipLaunchKernelGGL((conv_igemm3_kernel<1, true>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((conv_igemm3_kernel<1, false>), grid, dim3(256), 0, s, a);
                break;
        case 2: hipLaunchKernelGGL((conv_igemm3_kernel<2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((conv_igemm3_kernel<3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((conv_igemm3_kernel<4>), grid, dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((conv_igemm3_kernel<5>), grid, dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((conv_igemm3_kernel<6>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((conv_igemm3_kernel<7>), grid, dim3(256), 0, s, a); break;
    }
    return og_launch_status();
}

// Rows are covered by block rows of TM 32-row groups: `brows - 1` (or all) full-height block rows
// in one launch, plus one launch with a smaller TM for the remaining groups (388 rows = 13 groups
// -> 7 + 6; 194 -> 7; 768 -> 3 x 8), so that no block computes an empty row group.

struct Igemm2Plan { int nw, ng, TM, full_rows, rest, tiles_n, splits, ksplit_steps, full_cover; };

// Launch plan of run_igemm2 (also behind objgan_conv_igemm_ws_floats: the caller sizes the split-K workspace from it).
static Igemm2Plan igemm2_plan(const IgemmArgs& a, int y_prezeroed) {
    Igemm2Plan p;
    const int groups = og_cdiv(a.M, 32);
    const int Npix = a.N * a.PH * a.PW;
    const int nph = a.nphase > 1 ? a.nphase : 1;
    // bf16x3 runs against the L2 -> L1 fill rate (6 TB/s of fills at 190 TFLOP/s, profiles/r03_x3_pmc_objd_l3.txt),
    // and 2/3 of a workgroup's fills are its row tile: 8-wave workgroups (256 pixels per row tile) wherever the
    // grid still covers the chip twice
    // (tall tiles only: at TM <= 3 a 4-wave workgroup leaves room for three per CU and wins -- r03 A/B, 96-row
    // layers 179 vs 158 TFLOP/s; with 8 waves the block rows are as tall as the row count allows)
    p.nw = 4;
    p.ng = 1;
    const int tm_tall = og_cdiv(groups, og_cdiv(groups, 7));
    const int tm_cap = a.math == 5 ? og_rec_tmmax() : 7;
    if ((a.math == 2 || a.math >= 4 || (a.math == 1 && a.nhwc)) &&
        tm_tall >= (a.math == 5 ? og_rec_nw8_tm() : (a.math == 4 ? og_h2_nw8_tm() : 4)) &&
        og_nw8_min() > 0 &&
        (long)og_cdiv(groups, 7) * og_cdiv(Npix, 256) * nph >= og_nw8_min()) p.nw = 8;
    p.tiles_n = og_cdiv(Npix, 32 * p.nw);
    // (Carrying the 2 / 4 rows that 194 / 388 channels have beyond a multiple of 32 on the VALU next to the
    // MFMA stream -- as the weight-gradient kernels do -- was measured here in round 2 and bought nothing:
    // 106.0 vs 107.8 TFLOP/s on res1_128; interleaving the FMAs with the MFMAs cost 20 %.)
    og_row_plan(groups, p.tiles_n * nph, p.nw == 8, &p.TM, &p.full_rows, &p.rest, a.math >= 4 ? og_h2_pen_pct() : 100, tm_cap);
    if (a.math == 5 && (p.nw == 4 || og_rec_ng2_nw8()) && og_rec_ng2_maxtm() > 0) {
        // record form, short block rows: two 32-pixel groups per wave (every LDS row fragment feeds two MFMAs per
        // product) while the grid still fills the chip
        const int tn2 = og_cdiv(Npix, 64 * p.nw);
        int TM2, full2, rest2;
        const int cap2 = og_rec_ng2_maxtm() < 4 ? og_rec_ng2_maxtm() : 4;
        og_row_plan(groups, tn2 * nph, p.nw == 8, &TM2, &full2, &rest2, og_h2_pen_pct(), p.nw == 8 ? cap2 : tm_cap);
        if (TM2 <= og_rec_ng2_maxtm() && TM2 <= 4 && (long)(full2 + (rest2 ? 1 : 0)) * tn2 * nph >= og_rec_ng2_min()) {
            p.ng = 2; p.tiles_n = tn2; p.TM = TM2; p.full_rows = full2; p.rest = rest2;
        }
    }
    const int tiles = (p.full_rows + (p.rest ? 1 : 0)) * p.tiles_n;
    const int nk = a.math == 1 ? a.Krow / 32 : a.Kpad / 16;      // loop iterations of the kernel
    p.full_cover = ((a.osh == 1 && a.osw == 1 && a.PH == a.OHf && a.PW == a.OWf) || a.ring != nullptr) ? 1 : 0;
    int splits = 1;
    // (partial-coverage launches -- strided output phases into a pre-zeroed y -- are not split: their partials would
    // have to be accumulated with atomics; they are the two 3x3 stride-2 layers of G_HMAP, 0.1 ms per step)
    (void)y_prezeroed;
    if (tiles < 128 && nk >= 16 && p.full_cover && nph == 1) {
        splits = og_cdiv(512, tiles);
        if (splits > nk / 4) splits = nk / 4;
    } else if (og_split_target() > 0 && p.TM == 1 && p.rest == 0 && tiles < og_split_target() && nk >= 16 &&
               p.full_cover && nph == 1) {
        splits = og_cdiv(og_split_target(), tiles);
        if (splits > nk / 8) splits = nk / 8;
        if (splits < 1) splits = 1;
    }
    p.ksplit_steps = 0;
    if (splits > 1) {
        p.ksplit_steps = og_cdiv(nk, splits);
        splits = og_cdiv(nk, p.ksplit_steps);
    }
    p.splits = splits;
    return p;
}

// floats of split-K workspace run_igemm2 wants for this plan (0: one split, or a partial-coverage launch that
// accumulates into the pre-zeroed output)
// bf16 mode: floats of workspace the channel-blocked bf16 copy of the source takes (0: not this mode / too large for
// the 32-bit buffer range)
static long igemm2_nhwc_floats(int math, int N, int H, int W, int Cp) {
    if (math != 1 || (double)N * H * W * Cp * 2.0 >= 4.0e9) return 0;
    return ((long)N * H * W * Cp / 2 + 3) & ~3L;
}
static long igemm2_ws_floats(const IgemmArgs& a, const Igemm2Plan& p) {
    const long nh = a.nhwc == 1 ? igemm2_nhwc_floats(a.math, a.N, a.H, a.W, a.Cp) : 0;
    if (p.splits <= 1 || !p.full_cover) return nh;
    const long seg = (long)a.N * a.M * a.OHf * a.OWf + (a.ring ? (long)a.N * a.M * (2 * a.PW + 2 * a.PH) : 0);
    return nh + seg * p.splits;
}

static int run_igemm2(IgemmArgs a, hipStream_t s, int y_prezeroed, float* ws, long ws_floats, float* ymax = nullptr) {
    a.ymax = nullptr;
    if (a.nhwc == 1 && !ws) a.nhwc = 0;         // no workspace: fp32 NCHW gathers (conv_igemm3_kernel<.., 1, ..>)
    const Igemm2Plan p = igemm2_plan(a, y_prezeroed);
    if (a.nhwc == 1) {                          // bf16 channel-blocked copy of the source: first part of the workspace
        const long nh = igemm2_nhwc_floats(a.math, a.N, a.H, a.W, a.Cp);
        if (ws_floats < igemm2_ws_floats(a, p)) return OG_BAD_ARGS;
        hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(og_cdiv(a.H * a.W, 64), og_cdiv(a.Cp, 64), a.N), dim3(256), 0, s,
                           a.x, reinterpret_cast<__bf16*>(ws), a.C, a.H * a.W, a.Cp);
        a.x = ws;
        ws += nh; ws_floats -= nh;
    }
    const int Npix = a.N * a.PH * a.PW;
    const int nph = a.nphase > 1 ? a.nphase : 1;
    const int TM = p.TM, full_rows = p.full_rows, rest = p.rest, tiles_n = p.tiles_n, nw = p.nw, ng = p.ng;
    int splits = p.splits;
    const bool full_cover = p.full_cover != 0;
    const float* bias = a.bias;
    const int act = a.act;
    const bool act_later = (act == OG_ACT_TANH || act == OG_ACT_SIGMOID);
    const long y_elems = (long)a.N * a.M * a.OHf * a.OWf;
    const long ring_elems = a.ring ? (long)a.N * a.M * (2 * a.PW + 2 * a.PH) : 0;
    a.ws = nullptr; a.ws_stride = 0;
    if (splits > 1) {
        a.ksplit_steps = p.ksplit_steps;
        a.bias = nullptr; a.act = OG_ACT_NONE;
        const long need = igemm2_ws_floats(a, p) - (a.nhwc == 1 ? igemm2_nhwc_floats(a.math, a.N, a.H, a.W, a.Cp) : 0);
        if (need > 0) {                       // two-level reduction through the caller's workspace
            if (!ws || ws_floats < need) return OG_BAD_ARGS;
            a.ws = ws; a.ws_stride = y_elems + ring_elems;
        } else {
            return OG_BAD_ARGS;               // (not reached: the plan splits only launches that cover their output)
        }
    } else {
        a.ksplit_steps = 0;
        if (act_later) { a.bias = nullptr; a.act = OG_ACT_NONE; }
    }
    // |y| maxima in the epilogue: unsplit launches that write their final values and cover y (else a pass over y below)
    const bool emit = ymax && splits <= 1 && !act_later && full_cover && !a.ring;
    if (emit) a.ymax = ymax;
    int rc = OG_OK;
    if (full_rows > 0) {
        a.m_begin = 0; a.m_end = min(a.M, full_rows * TM * 32);
        ProfRec* pr = prof_begin((nw == 8 ? OG_CAT_IGEMM2_NW8(TM) : OG_CAT_IGEMM2(TM)) + (a.math == 5 ? (ng == 2 ? 144 : 96) : (a.math == 4 ? 48 : 0)),
                                 2.0 * (a.m_end - a.m_begin) * (double)a.K * (double)Npix * nph, s);
        prof_meta(pr, 0, TM, a.m_end - a.m_begin, a.C, a.T, a.N, a.PH * nph, a.PW, a.stride * (a.osh > 1 ? -1 : 1), splits);
        rc = launch_igemm2(a, TM, dim3(full_rows * tiles_n * nph, splits, 1), s, nw, ng);
        prof_end(pr, s);
        if (rc != OG_OK) return rc;
    }
    if (rest > 0) {
        a.m_begin = full_rows * TM * 32; a.m_end = a.M;
        ProfRec* pr = prof_begin((nw == 8 ? OG_CAT_IGEMM2_NW8(rest) : OG_CAT_IGEMM2(rest)) + (a.math == 5 ? (ng == 2 ? 144 : 96) : (a.math == 4 ? 48 : 0)),
                                 2.0 * (a.m_end - a.m_begin) * (double)a.K * (double)Npix * nph, s);
        prof_meta(pr, 0, rest, a.m_end - a.m_begin, a.C, a.T, a.N, a.PH * nph, a.PW, a.stride * (a.osh > 1 ? -1 : 1), splits);
        rc = launch_igemm2(a, rest, dim3(tiles_n * nph, splits, 1), s, nw, ng);
        prof_end(pr, s);
        if (rc != OG_OK) return rc;
    }
    if (a.ws) {         // second level: sum the splits in order, + bias, activation
        // (the maxima of y ride in the combine unless a later pass changes y: tanh / sigmoid heads, ring mode)
        const bool ymax_in_combine = ymax && !act_later && ring_elems == 0;
        hipLaunchKernelGGL(splitk_combine_kernel, dim3(og_stream_grid(y_elems, 256)), dim3(256), 0, s, a.ws, splits,
                           a.ws_stride, 0L, a.y, y_elems, bias, a.M, a.OHf * a.OWf, act, ymax_in_combine ? ymax : (float*)nullptr);
        if (ring_elems > 0)
            hipLaunchKernelGGL(splitk_combine_kernel, dim3(og_stream_grid(ring_elems, 256)), dim3(256), 0, s, a.ws, splits,
                               a.ws_stride, y_elems, a.ring, ring_elems, (const float*)nullptr, 1, 1, OG_ACT_NONE, (float*)nullptr);
        if (ymax && !ymax_in_combine) og_absmax_launch(a.y, y_elems, ymax, s);
        return og_launch_status();
    }
    if ((splits > 1 || act_later) && (bias || act != OG_ACT_NONE) && full_cover) {
        const long total = y_elems;
        hipLaunchKernelGGL(bias_act_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, s, a.y, bias,
                           total, a.M, a.OHf * a.OWf, act);
    }
    if (ymax && !emit) og_absmax_launch(a.y, y_elems, ymax, s);
    return og_launch_status();
}

// Row pitch of a packed bank in elements: fp32 Kpad floats; bf16 Kpad rounded up to 32 (one iteration = 32 k);
// bf16x3 three bf16 per k (the h / m / l pieces of a 16-deep step back to back: 96 bytes).
static inline int og_krow(int Kpad, int math) {
    return math == 1 ? (Kpad + 31) / 32 * 32 : (math == 2 ? 3 * Kpad : (math >= 4 ? 2 * Kpad : Kpad));
}

// Chunks per K group (og_kstep) of a row-major bank / conv_igemm3_kernel launch: a function of what both the pack job
// and the launch know (channels, taps, source and pixel-grid heights).  The resident workgroups of an XCD cover ~8192
// output pixels; one 16-channel chunk of their source pixels is ~0.8 MB at stride 1 (3 MB at stride 2), and the groups
// are sized so that a group's taps find their lines in the 4 MB L2.
static int og_kgroup(int C, int Tg, int H, int PH) {
    const int spt = (C + 15) / 16;
    if (Tg <= 1) return spt;
    const int G = (H > PH + PH / 2) ? og_kgroup_s2() : og_kgroup_s1();
    return (G <= 0 || G > spt) ? spt : G;
}
static int og_kgroup_phases(int C) {
    const int spt = (C + 15) / 16, G = og_kgroup_ph();
    return (G <= 0 || G > spt) ? spt : G;
}

// Which packed-bank layout (PackArgs::m_major) a call with these arguments uses; MT_out = accumulator
// count of the thin kernel when the answer is 2.  The single source of truth for
// objgan_conv_igemm and objgan_conv_bank_layout.
static int og_bank_layout(int N, int C, int H, int W, int M, int Tg, int PH, int PW, int act, int math,
                          int* MT_out) {
    const long Cp = ((long)C + 15) / 16 * 16;
    const bool v2 = !og_igemm_v1() && (double)N * C * H * W * 4.0 < 4.0e9 && (double)M * Tg * Cp * 4.0 < 4.0e9;
    const int MT = M <= 4 ? 4 : (M <= 12 ? 12 : (M <= 16 ? 16 : (M <= 24 ? 24 : 32)));
    if (MT_out) *MT_out = MT;
    if (!v2) return 0;
    // thin outputs: direct VALU kernel (full-coverage or strided-phase launches alike), always fp32
    const bool thin = !og_nothin() && M <= 32 && (Tg == 9 || Tg == 4) && (long)N * PH * PW >= 65536
                      && (MT <= 4 || (act != OG_ACT_TANH && act != OG_ACT_SIGMOID));
    if (thin) return 2;
    return math == 1 ? 3 : (math == 2 ? 4 : (math >= 4 ? 5 : 1));
}

// The PackArgs of objgan_conv_igemm for these arguments (single source of truth for the call itself and for
// objgan_conv_pack_job); returns the bank layout class, MT_out as og_bank_layout.
static int og_fill_pack(PackArgs& p, const float* w, float* wt, int N, int C, int H, int W, int Cout, int Cin,
                        int Torig, int transpose, int Tg, const int* src_tap, int PH, int PW, int act, int math,
                        int* MT_out) {
    const int M = transpose ? Cin : Cout;
    p.w = w; p.wt = wt; p.Cout = Cout; p.Cin = Cin; p.Torig = Torig; p.Tg = Tg;
    p.M = M; p.Mpad = (M + 127) / 128 * 128; p.Ck = C; p.Cp = (C + 15) / 16 * 16;
    p.transpose = transpose;
    int MT = 32;
    p.m_major = og_bank_layout(N, C, H, W, M, Tg, PH, PW, act, math, &MT);
    p.kgroup = og_kgroup(C, Tg, H, PH);
    p.wmax = wt ? wt + objgan_conv_packed_floats(M, C, Tg) - OG_AMAX_SLOTS : nullptr; p.wexp = 0;
    if (p.m_major == 2) p.Mpad = MT;
    for (int t = 0; t < OG_MAX_TAPS; ++t) p.src_tap[t] = (signed char)(t < Tg ? src_tap[t] : -1);
    if (MT_out) *MT_out = MT;
    return p.m_major;
}

// the four phase banks of objgan_conv_dgrad_s2_phases share one buffer of 4 * ceil(1.5 * M * Tg * Cp) + 1024 floats: the
// partial maxima of |w| sit behind the largest (bf16x3) bank size, whatever the arithmetic
static inline long og_phase_wmax_offset(int M, int Tg, int Cp) { return 4 * (((long)M * Tg * Cp * 3 + 1) / 2); }

static void og_fill_pack_phase(PackArgs& p, const float* w, float* wt, int Cout, int Cin, int Torig, int Tg,
                               const int* src_tap_phase, int phase, int math) {
    const int M = Cin, C = Cout;
    const int Cp = (C + 15) / 16 * 16;
    if (math == -2) {       // thin layout of conv_thin_ph4_kernel: [C + 1][Tg][MT] per phase
        const int MT = M <= 4 ? 4 : (M <= 12 ? 12 : (M <= 16 ? 16 : (M <= 24 ? 24 : 32)));
        p.w = w; p.wt = wt + (long)phase * (C + 1) * Tg * MT; p.Cout = Cout; p.Cin = Cin; p.Torig = Torig; p.Tg = Tg;
        p.M = M; p.Mpad = MT; p.Ck = C; p.Cp = Cp;
        p.transpose = 1; p.m_major = 2; p.kgroup = 0; p.wmax = nullptr; p.wexp = 0;
        for (int t = 0; t < OG_MAX_TAPS; ++t) p.src_tap[t] = (signed char)(t < Tg ? src_tap_phase[t] : -1);
        return;
    }
    const int Kpad = Tg * Cp;
    const int Krow = og_krow(Kpad, math);
    const long bank = math ? (long)M * Krow / 2 : (long)M * Krow;
    p.w = w; p.wt = wt + phase * bank; p.Cout = Cout; p.Cin = Cin; p.Torig = Torig; p.Tg = Tg;
    p.M = M; p.Mpad = (M + 127) / 128 * 128; p.Ck = C; p.Cp = Cp;
    p.transpose = 1; p.m_major = math == 1 ? 3 : (math == 2 ? 4 : (math >= 4 ? 5 : 1));
    p.kgroup = og_kgroup_phases(C);
    p.wmax = wt + og_phase_wmax_offset(M, Tg, Cp); p.wexp = 0;
    for (int t = 0; t < OG_MAX_TAPS; ++t) p.src_tap[t] = (signed char)(t < Tg ? src_tap_phase[t] : -1);
}


// Partial maxima of |x| over a tensor: out[0..OG_AMAX_SLOTS) (fp16x2's scale input; the consumers reduce the slots
// themselves -- no zeroed accumulator, no atomics, one launch; grid <= OG_AMAX_SLOTS workgroups own the slots).
__global__ __launch_bounds__(256) void absmax_partials_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    const long n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 256L * gridDim.x) {
        const f32x4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
    og_amax_own(og_block_max(m, red), out, blockIdx.x, gridDim.x);
}

static void og_absmax_launch(const float* x, long n, float* out, hipStream_t s) {
    long g = (n / 4 + 255) / 256;                   // one float4 per thread and trip
    g = g < 1 ? 1 : (g > OG_AMAX_SLOTS ? OG_AMAX_SLOTS : g);
    hipLaunchKernelGGL(absmax_partials_kernel, dim3((int)g), dim3(256), 0, s, x, n, out);
}

extern "C" {

// out[1024] = partial maxima of |x[0..n)| (x 16-byte aligned): the scale input of the fp16x2 arithmetic (math 4).
int objgan_absmax_partials(const float* x, long n, float* out, void* stream) {
    OG_ENTRY();
    if (!x || !out || n <= 0 || ((size_t)x & 15)) return OG_BAD_ARGS;
    og_absmax_launch(x, n, out, (hipStream_t)stream);
    return og_launch_status();
}

// ---- batched re-packing of cached filter banks ---------------------------------------------------
// A job is an opaque blob of objgan_conv_pack_job_bytes() bytes describing "pack w into wt exactly as
// objgan_conv_igemm (or phase `phase` of objgan_conv_dgrad_s2_phases) would for these arguments".  The
// caller keeps the blobs of all banks it caches for a network back to back in DEVICE memory and refreshes
// them with one launch after the network's weights changed.
int objgan_conv_pack_job_bytes() { return (int)sizeof(PackArgs); }

int objgan_conv_pack_job(void* job, const float* w, float* wt, int N, int C, int H, int W, int Cout, int Cin,
                         int Torig, int transpose, int Tg, const int* src_tap, int PH, int PW, int act, int math) {
    if (!job || Tg < 1 || Tg > OG_MAX_TAPS || Torig < 1 || Torig > 127) return OG_BAD_ARGS;
    if ((transpose ? Cout : Cin) != C) return OG_BAD_ARGS;
    PackArgs p;
    memset(&p, 0, sizeof(p));
    og_fill_pack(p, w, wt, N, C, H, W, Cout, Cin, Torig, transpose, Tg, src_tap, PH, PW, act, math, nullptr);
    memcpy(job, &p, sizeof(p));
    return OG_OK;
}

int objgan_conv_pack_job_phase(void* job, const float* w, float* wt, int Cout, int Cin, int Torig, int Tg,
                               const int* src_tap_phase, int phase, int math) {
    if (!job || Tg < 1 || Tg > 8 || phase < 0 || phase > 3 || Torig < 1 || Torig > 127) return OG_BAD_ARGS;
    PackArgs p;
    memset(&p, 0, sizeof(p));
    og_fill_pack_phase(p, w, wt, Cout, Cin, Torig, Tg, src_tap_phase, phase, math);
    memcpy(job, &p, sizeof(p));
    return OG_OK;
}

// jobs_dev: njobs blobs in device memory.
int objgan_conv_pack_jobs_run(const void* jobs_dev, int njobs, void* stream) {
    OG_ENTRY();
    if (njobs <= 0) return OG_OK;
    if (!jobs_dev || njobs > 65535) return OG_BAD_ARGS;
    // (fp16x2 jobs first leave the partial maxima of their weights behind their banks: the scale of the pack)
    hipLaunchKernelGGL(absmax_w_jobs_kernel, dim3(64, njobs), dim3(256), 0, (hipStream_t)stream, (const PackArgs*)jobs_dev);
    hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(OG_PACK_BLOCKS, njobs), dim3(256), 0, (hipStream_t)stream,
                       (const PackArgs*)jobs_dev);
    return og_launch_status();
}

// Layout of the packed bank objgan_conv_igemm would write / expect for these arguments: low byte = layout class
// (0..4), bits 8.. = chunks per K group of the row-major classes (og_kstep).  A caller that keeps packed banks
// (wt_packed = 1) must key them on this value as well: the same filter can be served by different kernels and K
// orders -- hence different bank layouts -- at different sizes.
int objgan_conv_bank_layout(int N, int C, int H, int W, int M, int Tg, int PH, int PW, int act, int math) {
    const int cls = og_bank_layout(N, C, H, W, M, Tg, PH, PW, act, math, nullptr);
    return (cls == 1 || cls >= 3) ? (cls | (og_kgroup(C, Tg, H, PH) << 8)) : cls;
}

// Size (in floats) of the packed-weight scratch for an M x K GEMM.
long objgan_conv_packed_floats(int M, int C, int T) {
    const long Mpad = ((long)M + 127) / 128 * 128;
    const long Cp = ((long)C + 15) / 16 * 16;
    // the pre-split bank of the bf16x3 mode takes 6 bytes per element; + OG_AMAX_SLOTS floats: partial maxima of |w| (fp16x2)
    return (Mpad * Cp * T * 3 + 1) / 2 + OG_AMAX_SLOTS;
}

// General entry: see the formula at the top of this file.
//   w        PyTorch-layout conv weight [Cout][Cin][Torig]
//   wt       scratch of objgan_conv_packed_floats(M, C*Tg) floats (overwritten)
//   transpose 0: M = Cout, C = Cin ; 1: M = Cin, C = Cout (data gradient)
//   src_tap[t] index of GEMM tap t in the Torig taps of w (or -1 for a zero tap)
// Fills the PackArgs / IgemmArgs of objgan_conv_igemm (shared with objgan_conv_igemm_ws_floats); returns OG_OK,
// OG_BAD_ARGS or 2 for "nothing to do".
static int og_igemm_setup(PackArgs& p, IgemmArgs& a, int& MT, const float* x, const float* w, const float* bias, float* y,
                          float* wt, int N, int C, int H, int W, int upsample, int pad_mode, int Cout, int Cin, int Torig,
                          int transpose, int Tg, const int* dh, const int* dw, const int* src_tap, int PH, int PW,
                          int stride, int OHf, int OWf, int osh, int osw, int ooh, int oow, int act, int math, float* ring,
                          const float* xmax = nullptr) {
    if (Tg < 1 || Tg > OG_MAX_TAPS) return OG_BAD_ARGS;
    if (ring && !(osh == 1 && osw == 1 && ooh == 0 && oow == 0 && PH == OHf + 2 && PW == OWf + 2 && !bias && !act))
        return OG_BAD_ARGS;
    if (math < 0 || math > 5) return OG_BAD_ARGS;
    // math 3 (round 6): the arithmetic of math 1 (bf16-rounded operands) with the pixel operand handed over AS its bf16
    // channel-blocked copy (objgan_nhwc_bf16) -- the copy of a tensor is made once and serves every convolution that reads it
    // (the forward call AND the weight gradient of the layer, every branch of an Inception block) instead of once per call
    const bool copy_in = math == 3;
    if (copy_in) math = 1;
    if (Torig < 1 || Torig > 127) return OG_BAD_ARGS;
    const int M = transpose ? Cin : Cout;
    const int Ck = transpose ? Cout : Cin;
    if (Ck != C) return OG_BAD_ARGS;
    if (N <= 0 || PH <= 0 || PW <= 0 || M <= 0) return 2;
    MT = 32;
    og_fill_pack(p, w, wt, N, C, H, W, Cout, Cin, Torig, transpose, Tg, src_tap, PH, PW, act, math, &MT);
    if (math == 5 && p.m_major != 5) return OG_BAD_ARGS;       // records: the MFMA implicit-GEMM kernel only (ask objgan_conv_bank_layout)
    const int kmath = p.m_major == 3 ? 1 : (p.m_major == 4 ? 2 : (p.m_major == 5 ? (math == 5 ? 5 : 4) : 0));   // arithmetic of the kernel that runs
    a.x = x; a.wt = wt; a.bias = bias; a.y = y;
    a.N = N; a.C = C; a.H = H; a.W = W;
    a.LH = upsample ? 2 * H : H; a.LW = upsample ? 2 * W : W;
    a.M = M; a.Mpad = p.Mpad; a.K = C * Tg; a.Kpad = Tg * p.Cp; a.T = Tg; a.Cp = p.Cp; a.kgroup = p.kgroup;
    a.math = kmath;
    a.xmax = xmax; a.wmax = p.wmax;
    if (kmath >= 4 && x && !xmax) return OG_BAD_ARGS;      // fp16x2 needs the maxima of its pixel operand (x == NULL: size query)
    a.nhwc = igemm2_nhwc_floats(kmath, N, H, W, p.Cp) > 0 ? 1 : 0;
    if (copy_in) {
        if (p.m_major != 3 || !a.nhwc) return OG_BAD_ARGS;    // the bf16 MFMA kernel only (ask objgan_conv_bank_layout: class 3)
        a.nhwc = 2;                                            // x IS the copy: nothing to make, nothing in the workspace
    }
    a.Krow = og_krow(a.Kpad, kmath);
    a.m_begin = 0; a.m_end = M;
    a.PH = PH; a.PW = PW; a.OHf = OHf; a.OWf = OWf;
    a.osh = osh; a.osw = osw; a.ooh = ooh; a.oow = oow;
    a.stride = stride; a.pad_mode = pad_mode; a.upsample = upsample; a.act = act;
    a.ksplit_steps = 0;
    a.nphase = 0;
    a.ring = ring;
    a.ws = nullptr; a.ws_stride = 0;
#ifdef OG_DEV
    a.ablate = og_ablate();
#endif
    for (int t = 0; t < OG_MAX_TAPS; ++t) {
        const int h = t < Tg ? dh[t] : 0, w_ = t < Tg ? dw[t] : 0;
        a.tap[t] = (int)(((unsigned)w_ << 16) | ((unsigned)h & 0xffffu));
    }
    if (!(osh == 1 && osw == 1 && PH == OHf && PW == OWf) && (bias || act)) return OG_BAD_ARGS;
    if (ring && p.m_major != 1 && p.m_major != 3 && p.m_major != 4 && p.m_major != 5) return OG_BAD_ARGS;     // ring mode: conv_igemm3_kernel only (ask objgan_conv_bank_layout)
    return OG_OK;
}

// Floats of split-K workspace objgan_conv_igemm needs for these arguments (0: none).  Small-grid / long-K launches
// are split along K: every split stores its partial output tile into its own workspace slot and a second kernel sums
// the slots in split order (+ bias, activation) -- bit-reproducible, no zero-fill of y, no atomics.  Host-only.
long objgan_conv_igemm_ws_floats(int N, int C, int H, int W, int upsample, int pad_mode,
                                 int Cout, int Cin, int Torig, int transpose, int Tg,
                                 int PH, int PW, int stride, int OHf, int OWf, int osh, int osw,
                                 int act, int y_prezeroed, int math, int ring) {
    PackArgs p;
    IgemmArgs a;
    int MT = 32;
    int zeros[OG_MAX_TAPS] = {0};
    float dummy = 0.f;
    memset(&p, 0, sizeof(p));
    const int rc = og_igemm_setup(p, a, MT, nullptr, nullptr, nullptr, nullptr, nullptr, N, C, H, W, upsample, pad_mode, Cout, Cin,
                                  Torig, transpose, Tg, zeros, zeros, zeros, PH, PW, stride, OHf, OWf, osh, osw, 0, 0, act,
                                  math, ring ? &dummy : nullptr);
    if (rc != OG_OK || p.m_major == 0 || p.m_major == 2) return 0;
    return igemm2_ws_floats(a, igemm2_plan(a, y_prezeroed));
}

int objgan_conv_igemm(const float* x, const float* w, const float* bias, float* y, float* wt,
                      int N, int C, int H, int W, int upsample, int pad_mode,
                      int Cout, int Cin, int Torig, int transpose,
                      int Tg, const int* dh, const int* dw, const int* src_tap,
                      int PH, int PW, int stride,
                      int OHf, int OWf, int osh, int osw, int ooh, int oow,
                      int act, int y_prezeroed, int wt_packed, int math, float* ring, const float* xmax, float* ymax,
                      float* ws, long ws_floats, void* stream) {
    OG_ENTRY();
    PackArgs p;
    IgemmArgs a;
    int MT = 32;
    const int rc0 = og_igemm_setup(p, a, MT, x, w, bias, y, wt, N, C, H, W, upsample, pad_mode, Cout, Cin, Torig, transpose,
                                   Tg, dh, dw, src_tap, PH, PW, stride, OHf, OWf, osh, osw, ooh, oow, act, math, ring, xmax);
    if (rc0 == 2) return OG_OK;
    if (rc0 != OG_OK) return rc0;
    hipStream_t s = (hipStream_t)stream;
    const bool v2 = p.m_major != 0, thin = p.m_major == 2;
    if (!wt_packed) {       // wt_packed: the caller kept wt from an earlier call with the same
        if (p.m_major == 5)
            hipLaunchKernelGGL(absmax_w_kernel, dim3(64), dim3(256), 0, s, w, (long)Cout * Cin * Torig, const_cast<float*>(p.wmax));
        const long ptotal = (long)Tg * p.Cp * p.Mpad;   // filter bank, taps, math and geometry class
        hipLaunchKernelGGL(pack_weights_kernel, dim3(og_stream_grid(ptotal, 256)), dim3(256), 0, s, p);
        int rc = og_launch_status();
        if (rc != OG_OK) return rc;
    }
    if (thin || !v2) {
        a.ymax = nullptr;
        const int rc = thin ? run_thin(a, MT, s) : run_igemm(a, s, y_prezeroed);
        if (rc == OG_OK && ymax) og_absmax_launch(y, (long)N * a.M * OHf * OWf, ymax, s);
        return rc == OG_OK ? og_launch_status() : rc;
    }
    return run_igemm2(a, s, y_prezeroed, ws, ws_floats, ymax);
}

// Data gradient of a stride-2 convolution whose four output parity phases have the same tap count
// (k = 4, pad 1, even sizes: 2x2 taps each): ONE launch, the phase is the fastest digit of the workgroup id.  x = dY [N, Cout, OH, OW],
// y = dX [N, Cin, 2*PH, 2*PW] (every element is written by exactly one phase: no pre-zeroing).
// dh/dw/src_tap: 4 phases x Tg entries, phase p = (row parity << 1) | column parity.
// wt: 4 * ceil(1.5 * Cin * Tg * ceil16(Cout)) + 1024 floats (the four phase banks, then the partial maxima of |w|).
int objgan_conv_dgrad_s2_phases(const float* x, const float* w, float* y, float* wt,
                                int N, int Cout, int OH, int OW, int Cin, int Torig,
                                int Tg, const int* dh, const int* dw, const int* src_tap,
                                int PH, int PW, int wt_packed, int math, const float* xmax, float* ws, long ws_floats,
                                void* stream) {
    OG_ENTRY();
    if (Tg < 1 || Tg > 8) return OG_BAD_ARGS;
    if (math < 0 || math > 5 || math == 3 || (math >= 4 && !xmax)) return OG_BAD_ARGS;
    if (Torig < 1 || Torig > 127) return OG_BAD_ARGS;
    if (N <= 0 || PH <= 0 || PW <= 0 || Cin <= 0) return OG_OK;
    const int M = Cin, C = Cout;
    const int Cp = (C + 15) / 16 * 16;
    if ((double)N * C * OH * OW * 4.0 >= 4.0e9 || (double)M * Tg * Cp * 4.0 >= 4.0e9) return OG_BAD_ARGS;
    hipStream_t s = (hipStream_t)stream;
    const int Kpad = Tg * Cp;
    const int Krow = og_krow(Kpad, math);
    // phase banks are stored back to back: `bank` floats apart (Krow counts bf16 elements in the bf16 modes)
    const long bank = math ? (long)M * Krow / 2 : (long)M * Krow;
    if (!wt_packed) {
        if (math >= 4)
            hipLaunchKernelGGL(absmax_w_kernel, dim3(64), dim3(256), 0, s, w, (long)Cout * Cin * Torig,
                               wt + og_phase_wmax_offset(M, Tg, Cp));
        for (int ph = 0; ph < 4; ++ph) {
            PackArgs p;
            og_fill_pack_phase(p, w, wt, Cout, Cin, Torig, Tg, src_tap + ph * Tg, ph, math);
            hipLaunchKernelGGL(pack_weights_kernel, dim3(og_stream_grid((long)M * Krow, 256)), dim3(256), 0, s, p);
            int rc = og_launch_status();
            if (rc != OG_OK) return rc;
        }
    }
    IgemmArgs a;
    a.x = x; a.wt = wt; a.bias = nullptr; a.y = y;
    a.N = N; a.C = C; a.H = OH; a.W = OW; a.LH = OH; a.LW = OW;
    a.M = M; a.Mpad = (M + 127) / 128 * 128; a.K = C * Tg; a.Kpad = Kpad; a.T = Tg; a.Cp = Cp;
    a.kgroup = og_kgroup_phases(C);
    a.math = math; a.Krow = Krow; a.xmax = xmax; a.wmax = wt + og_phase_wmax_offset(M, Tg, Cp);
    a.nhwc = igemm2_nhwc_floats(math, N, OH, OW, Cp) > 0 ? 1 : 0;
    a.m_begin = 0; a.m_end = M;
    a.PH = PH; a.PW = PW; a.OHf = 2 * PH; a.OWf = 2 * PW;
    a.osh = 2; a.osw = 2; a.ooh = 0; a.oow = 0;
    a.stride = 1; a.pad_mode = 0; a.upsample = 0; a.act = OG_ACT_NONE;
    a.ksplit_steps = 0;
    a.ring = nullptr;
    a.ws = nullptr; a.ws_stride = 0;
    a.nphase = 4;
#ifdef OG_DEV
    a.ablate = og_ablate();
#endif
    for (int t = 0; t < OG_MAX_TAPS; ++t) a.tap[t] = 0;
    for (int ph = 0; ph < 4; ++ph)
        for (int t = 0; t < Tg; ++t)
            a.tap[ph * 8 + t] = (int)(((unsigned)dw[ph * Tg + t] << 16) | ((unsigned)dh[ph * Tg + t] & 0xffffu));
    return run_igemm2(a, s, 0, ws, ws_floats);  // (four phases in one launch: never split along K)
}

// floats of workspace objgan_conv_dgrad_s2_phases takes (bf16 mode: the channel-blocked bf16 copy of dY; else 0)
long objgan_conv_dgrad_s2_phases_ws_floats(int N, int Cout, int OH, int OW, int math) {
    return igemm2_nhwc_floats(math, N, OH, OW, (Cout + 15) / 16 * 16);
}

// Data gradient of a 4 x 4 / stride-2 / pad-1 convolution w.r.t. an input of Cin <= 32 channels, all four output parity
// phases in ONE launch of the fp32 VALU kernel (conv_thin_ph4_kernel: dy read twice instead of four times): dy [N, Cout, OH, OW]
// -> dx [N, Cin, 2 OH, 2 OW], every element written exactly once (no pre-zeroing).  w [Cout][Cin][16]; wt: objgan_conv_dgrad_s2_thin_floats(Cout, Cin) floats
// (the four phase banks of the thin layout), packed by the call unless wt_packed.
static inline int og_thin_mt(int M) { return M <= 4 ? 4 : (M <= 12 ? 12 : (M <= 16 ? 16 : (M <= 24 ? 24 : 32))); }
static void og_thin_phase_taps(int phase, int* st) {      // source taps of phase (pa, pb), t = i * 2 + j (see the kernel)
    const int pa = phase >> 1, pb = phase & 1;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const int kh = pa ? 2 * i : 1 + 2 * i, kw = pb ? 2 * j : 1 + 2 * j;     // dh = (pa + 1 - kh) / 2: 0, -1 | 1, 0
            st[i * 2 + j] = kh * 4 + kw;
        }
}
long objgan_conv_dgrad_s2_thin_floats(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin > 32) return 0;
    return 4L * (Cout + 1) * 4 * og_thin_mt(Cin);
}
// the pack job of phase `phase` of that bank set (for objgan_conv_pack_jobs_run)
int objgan_conv_pack_job_thin_phase(void* job, const float* w, float* wt, int Cout, int Cin, int phase) {
    if (!job || phase < 0 || phase > 3 || Cin > 32 || Cin <= 0 || Cout <= 0) return OG_BAD_ARGS;
    int st[4];
    og_thin_phase_taps(phase, st);
    PackArgs p;
    memset(&p, 0, sizeof(p));
    og_fill_pack_phase(p, w, wt, Cout, Cin, 16, 4, st, phase, -2);
    memcpy(job, &p, sizeof(p));
    return OG_OK;
}
int objgan_conv_dgrad_s2_thin(const float* dy, const float* w, float* dx, float* wt, int N, int Cout, int OH, int OW,
                              int Cin, int wt_packed, void* stream) {
    OG_ENTRY();
    if (Cin <= 0 || Cin > 32 || Cout <= 0) return OG_BAD_ARGS;
    if (N <= 0 || OH <= 0 || OW <= 0) return OG_OK;
    if ((double)N * Cout * OH * OW * 4.0 >= 4.0e9 || (double)N * OH * OW >= 2.0e9) return OG_BAD_ARGS;
    hipStream_t s = (hipStream_t)stream;
    const int MT = og_thin_mt(Cin);
    if (!wt_packed) {
        for (int ph = 0; ph < 4; ++ph) {
            int st[4];
            og_thin_phase_taps(ph, st);
            PackArgs p;
            memset(&p, 0, sizeof(p));
            og_fill_pack_phase(p, w, wt, Cout, Cin, 16, 4, st, ph, -2);
            hipLaunchKernelGGL(pack_weights_kernel, dim3(og_stream_grid((long)(Cout + 1) * 4 * MT, 256)), dim3(256), 0, s, p);
            int rc = og_launch_status();
            if (rc != OG_OK) return rc;
        }
    }
    const long Npos = (long)N * OH * OW;
    const int PX = MT <= 16 ? 2 : 1;
    dim3 grid(og_cdiv(Npos, 256 * PX), 2);           // y: row parity of the output rows
    ProfRec* pr = prof_begin(OG_CAT_THIN, 2.0 * Cin * (double)Cout * 4.0 * (double)Npos * 4.0, s);
    prof_meta(pr, 2, MT, Cin, Cout, 4, N, 4 * OH, OW, -1, 1);
    switch (MT) {
        case 4: hipLaunchKernelGGL((conv_thin_ph4_kernel<4, 2>), grid, dim3(256), 0, s, dy, wt, dx, N, Cout, OH, OW, Cin); break;
        case 12: hipLaunchKernelGGL((conv_thin_ph4_kernel<12, 2>), grid, dim3(256), 0, s, dy, wt, dx, N, Cout, OH, OW, Cin); break;
        case 16: hipLaunchKernelGGL((conv_thin_ph4_kernel<16, 2>), grid, dim3(256), 0, s, dy, wt, dx, N, Cout, OH, OW, Cin); break;
        case 24: hipLaunchKernelGGL((conv_thin_ph4_kernel<24, 1>), grid, dim3(256), 0, s, dy, wt, dx, N, Cout, OH, OW, Cin); break;
        default: hipLaunchKernelGGL((conv_thin_ph4_kernel<32, 1>), grid, dim3(256), 0, s, dy, wt, dx, N, Cout, OH, OW, Cin); break;
    }
    prof_end(pr, s);
    return og_launch_status();
}

// y [planes, H, W] += mirror of ring [planes, 2*(W+2) + 2*(H+2)] (written by objgan_conv_igemm in ring mode).
int objgan_reflect_ring_fold(const float* ring, float* y, long planes, int H, int W, void* stream) {
    OG_ENTRY();
    if (H < 3 || W < 3) return OG_BAD_ARGS;
    if (planes <= 0) return OG_OK;
    const long total = planes * (2 * W + 2 * H);
    hipLaunchKernelGGL(reflect_ring_fold_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       ring, y, planes, H, W);
    return og_launch_status();
}

}  // extern "C"

// Does the record form of the weight gradient (conv_wgrad_rec_kernel, math 5) take this geometry?
static bool og_wgrad_rec_geometry(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ksize) {
    const long OHW = (long)OH * OW;
    const long Cp = ((long)Cin + 15) / 16 * 16;
    return !og_igemm_v1() && (OW % 8 == 0) && (OHW % 32 == 0) && OH <= 256 && OW <= 256 && (long)(H - 1) * W < 65535
           && (ksize == 1 || ksize == 3 || ksize == 4)
           && (double)N * Cp * H * W * 4.0 < 4.0e9 && (double)N * Cout * OHW * 4.0 < 4.0e9;
}

// Plan and (unless ws_need != nullptr: plan only) run the weight gradient.  The reduction over pixels is split across
// gridDim.y; each split writes its partial tile into its own workspace slot and wgrad_combine_kernel sums the slots
// in split order into dw (stored, or added when `accumulate`): bit-reproducible, dw needs no zero-fill.
static int og_wgrad(const float* x, const float* dy, float* dw,
                    int N, int Cin, int H, int W, int upsample, int pad_mode,
                    int Cout, int OH, int OW, int ksize, int stride, int pad,
                    int math, int accumulate, float* ws, long ws_floats, long* ws_need, hipStream_t s,
                    const float* xmax = nullptr, const float* dymax = nullptr) {
    const bool plan_only = ws_need != nullptr;
    long ws_used = 0;
    if (plan_only) *ws_need = 0;
    if (ksize != 1 && ksize != 3 && ksize != 4) return OG_BAD_ARGS;
    if (math < 0 || math > 7) return OG_BAD_ARGS;
    // math 3 (round 6): math 1 with x handed over AS its bf16 channel-blocked copy (objgan_nhwc_bf16; usually the one the
    // forward call of the layer read): only the bf16 copy of dy is made here.  objgan_conv_wgrad_bfb_ok says where.
    const bool x_copy_in = math == 3;
    if (x_copy_in) math = 1;
    if (math >= 4 && !plan_only && (!xmax || !dymax)) return OG_BAD_ARGS;
    if (N <= 0 || Cout <= 0 || Cin <= 0) return OG_OK;
    // math 5: x is the fp16 record of the source (conv_igemm_rec.hip), dy the fp32 tensor: conv_wgrad_rec_kernel.  The
    // caller asks objgan_conv_wgrad_rec_ok first; a geometry the record kernel does not take is an argument error here.
    // math 6 (round 6): the same with dy pre-split too -- its fp16 pair is written into the workspace by one pass
    // (h2_pair_kernel) and the kernel's K loop carries no operand split at all.
    // math 7 (round 6): the record form with TWO column groups per wave (conv_wgrad_rec2_kernel: block rows <= 128 rows,
    // 8-wave workgroups); launches of fewer than 16 384 pixels keep the one-group kernel.
    const bool rec = math == 5 || math == 6 || math == 7;
    const bool dyp = math == 6;
    const bool rec2 = math == 7 && (long)N * OH * OW >= 16384;
    if (rec && !og_wgrad_rec_geometry(N, Cin, H, W, Cout, OH, OW, ksize)) return OG_BAD_ARGS;
    // fp16x2 lives in the register-fragment kernel; launches that plan the LDS-staged / first-generation kernels run
    // bf16x3 (both are fp32-result arithmetics)
    const bool h2 = math == 4;
    if (h2 || rec) math = 2;
    WgradArgs a;
    a.xmax = xmax; a.dymax = dymax;
    a.ws = nullptr; a.ws_stride = 0; a.accumulate = accumulate;
    a.x = x; a.dy = dy; a.dw = dw;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W;
    a.LH = upsample ? 2 * H : H; a.LW = upsample ? 2 * W : W;
    a.Cout = Cout; a.OH = OH; a.OW = OW;
    a.stride = stride; a.pad = pad; a.pad_mode = pad_mode; a.upsample = upsample;
    a.ncol = Cin * ksize * ksize;
    a.xr_begin = 0; a.xr_count = 0;
    const int Npix = N * OH * OW;

    const int OHW = OH * OW;
    const bool v2 = !og_igemm_v1() && (OW % 8 == 0) && (OHW % 16 == 0)
                    && (double)N * Cin * H * W * 4.0 < 4.0e9 && (double)N * Cout * OHW * 4.0 < 4.0e9;
    a.math = math;
    if (x_copy_in && !v2) return OG_BAD_ARGS;
    if (v2) {
        const bool bf = math == 1, sp = math == 2;
        // bf16 mode: bf16 operands, x from its channel-blocked copy in the workspace (conv_wgrad_bfb_kernel); without a
        // workspace (or on maps of fewer than 32 pixels) the fp32-gather kernels below
        const int Cpb = (Cin + 15) / 16 * 16;
        bool bfb = bf && OHW % 32 == 0 && OH <= 256 && OW <= 256 && (long)(H - 1) * W < 65535 && ksize <= 4
                   && (double)N * Cpb * H * W * 2.0 < 4.0e9 && (plan_only || ws != nullptr);
        if (x_copy_in && !bfb) return OG_BAD_ARGS;
        const long xb_floats = (bfb && !x_copy_in) ? (((long)N * H * W * Cpb / 2 + 3) & ~3L) : 0;
        const long dyb_floats = bfb ? (((long)N * Cout * OHW / 2 + 3) & ~3L) : 0;       // (OHW % 32 == 0: a multiple of 4)
        const __bf16* xb = nullptr;
        const __bf16* dyb = nullptr;
        if (bfb) {
            if (plan_only) { *ws_need += xb_floats + dyb_floats; }
            else {
                if (ws_floats < xb_floats + dyb_floats) return OG_BAD_ARGS;
                if (!x_copy_in)
                    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(og_cdiv(H * W, 64), og_cdiv(Cpb, 64), N), dim3(256), 0, s,
                                       x, reinterpret_cast<__bf16*>(ws), Cin, H * W, Cpb);
                const long n4 = (long)N * Cout * OHW / 4;
                hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(og_stream_grid(n4, 256)), dim3(256), 0, s, dy,
                                   reinterpret_cast<__bf16*>(ws + xb_floats), n4);
                xb = x_copy_in ? reinterpret_cast<const __bf16*>(x) : reinterpret_cast<const __bf16*>(ws);
                dyb = reinterpret_cast<const __bf16*>(ws + xb_floats);
                ws_used = xb_floats + dyb_floats;
            }
        }
        if (dyp) {                                         // the fp16 pair of dy: N * Cout * OHW floats' worth of workspace
            const long dyp_floats = ((long)N * Cout * OHW + 3) & ~3L;
            if (plan_only) { *ws_need += dyp_floats; }
            else {
                if (!ws || ws_floats < dyp_floats) return OG_BAD_ARGS;
                og_launch_h2_pair(dy, dymax, ws, (long)N * Cout * OHW, s);
                a.dy = ws;
                ws_used = dyp_floats;
            }
        }
        int groups = og_cdiv(Cout, 32);
        const int tiles_n0 = og_cdiv(a.ncol, 128);
        // 1..4 output channels beyond a multiple of 32: on the VALU of block row 0 (see WgradArgs)
        const int tail_rows = Cout & 31;
        bool xrows = !og_no_xrows() && math == 0 && tail_rows >= 1 && tail_rows <= 4 && Cout >= 64;
        int TM, full_rows, rest;
        if (xrows) {
            og_row_plan(groups - 1, tiles_n0, 1, &TM, &full_rows, &rest);
            if (TM >= 2 && full_rows >= 1) groups -= 1; else xrows = false;
        }
        og_row_plan(groups, tiles_n0, 1, &TM, &full_rows, &rest, 100, rec2 ? 4 : (rec ? og_wgrad_rec_tmmax() : 7));      // (tall tiles: independent of the column tiling)
        a.xr_begin = groups * 32;
        for (int part = 0; part < 2; ++part) {
            const int tm = part == 0 ? TM : rest;
            const int rows = part == 0 ? full_rows : (rest ? 1 : 0);
            if (rows == 0) continue;
            const int m_cap = xrows ? groups * 32 : Cout;
            a.m_begin = part == 0 ? 0 : full_rows * TM * 32;
            a.m_end = part == 0 ? (m_cap < full_rows * TM * 32 ? m_cap : full_rows * TM * 32) : m_cap;
            a.xr_count = (part == 0 && xrows) ? tail_rows : 0;
            // split K (pixels): the launch takes about (workgroups per CU, rounded up) x (K steps per
            // split + a fixed prologue / atomic-epilogue cost); pick the split count that minimises it
            // (r02: `slots / workgroups` left the 288-workgroup launches of the 16x16 maps at 1 split --
            // 32 CUs with two workgroups, 224 with one -- 63 TFLOP/s).  OG_WGRAD_OLDSPLIT=1: previous rule.
            // which form (see the comment at the launch below) and how many waves per workgroup: decided here because
            // the column-tile count of the launch depends on it
            const bool wide_s1 = stride == 1 && !upsample && OW >= 64;
            const bool b128 = wide_s1 && !og_wgrad_nob128() && !bf;
            // (bf16x3: the LDS-staged form is instruction-issue bound -- 8 VALU per MFMA for per-element gather addresses
            // plus the split of both operands, profiles/r03_x3_pmc_objd_l3.txt -- so the register-fragment form also
            // takes the tall tiles wherever its constant-stride gather path applies: zero padding, no upsampling;
            // r03 A/B: objd_l2 / objd_l3 155 -> 170 TFLOP/s, upsampled sources 119 -> 109)
            const bool x3_frag = tm <= og_x3_wgrad3_maxtm() || b128 || (!upsample && !pad_mode && og_x3_wgrad3_maxtm() >= 0);
            const bool use3 = bf ? tm <= 2 : (sp ? x3_frag : (tm <= og_wgrad3_maxtm() || b128));
            // bf16x3, register-fragment form: 8-wave workgroups (256 columns per dy row tile), as in run_igemm2
            const int nw = bfb ? ((tm <= 6 && Npix >= 16384) ? 8 : 4)
                               : (rec2 ? 8 : rec ? ((tm <= 6 && Npix >= 16384 && og_wgrad_rec_nw8()) ? 8 : 4)
                                      : ((sp && use3 && tm >= 4 && og_nw8_min() > 0 && Npix >= 16384) ? 8 : 4));
            // column tiles: 32 columns (ci * T + t) per wave; bfb / rec: one (tap, 32-channel group) per wave
            const int tiles_n = (bfb || rec) ? og_cdiv(ksize * ksize * og_cdiv(Cpb, 32), rec2 ? 2 * nw : nw) : og_cdiv(a.ncol, 32 * nw);
            int splits;
            const int max_splits = og_cdiv(Npix, 512);       // >= 32 K steps per split
            if (og_wgrad_oldsplit()) {
                const int slots = 256 * (tm == 1 ? 6 : (tm <= 4 ? 3 : 2));   // LDS-free TM = 1: more waves
                splits = slots / (rows * tiles_n);
                if (splits > max_splits) splits = max_splits;
                if (splits < 1) splits = 1;
            } else {
                const long wgs = (long)rows * tiles_n;
                const int nsteps = og_cdiv(Npix, 16);
                const int resident = nw == 8 ? 1 : (tm == 1 ? 6 : (tm <= 4 ? 3 : 2));
                double best = -1;
                splits = 1;
                for (int sp = 1; sp <= max_splits && sp <= 1024; ++sp) {
                    const long per_cu = og_cdiv(wgs * sp, 256);
                    // fewer co-resident workgroups than the CU can hold: nothing hides the memory latency
                    const double lat = per_cu < resident ? 1.0 + 0.15 * (resident - per_cu) : 1.0;
                    const double cost = (double)per_cu * (og_cdiv(nsteps, sp) + 10.0) * lat;
                    if (best < 0 || cost < best * 0.985) { best = cost; splits = sp; }
                }
            }
            int pps = og_cdiv(Npix, splits);
            pps = (bfb || rec) ? (pps + 31) / 32 * 32 : (pps + 15) / 16 * 16;
            splits = og_cdiv(Npix, pps);
            a.pix_per_split = pps;
            {
                const long slot = (long)(a.m_end - a.m_begin + a.xr_count) * a.ncol;
                const long need = splits > 1 ? slot * splits : 0;
                if (plan_only) { *ws_need += need; continue; }
                a.ws = nullptr; a.ws_stride = 0;
                if (splits > 1) {
                    if (!ws || ws_used + need > ws_floats) return OG_BAD_ARGS;
                    a.ws = ws + ws_used; a.ws_stride = slot; ws_used += need;
                }
            }
            dim3 grid(rows * tiles_n, splits);
            if (og_trace())
                fprintf(stderr, "OGTRACE wgrad TM=%d NW=%d form=%d Cout=%d Cin=%d k=%d N=%d OH=%d OW=%d stride=%d grid=%u,%u math=%d\n", tm, nw,
                        use3 ? 3 : 2, Cout, Cin, ksize, N, OH, OW, stride, grid.x, grid.y, math);
            ProfRec* pr = prof_begin(bfb ? OG_CAT_WGRAD2(tm)
                                         : (rec ? (nw == 8 ? OG_CAT_WGRAD3_NW8(tm) : OG_CAT_WGRAD3(tm)) + 96
                                                : (use3 ? (nw == 8 ? OG_CAT_WGRAD3_NW8(tm) : OG_CAT_WGRAD3(tm)) + (h2 ? 48 : 0)
                                                        : OG_CAT_WGRAD2(tm))),
                                     2.0 * (a.m_end - a.m_begin + a.xr_count) * (double)a.ncol * (double)Npix, s);
            prof_meta(pr, 1, tm, a.m_end - a.m_begin + a.xr_count, Cin, ksize * ksize, N, OH, OW,
                      stride * (upsample ? 10 : 1) * (pad_mode ? -1 : 1), splits);
            // LDS-free register-fragment form for short tiles, LDS-staged form for tall ones
            // register-fragment form for short tiles, LDS-staged form for tall ones (measured equal or
            // better there: the up-sampling / reflecting gathers keep their per-element address math)
            // Which form: the register-fragment kernel (wgrad3) with 16-byte gathers wins on wide stride-1
            // maps without upsampling (r02 A/B: res1_128 100 -> 107 TF, shp_512 33 -> 37), the LDS-staged
            // kernel (wgrad2) elsewhere: narrow maps spend half of their spans on the border path, the
            // up-sampling gather keeps its per-element address math.
#define OG_WG2X(TMv) if (use3 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 0, true, 4>), grid, dim3(256), 0, s, a, ksize); \
                     else if (use3) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 0, false, 4>), grid, dim3(256), 0, s, a, ksize); \
                     else hipLaunchKernelGGL((conv_wgrad2_kernel<TMv, 0, 4>), grid, dim3(256), 0, s, a, ksize);
#define OG_WG2(TMv) if (bf && TMv <= 2) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 1>), grid, dim3(256), 0, s, a, ksize); \
                    else if (bf) hipLaunchKernelGGL((conv_wgrad2_kernel<TMv, 1>), grid, dim3(256), 0, s, a, ksize); \
                    else if (h2 && use3 && nw == 8 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<(TMv > 1 ? TMv : 2), 4, true, 0, 8>), grid, dim3(512), 0, s, a, ksize); \
                    else if (h2 && use3 && nw == 8) hipLaunchKernelGGL((conv_wgrad3_kernel<(TMv > 1 ? TMv : 2), 4, false, 0, 8>), grid, dim3(512), 0, s, a, ksize); \
                    else if (h2 && use3 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 4, true>), grid, dim3(256), 0, s, a, ksize); \
                    else if (h2 && use3) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 4>), grid, dim3(256), 0, s, a, ksize); \
                    else if (sp && use3 && nw == 8 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<(TMv > 1 ? TMv : 2), 2, true, 0, 8>), grid, dim3(512), 0, s, a, ksize); \
                    else if (sp && use3 && nw == 8) hipLaunchKernelGGL((conv_wgrad3_kernel<(TMv > 1 ? TMv : 2), 2, false, 0, 8>), grid, dim3(512), 0, s, a, ksize); \
                    else if (sp && use3 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 2, true>), grid, dim3(256), 0, s, a, ksize); \
                    else if (sp && use3) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 2>), grid, dim3(256), 0, s, a, ksize); \
                    else if (sp) hipLaunchKernelGGL((conv_wgrad2_kernel<TMv, 2>), grid, dim3(256), 0, s, a, ksize); \
                    else if (use3 && b128) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv, 0, true>), grid, dim3(256), 0, s, a, ksize); \
                    else if (use3) hipLaunchKernelGGL((conv_wgrad3_kernel<TMv>), grid, dim3(256), 0, s, a, ksize); \
                    else hipLaunchKernelGGL((conv_wgrad2_kernel<TMv>), grid, dim3(256), 0, s, a, ksize);
            if (rec2) {
                const int rc_ = og_launch_wgrad_rec2(a, tm, grid, ksize, Cpb, s);
                if (rc_ != OG_OK) { prof_end(pr, s); return rc_; }
            } else if (rec) {
                const int rc_ = og_launch_wgrad_rec(a, tm, nw, grid, ksize, Cpb, dyp ? 1 : 0, s);
                if (rc_ != OG_OK) { prof_end(pr, s); return rc_; }
            } else if (bfb) {
#define OG_WGB(TMv) if (nw == 8) hipLaunchKernelGGL((conv_wgrad_bfb_kernel<(TMv <= 6 ? TMv : 6), 8>), grid, dim3(512), 0, s, a, xb, dyb, ksize, Cpb); \
                    else hipLaunchKernelGGL((conv_wgrad_bfb_kernel<TMv, 4>), grid, dim3(256), 0, s, a, xb, dyb, ksize, Cpb);
                switch (tm) {
                    case 1: OG_WGB(1) break;
                    case 2: OG_WGB(2) break;
                    case 3: OG_WGB(3) break;
                    case 4: OG_WGB(4) break;
                    case 5: OG_WGB(5) break;
                    case 6: OG_WGB(6) break;
                    default: OG_WGB(7) break;
                }
#undef OG_WGB
            } else if (a.xr_count > 0) {
                switch (tm) {
                    case 2: OG_WG2X(2) break;
                    case 3: OG_WG2X(3) break;
                    case 4: OG_WG2X(4) break;
                    case 5: OG_WG2X(5) break;
                    case 6: OG_WG2X(6) break;
                    default: OG_WG2X(7) break;
                }
            } else {
            switch (tm) {
                case 1: OG_WG2(1) break;
                case 2: OG_WG2(2) break;
                case 3: OG_WG2(3) break;
                case 4: OG_WG2(4) break;
                case 5: OG_WG2(5) break;
                case 6: OG_WG2(6) break;
                default: OG_WG2(7) break;
            }
            }
#undef OG_WG2
#undef OG_WG2X
            prof_end(pr, s);
            int rc = og_launch_status();
            if (rc != OG_OK) return rc;
            if (a.ws) {
                const long slot = a.ws_stride;
                hipLaunchKernelGGL(wgrad_combine_kernel, dim3(og_stream_grid(slot, 256)), dim3(256), 0, s, a.ws, splits, slot,
                                   dw, a.ncol, a.m_begin, a.m_end - a.m_begin, a.xr_begin, slot, accumulate);
                rc = og_launch_status();
                if (rc != OG_OK) return rc;
            }
        }
        return OG_OK;
    }

    if (og_trace())
        fprintf(stderr, "OGTRACE wgrad(v1) Cout=%d Cin=%d k=%d N=%d OH=%d OW=%d stride=%d\n", Cout, Cin, ksize, N, OH, OW, stride);
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
        {
            const long slot = (long)(a.m_end - a.m_begin) * a.ncol;
            const long need = splits > 1 ? slot * splits : 0;
            if (plan_only) { *ws_need += need; continue; }
            a.ws = nullptr; a.ws_stride = 0;
            if (splits > 1) {
                if (!ws || ws_used + need > ws_floats) return OG_BAD_ARGS;
                a.ws = ws + ws_used; a.ws_stride = slot; ws_used += need;
            }
        }
        dim3 grid(tiles, splits);
#define OG_WG(KS)                                                                              \
        if (cfg == 0) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 2, 2>), grid, dim3(256), 0, s, a);       \
        else if (cfg == 1) hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 2>), grid, dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((conv_wgrad_kernel<KS, 1, 1>), grid, dim3(256), 0, s, a);
        ProfRec* pr = prof_begin(OG_CAT_WGRAD1,
                                 2.0 * (a.m_end - a.m_begin) * (double)a.ncol * (double)Npix, s);
        prof_meta(pr, 1, 0, a.m_end - a.m_begin, Cin, ksize * ksize, N, OH, OW, stride, splits);
        if (ksize == 1) { OG_WG(1) } else if (ksize == 3) { OG_WG(3) } else { OG_WG(4) }
        prof_end(pr, s);
#undef OG_WG
        int rc = og_launch_status();
        if (rc != OG_OK) return rc;
        if (a.ws) {
            const long slot = a.ws_stride;
            hipLaunchKernelGGL(wgrad_combine_kernel, dim3(og_stream_grid(slot, 256)), dim3(256), 0, s, a.ws, splits, slot,
                               dw, a.ncol, a.m_begin, a.m_end - a.m_begin, 0, slot, accumulate);
            rc = og_launch_status();
            if (rc != OG_OK) return rc;
        }
    }
    return OG_OK;
}

extern "C" {

// Floats of workspace objgan_conv_wgrad needs for these arguments (0: every launch runs as one split).  Host-only.
long objgan_conv_wgrad_ws_floats(int N, int Cin, int H, int W, int upsample, int pad_mode,
                                 int Cout, int OH, int OW, int ksize, int stride, int pad, int math) {
    long need = 0;
    (void)og_wgrad(nullptr, nullptr, nullptr, N, Cin, H, W, upsample, pad_mode, Cout, OH, OW, ksize, stride, pad, math, 0,
                   nullptr, 0, &need, nullptr);
    return need;
}

// 1 if objgan_conv_wgrad takes math 5 (x as its fp16 record, see objgan_h2_records) for this geometry.  Host-only.
int objgan_conv_wgrad_rec_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ksize) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return 0;
    return og_wgrad_rec_geometry(N, Cin, H, W, Cout, OH, OW, ksize) ? 1 : 0;
}

// 1 if objgan_conv_wgrad takes math 3 (x as its bf16 channel-blocked copy) for this geometry.  Host-only.
int objgan_conv_wgrad_bfb_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ksize) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return 0;
    const long OHW = (long)OH * OW;
    const long Cpb = ((long)Cin + 15) / 16 * 16;
    const bool v2 = !og_igemm_v1() && (OW % 8 == 0) && (OHW % 16 == 0)
                    && (double)N * Cin * H * W * 4.0 < 4.0e9 && (double)N * Cout * OHW * 4.0 < 4.0e9;
    return (v2 && OHW % 32 == 0 && OH <= 256 && OW <= 256 && (long)(H - 1) * W < 65535 && ksize <= 4
            && (ksize == 1 || ksize == 3 || ksize == 4) && (double)N * Cpb * H * W * 2.0 < 4.0e9) ? 1 : 0;
}

// The bf16 channel-blocked copy [N][Cp/16][HW][16] (RNE) of an fp32 [N][C][HW] tensor: what the bf16-mode kernels read
// (math 3).  objgan_nhwc_bf16_floats: its size in floats (0: too large for the 32-bit buffer range -- use math 1).
long objgan_nhwc_bf16_floats(int N, int C, long HW) {
    const long Cp = ((long)C + 15) / 16 * 16;
    if (N <= 0 || C <= 0 || HW <= 0 || (double)N * HW * Cp * 2.0 >= 4.0e9 || HW >= (1L << 31)) return 0;
    return ((long)N * HW * Cp / 2 + 3) & ~3L;
}
int objgan_nhwc_bf16(const float* x, float* out, int N, int C, long HW, void* stream) {
    OG_ENTRY();
    if (!x || !out || objgan_nhwc_bf16_floats(N, C, HW) == 0) return OG_BAD_ARGS;
    const int Cp = (C + 15) / 16 * 16;
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(og_cdiv(HW, 64), og_cdiv(Cp, 64), N), dim3(256), 0, (hipStream_t)stream,
                       x, reinterpret_cast<__bf16*>(out), C, (int)HW, Cp);
    return og_launch_status();
}

// dw [Cout][Cin][k][k] = (accumulate ? dw : 0) + sum dy * x.  ws: objgan_conv_wgrad_ws_floats(...) floats of scratch.
int objgan_conv_wgrad(const float* x, const float* dy, float* dw,
                      int N, int Cin, int H, int W, int upsample, int pad_mode,
                      int Cout, int OH, int OW, int ksize, int stride, int pad,
                      int math, int accumulate, const float* xmax, const float* dymax, float* ws, long ws_floats,
                      void* stream) {
    OG_ENTRY();
    return og_wgrad(x, dy, dw, N, Cin, H, W, upsample, pad_mode, Cout, OH, OW, ksize, stride, pad, math, accumulate ? 1 : 0,
                    ws, ws_floats, nullptr, (hipStream_t)stream, xmax, dymax);
}

// ---- profiling control (see the note above the host section) ---------------------------------
int objgan_prof_enable(int on) {
    OG_ENTRY();
    g_prof_on = on ? 1 : 0;
    if (on) {
        g_prof_n = 0;
        // create the whole event pool up front: hipEventCreate inside the measured region would
        // cost the host tens of milliseconds per step
        if (!g_prof) g_prof = (ProfRec*)calloc(OG_PROF_MAX, sizeof(ProfRec));
        while (g_prof && g_prof_made < OG_PROF_MAX) {
            ProfRec* r = &g_prof[g_prof_made];
            if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) break;
            g_prof_made++;
        }
    }
    return OG_OK;
}

// Sums the recorded launches per category (the caller must have synchronised the device).
// ms, flops, count: arrays of OG_PROF_CATS = 192.  Categories = kernel instances (see OG_CAT_* above).
int objgan_prof_collect(double* ms, double* flops, long* count) {
    OG_ENTRY();
    for (int i = 0; i < OG_PROF_CATS; ++i) { ms[i] = 0; flops[i] = 0; count[i] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
        ms[g_prof[i].cat] += t; flops[g_prof[i].cat] += g_prof[i].flops; count[g_prof[i].cat] += 1;
    }
    g_prof_n = 0;
    return OG_OK;
}

// Per-launch records of the last profiling window (device must be idle): ms[i], flops[i], meta[10*i..]
// (see ProfRec::meta), at most max_records; *n_out = number written.  Does not reset the window.
int objgan_prof_dump(float* ms, double* flops, int* meta, int max_records, int* n_out) {
    OG_ENTRY();
    int n = 0;
    for (int i = 0; i < g_prof_n && n < max_records; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
        ms[n] = t; flops[n] = g_prof[i].flops;
        for (int j = 0; j < 10; ++j) meta[10 * n + j] = g_prof[i].meta[j];
        ++n;
    }
    if (n_out) *n_out = n;
    return OG_OK;
}

}  // extern "C"
