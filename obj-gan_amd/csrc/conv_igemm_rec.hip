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

// ---- weight gradient on records ------------------------------------------------------------------------------------
//   dw[m][ci * T + t] = sum over pixels p of dy[m][p] * x[ci][tap t of p]          GEMM: M = Cout, N = (tap, ci), K = pixels
// The register-fragment weight gradient (conv_wgrad3_kernel, math 4) gathers x from the fp32 NCHW tensor: the lanes of a
// wave sit on 32 different (channel, tap) planes, every gather instruction touches 32+ cache lines, and both operands
// are split on the VALU inside the loop (MFMA pipe 34-38 % busy, profiles/r04_roofline_table.md).  Here x arrives as its
// fp16 record (the one the forward convolution of the layer already read): a wave owns ONE tap and 32 channels (two
// 16-channel chunks); per iteration of 32 pixels a lane fetches the 16-byte half records of its pixels -- piece h and
// piece l, 4 loads, every geometry (padding, reflection, stride, upsampling) is just the record address, taken from two
// small LDS tables -- the wave parks them in a wave-private LDS image [piece][half][16 pixels][2 chunks x 16 channels]
// and reads them back TRANSPOSED with ds_read_b64_tr_b16 (gfx950): four consecutive pixels of one channel per lane,
// i.e. the MFMA's B fragment.  The scheme is conv_wgrad_bfb_kernel's (bf16 mode, conv_igemm.hip), with two pieces.
// dy comes from the fp32 tensor: the loader thread scales and splits its four pixels on the way into LDS (row image
// [h 32 px | l 32 px], 144-byte pitch).  Products and their order per 16-pixel step as conv_wgrad3_kernel<.., 4>:
// al.bh, ah.bh, ah.bl with fp32 accumulation; scales undone in the epilogue.
// Requires OH * OW % 32 == 0, OH, OW <= 256, (H - 1) * W < 65535, k <= 4 (og_wgrad asks objgan_conv_wgrad_rec_ok).
// DYP (round 6): dy ALSO arrives pre-split -- a.dy is the fp16 pair of the tensor in its own NCHW layout, plane h
// followed by plane l (h2_pair_kernel below, same scale and the same two conversions as the loader's split) -- and
// the loader threads copy 16-byte pieces (8 pixels of one plane) straight into the same LDS row image: no operand split is
// left in the K loop (the PMC passes of round 5 charged 9.5 VALU instructions per MFMA to the in-loop splits of both
// operands, profiles/r05_pmc_objd_l3_fp16x2.txt).  Same pieces, same products, same order: bit-identical to DYP = false.
template <int TM, int NW, bool DYP = false>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_rec_kernel(const WgradArgs a, const int KS, const int Cp) {
    constexpr int NT = 64 * NW;
    constexpr int BM = 32 * TM;
    constexpr int BK = 32;
    constexpr int ALD = 36;                          // floats per dy row in LDS: 2 x 64 bytes of fp16 + 16 (odd multiple of 16)
    constexpr int ATILE = BM * ALD;
    constexpr int NA4 = BM * 8;                      // 16-byte fp32 pieces (4 pixels) of a row tile per iteration
    constexpr int NA_PER = (NA4 + NT - 1) / NT;
    constexpr int BTILE = 1024;                      // floats per wave: [piece h | l][half][64 lanes x 16 bytes]
    constexpr int TAB = 256;                         // table pitch: OH, OW <= 256
    static_assert((2 * ATILE + NW * BTILE) * 4 + 2 * 4 * TAB * 2 <= 160 * 1024, "LDS");
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
    // XCD placement over both grid dimensions: all tiles of a pixel split on one XCD (see conv_wgrad_bfb_kernel)
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

    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;
    const int Npix = a.N * OHW;
    const int p_begin = split * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    const int nk = (p_end - p_begin + BK - 1) / BK;

    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * (unsigned)Cc * (unsigned)HW * 64u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.N * a.Cout * OHW * 4u), OG_BUF_FLAGS);      // (DYP: two fp16 planes, the same bytes)
    const unsigned dy_plane = (unsigned)a.N * (unsigned)a.Cout * (unsigned)OHW * 2u;  // bytes of one fp16 plane

    og_fp16_saturate();
    const int sx = og_h2_exponent(a.xmax, lane), sd = og_h2_exponent(a.dymax, lane);
    const float h2_dys = og_pow2(sd), h2_inv = og_pow2_sum(-sx, -sd);

    // ---- x records: lane = (pixel j = l >> 2 of a 16-pixel half, chunk (l >> 1) & 1, 16-byte half l & 1)
    const int chunk = cg * 2 + ((lane >> 1) & 1);
    const bool rec_ok = grp_ok && chunk < Cc;
    const unsigned rec_lane = (unsigned)chunk * 2u * (unsigned)HW;     // h plane of this chunk; l plane: + HW
    const unsigned img_recs = (unsigned)Cc * 2u * (unsigned)HW;        // records per image
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    for (int i = tid; i < KS * (a.OH + a.OW); i += NT) {             // tap geometry tables: source row offset per (kh, oh),
        const bool is_row = i < KS * a.OH;                           // source column per (kw, ow); 0xffff = padding
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
    auto load_b = [&](f32x4 (&rb)[4]) {              // rb[piece * 2 + half]
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
            const unsigned rec = (unsigned)n * img_recs + rec_lane + r + c;
            const unsigned off = ok ? rec * 32u + half16 : OG_OOB;
            rb[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
            rb[2 + h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, HW * 32, 0));
        }
        p_ld += BK;
        pow_ += cols32;
        const int c = pow_ >= a.OW ? 1 : 0;
        pow_ -= c ? a.OW : 0;
        poh += rows32 + c;
        while (poh >= a.OH) { poh -= a.OH; pn += 1; }
    };
    auto store_b = [&](const f32x4 (&rb)[4]) {
        float* Bs = ldsB + wid * BTILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(Bs + q * 256 + lane * 4) = rb[q];
    };
    // transposing read: 16-lane group g = l >> 4: chunk g & 1, pixel half-octet g >> 1 (conv_wgrad_bfb_kernel)
    const int b_rd = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;   // bytes

    // ---- dy rows (fp32 in HBM; scaled and split into two fp16 pieces by the loader thread)
    unsigned avoff[NA_PER];
    int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + NT * i;
        const int row = idx >> 3, q = idx & 7;
        const bool on = (NA4 % NT == 0 || idx < NA4) && (m0 + row) < a.m_end;
        if (DYP) {          // piece q < 4: pixels 8 q .. 8 q + 7 of plane h; q >= 4: the same of plane l -> LDS row image [h | l]
            avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + (q & 3) * 8u) * 2u + (q >> 2) * dy_plane : OG_OOB;
            alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * ALD + q * 4 : -1;
        } else {
            avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + q * 4u) * 4u : OG_OOB;
            alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * ALD + q * 2 : -1;
        }
    }
    int n_ld = p_begin / OHW;                        // scalar (image, offset) of the next dy iteration
    int rem_ld = p_begin - n_ld * OHW;
    f32x4 ra[NA_PER];
    auto load_a = [&]() {
        const int so = (n_ld * a.Cout * OHW + rem_ld) * (DYP ? 2 : 4);
        rem_ld += BK;
        if (rem_ld >= OHW) { rem_ld = 0; n_ld += 1; }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, avoff[i], so, 0));
    };
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto store_a = [&](int buf) {
        float* As = ldsA + buf * ATILE;
        if (DYP) {
#pragma unroll
            for (int i = 0; i < NA_PER; ++i)
                if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sv = ra[i][j] * h2_dys;
                h[j] = (_Float16)sv;
                l[j] = (_Float16)og_sub(sv, (float)h[j]);
            }
            if (NA4 % NT == 0 || alds[i] >= 0) {
                *reinterpret_cast<f16x4*>(As + alds[i]) = h;
                *reinterpret_cast<f16x4*>(As + alds[i] + 16) = l;
            }
        }
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
            const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024));
            const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024 + 256));
            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + 2048 + h * 1024));
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + 2048 + h * 1024 + 256));
            // (whole-vector casts: see conv_wgrad_bfb_kernel)
            const f16x8 bh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
            const f16x8 bl = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                al[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * ALD + 16 + h * 8);
                ah[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * ALD + h * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(al[i], bh, acc[i]);
            if (h == 1) {                 // the refill behind the first TM MFMAs of the second half
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bl, acc[i]);
        }
    };

    // prologue: iteration 0 in LDS buffer 0, iteration 1 in registers
    f32x4 rb[4];
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
            if (m < a.m_end) og_wgrad_store(a, m, ocol, acc[i][r] * h2_inv, split);
        }
    }
}

// launch of the record-reading weight gradient (called by og_wgrad in conv_igemm.hip)
// ---- weight gradient on records, TWO column groups per wave (round 6) --------------------------------------------------
// conv_wgrad_rec_kernel above is bound by its LDS fragment reads: per 32-pixel iteration the eight waves of a 192-row tile
// read 8 x 24 KB of dy row fragments (every wave the same rows, for its own 32 columns) -- ~2 060 cycles at 128 B/clk against
// 2 300 cycles of MFMA per SIMD, and with two waves per SIMD the two do not overlap (MFMA pipe 35-40 % busy,
// profiles/r05_roofline_table.md).  Here a wave owns TWO (tap, 32-channel) column groups and a 32 * TM <= 128 row tile:
// one set of row fragments feeds twice the MFMAs (LDS bytes per MAC: 0.047 against 0.073), the x records of both groups go
// through two wave-private LDS images.  Same products per 16-pixel step, same order inside a column group: a column of dw
// is the same sum as in the one-group kernel with the same pixel splits (bit-identical when the plans agree).
template <int TM>
__global__ __launch_bounds__(512) void conv_wgrad_rec2_kernel(const WgradArgs a, const int KS, const int Cp) {
    constexpr int NW = 8, NCG = 2;
    constexpr int NT = 64 * NW;
    constexpr int BM = 32 * TM;
    constexpr int BK = 32;
    constexpr int ALD = 36;
    constexpr int ATILE = BM * ALD;
    constexpr int NA4 = BM * 8;
    constexpr int NA_PER = (NA4 + NT - 1) / NT;
    constexpr int BTILE = 1024;
    constexpr int TAB = 256;
    static_assert((2 * ATILE + NW * NCG * BTILE) * 4 + 2 * 4 * TAB * 2 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) float ldsA[2 * ATILE];
    __shared__ __attribute__((aligned(16))) float ldsB[NW * NCG * BTILE];
    __shared__ unsigned short rtab[4 * TAB], ctab[4 * TAB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5;
    const int lcol = lane & 31;

    const int T = KS * KS;
    const int Cc = Cp >> 4;
    const int CG = (Cc + 1) >> 1;
    const int ngroups = T * CG;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (ngroups + NW * NCG - 1) / (NW * NCG);
    const int nwg = tiles_m * tiles_n;
    const int vid = og_xcd_remap(blockIdx.x + nwg * blockIdx.y, nwg * gridDim.y);
    const int split = vid / nwg;
    const int wg = vid - split * nwg;
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;

    const int OHW = a.OH * a.OW;
    const int HW = a.H * a.W;
    const int Npix = a.N * OHW;
    const int p_begin = split * a.pix_per_split;
    const int p_end = min(Npix, p_begin + a.pix_per_split);
    const int nk = (p_end - p_begin + BK - 1) / BK;

    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (int)((unsigned)a.N * (unsigned)Cc * (unsigned)HW * 64u), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t dyres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (int)((unsigned)a.N * a.Cout * OHW * 4u), OG_BUF_FLAGS);

    og_fp16_saturate();
    const int sx = og_h2_exponent(a.xmax, lane), sd = og_h2_exponent(a.dymax, lane);
    const float h2_dys = og_pow2(sd), h2_inv = og_pow2_sum(-sx, -sd);

    // ---- the wave's two column groups
    bool grp_ok[NCG], rec_ok[NCG];
    int tg[NCG], cgg[NCG];
    unsigned rec_lane[NCG];
    const unsigned short* rt[NCG];
    const unsigned short* ct[NCG];
#pragma unroll
    for (int j = 0; j < NCG; ++j) {
        const int group = (tile_n * NW + wid) * NCG + j;
        grp_ok[j] = group < ngroups;
        tg[j] = grp_ok[j] ? group / CG : 0;
        cgg[j] = grp_ok[j] ? group - tg[j] * CG : 0;
        const int kh = tg[j] / KS;
        const int chunk = cgg[j] * 2 + ((lane >> 1) & 1);
        rec_ok[j] = grp_ok[j] && chunk < Cc;
        rec_lane[j] = (unsigned)chunk * 2u * (unsigned)HW;
        rt[j] = rtab + kh * TAB;
        ct[j] = ctab + (tg[j] - kh * KS) * TAB;
    }
    const unsigned img_recs = (unsigned)Cc * 2u * (unsigned)HW;
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    for (int i = tid; i < KS * (a.OH + a.OW); i += NT) {             // tap geometry tables (see conv_wgrad_rec_kernel)
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
    const int rows16 = 16 / a.OW, cols16 = 16 - rows16 * a.OW;
    const int rows32 = 32 / a.OW, cols32 = 32 - rows32 * a.OW;
    int pn, poh, pow_;
    {
        const int p = p_begin + (lane >> 2);
        pn = p / OHW;
        const int r = p - pn * OHW;
        poh = r / a.OW;
        pow_ = r - poh * a.OW;
    }
    int p_ld = p_begin + (lane >> 2);
    const unsigned half16 = (unsigned)((lane & 1) * 16);
    auto load_b = [&](f32x4 (&rb)[NCG][4]) {         // rb[group][piece * 2 + half]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int n = pn, oh = poh, ow = pow_;
            if (h == 1) {
                ow += cols16;
                const int c = ow >= a.OW ? 1 : 0;
                ow -= c ? a.OW : 0;
                oh += rows16 + c;
                while (oh >= a.OH) { oh -= a.OH; n += 1; }
            }
            const bool in = p_ld + 16 * h < p_end;
#pragma unroll
            for (int j = 0; j < NCG; ++j) {
                const unsigned r = rt[j][oh], c = ct[j][ow];
                const bool ok = rec_ok[j] && in && r != 0xffffu && c != 0xffffu;
                const unsigned rec = (unsigned)n * img_recs + rec_lane[j] + r + c;
                const unsigned off = ok ? rec * 32u + half16 : OG_OOB;
                rb[j][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, 0, 0));
                rb[j][2 + h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, off, HW * 32, 0));
            }
        }
        p_ld += BK;
        pow_ += cols32;
        const int c = pow_ >= a.OW ? 1 : 0;
        pow_ -= c ? a.OW : 0;
        poh += rows32 + c;
        while (poh >= a.OH) { poh -= a.OH; pn += 1; }
    };
    auto store_b = [&](const f32x4 (&rb)[NCG][4]) {
#pragma unroll
        for (int j = 0; j < NCG; ++j) {
            float* Bs = ldsB + (wid * NCG + j) * BTILE;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(Bs + q * 256 + lane * 4) = rb[j][q];
        }
    };
    const int b_rd = ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;   // bytes

    // ---- dy rows (fp32 in HBM; scaled and split into two fp16 pieces by the loader thread)
    unsigned avoff[NA_PER];
    int alds[NA_PER];
#pragma unroll
    for (int i = 0; i < NA_PER; ++i) {
        const int idx = tid + NT * i;
        const int row = idx >> 3, q = idx & 7;
        const bool on = (NA4 % NT == 0 || idx < NA4) && (m0 + row) < a.m_end;
        avoff[i] = on ? ((unsigned)(m0 + row) * (unsigned)OHW + q * 4u) * 4u : OG_OOB;
        alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * ALD + q * 2 : -1;
    }
    int n_ld = p_begin / OHW;
    int rem_ld = p_begin - n_ld * OHW;
    f32x4 ra[NA_PER];
    auto load_a = [&]() {
        const int so = (n_ld * a.Cout * OHW + rem_ld) * 4;
        rem_ld += BK;
        if (rem_ld >= OHW) { rem_ld = 0; n_ld += 1; }
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyres, avoff[i], so, 0));
    };
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    auto store_a = [&](int buf) {
        float* As = ldsA + buf * ATILE;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sv = ra[i][j] * h2_dys;
                h[j] = (_Float16)sv;
                l[j] = (_Float16)og_sub(sv, (float)h[j]);
            }
            if (NA4 % NT == 0 || alds[i] >= 0) {
                *reinterpret_cast<f16x4*>(As + alds[i]) = h;
                *reinterpret_cast<f16x4*>(As + alds[i] + 16) = l;
            }
        }
    };

    f32x16 acc[NCG][TM];
#pragma unroll
    for (int j = 0; j < NCG; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
    auto mma = [&](int buf, auto&& mid) {
        const float* As = ldsA + buf * ATILE + lcol * ALD + lrow * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f16x8 ah[TM], al[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                al[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * ALD + 16 + h * 8);
                ah[i] = *reinterpret_cast<const f16x8*>(As + i * 32 * ALD + h * 8);
            }
#pragma unroll
            for (int j = 0; j < NCG; ++j) {
                const char* Bs = reinterpret_cast<const char*>(ldsB + (wid * NCG + j) * BTILE) + b_rd;
                const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024));
                const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + h * 1024 + 256));
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + 2048 + h * 1024));
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + 2048 + h * 1024 + 256));
                const f16x8 bh = __builtin_bit_cast(f16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                const f16x8 bl = __builtin_bit_cast(f16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(al[i], bh, acc[j][i]);
                if (h == 1 && j == NCG - 1) { // the refill of the (single-buffered, wave-private) x images: behind the LAST
                                              // transposing reads of this iteration -- the second half of the last group
                    __builtin_amdgcn_sched_barrier(0);
                    mid();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bh, acc[j][i]);
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(ah[i], bl, acc[j][i]);
            }
        }
    };

    f32x4 rb[NCG][4];
    __syncthreads();                                  // tables
    load_a(); load_b(rb);
    store_a(0); store_b(rb);
    load_a(); load_b(rb);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        mma(0, [&]() { store_a(1); store_b(rb); load_a(); load_b(rb); });
        __syncthreads();
        mma(1, [&]() { store_a(0); store_b(rb); load_a(); load_b(rb); });
        __syncthreads();
    }
    if (kt < nk) mma(0, [] {});

#pragma unroll
    for (int j = 0; j < NCG; ++j) {
        const int ci = cgg[j] * 32 + lcol;
        if (!grp_ok[j] || ci >= a.Cin) continue;
        const int ocol = ci * T + tg[j];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (m < a.m_end) og_wgrad_store(a, m, ocol, acc[j][i][r] * h2_inv, split);
            }
        }
    }
}

int og_launch_wgrad_rec2(const WgradArgs& a, int tm, dim3 grid, int ksize, int Cp, hipStream_t s) {
    switch (tm) {
        case 1: hipLaunchKernelGGL((conv_wgrad_rec2_kernel<1>), grid, dim3(512), 0, s, a, ksize, Cp); break;
        case 2: hipLaunchKernelGGL((conv_wgrad_rec2_kernel<2>), grid, dim3(512), 0, s, a, ksize, Cp); break;
        case 3: hipLaunchKernelGGL((conv_wgrad_rec2_kernel<3>), grid, dim3(512), 0, s, a, ksize, Cp); break;
        case 4: hipLaunchKernelGGL((conv_wgrad_rec2_kernel<4>), grid, dim3(512), 0, s, a, ksize, Cp); break;
        default: return OG_BAD_ARGS;
    }
    return og_launch_status();
}

// fp32 -> its fp16 pair under the tensor's scale, same layout, plane h followed by plane l (n4 = elements / 4): the dy
// operand of conv_wgrad_rec_kernel<.., true>
__global__ __launch_bounds__(256) void h2_pair_kernel(const float* __restrict__ x, const float* __restrict__ xmax,
                                                      _Float16* __restrict__ out, long n4) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    og_fp16_saturate();
    const float xs = og_pow2(og_h2_exponent(xmax, threadIdx.x & 63));
    _Float16* ol = out + 4 * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
        f16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = v[j] * xs;
            h[j] = (_Float16)sv;
            l[j] = (_Float16)og_sub(sv, (float)h[j]);
        }
        *reinterpret_cast<f16x4*>(out + 4 * i) = h;
        *reinterpret_cast<f16x4*>(ol + 4 * i) = l;
    }
}

void og_launch_h2_pair(const float* x, const float* xmax, float* out, long n, hipStream_t s) {
    const long n4 = n / 4;
    hipLaunchKernelGGL(h2_pair_kernel, dim3(og_stream_grid(n4, 256)), dim3(256), 0, s, x, xmax,
                       reinterpret_cast<_Float16*>(out), n4);
}

int og_launch_wgrad_rec(const WgradArgs& a, int tm, int nw, dim3 grid, int ksize, int Cp, int dyp, hipStream_t s) {
#define OG_WGR(TMv)                                                                                                   \
        if (dyp && nw == 8) hipLaunchKernelGGL((conv_wgrad_rec_kernel<(TMv <= 6 ? TMv : 6), 8, true>), grid, dim3(512), 0, s, a, ksize, Cp); \
        else if (dyp) hipLaunchKernelGGL((conv_wgrad_rec_kernel<TMv, 4, true>), grid, dim3(256), 0, s, a, ksize, Cp); \
        else if (nw == 8) hipLaunchKernelGGL((conv_wgrad_rec_kernel<(TMv <= 6 ? TMv : 6), 8>), grid, dim3(512), 0, s, a, ksize, Cp); \
        else hipLaunchKernelGGL((conv_wgrad_rec_kernel<TMv, 4>), grid, dim3(256), 0, s, a, ksize, Cp);
    if (nw == 8 && tm > 6) return OG_BAD_ARGS;
    switch (tm) {
        case 1: OG_WGR(1) break;
        case 2: OG_WGR(2) break;
        case 3: OG_WGR(3) break;
        case 4: OG_WGR(4) break;
        case 5: OG_WGR(5) break;
        case 6: OG_WGR(6) break;
        default: OG_WGR(7) break;
    }
#undef OG_WGR
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
    // (round 6 tried 64 x 128 tiles fetched as 16-byte pieces: 7.8 ms per step against 7.0 for this form in the same profile
    //  -- the larger LDS tile halves the workgroups per CU; rejected, profiles/r06_ab_variants.txt)
    hipLaunchKernelGGL(h2_records_kernel, dim3(og_cdiv(HW, 64), og_cdiv(Cp, 64), N), dim3(256), 0, (hipStream_t)stream,
                       x, xmax, reinterpret_cast<_Float16*>(rec), C, (int)HW, Cp);
    return og_launch_status();
}

}  // extern "C"
