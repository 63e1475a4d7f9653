// The register-fragment implicit-GEMM kernel (conv_igemm3_kernel) with its argument block and arithmetic helpers:
// shared by conv_igemm.hip (fp32 MFMA, bf16, bf16x3, fp16x2 instances) and conv_igemm_rec.hip (the fp16x2 instances
// that read pre-split fp16 records).  See conv_igemm.hip for the GEMM view of a convolution.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

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
    const float* xmax;      // math 4: the OG_AMAX_SLOTS partial maxima of |x| (objgan_absmax_partials or a producer)
    const float* wmax;      // math 4: the OG_AMAX_SLOTS partial maxima of |w| behind the bank (absmax_w_*)
    float* ymax;            // != nullptr (unsplit conv_igemm3_kernel launches): OG_AMAX_SLOTS zero-filled slots that receive
                            // the partial maxima of |y| (the scale input of an fp16x2 convolution that reads y next)
    int math;               // 0: fp32 MFMA, 1: bf16 inputs (RNE) on the bf16 MFMA, fp32 accumulation; 2: bf16x3; 4: fp16x2
    int nhwc;               // bf16 mode: 1 = the pixel operand comes from a bf16 [N][Cp/16][H][W][16] copy of x that
                            // run_igemm2 makes in the caller's workspace (conv_igemm3_kernel<.., 3, ..>)
    int Krow;               // row pitch of the [M][Krow] bank in elements (= Kpad; bf16 bank: Kpad rounded up to 32)
    int T, Cp;              // taps, channels rounded up to 16
    int kgroup;             // conv_igemm3_kernel: K walk order -- groups of `kgroup` 16-channel chunks, all taps of a group
                            // before the next group (og_kstep; = Cp / 16: plain tap-major).  The bank is packed in the
                            // same order (PackArgs::kgroup).
    int m_begin, m_end;  // output-channel rows covered by this launch
    int PH, PW;          // GEMM pixel grid per image
    int OHf, OWf;        // physical output dims
    int osh, osw, ooh, oow;
    int stride;
    int pad_mode;        // 0 = zeros outside [0,LH)x[0,LW), 1 = reflect
    int upsample;        // 1 = source index = logical index >> 1
    int act;
    int ksplit_steps;    // > 0: split-K -- blockIdx.y owns this many BK steps; epilogue: its partial tile goes to
                         // ws[blockIdx.y][...] (conv_igemm3_kernel; summed in split order by splitk_combine_kernel,
                         // which also applies bias / activation: bit-reproducible, no zero-fill, no atomics)
    float* ws;           // split-K partials [splits][ws_stride]: y-shaped, then (ring mode) ring-shaped
    long ws_stride;
    int nphase;          // > 1 (conv_igemm3_kernel only): workgroup id % nphase = output phase p with its own tap table
                         // tap[p*8 ..], packed bank wt + p*M*Kpad and output offset (ooh, oow) = (p>>1, p&1)
    float* ring;         // != nullptr (conv_igemm3_kernel only): data gradient of a ReflectionPad2d(1) convolution.
                         // The GEMM runs over the PADDED grid (PH = OHf + 2, PW = OWf + 2); interior pixels are
                         // stored straight into the unpadded y, the one-pixel ring into ring[n*M + m][2*PW + 2*PH]
                         // (top row, bottom row, left column, right column) for objgan_reflect_ring_fold to add
                         // back -- instead of writing the padded tensor and folding it in a second full pass.
    int tap[OG_MAX_TAPS];   // (dw << 16) | (dh & 0xffff): one scalar load per (uniform) tap
#ifdef OG_DEV
    int ablate;             // development builds: main-loop ablation of the bf16x3 kernel (timing only, wrong results):
                            // 1 no m/l products, 2 no pixel gathers after the prologue, 4 no row-tile loads after the
                            // prologue, 8 no barriers, 16 no LDS stores, 32 no split VALU, 64 no LDS fragment reads
#endif
};
#ifdef OG_DEV
#define OG_ABL(bit) (a.ablate & (bit))
#else
#define OG_ABL(bit) false
#endif

__device__ __forceinline__ float og_act(float v, int act) {
    if (act == OG_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == OG_ACT_TANH) return tanhf(v);
    if (act == OG_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    if (act == OG_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// ---- fp32 on the bf16 matrix pipe ("bf16x3": the name oneMKL uses for the same scheme) -------------------
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of v_mfma_f32_32x32x2_f32.  An fp32 value is EXACTLY the
// sum of three bf16 values (24 significand bits = 8 + 8 + 8, round-to-nearest at every cut, the residuals are
// exact in fp32): x = h + m + l.  A product of two such values is the sum of nine bf16 x bf16 products, each
// exact in the fp32 accumulator; the six largest -- hh, hm, mh, mm, hl, lh -- leave out terms below
// 2^-24 |x||y| per product (measured on K = 1746 rows: 5.9e-9 relative against 1.9e-7 of fp32 accumulation
// rounding itself; tests/test_kernels_gpu.py pins split-mode error <= 1.1 x native fp32 error against an fp64
// evaluation).  Six MFMAs per 16-deep K step instead of eight f32 ones at 1/16 of the rate: 2.67x the fp32
// matrix peak (416.7 TFLOP/s of fp32-equivalent work).
// (plain v_sub_f32 through asm: hipcc SLP-packs adjacent fp32 subtractions into v_pk_add_f32, which costs the
// MFMA stream beside it ~13 cycles each -- MI355X_MICROARCH.md, "price of one filler beside MFMAs")
__device__ __forceinline__ float og_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void og_split8_h(const float* v, bf16x8& h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[j];
}
__device__ __forceinline__ void og_split8_ml(const float* v, const bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r1 = og_sub(v[j], (float)h[j]);
        const __bf16 mj = (__bf16)r1;
        m[j] = mj; l[j] = (__bf16)og_sub(r1, (float)mj);
    }
}
__device__ __forceinline__ void og_split8(const float* v, bf16x8& h, bf16x8& m, bf16x8& l) {
    og_split8_h(v, h);
    og_split8_ml(v, h, m, l);
}
// after a sched_barrier: pin `pairs` x (one MFMA, then `valu` VALU instructions) in issue order
template <int PAIRS, int VALU>
__device__ __forceinline__ void og_interleave() {
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
    }
}
__device__ __forceinline__ void og_split4(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 hj = (__bf16)v[j];
        const float r1 = og_sub(v[j], (float)hj);
        const __bf16 mj = (__bf16)r1;
        h[j] = hj; m[j] = mj; l[j] = (__bf16)og_sub(r1, (float)mj);
    }
}
// the six products of one row group and K step, smallest terms first
#define OG_MFMA_BF(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

// ---- fp32 on the fp16 matrix pipe ("fp16x2", math 4) --------------------------------------------------------------
// v_mfma_f32_32x32x16_f16 runs at the rate of the bf16 MFMA, and fp16 carries 11 significand bits: TWO pieces
// x * 2^s = h + l (round-to-nearest at each cut) leave |residual| <= 2^-23 |x| -- one fp32 ulp: h keeps 11 significand
// bits, l the next 11 plus its sign (tests/test_kernels_gpu.py::test_h2_records_layout_and_split) -- as long as l stays
// a normal fp16, which a power-of-two scale 2^s per operand tensor arranges (max |x| * 2^s in [2^14, 2^15): activations
// and gradients from the per-workgroup maxima of objgan_absmax_partials, filter banks from the partial maxima the pack
// path leaves behind the bank; an
// element below 2^-10 of its tensor's maximum keeps an ABSOLUTE error of 2^-39 of that maximum instead).  Three
// products hh, hl, lh (ll is below 2^-24 of a product) are three MFMAs per 16-deep K step and row group instead of
// the six of bf16x3; the scales are undone exactly in the epilogue.  Measured against fp64 the error is BELOW
// bf16x3's (the MFMA adds 16 products before it rounds, and there are half as many accumulations): loop laboratory
// K = 3072: 6.1e-7 vs 8.7e-7, at 295-302 vs 190-193 TFLOP/s (profiles/r04_loop_lab.txt).
#define OG_MFMA_H(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
// scale exponent s of a tensor from its OG_AMAX_SLOTS partial maxima: max * 2^s in [2^14, 2^15)  (0 for an all-zero tensor)
__device__ __forceinline__ int og_h2_exponent(const float* __restrict__ pm, int lane) {
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < OG_AMAX_SLOTS / 64; ++k) m = fmaxf(m, pm[lane + 64 * k]);
    m = og_wave_max(m);
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);
    int sx = (e > 0 && e < 255) ? 127 + 14 - e : 0;
    sx = sx > 100 ? 100 : (sx < -100 ? -100 : sx);
    return __builtin_amdgcn_readfirstlane(sx);
}
__device__ __forceinline__ float og_pow2(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }
// 2^(a + b) for two exponents that are each within +-100 but whose sum may leave the normal range (two vanishing
// tensors): the exponent field must not wrap into the sign bit -- the product of two in-range powers underflows to a
// subnormal / zero or overflows to inf like any other float product
__device__ __forceinline__ float og_pow2_sum(int a, int b) { return og_pow2(a) * og_pow2(b); }
__device__ __forceinline__ void og_h2_split_h(const float* v, float xs, float* sc, f16x8& h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = v[j] * xs; h[j] = (_Float16)sc[j]; }
}
__device__ __forceinline__ void og_h2_split_l(const float* sc, const f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = (_Float16)og_sub(sc[j], (float)h[j]);
}

// bijective XCD-aware remap of a linear workgroup id (dispatcher places id b on XCD b % 8)
__device__ __forceinline__ int og_xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, j = id >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + j;
}

template <int V> struct OgInt { static constexpr int value = V; };
#define OG_BUF_FLAGS 0x00020000
#define OG_OOB 0x7ffffff0u

// =============================================================================================
// v3: the pixel / column operand never touches LDS.  In the (32*TM) x 128 tiling a wave owns its 32
// pixel (or weight-gradient column) positions exclusively, and the eight consecutive k values a
// thread gathers for one position are exactly the lane's MFMA operand for a K step (lane = (position
// l & 31, k half l >> 5)): the gather registers ARE the fragment.  Only the row operand (filter bank /
// dy), which all four waves share, is staged through LDS -- and for TM = 1 (thin outputs) it is read
// straight from L1/L2 as two 16-byte loads per lane, so those launches run without LDS and without
// barriers at all: waves are independent and latency is hidden by occupancy.
// (A chunk-major K order -- all taps of a 16-channel chunk back to back, hoping for L1 hits between the
// shifted windows -- was measured in round 2 and lost on every shape: profiles/r02_ab_convbench_variants.txt.)
template <int TM, bool ADIRECT = false, int MATH = 0, int NW = 4, int NG = 1>
__global__ __launch_bounds__(64 * NW) void conv_igemm3_kernel(const IgemmArgs a) {
    // NW: waves per workgroup = 32-pixel column groups sharing one row tile (4: 128 pixels; 8: 256 pixels -- the
    // bf16x3 kernels run against the L2 -> L1 fill rate, and the row tile is 2/3 of a workgroup's fills)
    constexpr int NT = 64 * NW;
    // MATH 0: fp32 MFMA; 1 (BF): bf16-rounded operands; 2 (SP): fp32 operands split three ways on the bf16 MFMA
    // (og_split8).  SP: one iteration = one 16-deep K step like fp32; the bank holds the PRE-SPLIT filter rows,
    // [M][Kpad/16][h,m,l][16] bf16 = 96 bytes per row and step, LDS row pitch 112 bytes (an odd multiple of 16:
    // the 16 lanes of a ds_read_b128 group fall on 16 different 16-byte slots); the pixel fragment is split in
    // registers (~44 VALU instructions per step next to 6 * TM MFMAs).
    // 3 (NH): as 1, with the pixel operand read from a bf16 channel-blocked copy of the source, [N][Cp/16][H][W][16]
    // (nchw_to_nhwc_bf16_kernel): the eight consecutive k of a lane are eight consecutive channels of its pixel -- one
    // 16-byte load, no conversion -- instead of eight channel-strided dword gathers and eight conversions; the 32
    // pixels of a wave that are neighbours in a row read 1 KiB of contiguous memory per instruction (a plain
    // channels-last [N][H][W][Cp] copy made every such load touch 32 cache lines: 517 -> 603 TFLOP/s, LAB.md §4).
    // 4 (H2): fp32 operands as two fp16 pieces on the fp16 MFMA, three products (see OG_MFMA_H): the SP pipeline with a
    // 64-byte bank record [h16 | l16] per row and step and the scales of the two operands undone in the epilogue.
    // 5 (HR): the arithmetic of 4 with the pixel operand read from PRE-SPLIT fp16 records (objgan_h2_records, or a
    // producer that wrote them): [N][Cp/16][h | l][H*W][16] fp16, x * 2^s = h + l with s from the same partial maxima
    // a.xmax the epilogue undoes it with -- the eight k of a lane are two 16-byte loads (h, l), no dword gathers, no
    // split VALU in the loop; the 32 pixels of a wave read 1 KiB of contiguous memory per instruction.  Same products
    // in the same order as 4: bit-identical results.
    // NG (HR only): 32-pixel groups per wave -- every row fragment read from LDS feeds NG MFMAs per product
    // (tile = 32 TM rows x 32 NW NG pixels; profiles/r04_loop_lab.txt (d): +42-53 % on the 96- / 32-row classes).
    constexpr bool NH = MATH == 3, BF = MATH == 1 || NH, SP = MATH == 2, HR = MATH == 5, H2 = MATH == 4 || HR;
    static_assert(NG == 1 || HR, "pixel groups: record form only");
    static_assert(!(HR && ADIRECT), "record form: row operand through LDS");
    // P3: three LDS row tiles / three pixel-fragment register sets, loads two steps ahead (see the main loop).  SP: a
    // step is 6 TM MFMAs; NH: 2 TM MFMAs -- 0.2 us at TM = 6, far below a loaded L2 round trip, and the fragment of a
    // step is only 8 registers.
    constexpr bool P3 = SP || NH || H2;
    constexpr int BM = 32 * TM;
    constexpr int BN = 32 * NW * NG;
    constexpr int BK = 16;
    constexpr int LD = SP ? 28 : BK + 4;              // floats per LDS row
    constexpr int PIECES = SP ? 6 : 4;                // 16-byte pieces of a bank row per iteration
    constexpr int ABYTES = PIECES * 16;
    constexpr int NA4 = BM * PIECES;
    constexpr int NA_PER = (NA4 + NT - 1) / NT;
    constexpr int TILE = BM * LD;
    constexpr int NAD = SP ? 3 : 2;                   // direct row fragments per lane (TM = 1 LDS-free form)
    constexpr bool ALDS = !(ADIRECT && TM == 1);      // row operand through LDS (shared by 4 waves)
    // BF: bf16 inputs (round-to-nearest-even of the fp32 operands) on v_mfma_f32_32x32x16_bf16, fp32
    // accumulation.  One loop iteration then covers 32 k (two 16-channel gathers, two MFMAs per row
    // group); the bank is bf16 [M][Krow], so a row piece is again 64 bytes per iteration and the
    // LDS image / fragment reads keep their 80-byte pitch.
    constexpr int ESZ = (BF || SP || H2) ? 2 : 4;
    constexpr int NBC = NH ? 4 : 8;                   // registers per 16-channel chunk of the pixel operand
    constexpr int NB = BF ? 2 * NBC : 8;              // pixel-operand registers per lane and iteration

    __shared__ __attribute__((aligned(16))) float lds[ALDS ? (P3 ? 3 : 2) * TILE : 4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 5;
    const int lcol = lane & 31;

    const int Npix = a.N * a.PH * a.PW;
    const int tiles_m = (a.m_end - a.m_begin + BM - 1) / BM;
    const int tiles_n = (Npix + BN - 1) / BN;
    // Phase-fastest workgroup order: the nphase output phases of a stride-2 data gradient / up-convolution read the SAME
    // source pixels, so the phases of a tile sit next to each other in the XCD-remapped id (same L2, same time) -- with
    // the phase on blockIdx.z they ran a whole grid apart and every phase re-fetched the source from HBM / MALL
    // (r03: 562 MB fetched per launch of the 4-phase 192 -> 96 up-convolution for 136 MB algorithmic).
    const int nph = a.nphase > 1 ? a.nphase : 1;
    const int nwg = tiles_m * tiles_n;
    const int wgp = og_xcd_remap(blockIdx.x, nwg * nph);
    const int phase = wgp % nph;
    const int wg = wgp / nph;
    const int tile_m = wg % tiles_m;
    const int tile_n = wg / tiles_m;
    const int m0 = a.m_begin + tile_m * BM;
    const int n0 = tile_n * BN;

    const int tapbase = phase * 8;
    const int HW = a.H * a.W;
    __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, NH ? (int)((unsigned)a.N * HW * a.Cp * 2u)
                          : (HR ? (int)((unsigned)a.N * HW * a.Cp * 4u) : (int)((unsigned)a.N * a.C * HW * 4u)), OG_BUF_FLAGS);
    __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.wt + (size_t)phase * a.M * a.Krow * ESZ), 0,
        (int)((unsigned)a.M * a.Krow * (unsigned)ESZ), OG_BUF_FLAGS);

    // ---- pixel operand: this lane's pixel and k half
    // (group g of wave `wid`: pixels n0 + (wid * NG + g) * 32 ..)
    int pix[NG];
    bool pix_ok[NG];
    int ihb[NG], iwb[NG];
    unsigned img_off[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        pix[g] = n0 + (wid * NG + g) * 32 + lcol;
        pix_ok[g] = pix[g] < Npix;
        const int ppi = a.PH * a.PW;
        const int pp = pix_ok[g] ? pix[g] : 0;
        const int n = pp / ppi;
        const int rem = pp - n * ppi;
        const int pa = rem / a.PW;
        const int pb = rem - pa * a.PW;
        ihb[g] = pa * a.stride;
        iwb[g] = pb * a.stride;
        img_off[g] = NH ? (unsigned)n * (unsigned)(a.Cp / 16) * (unsigned)HW
                        : (HR ? (unsigned)n * (unsigned)(a.Cp / 8) * (unsigned)HW
                              : (unsigned)n * (unsigned)a.C * (unsigned)HW + (unsigned)(lrow * 8) * (unsigned)HW);
    }
    const int us = a.upsample ? 1 : 0;
    const bool refl = a.pad_mode == 1;
    unsigned bvoff[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) bvoff[g] = OG_OOB;
    auto tap_geometry = [&](int t) {
        const int tp = a.tap[tapbase + t];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ih = ihb[g] + ((tp << 16) >> 16);
            const int iw = iwb[g] + (tp >> 16);
            int ihr = ih < 0 ? -ih : ih;
            int iwr = iw < 0 ? -iw : iw;
            ihr = ihr >= a.LH ? 2 * (a.LH - 1) - ihr : ihr;
            iwr = iwr >= a.LW ? 2 * (a.LW - 1) - iwr : iwr;
            const bool inb = ((unsigned)ih < (unsigned)a.LH) && ((unsigned)iw < (unsigned)a.LW);
            const bool ok = pix_ok[g] && (refl || inb);
            const int ihs = (refl ? ihr : ih) >> us;
            const int iws = (refl ? iwr : iw) >> us;
            if (NH || HR) bvoff[g] = ok ? (img_off[g] + (unsigned)(ihs * a.W + iws)) * 32u + (unsigned)(lrow * 16) : OG_OOB;
            else bvoff[g] = ok ? (img_off[g] + (unsigned)(ihs * a.W + iws)) * 4u : OG_OOB;
        }
    };
    int t_ld, cb_ld;
    int gb_ld, ge_ld;                                   // channel range of the current K group (og_kstep)
    const int spt = a.Cp / BK;
    // channels past C (padding of the last 16-channel chunk) read finite neighbouring data or the
    // range-check zero; their filter entries are zero  (NH: the copy holds zeros there)
    auto load_b8 = [&](float* rb) {
        if (HR) {                                       // the h and the l plane of this 16-channel chunk, every pixel group
            const int so = (cb_ld >> 4) * HW * 64;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 vh = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[g], so, 0));
                const f32x4 vl = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[g], so + HW * 32, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) { rb[g * 8 + i] = vh[i]; rb[g * 8 + 4 + i] = vl[i]; }
            }
        } else if (NH) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xres, bvoff[0], (cb_ld >> 4) * HW * 32, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xres, bvoff[0], (cb_ld + i) * HW * 4, 0));
        }
        cb_ld += BK;
        if (cb_ld >= ge_ld) {                           // end of this tap's run of the group
            t_ld += 1;
            if (t_ld >= a.T) {                          // all taps done: next group of channels (past the last one:
                t_ld = 0;                               // channel offsets beyond the tensor, the range check returns 0)
                gb_ld = ge_ld;
                ge_ld = min(a.Cp, ge_ld + a.kgroup * BK);
            }
            cb_ld = gb_ld;
            if (gb_ld < a.Cp) tap_geometry(t_ld);
        }
    };
    auto load_b = [&](float (&rb)[NG * NB]) {
        load_b8(&rb[0]);
        if (BF) load_b8(&rb[NBC]);         // second 16-channel chunk (past the last one: zero filter entries)
    };

    // ---- row operand (filter bank [M][Kpad])
    unsigned avoff[NA_PER];
    int alds[NA_PER];
    if (ALDS) {
#pragma unroll
        for (int i = 0; i < NA_PER; ++i) {
            const int idx = tid + NT * i;
            const int row = idx / PIECES, q = idx - row * PIECES;
            const bool on = (NA4 % NT == 0 || idx < NA4) && (m0 + row) < a.m_end;
            avoff[i] = on ? (unsigned)(m0 + row) * (unsigned)a.Krow * (unsigned)ESZ + q * 16u : OG_OOB;
            alds[i] = (NA4 % NT == 0 || idx < NA4) ? row * LD + q * 4 : -1;
        }
    }
    const unsigned adir = (m0 + lcol) < a.m_end
        ? (unsigned)(m0 + lcol) * (unsigned)a.Krow * (unsigned)ESZ + lrow * ((BF || SP || H2) ? 16u : 32u) : OG_OOB;
    f32x4 ra[NA_PER];
    auto load_a = [&](int kt) {
        // (SP: hipcc keeps the strength-reduced offset of the three-step loop in a VGPR -- SGPR pressure -- and would
        // wrap every load in a waterfall loop; one readfirstlane instead)
        const int so = (SP || H2) ? __builtin_amdgcn_readfirstlane(kt * ABYTES) : kt * ABYTES;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, avoff[i], so, 0));
    };
    auto store_a = [&](int buf) {
        float* As = lds + buf * TILE;
#pragma unroll
        for (int i = 0; i < NA_PER; ++i)
            if (NA4 % NT == 0 || alds[i] >= 0) *reinterpret_cast<f32x4*>(As + alds[i]) = ra[i];
    };

    const int nk_all = BF ? a.Krow / 32 : a.Kpad / BK;
    const int kt0 = a.ksplit_steps > 0 ? blockIdx.y * a.ksplit_steps : 0;
    const int nk = a.ksplit_steps > 0 ? min(nk_all, kt0 + a.ksplit_steps) : nk_all;
    {
        const int sub0 = BF ? 2 * kt0 : kt0;            // first 16-channel chunk of this block
        const int G = a.kgroup, TG = a.T * G, nfull = spt / G;
        int g = sub0 / TG, r = sub0 - g * TG, Gg = G;
        if (g >= nfull) {                               // the (shorter) last group, or past the end
            g = nfull;
            r = sub0 - nfull * TG;
            Gg = spt - nfull * G;
        }
        t_ld = Gg > 0 ? r / Gg : a.T;
        gb_ld = g * G * BK;
        ge_ld = min(a.Cp, gb_ld + G * BK);
        cb_ld = gb_ld + (Gg > 0 ? r - t_ld * Gg : 0) * BK;
    }
    tap_geometry(min(t_ld, a.T - 1));

    float h2_xs = 1.f, h2_inv = 1.f;                  // H2: scale of the pixel operand, inverse of both scales
    if (H2) {
        if (!HR) og_fp16_saturate();                  // (the in-loop split converts to fp16: clamp, never inf)
        const int sx = og_h2_exponent(a.xmax, lane);
        h2_xs = og_pow2(sx);
        h2_inv = og_pow2_sum(-sx, -og_h2_exponent(a.wmax, lane));
    }
    f32x16 acc[TM * NG];                               // row group i, pixel group g: acc[i * NG + g]
#pragma unroll
    for (int i = 0; i < TM * NG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int a_rd = lcol * LD + lrow * 8;
    float rb0[NG * NB], rb1[NG * NB];
    f32x4 ad0[NAD], ad1[NAD];                          // TM == 1: direct row fragments (ping-pong)
    auto load_adir = [&](f32x4 (&ad)[NAD], int kt) {
        const int so = (SP || H2) ? __builtin_amdgcn_readfirstlane(kt * ABYTES) : kt * ABYTES;
        ad[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, adir, so, 0));
        ad[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, adir + ((BF || SP || H2) ? 32u : 16u), so, 0));
        if (SP) ad[NAD - 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wres, adir + 64u, so, 0));
    };
    // `mid`: the refill of the software pipeline (LDS store of the next row tile, next pixel gather, next
    // row-tile load).  Issue is in order, so work placed in FRONT of a step's MFMAs is exposed every step;
    // in the fp32 LDS form it goes behind the first TM MFMAs and runs in the shadow of the rest (r02:
    // weight-gradient kernel 110 -> 119 TFLOP/s with the same move).
    bf16x8 ah[SP ? TM : 1], am[SP ? TM : 1], al[SP ? TM : 1];     // SP: row fragments of the current step
    f16x8 hh[H2 ? TM : 1], hl[H2 ? TM : 1];                         // H2: the two pieces of the row fragments
#ifdef OG_DEV
    if (SP) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { ah[i][j] = (__bf16)(0.001f * (lane + j)); am[i][j] = ah[i][j]; al[i][j] = ah[i][j]; }
    }
#endif
    auto mma = [&](const float (&rb)[NG * NB], const f32x4 (&ad)[NAD], int cur, auto&& mid) {
        if ((BF && !NH) || !ALDS) mid();       // (SP / NH with LDS: behind their first TM MFMAs, below)
        if (HR) {
            // the pieces arrive split: products and order of H2 (hl.bh, hh.bh, hh.bl per accumulator), every row
            // fragment feeding the NG pixel groups; the refill behind the first TM * NG MFMAs
            f16x8 bh[NG], bl[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x4 vh, vl;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vh[j] = rb[g * 8 + j]; vl[j] = rb[g * 8 + 4 + j]; }
                bh[g] = __builtin_bit_cast(f16x8, vh);
                bl[g] = __builtin_bit_cast(f16x8, vl);
            }
            const char* T = reinterpret_cast<const char*>(lds + cur * TILE) + lcol * (LD * 4) + lrow * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                hl[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4 + 32);
                hh[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(hl[i], bh[g], acc[i * NG + g]);
            __builtin_amdgcn_sched_barrier(0);
            mid();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(hh[i], bh[g], acc[i * NG + g]);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i) OG_MFMA_H(hh[i], bl[g], acc[i * NG + g]);
            return;
        }
        if (SP) {
            // Order: the three products that need only the h piece of the pixel fragment first (4 conversions), the
            // refill behind the first TM of them, the m / l pieces (~40 VALU) pinned between the next 2 TM MFMAs.
            bf16x8 bh, bm, bl;
            og_split8_h(rb, bh);
            if (ALDS && !OG_ABL(64)) {
                const char* T = reinterpret_cast<const char*>(lds + cur * TILE) + lcol * (LD * 4) + lrow * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    al[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 64);
                    am[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4 + 32);
                    ah[i] = *reinterpret_cast<const bf16x8*>(T + i * 32 * LD * 4);
                }
            } else if (!ALDS) {
                ah[0] = __builtin_bit_cast(bf16x8, ad[0]);
                am[0] = __builtin_bit_cast(bf16x8, ad[1]);
                al[0] = __builtin_bit_cast(bf16x8, ad[NAD - 1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(al[i], bh, acc[i]);
            if (ALDS) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (OG_ABL(32)) { bm = bh; bl = bh; } else og_split8_ml(rb, bh, bm, bl);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bh, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bh, acc[i]);
            if (ALDS) og_interleave<2 * TM, (40 + 2 * TM - 1) / (2 * TM)>();
            if (OG_ABL(1)) return;
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(am[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bm, acc[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_BF(ah[i], bl, acc[i]);
            return;
        }
        if (H2) {
            // Order as SP: the product that needs only the h piece of the pixel fragment first, the refill behind the
            // first TM MFMAs, the l piece (16 VALU) between the next TM.
            f16x8 bh, bl;
            float sc[8];
            og_h2_split_h(rb, h2_xs, sc, bh);
            if (ALDS) {
                const char* T = reinterpret_cast<const char*>(lds + cur * TILE) + lcol * (LD * 4) + lrow * 16;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    hl[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4 + 32);
                    hh[i] = *reinterpret_cast<const f16x8*>(T + i * 32 * LD * 4);
                }
            } else {
                hh[0] = __builtin_bit_cast(f16x8, ad[0]);
                hl[0] = __builtin_bit_cast(f16x8, ad[1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(hl[i], bh, acc[i]);
            if (ALDS) {
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
            }
            og_h2_split_l(sc, bh, bl);
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(hh[i], bh, acc[i]);
            if (ALDS) og_interleave<TM, (16 + TM - 1) / TM>();
#pragma unroll
            for (int i = 0; i < TM; ++i) OG_MFMA_H(hh[i], bl, acc[i]);
            return;
        }
        if (BF) {
            bf16x8 bq[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (NH) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rb[NBC * h + j];
                    bq[h] = __builtin_bit_cast(bf16x8, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bq[h][j] = (__bf16)rb[NBC * h + j];
                }
            }
            if (ALDS) {
                const float* T = lds + cur * TILE;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 aq[TM];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        aq[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(T + lcol * LD + h * 8 + lrow * 4 + i * 32 * LD));
                    if (NH && h == 1) {        // the refill (LDS store of the next row tile, loads two steps ahead)
                        __builtin_amdgcn_sched_barrier(0);
                        mid();
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[i], bq[h], acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ad[h]), bq[h], acc[0], 0, 0, 0);
            }
            return;
        }
        if (ALDS) {
            const float* T = lds + cur * TILE;
            f32x4 a0[TM], a1[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a0[i] = *reinterpret_cast<const f32x4*>(T + a_rd + i * 32 * LD);
                a1[i] = *reinterpret_cast<const f32x4*>(T + a_rd + i * 32 * LD + 4);
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
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[0][kk], rb[kk], acc[0], 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[1][kk], rb[4 + kk], acc[0], 0, 0, 0);
        }
    };

    // prologue: row tile of step kt0 in LDS buffer 0, of kt0+1 in flight; pixel fragment of kt0 in rb0
    if (ALDS) {
        load_a(kt0);
        store_a(0);
        if (P3 || kt0 + 1 < nk) load_a(kt0 + 1);
    } else {
        load_adir(ad0, kt0);
    }
    load_b(rb0);
    if (P3 && ALDS) load_b(rb1);
    if (ALDS) __syncthreads();
    int cur = 0;
    // Two steps per trip with the fragment registers in fixed ping-pong roles (rb0/ad0 even, rb1/ad1
    // odd): the gather of step k+1 is only waited for at its own MFMAs, i.e. it overlaps the MFMAs of
    // step k.  (A single-step loop with a register copy at its end makes hipcc wait for the fresh
    // loads BEFORE the MFMAs of the current step -- the whole memory latency exposed every step.)
    int kt = kt0;
    if (!ALDS) {
        // LDS-free form: nothing couples the waves, so the only latency cover is the prefetch depth.
        // Three fragment sets in fixed rotation keep the loads TWO steps (16 MFMAs) ahead; one step
        // (8 MFMAs, ~0.25 us) is shorter than a loaded L2 round trip whenever fewer than ~5 waves
        // share a SIMD, which is exactly the small-grid case this form serves.
        float rb2[NG * NB];
        f32x4 ad2[NAD];
        // (the loads run up to two steps past the end, unconditionally: a branch around them makes
        // hipcc's wait-count insertion drain everything at the next MFMA block; the surplus reads
        // land inside the buffers or on the range check and are never used)
        load_adir(ad1, kt0 + 1); load_b(rb1);
        for (; kt + 2 < nk; kt += 3) {
            load_adir(ad2, kt + 2); load_b(rb2);
            mma(rb0, ad0, 0, [] {});
            load_adir(ad0, kt + 3); load_b(rb0);
            mma(rb1, ad1, 0, [] {});
            load_adir(ad1, kt + 4); load_b(rb1);
            mma(rb2, ad2, 0, [] {});
        }
        if (kt < nk) mma(rb0, ad0, 0, [] {});
        if (kt + 1 < nk) mma(rb1, ad1, 0, [] {});
        kt = nk;
    }
    if (P3 && ALDS) {
        // Split mode: a step is 6 * TM MFMAs of 32 cycles -- shorter than a loaded gather round trip for the
        // short tiles -- so the pixel gather runs TWO steps ahead (three fragment sets in fixed rotation, three
        // steps per trip).  Order inside the refill: LDS store of the row tile loaded one step ago, load of the
        // next row tile, THEN the gather: vector-memory results return in order, so the wait for the row tile in
        // the next step leaves the younger gather in flight.  Loads run up to two steps past the end,
        // unconditionally (inside the buffers or on the range check, never used).  Three LDS row tiles in the
        // same rotation: every buffer index is a literal (a run-time `cur` ended up in a VGPR here -- hipcc merged
        // its initial 0 with the zero of the accumulator init -- and with it every LDS address and the scalar
        // offsets of the row loads: waterfall loops around each buffer_load).
        float rb2[NG * NB];
        int ks = kt0;
#ifdef OG_DEV
        if (a.ablate & 0x7e) {          // ablation form of the main loop (development builds only)
#pragma unroll
            for (int j = 0; j < NB; ++j) rb2[j] = rb0[j];
            for (; ks + 2 < nk; ks += 3) {
                mma(rb0, ad0, 0, [&]() { if (!OG_ABL(16)) store_a(1); if (!OG_ABL(4)) load_a(ks + 2); if (!OG_ABL(2)) load_b(rb2); });
                if (!OG_ABL(8)) __syncthreads();
                mma(rb1, ad0, 1, [&]() { if (!OG_ABL(16)) store_a(2); if (!OG_ABL(4)) load_a(ks + 3); if (!OG_ABL(2)) load_b(rb0); });
                if (!OG_ABL(8)) __syncthreads();
                mma(rb2, ad0, 2, [&]() { if (!OG_ABL(16)) store_a(0); if (!OG_ABL(4)) load_a(ks + 4); if (!OG_ABL(2)) load_b(rb1); });
                if (!OG_ABL(8)) __syncthreads();
            }
        } else
#endif
        if (ks + 2 < nk) {
            do {
                mma(rb0, ad0, 0, [&]() { store_a(1); load_a(ks + 2); load_b(rb2); });
                __syncthreads();
                mma(rb1, ad0, 1, [&]() { store_a(2); load_a(ks + 3); load_b(rb0); });
                __syncthreads();
                mma(rb2, ad0, 2, [&]() { store_a(0); load_a(ks + 4); load_b(rb1); });
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
            if (ALDS) store_a(cur ^ 1); else load_adir(ad1, kt + 1);
            load_b(rb1);
            if (ALDS && kt + 2 < nk) load_a(kt + 2);
        });
        if (ALDS) __syncthreads();
        cur ^= 1;
        mma(rb1, ad1, cur, [&]() {
            if (kt + 2 < nk) {
                if (ALDS) store_a(cur ^ 1); else load_adir(ad0, kt + 2);
                load_b(rb0);
            }
            if (ALDS && kt + 3 < nk) load_a(kt + 3);
        });
        if (ALDS) __syncthreads();
        cur ^= 1;
    }
    if (kt < nk) mma(rb0, ad0, cur, [] {});            // odd step count: last step

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int ppi = a.PH * a.PW;
    const bool split = a.ksplit_steps > 0;
    const bool lrelu = a.act == OG_ACT_LRELU, relu = a.act == OG_ACT_RELU;
    const bool to_ws = split && a.ws;
    float vmax = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (!pix_ok[g]) continue;           // (with ymax every lane of the wave stays for the maximum below)
        const int n = pix[g] / ppi;
        const int rem = pix[g] - n * ppi;
        const int pa = rem / a.PW;
        const int pb = rem - pa * a.PW;
        const int oh = pa * a.osh + (a.nphase > 1 ? (phase >> 1) : a.ooh);
        const int ow = pb * a.osw + (a.nphase > 1 ? (phase & 1) : a.oow);
        size_t plane = (size_t)a.OHf * a.OWf;
        float* yb = a.y + (size_t)n * a.M * plane + (size_t)oh * a.OWf + ow;
        bool in_ring = false;
        if (a.ring) {        // padded grid -> unpadded y (interior) or the ring buffer (see IgemmArgs::ring)
            const bool inner = pa >= 1 && pa <= a.PH - 2 && pb >= 1 && pb <= a.PW - 2;
            in_ring = !inner;
            if (inner) {
                yb = a.y + (size_t)n * a.M * plane + (size_t)(pa - 1) * a.OWf + (pb - 1);
            } else {
                const int R = 2 * a.PW + 2 * a.PH;
                const int ri = pa == 0 ? pb : (pa == a.PH - 1 ? a.PW + pb : (pb == 0 ? 2 * a.PW + pa : 2 * a.PW + a.PH + pa));
                yb = a.ring + (size_t)n * a.M * R + ri;
                plane = (size_t)R;
            }
        }
        if (to_ws) {      // partial tile of this K range -> workspace slot of this split (same element offsets as y / ring)
            const size_t off = in_ring ? (size_t)a.N * a.M * a.OHf * a.OWf + (size_t)(yb - a.ring) : (size_t)(yb - a.y);
            yb = a.ws + (size_t)blockIdx.y * a.ws_stride + off;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lrow;
                if (m < a.m_end) {
                    float v = H2 ? acc[i * NG + g][r] * h2_inv : acc[i * NG + g][r];
                    if (to_ws) {           // (a split launch always has its workspace: run_igemm2)
                        yb[(size_t)m * plane] = v;
                    } else {
                        if (a.bias) v += a.bias[m];
                        v = lrelu ? (v > 0.f ? v : 0.2f * v) : (relu ? fmaxf(v, 0.f) : v);
                        yb[(size_t)m * plane] = v;
                        vmax = fmaxf(vmax, fabsf(v));
                    }
                }
            }
        }
    }
    if (a.ymax) {       // partial maxima of |y| for an fp16x2 consumer: one integer atomicMax per wave into zeroed slots
        const float wm = og_wave_max(vmax);
        if (lane == 0)
            atomicMax(reinterpret_cast<unsigned*>(a.ymax) + ((blockIdx.x * NW + wid) & (OG_AMAX_SLOTS - 1)), __float_as_uint(wm));
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
    int math;          // 0 fp32, 1 bf16 inputs, 2 bf16x3, 4 fp16x2 (register-fragment kernel only; else bf16x3 runs)
    const float* xmax;   // math 4: per-workgroup maxima of |x| and |dy| (objgan_absmax_partials)
    const float* dymax;
    // Where a workgroup's tile goes.  One split (gridDim.y == 1): straight into dw -- stored, or added to what is there
    // when `accumulate` -- every element by exactly one thread.  Several splits: the partial tile of split s goes to
    // ws[s * ws_stride + (m - m_begin) * ncol + col] (extra rows behind the block rows); wgrad_combine_kernel sums the
    // splits in order.  No atomics, no zero-fill: the weight gradient is bit-reproducible.
    float* ws;
    long ws_stride;
    int accumulate;
    int xr_begin, xr_count;   // XR kernels: dy rows [xr_begin, xr_begin + xr_count) (<= 4) are carried by block row 0 on
                              // the fp32 VALU instead of costing a 32-row MFMA group (194 = 6*32 + 2, 388 = 12*32 + 4):
                              // the lane's eight gathered values (its MFMA operand) meet the extra rows' dy values read
                              // as LDS broadcasts; the two pixel halves of a column meet in one shuffle in the epilogue.
                              // r02: -8 % on the 194 / 388-channel weight gradients at 128^2, -20 % at 32^2.
};

__device__ __forceinline__ void og_wgrad_store(const WgradArgs& a, int m, int col, float v, int split) {
    if (a.ws) {
        a.ws[(size_t)split * a.ws_stride + (size_t)(m - a.m_begin) * a.ncol + col] = v;
    } else {
        float* p = a.dw + (size_t)m * a.ncol + col;
        *p = a.accumulate ? *p + v : v;
    }
}
// extra row j of the XR kernels: local row (m_end - m_begin) + j of the workspace slot
