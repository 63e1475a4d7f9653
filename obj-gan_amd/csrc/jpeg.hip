// Baseline JPEG decode on the MI355X: the loader side of the hot path.
//
// The reference decodes every training image on the host with PIL.Image.open(...).convert('RGB') (reference
// image_generation/miscc/load.py:141-151); Pillow's decoder is libjpeg-turbo with the library defaults (JDCT_ISLOW,
// fancy upsampling, RGB output).  Here the file BYTES cross PCIe and the image is decoded on the device, bit for bit what
// Pillow returns, in three kernels:
//
//   jpeg_entropy_kernel   The Huffman-coded scan is a serial bit stream (no restart markers
//                         in ordinary files): a file seen for the FIRST time is walked by one lane -- JPEG Annex F /
//                         jdhuff.c decode_mcu: 9-bit look-ahead table, canonical-code walk for longer codes, DC prediction,
//                         EOB / ZRL, restart intervals; tables and a per-lane 128-byte window of the file in LDS -- and on the
//                         way the lane records its state (byte position, bit accumulator, DC predictors) at every MCU-row
//                         start: the file's ENTROPY INDEX (48 bytes per MCU row).  A training set is decoded once per epoch,
//                         so from the second epoch on the file arrives with its index and every MCU row is decoded by its own
//                         lane (the 60 rows of a 640 x 480 image at once, four decoding lanes per wave: lanes of one wave that
//                         walk different bit streams serialise on every data-dependent branch).  Same state machine, same
//                         state: bit-identical.
//                         Output: quantised coefficients, int16 [block][64] in natural order, written sparsely into a
//                         zero-filled buffer.
//   jpeg_idct_kernel      one thread per 8 x 8 block: de-quantisation and the accurate integer inverse DCT of jidctint.c
//                         (jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2), samples clamped to 0..255 into the component's
//                         plane.  Integer arithmetic: exactly libjpeg's values.
//   jpeg_color_kernel     one thread per four output pixels: "fancy" triangle-filter chroma upsampling (jdsample.c
//                         h2v1_fancy_upsample / h2v2_fancy_upsample, edge rows and columns replicated as jdmainct.c's
//                         context rows do) fused with the YCbCr -> RGB conversion of jdcolor.c (16-bit fixed point),
//                         interleaved RGB bytes out.
//
// Headers are parsed on the host (objgan_jpeg_parse: markers, quantisation and Huffman tables -> a fixed-size descriptor):
// a few hundred bytes per file, no arithmetic.  Supported: baseline / extended sequential Huffman (SOF0 / SOF1), 8-bit, one
// interleaved scan, 1 component or 3 components with luma sampling 1x1 / 2x1 / 2x2 over 1x1 chroma.  Anything else
// (progressive, arithmetic, CMYK, 4:4:0 ...) is REFUSED by the parser (return 0, reason in the descriptor) so the caller can
// route that file to its host decoder knowingly; nothing is decoded approximately.
#include "common.h"
#include <string.h>

#define OG_JPEG_LOOK 9
#define OG_JPEG_MAXC 3

struct JpegHuff {                      // derived table (jdhuff.c jpeg_make_d_derived_tbl)
    unsigned short look[1 << OG_JPEG_LOOK];   // (code length << 8) | symbol for codes of <= 9 bits, 0 = longer
    int maxcode[18];                   // largest code of length l (-1: none); [17] = sentinel
    int valoff[17];                    // vals index of the first code of length l, minus that code
    unsigned char vals[256];
};

struct JpegDesc {                      // first fields mirrored by ctypes (objgan_hip/ops.py _JpegHead)
    int width, height, ncomp, hmax, vmax, mcux, mcuy, restart_interval;
    int reason;                        // 0 = supported; else why objgan_jpeg_parse refused the file
    int scan_offset;                   // byte offset of the entropy-coded data in the file
    long file_offset, nbytes;          // where the file sits in the batch's byte buffer (set by the caller)
    long out_offset;                   // byte offset of this image's RGB output (set by the caller)
    long coef_offset, plane_offset;    // element / byte offsets into the workspace (set by objgan_jpeg_plan)
    long seg_offset, idx_offset;       // first entry of this image in the batch's index (in) / index (out) arrays
    int nseg, pad0;                    // entries of the index handed in (0: none -- one lane decodes the whole scan)
    int ch[OG_JPEG_MAXC], cv[OG_JPEG_MAXC], ctq[OG_JPEG_MAXC], ctd[OG_JPEG_MAXC], cta[OG_JPEG_MAXC];
    int cblk_w[OG_JPEG_MAXC], cblk_h[OG_JPEG_MAXC];      // blocks per row / column of the component (MCU-padded)
    long ccoef[OG_JPEG_MAXC], cplane[OG_JPEG_MAXC];      // per-component offsets (elements of int16 / bytes) inside this image's slices
    unsigned short qt[4][64];          // natural order
    JpegHuff dc[4], ac[4];
};

__constant__ int c_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
                                 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52,
                                 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// --------------------------------------------------------------------------------------------------------------
// entropy decode
// --------------------------------------------------------------------------------------------------------------
// A SEGMENT is a run of consecutive MCUs with the decoder state at its first MCU.  A file seen for the first time is one
// segment (state: start of the scan) walked by one lane, which writes the state at every MCU-row start into the file's
// INDEX as it passes; a file that comes with its index is decoded by up to 64 lanes at once, one MCU row each.
#define OG_WIN 128                     // bytes of the file each lane holds in LDS (refilled by the lane itself, 16-byte loads)
#define OG_JPEG_LPW 4                  // decoding lanes per wave (see the kernel)

struct JpegSeg {                       // decoder state in front of MCU `mcu` (48 bytes)
    unsigned long long acc;            // bit accumulator (the low `nbits` bits are the unread bits)
    int pos;                           // next byte of the file to read
    int nbits;
    int mcu;                           // first MCU of the segment; -1: entry not written (yet)
    int todo;                          // MCUs until the next restart marker
    int pred[OG_JPEG_MAXC];            // DC predictors
    int marker;                        // a marker has been reached
    int pad;
};

struct LdsHuff {
    unsigned short look[1 << OG_JPEG_LOOK];
    int maxcode[18];
    int valoff[17];
    unsigned char vals[256];
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) unsigned lds_u32;

struct BitReader {
    const unsigned char* file;         // global
    lds_u8* win;                       // this lane's LDS window: file[wbase .. wbase + OG_WIN) (LDS address space: ds_read, not flat)
    long wbase, nbytes, pos;           // pos: next byte of the file to read
    unsigned long long acc;
    int nbits;
    bool marker;                       // a marker was reached: the rest of the segment reads as zeros (jdhuff.c)
};

// make file[p .. p + need) resident in the lane's window (need <= 8)
__device__ __forceinline__ void br_window(BitReader& b, long p, int need) {
    if (p >= b.wbase && p + need <= b.wbase + OG_WIN) return;
    b.wbase = p & ~15L;
#pragma unroll
    for (int i = 0; i < OG_WIN; i += 16) {
        uint4 v = {0xD9D9D9D9u, 0xD9D9D9D9u, 0xD9D9D9D9u, 0xD9D9D9D9u};
        // (whole 16-byte pieces inside the batch buffer: files start on 16-byte boundaries and the buffer is padded)
        if (b.wbase + i < ((b.nbytes + 15) & ~15L)) v = *reinterpret_cast<const uint4*>(b.file + b.wbase + i);
        lds_u32* w = (lds_u32*)(b.win + i);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
}

__device__ __forceinline__ unsigned br_byte(BitReader& b, long p) {
    if (p >= b.nbytes) return 0xD9u;
    br_window(b, p, 1);
    return b.win[p - b.wbase];
}

__device__ __forceinline__ void br_fill(BitReader& b) {
    // fast path: four bytes at once when none of them is 0xFF (no stuffing, no marker) -- two aligned LDS dwords, one byte
    // alignment, one byte swap instead of four byte loads with their checks
    if (!b.marker && b.nbits <= 31 && b.pos + 4 <= b.nbytes) {
        br_window(b, b.pos, 8);
        const int off = (int)(b.pos - b.wbase);
        lds_u32* w = (lds_u32*)(b.win + (off & ~3));
        const unsigned v = __builtin_amdgcn_alignbyte(w[1], w[0], off & 3);     // bytes pos .. pos + 3, first byte lowest
        const unsigned inv = ~v;
        if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {                   // no byte equals 0xFF
            b.acc = (b.acc << 32) | __builtin_bswap32(v);
            b.nbits += 32;
            b.pos += 4;
            return;
        }
    }
    while (b.nbits <= 48) {
        unsigned v = 0;
        if (!b.marker) {
            if (b.pos >= b.nbytes) {
                b.marker = true;
            } else {
                v = br_byte(b, b.pos);
                if (v == 0xFFu) {
                    const unsigned nx = br_byte(b, b.pos + 1);
                    if (nx == 0) b.pos += 2;                 // stuffed zero
                    else { b.marker = true; v = 0; }         // RSTn / EOI / anything else: stop consuming
                } else {
                    b.pos += 1;
                }
            }
        }
        b.acc = (b.acc << 8) | v;
        b.nbits += 8;
    }
}

__device__ __forceinline__ unsigned br_peek(BitReader& b, int k) {       // k <= 16, after br_fill
    return (unsigned)(b.acc >> (b.nbits - k)) & ((1u << k) - 1u);
}

__device__ __forceinline__ int br_huff(BitReader& b, const LdsHuff& t) {
    if (b.nbits < 16) br_fill(b);
    const unsigned c16 = br_peek(b, 16);
    const unsigned e = t.look[c16 >> (16 - OG_JPEG_LOOK)];
    if (e) { b.nbits -= (int)(e >> 8); return (int)(e & 255u); }
    for (int l = OG_JPEG_LOOK + 1; l <= 16; ++l) {          // canonical-code walk (jdhuff.c jpeg_huff_decode)
        const int code = (int)(c16 >> (16 - l));
        if (code <= t.maxcode[l]) {
            b.nbits -= l;
            return t.vals[(code + t.valoff[l]) & 255];
        }
    }
    b.nbits -= 16;                                          // corrupt code: libjpeg warns and returns 0
    return 0;
}

__device__ __forceinline__ int br_receive_extend(BitReader& b, int s) {
    if (s == 0) return 0;
    if (b.nbits < s) br_fill(b);
    const int v = (int)br_peek(b, s);
    b.nbits -= s;
    return v < (1 << (s - 1)) ? v - ((1 << s) - 1) : v;
}

// grid: images.  seg_in: per image `nseg` entries at d.seg_offset (nseg = 0: the file has no index -- one segment from the
// start of the scan); seg_out (may be null): one entry per MCU row at d.idx_offset, written by whichever lane passes the row.
__global__ __launch_bounds__(64) void jpeg_entropy_kernel(const unsigned char* __restrict__ files,
                                                          const JpegDesc* __restrict__ descs, short* __restrict__ coef,
                                                          const JpegSeg* __restrict__ seg_in, JpegSeg* __restrict__ seg_out) {
    __shared__ __attribute__((aligned(16))) unsigned char wins[64 * OG_WIN];
    __shared__ LdsHuff tabs[4];        // slots 0, 1: the scan's DC tables; 2, 3: its AC tables
    __shared__ unsigned char zz[64];
    const JpegDesc& d = descs[blockIdx.x];
    if (d.reason != 0) return;
    if ((int)blockIdx.y * OG_JPEG_LPW >= (d.nseg > 0 ? d.nseg : 1)) return;     // (grid.y is sized for the image with most rows)
    const int lane = threadIdx.x;
    const unsigned char* file = files + d.file_offset;
    // Huffman tables used by the components (at most two DC + two AC in one scan; components that share ids share a slot)
    int dslot[OG_JPEG_MAXC], aslot[OG_JPEG_MAXC];
    {
        int nd = 0, na = 0, did[2] = {-1, -1}, aid[2] = {-1, -1};
        for (int c = 0; c < d.ncomp; ++c) {
            int k = 0;
            for (; k < nd; ++k) if (did[k] == d.ctd[c]) break;
            if (k == nd && nd < 2) did[nd++] = d.ctd[c];
            dslot[c] = k < 2 ? k : 0;
            k = 0;
            for (; k < na; ++k) if (aid[k] == d.cta[c]) break;
            if (k == na && na < 2) aid[na++] = d.cta[c];
            aslot[c] = 2 + (k < 2 ? k : 0);
        }
        for (int k = 0; k < 2; ++k) {
            const JpegHuff* srcs[2] = {did[k] >= 0 ? &d.dc[did[k]] : nullptr, aid[k] >= 0 ? &d.ac[aid[k]] : nullptr};
            for (int w = 0; w < 2; ++w) {
                if (!srcs[w]) continue;
                LdsHuff& t = tabs[w * 2 + k];
                for (int i = lane; i < (1 << OG_JPEG_LOOK); i += 64) t.look[i] = srcs[w]->look[i];
                for (int i = lane; i < 256; i += 64) t.vals[i] = srcs[w]->vals[i];
                if (lane < 18) t.maxcode[lane] = srcs[w]->maxcode[lane];
                if (lane < 17) t.valoff[lane] = srcs[w]->valoff[lane];
            }
        }
    }
    zz[lane] = (unsigned char)c_zigzag[lane];
    __syncthreads();

    // the descriptor fields the serial loop needs, in registers (a global load per use would sit on the lane's critical path)
    const int ncomp = d.ncomp, mcux = d.mcux, rint = d.restart_interval;
    const int nmcu = d.mcux * d.mcuy;
    int c_h[OG_JPEG_MAXC], c_v[OG_JPEG_MAXC], c_bw[OG_JPEG_MAXC];
    long c_co[OG_JPEG_MAXC];
    for (int c = 0; c < OG_JPEG_MAXC; ++c) { c_h[c] = d.ch[c]; c_v[c] = d.cv[c]; c_bw[c] = d.cblk_w[c]; c_co[c] = d.ccoef[c]; }
    short* cimg = coef + d.coef_offset;
    const int nseg = d.nseg > 0 ? d.nseg : 1;
    JpegSeg* iout = seg_out ? seg_out + d.idx_offset : nullptr;

    // OG_JPEG_LPW lanes of a wave decode (one segment each): lanes of one wave walk DIFFERENT bit streams, and every
    // data-dependent branch of the decoder serialises them -- measured with 60 rows on the 64 lanes of one wave: 13 ms per
    // batch of sixteen 640 x 480 files, little better than the 60 rows one after the other; a few lanes per wave and many
    // waves (blockIdx.y) spread the rows over the CUs instead
    const int sg = (int)blockIdx.y * OG_JPEG_LPW + lane;
    if (lane < OG_JPEG_LPW && sg < nseg) {
        BitReader br;
        br.file = file; br.win = (lds_u8*)(wins + lane * OG_WIN); br.wbase = -OG_WIN; br.nbytes = d.nbytes;
        int pred[OG_JPEG_MAXC] = {0, 0, 0};
        int mcu, mcu_end, todo;
        if (d.nseg > 0) {
            const JpegSeg e = seg_in[d.seg_offset + sg];
            br.pos = e.pos; br.acc = e.acc; br.nbits = e.nbits; br.marker = e.marker != 0;
            pred[0] = e.pred[0]; pred[1] = e.pred[1]; pred[2] = e.pred[2];
            mcu = e.mcu; todo = e.todo;
            mcu_end = sg + 1 < d.nseg ? seg_in[d.seg_offset + sg + 1].mcu : nmcu;
        } else {
            br.pos = d.scan_offset; br.acc = 0; br.nbits = 0; br.marker = false;
            mcu = 0; mcu_end = nmcu; todo = rint;
        }
        while (mcu < mcu_end) {
            if (rint && todo == 0) {
                // byte-align, skip to behind the RSTn marker (jdhuff.c process_restart)
                br.acc = 0; br.nbits = 0; br.marker = false;
                while (br.pos + 1 < br.nbytes) {
                    const unsigned a0 = br_byte(br, br.pos), a1 = br_byte(br, br.pos + 1);
                    if (a0 == 0xFFu && a1 >= 0xD0u && a1 <= 0xD7u) break;
                    br.pos += 1;
                }
                br.pos += 2;
                pred[0] = pred[1] = pred[2] = 0;
                todo = rint;
            }
            const int my = mcu / mcux, mx = mcu - my * mcux;
            if (iout && mx == 0) {                          // the index entry of this MCU row: the state right here
                JpegSeg e;
                e.acc = br.acc; e.pos = (int)br.pos; e.nbits = br.nbits; e.mcu = mcu; e.todo = todo;
                e.pred[0] = pred[0]; e.pred[1] = pred[1]; e.pred[2] = pred[2]; e.marker = br.marker ? 1 : 0; e.pad = 0;
                iout[my] = e;
            }
            for (int c = 0; c < ncomp; ++c) {
                const LdsHuff& td = tabs[dslot[c]];
                const LdsHuff& ta = tabs[aslot[c]];
                for (int by = 0; by < c_v[c]; ++by)
                    for (int bx = 0; bx < c_h[c]; ++bx) {
                        short* blk = cimg + c_co[c] + ((long)(my * c_v[c] + by) * c_bw[c] + (mx * c_h[c] + bx)) * 64;
                        int s = br_huff(br, td);
                        pred[c] += br_receive_extend(br, s & 15);
                        blk[0] = (short)pred[c];
                        int k = 1;
                        while (k < 64) {
                            const int rs = br_huff(br, ta);
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s) {
                                k += r;
                                const int v = br_receive_extend(br, s);
                                if (k > 63) break;
                                blk[zz[k]] = (short)v;
                                k += 1;
                            } else if (r == 15) {
                                k += 16;
                            } else {
                                break;
                            }
                        }
                    }
            }
            mcu += 1;
            todo -= 1;
        }
    }
}

// --------------------------------------------------------------------------------------------------------------
// inverse DCT (jidctint.c jpeg_idct_islow), one thread per block
// --------------------------------------------------------------------------------------------------------------
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

__device__ __forceinline__ void idct8(const int (&d)[8], int (&o)[8], const int shift) {
    int z2 = d[2], z3 = d[6];
    int z1 = (z2 + z3) * FIX_0_541196100;
    const int tmp2 = z1 + z3 * (-FIX_1_847759065);
    const int tmp3 = z1 + z2 * FIX_0_765366865;
    const int tmp0 = (d[0] + d[4]) << 13;
    const int tmp1 = (d[0] - d[4]) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int t0 = d[7], t1 = d[5], t2 = d[3], t3 = d[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    int z4 = t1 + t3;
    const int z5 = (z3 + z4) * FIX_1_175875602;
    t0 *= FIX_0_298631336; t1 *= FIX_2_053119869; t2 *= FIX_3_072711026; t3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 = z3 * (-FIX_1_961570560) + z5;
    z4 = z4 * (-FIX_0_390180644) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    o[0] = (tmp10 + t3 + rnd) >> shift; o[7] = (tmp10 - t3 + rnd) >> shift;
    o[1] = (tmp11 + t2 + rnd) >> shift; o[6] = (tmp11 - t2 + rnd) >> shift;
    o[2] = (tmp12 + t1 + rnd) >> shift; o[5] = (tmp12 - t1 + rnd) >> shift;
    o[3] = (tmp13 + t0 + rnd) >> shift; o[4] = (tmp13 - t0 + rnd) >> shift;
}

struct IdctJob { int image, comp; long first_block; };      // (host-built prefix table is avoided: blocks are indexed per image)

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegDesc* __restrict__ descs, const short* __restrict__ coef,
                                                        unsigned char* __restrict__ planes, int max_blocks) {
    const JpegDesc& d = descs[blockIdx.y];
    if (d.reason != 0) return;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= max_blocks) return;
    int c = 0, rel = b;
    for (; c < d.ncomp; ++c) {
        const int nb = d.cblk_w[c] * d.cblk_h[c];
        if (rel < nb) break;
        rel -= nb;
    }
    if (c >= d.ncomp) return;
    const short* src = coef + d.coef_offset + d.ccoef[c] + (long)rel * 64;
    const unsigned short* q = d.qt[d.ctq[c]];
    int ws[8][8];
#pragma unroll
    for (int col = 0; col < 8; ++col) {                 // pass 1: columns
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) in[r] = (int)src[r * 8 + col] * (int)q[r * 8 + col];
        idct8(in, o, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[r][col] = o[r];
    }
    const int by = rel / d.cblk_w[c], bx = rel - by * d.cblk_w[c];
    const int pitch = d.cblk_w[c] * 8;
    unsigned char* dst = planes + d.plane_offset + d.cplane[c] + (long)(by * 8) * pitch + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {                       // pass 2: rows
        int o[8];
        idct8(ws[r], o, 13 + 2 + 3);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (unsigned)min(255, max(0, o[k] + 128)) << (8 * k);
            hi |= (unsigned)min(255, max(0, o[4 + k] + 128)) << (8 * k);
        }
        *reinterpret_cast<uint2*>(dst + (long)r * pitch) = make_uint2(lo, hi);
    }
}

// --------------------------------------------------------------------------------------------------------------
// chroma upsampling + colour conversion
// --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int chroma_at(const unsigned char* __restrict__ p, int pitch, int dw, int dh, int hs, int vs,
                                         int y, int x) {
    // value of the component upsampled to full resolution at (y, x): hs / vs = 1 (no upsampling) or 2 (fancy)
    if (hs == 1 && vs == 1) return p[(long)y * pitch + x];
    // jdsample.c jinit_upsampler selects the fancy filters only when downsampled_width > 2: narrower components are
    // replicated (h2v1_upsample / h2v2_upsample)
    if (dw <= 2) return p[(long)(vs == 2 ? y >> 1 : y) * pitch + (x >> 1)];
    if (vs == 1) {                                          // h2v1_fancy_upsample
        const int cx = x >> 1;
        const int v = p[(long)y * pitch + cx];
        if (x & 1) { return cx == dw - 1 ? v : (v * 3 + p[(long)y * pitch + cx + 1] + 2) >> 2; }
        return cx == 0 ? v : (v * 3 + p[(long)y * pitch + cx - 1] + 1) >> 2;
    }
    // h2v2_fancy_upsample: nearer row cy, farther row above (even y) / below (odd y), replicated at the edges
    const int cy = y >> 1, cx = x >> 1;
    int fy = (y & 1) ? cy + 1 : cy - 1;
    fy = fy < 0 ? 0 : (fy > dh - 1 ? dh - 1 : fy);
    const unsigned char* r0 = p + (long)cy * pitch;
    const unsigned char* r1 = p + (long)fy * pitch;
    const int cs = r0[cx] * 3 + r1[cx];
    if (x & 1) {
        if (cx == dw - 1) return (cs * 4 + 7) >> 4;
        return (cs * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
    }
    if (cx == 0) return (cs * 4 + 8) >> 4;
    return (cs * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const JpegDesc* __restrict__ descs,
                                                         const unsigned char* __restrict__ planes,
                                                         unsigned char* __restrict__ out, int max_quads) {
    const JpegDesc& d = descs[blockIdx.y];
    if (d.reason != 0) return;
    const int W = d.width, H = d.height;
    const int qw = (W + 3) >> 2;                            // groups of four pixels per row
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= max_quads || g >= qw * H) return;
    const int y = g / qw, x0 = (g - y * qw) * 4;
    const unsigned char* pl = planes + d.plane_offset;
    unsigned char* o = out + d.out_offset + ((long)y * W + x0) * 3;
    const int p0 = d.cblk_w[0] * 8;
    if (d.ncomp == 1) {
        for (int i = 0; i < 4 && x0 + i < W; ++i) {
            const unsigned char v = pl[d.cplane[0] + (long)y * p0 + x0 + i];
            o[3 * i] = v; o[3 * i + 1] = v; o[3 * i + 2] = v;
        }
        return;
    }
    const int hs = d.hmax / d.ch[1], vs = d.vmax / d.cv[1];
    const int dw = (W * d.ch[1] + d.hmax - 1) / d.hmax, dh = (H * d.cv[1] + d.vmax - 1) / d.vmax;
    const int p1 = d.cblk_w[1] * 8, p2 = d.cblk_w[2] * 8;
    for (int i = 0; i < 4 && x0 + i < W; ++i) {
        const int x = x0 + i;
        const int yy = pl[d.cplane[0] + (long)y * p0 + x];
        const int cb = chroma_at(pl + d.cplane[1], p1, dw, dh, hs, vs, y, x) - 128;
        const int cr = chroma_at(pl + d.cplane[2], p2, dw, dh, hs, vs, y, x) - 128;
        // jdcolor.c build_ycc_rgb_table: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 1 << 15
        const int r = yy + ((91881 * cr + 32768) >> 16);
        const int gg = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        const int b = yy + ((116130 * cb + 32768) >> 16);
        o[3 * i] = (unsigned char)min(255, max(0, r));
        o[3 * i + 1] = (unsigned char)min(255, max(0, gg));
        o[3 * i + 2] = (unsigned char)min(255, max(0, b));
    }
}

// --------------------------------------------------------------------------------------------------------------
// host side: header parse, workspace plan, launch
// --------------------------------------------------------------------------------------------------------------
static const int h_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
                                 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52,
                                 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static bool og_jpeg_derive(const unsigned char* bits, const unsigned char* vals, int nvals, JpegHuff& t) {
    memset(&t, 0, sizeof(t));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        const int n = bits[l - 1];
        if (n) {
            t.valoff[l] = k - code;
            for (int i = 0; i < n; ++i, ++k, ++code) {
                if (k >= nvals || code >= (1 << l)) return false;
                if (l <= OG_JPEG_LOOK) {
                    const int first = code << (OG_JPEG_LOOK - l);
                    for (int j = 0; j < (1 << (OG_JPEG_LOOK - l)); ++j)
                        t.look[first + j] = (unsigned short)((l << 8) | vals[k]);
                }
            }
            t.maxcode[l] = code - 1;
        } else {
            t.maxcode[l] = -1;
        }
        code <<= 1;
    }
    t.maxcode[17] = 0xFFFFF;
    memcpy(t.vals, vals, nvals < 256 ? nvals : 256);
    return true;
}

extern "C" {

long objgan_jpeg_desc_bytes(void) { return (long)sizeof(JpegDesc); }

// Reasons objgan_jpeg_parse refuses a file (JpegDesc::reason)
enum { OG_JPEG_OK = 0, OG_JPEG_NOT_JPEG = 1, OG_JPEG_PROGRESSIVE = 2, OG_JPEG_PRECISION = 3, OG_JPEG_COMPONENTS = 4,
       OG_JPEG_SAMPLING = 5, OG_JPEG_SCAN = 6, OG_JPEG_TABLES = 7, OG_JPEG_TRUNCATED = 8, OG_JPEG_COLORSPACE = 9 };

int objgan_jpeg_parse(const unsigned char* f, long n, void* desc_out) {
    JpegDesc& d = *reinterpret_cast<JpegDesc*>(desc_out);
    memset(&d, 0, sizeof(d));
    d.nbytes = n;
    bool have_q[4] = {false, false, false, false}, have_dc[4] = {false, false, false, false},
         have_ac[4] = {false, false, false, false};
    int cid[OG_JPEG_MAXC] = {0, 0, 0};
    bool have_sof = false;
    bool saw_jfif = false, saw_adobe = false;          // colour-space guess of libjpeg (jdapimin.c default_decompress_parms)
    int adobe_transform = 1;
#define OG_REFUSE(why) do { d.reason = (why); return OG_BAD_ARGS; } while (0)
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) OG_REFUSE(OG_JPEG_NOT_JPEG);
    long pos = 2;
    while (pos + 4 <= n) {
        if (f[pos] != 0xFF) OG_REFUSE(OG_JPEG_NOT_JPEG);
        while (pos < n && f[pos] == 0xFF) ++pos;
        if (pos >= n) break;
        const int m = f[pos++];
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (pos + 2 > n) OG_REFUSE(OG_JPEG_TRUNCATED);
        const long L = ((long)f[pos] << 8) | f[pos + 1];
        if (L < 2 || pos + L > n) OG_REFUSE(OG_JPEG_TRUNCATED);
        const unsigned char* s = f + pos + 2;
        const long sl = L - 2;
        if (m == 0xDB) {
            long i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                if (pq != 0 || tq > 3 || i + 65 > sl) OG_REFUSE(OG_JPEG_TABLES);
                for (int k = 0; k < 64; ++k) d.qt[tq][h_zigzag[k]] = s[i + 1 + k];
                have_q[tq] = true;
                i += 65;
            }
        } else if (m == 0xC4) {
            long i = 0;
            while (i < sl) {
                if (i + 17 > sl) OG_REFUSE(OG_JPEG_TABLES);
                const int tc = s[i] >> 4, th = s[i] & 15;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += s[i + 1 + k];
                if (tc > 1 || th > 3 || cnt > 256 || i + 17 + cnt > sl) OG_REFUSE(OG_JPEG_TABLES);
                if (!og_jpeg_derive(s + i + 1, s + i + 17, cnt, tc ? d.ac[th] : d.dc[th])) OG_REFUSE(OG_JPEG_TABLES);
                (tc ? have_ac : have_dc)[th] = true;
                i += 17 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6) OG_REFUSE(OG_JPEG_TRUNCATED);
            if (s[0] != 8) OG_REFUSE(OG_JPEG_PRECISION);
            d.height = (s[1] << 8) | s[2];
            d.width = (s[3] << 8) | s[4];
            d.ncomp = s[5];
            if ((d.ncomp != 1 && d.ncomp != 3) || sl < 6 + 3 * d.ncomp) OG_REFUSE(OG_JPEG_COMPONENTS);
            if (d.width < 1 || d.height < 1) OG_REFUSE(OG_JPEG_TRUNCATED);
            for (int c = 0; c < d.ncomp; ++c) {
                cid[c] = s[6 + 3 * c];
                d.ch[c] = s[7 + 3 * c] >> 4;
                d.cv[c] = s[7 + 3 * c] & 15;
                d.ctq[c] = s[8 + 3 * c];
                if (d.ctq[c] > 3) OG_REFUSE(OG_JPEG_TABLES);
            }
            have_sof = true;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
            OG_REFUSE(OG_JPEG_PROGRESSIVE);
        } else if (m == 0xE0) {
            if (sl >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) saw_jfif = true;
        } else if (m == 0xEE) {
            if (sl >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') {
                saw_adobe = true;
                adobe_transform = s[11];
            }
        } else if (m == 0xDD) {
            if (sl < 2) OG_REFUSE(OG_JPEG_TRUNCATED);
            d.restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!have_sof || sl < 1) OG_REFUSE(OG_JPEG_SCAN);
            const int ns = s[0];
            if (ns != d.ncomp || sl < 1 + 2 * ns + 3) OG_REFUSE(OG_JPEG_SCAN);
            for (int k = 0; k < ns; ++k) {
                if (s[1 + 2 * k] != cid[k]) OG_REFUSE(OG_JPEG_SCAN);       // components in frame order (every encoder's)
                d.ctd[k] = s[2 + 2 * k] >> 4;
                d.cta[k] = s[2 + 2 * k] & 15;
                if (d.ctd[k] > 3 || d.cta[k] > 3 || !have_dc[d.ctd[k]] || !have_ac[d.cta[k]] || !have_q[d.ctq[k]])
                    OG_REFUSE(OG_JPEG_TABLES);
            }
            // at most two DC and two AC tables in one scan (the entropy kernel keeps two of each in LDS)
            {
                int nd = 0, na = 0, did[3], aid[3];
                for (int k = 0; k < ns; ++k) {
                    int j = 0;
                    for (; j < nd; ++j) if (did[j] == d.ctd[k]) break;
                    if (j == nd) did[nd++] = d.ctd[k];
                    j = 0;
                    for (; j < na; ++j) if (aid[j] == d.cta[k]) break;
                    if (j == na) aid[na++] = d.cta[k];
                }
                if (nd > 2 || na > 2) OG_REFUSE(OG_JPEG_TABLES);
            }
            if (d.ncomp == 3) {
                // libjpeg decides the colour space of a three-component file like this: JFIF -> YCbCr; else an Adobe marker with
                // transform 0 -> RGB (no conversion); else component ids 'R','G','B' -> RGB; else YCbCr.  The device path
                // converts YCbCr: an RGB-coded file is refused and decoded on the host.
                const bool rgb_ids = cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B';
                if (!saw_jfif && ((saw_adobe && adobe_transform == 0) || (!saw_adobe && rgb_ids))) OG_REFUSE(OG_JPEG_COLORSPACE);
            }
            if (d.ncomp == 1) { d.ch[0] = d.cv[0] = 1; }                   // a single-component scan is never interleaved
            d.hmax = d.vmax = 1;
            for (int c = 0; c < d.ncomp; ++c) { d.hmax = d.ch[c] > d.hmax ? d.ch[c] : d.hmax; d.vmax = d.cv[c] > d.vmax ? d.cv[c] : d.vmax; }
            if (d.ncomp == 3) {
                const bool luma_ok = (d.ch[0] == 1 || d.ch[0] == 2) && (d.cv[0] == 1 || d.cv[0] == 2) && !(d.ch[0] == 1 && d.cv[0] == 2);
                if (!luma_ok || d.ch[1] != 1 || d.cv[1] != 1 || d.ch[2] != 1 || d.cv[2] != 1) OG_REFUSE(OG_JPEG_SAMPLING);
            }
            d.mcux = (d.width + 8 * d.hmax - 1) / (8 * d.hmax);
            d.mcuy = (d.height + 8 * d.vmax - 1) / (8 * d.vmax);
            long co = 0, po = 0;
            for (int c = 0; c < d.ncomp; ++c) {
                d.cblk_w[c] = d.mcux * d.ch[c];
                d.cblk_h[c] = d.mcuy * d.cv[c];
                d.ccoef[c] = co;
                d.cplane[c] = po;
                co += (long)d.cblk_w[c] * d.cblk_h[c] * 64;
                po += (long)d.cblk_w[c] * d.cblk_h[c] * 64;
            }
            d.scan_offset = (int)(pos + L);
            d.reason = OG_JPEG_OK;
            return OG_OK;
        }
        pos += L;
    }
#undef OG_REFUSE
    d.reason = OG_JPEG_SCAN;
    return OG_BAD_ARGS;
}

static long og_jpeg_blocks(const JpegDesc& d) {
    long b = 0;
    for (int c = 0; c < d.ncomp; ++c) b += (long)d.cblk_w[c] * d.cblk_h[c];
    return b;
}

// Lay the batch out: file / output offsets from the caller's arrays, workspace slices from the geometry, index slices from
// nsegs (entries of the entropy index handed in per image; NULL or 0: none).
// -> workspace bytes: [int16 coefficients of all images | uint8 planes of all images], or 0 on a bad descriptor.
long objgan_jpeg_plan(void* descs, int n, const long* file_offsets, const long* out_offsets, const int* nsegs) {
    JpegDesc* d = reinterpret_cast<JpegDesc*>(descs);
    long blocks = 0, segs = 0, rows = 0;
    for (int i = 0; i < n; ++i) {
        if (d[i].reason != 0) return 0;
        d[i].file_offset = file_offsets[i];
        d[i].out_offset = out_offsets[i];
        d[i].coef_offset = blocks * 64;
        blocks += og_jpeg_blocks(d[i]);
        d[i].nseg = nsegs ? nsegs[i] : 0;
        if (d[i].nseg < 0 || d[i].nseg > d[i].mcuy) return 0;
        d[i].seg_offset = segs;
        segs += d[i].nseg;
        d[i].idx_offset = rows;
        rows += d[i].mcuy;
    }
    long done = 0;
    for (int i = 0; i < n; ++i) {
        d[i].plane_offset = blocks * 128 + done * 64;      // bytes, behind the coefficient region
        done += og_jpeg_blocks(d[i]);
    }
    return blocks * 128 + blocks * 64;
}

long objgan_jpeg_seg_bytes(void) { return (long)sizeof(JpegSeg); }

// files: the batch's JPEG files back to back (device, every file on a 16-byte boundary, the buffer padded to 16 bytes);
// descs_host / descs_dev: the planned descriptors (the device copy is what the kernels read, the host copy gives the launch
// geometry); out: RGB bytes, image i [height][width][3] at its out_offset; ws: objgan_jpeg_plan's byte count.
// index_in (may be NULL): the entropy-index entries the plan counted (image i: nsegs[i] entries of objgan_jpeg_seg_bytes()
// bytes, back to back) -- a file with an index is decoded by one lane per MCU row.  index_out (may be NULL): receives one
// entry per MCU row of every image (image i at the sum of the MCU rows before it): what to hand in next time.
// Asynchronous on `stream`.
int objgan_jpeg_decode(const unsigned char* files, const void* descs_host, const void* descs_dev, int n,
                       unsigned char* out, void* ws, long ws_bytes, const void* index_in, void* index_out, void* stream) {
    OG_ENTRY();
    if (n <= 0 || !files || !descs_host || !descs_dev || !out || !ws) return OG_BAD_ARGS;
    const JpegDesc* h = reinterpret_cast<const JpegDesc*>(descs_host);
    long blocks = 0, max_blocks = 0, max_quads = 0;
    for (int i = 0; i < n; ++i) {
        if (h[i].reason != 0) return OG_BAD_ARGS;
        if (h[i].nseg > 0 && !index_in) return OG_BAD_ARGS;
        const long b = og_jpeg_blocks(h[i]);
        blocks += b;
        max_blocks = b > max_blocks ? b : max_blocks;
        const long q = (long)((h[i].width + 3) / 4) * h[i].height;
        max_quads = q > max_quads ? q : max_quads;
    }
    if (ws_bytes < blocks * 192 || max_blocks >= (1L << 30) || max_quads >= (1L << 30)) return OG_BAD_ARGS;
    hipStream_t s = (hipStream_t)stream;
    const JpegDesc* dd = reinterpret_cast<const JpegDesc*>(descs_dev);
    short* coef = reinterpret_cast<short*>(ws);
    unsigned char* planes = reinterpret_cast<unsigned char*>(ws);
    if (hipMemsetAsync(coef, 0, (size_t)blocks * 128, s) != hipSuccess) return og_launch_status();
    int max_seg = 1;
    for (int i = 0; i < n; ++i) max_seg = h[i].nseg > max_seg ? h[i].nseg : max_seg;
    hipLaunchKernelGGL(jpeg_entropy_kernel, dim3(n, og_cdiv(max_seg, OG_JPEG_LPW)), dim3(64), 0, s, files, dd, coef,
                       reinterpret_cast<const JpegSeg*>(index_in), reinterpret_cast<JpegSeg*>(index_out));
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3(og_cdiv(max_blocks, 256), n), dim3(256), 0, s, dd, coef, planes, (int)max_blocks);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(og_cdiv(max_quads, 256), n), dim3(256), 0, s, dd, planes, out, (int)max_quads);
    return og_launch_status();
}

}  // extern "C"
