"""ORACLE (test infrastructure): baseline JPEG decode restated in numpy, bit for bit what Pillow returns.

The reference loader decodes every training image with `PIL.Image.open(...).convert('RGB')` (reference
image_generation/miscc/load.py:141-151, get_imgs).  Pillow's decoder is libjpeg-turbo in its libjpeg-6b configuration
(PIL.features: jpg 6.2, libjpeg_turbo) with the library defaults: dct_method JDCT_ISLOW, do_fancy_upsampling on, no
merged upsampling, output colour space RGB.  libjpeg-turbo is a third-party dependency that is NOT in /root/reference
(and not vendored here); its published algorithm for this path is restated below, stage by stage, with the source file
of libjpeg(-turbo) each stage follows:

  * marker parsing, Huffman tables (JPEG Annex C / F; jdmarker.c, jdhuff.c `jpeg_make_d_derived_tbl`)
  * sequential Huffman entropy decode with DC prediction, EOB / ZRL, restart intervals (jdhuff.c `decode_mcu`)
  * de-quantisation + the accurate integer inverse DCT (jidctint.c `jpeg_idct_islow`, CONST_BITS 13, PASS1_BITS 2)
  * "fancy" triangle-filter chroma upsampling (jdsample.c `h2v1_fancy_upsample`, `h2v2_fancy_upsample`; edge rows /
    columns replicated as jdmainct.c's context rows do)
  * YCbCr -> RGB with the 16-bit fixed-point tables (jdcolor.c `build_ycc_rgb_table`, `ycc_rgb_convert`)

Pinned: tests/test_jpeg_cpu.py compares this module with Pillow itself on encoded test images (4:4:4, 4:2:2, 4:2:0,
grey, odd sizes, restart intervals, optimised tables) -- bit-exact.  The product (csrc/jpeg.hip) is compared with Pillow
and with this module in tests/test_jpeg_gpu.py.  Only tests import this file.
"""
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                   14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39,
                   46, 53, 60, 61, 54, 47, 55, 62, 63], np.int32)


class Unsupported(ValueError):
    """not a baseline sequential 8-bit Huffman JPEG with 1 or 3 components in the supported samplings"""


def parse(data):
    """-> dict(width, height, comps=[(id, h, v, tq)], qt={id: [64] natural order}, dc/ac = {id: (bits[16], vals)},
    scan=(component order [(comp index, td, ta)]), restart_interval, scan_offset, scan_end)"""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("no SOI")
    pos = 2
    out = {"qt": {}, "dc": {}, "ac": {}, "restart_interval": 0}
    saw_jfif = saw_adobe = False
    adobe_transform = 1
    n = len(data)
    while pos < n:
        if data[pos] != 0xFF:
            raise Unsupported("marker expected at %d" % pos)
        while pos < n and data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m in (0x01,) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        L = struct.unpack(">H", data[pos:pos + 2])[0]
        seg = data[pos + 2:pos + L]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    raise Unsupported("16-bit quantisation table")
                q = np.zeros(64, np.int32)
                q[ZIGZAG] = np.frombuffer(seg[i:i + 64], np.uint8)
                out["qt"][tq] = q
                i += 64
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = list(seg[i + 1:i + 17])
                cnt = sum(bits)
                vals = list(seg[i + 17:i + 17 + cnt])
                out["ac" if tc else "dc"][th] = (bits, vals)
                i += 17 + cnt
        elif m == 0xC0 or m == 0xC1:
            if seg[0] != 8:
                raise Unsupported("precision %d" % seg[0])
            out["height"], out["width"] = struct.unpack(">HH", seg[1:5])
            nc = seg[5]
            out["comps"] = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(nc)]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("SOF%d (progressive / lossless / arithmetic)" % (m - 0xC0))
        elif m == 0xE0 and seg[:5] == b"JFIF\x00":
            saw_jfif = True
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            saw_adobe, adobe_transform = True, seg[11]
        elif m == 0xDD:
            out["restart_interval"] = struct.unpack(">H", seg[:2])[0]
        elif m == 0xDA:
            ns = seg[0]
            ids = [c[0] for c in out["comps"]]
            out["scan"] = [(ids.index(seg[1 + 2 * k]), seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(ns)]
            if ns != len(out["comps"]):
                raise Unsupported("non-interleaved scans")
            # libjpeg's colour-space guess (jdapimin.c default_decompress_parms): JFIF -> YCbCr; else Adobe transform 0 -> RGB;
            # else component ids 'R', 'G', 'B' -> RGB; else YCbCr.  RGB-coded files are outside this restatement.
            if len(ids) == 3 and not saw_jfif and ((saw_adobe and adobe_transform == 0) or (not saw_adobe and ids == [82, 71, 66])):
                raise Unsupported("RGB-coded file (no colour transform)")
            out["scan_offset"] = pos + L
            return out
        pos += L
    raise Unsupported("no SOS")


def derive(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: -> (mincode[17], maxcode[18], valptr[17]) of the canonical code"""
    code, k = 0, 0
    mincode, maxcode, valptr = [0] * 17, [-1] * 18, [0] * 17
    for l in range(1, 17):
        if bits[l - 1]:
            valptr[l] = k
            mincode[l] = code
            code += bits[l - 1]
            k += bits[l - 1]
            maxcode[l] = code - 1
        code <<= 1
    maxcode[17] = 0xFFFFF
    return mincode, maxcode, valptr


class _Bits(object):
    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0
        self.hit_marker = False

    def _fill(self):
        while self.n <= 24:
            if self.hit_marker or self.p >= len(self.d):
                b = 0
            else:
                b = self.d[self.p]
                if b == 0xFF:
                    nb = self.d[self.p + 1] if self.p + 1 < len(self.d) else 0xD9
                    if nb == 0:
                        self.p += 2
                    else:                           # a marker: feed zeros (jdhuff.c jpeg_fill_bit_buffer, no more bytes)
                        self.hit_marker = True
                        b = 0
                else:
                    self.p += 1
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def huff(self, tbl, vals):
        mincode, maxcode, valptr = tbl
        code = 0
        for l in range(1, 17):
            code = (code << 1) | self.get(1)
            if code <= maxcode[l]:
                return vals[valptr[l] + code - mincode[l]]
        return 0

    def restart(self):
        """byte-align, skip the RSTn marker"""
        self.acc = self.n = 0
        self.hit_marker = False
        while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        self.p += 2


def _extend(v, s):
    return v - ((1 << s) - 1) if s and v < (1 << (s - 1)) else v


def decode_coefficients(data, hdr):
    """-> per component int16 [blocks_h, blocks_w, 64] quantised coefficients in natural order"""
    comps = hdr["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux = -(-hdr["width"] // (8 * hmax))
    mcuy = -(-hdr["height"] // (8 * vmax))
    coef = [np.zeros((mcuy * c[2], mcux * c[1], 64), np.int16) for c in comps]
    dct = {k: derive(*v) for k, v in hdr["dc"].items()}
    act = {k: derive(*v) for k, v in hdr["ac"].items()}
    br = _Bits(data, hdr["scan_offset"])
    pred = [0] * len(comps)
    ri = hdr["restart_interval"]
    todo = ri
    for my in range(mcuy):
        for mx in range(mcux):
            if ri and todo == 0:
                br.restart()
                pred = [0] * len(comps)
                todo = ri
            for ci, td, ta in hdr["scan"]:
                _, h, v, _ = comps[ci]
                for by in range(v):
                    for bx in range(h):
                        blk = coef[ci][my * v + by, mx * h + bx]
                        s = br.huff(dct[td], hdr["dc"][td][1])
                        pred[ci] += _extend(br.get(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.huff(act[ta], hdr["ac"][ta][1])
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                if k > 63:
                                    break
                                blk[ZIGZAG[k]] = _extend(br.get(s), s)
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
            todo -= 1
    return coef


# jidctint.c constants (CONST_BITS = 13)
_F = {"0.298631336": 2446, "0.390180644": 3196, "0.541196100": 4433, "0.765366865": 6270, "0.899976223": 7373,
      "1.175875602": 9633, "1.501321110": 12299, "1.847759065": 15137, "1.961570560": 16069, "2.053119869": 16819,
      "2.562915447": 20995, "3.072711026": 25172}


def _idct_1d(d, shift):
    """one pass of jpeg_idct_islow over the LAST axis being the 8 inputs: d [..., 8] int64 -> [..., 8]"""
    d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
    z2, z3 = d2, d6
    z1 = (z2 + z3) * _F["0.541196100"]
    tmp2 = z1 + z3 * (-_F["1.847759065"])
    tmp3 = z1 + z2 * _F["0.765366865"]
    tmp0 = (d0 + d4) << 13
    tmp1 = (d0 - d4) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    t0, t1, t2, t3 = d7, d5, d3, d1
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * _F["1.175875602"]
    t0 = t0 * _F["0.298631336"]
    t1 = t1 * _F["2.053119869"]
    t2 = t2 * _F["3.072711026"]
    t3 = t3 * _F["1.501321110"]
    z1 = z1 * (-_F["0.899976223"])
    z2 = z2 * (-_F["2.562915447"])
    z3 = z3 * (-_F["1.961570560"]) + z5
    z4 = z4 * (-_F["0.390180644"]) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    rnd = 1 << (shift - 1)
    out = [tmp10 + t3, tmp11 + t2, tmp12 + t1, tmp13 + t0, tmp13 - t0, tmp12 - t1, tmp11 - t2, tmp10 - t3]
    return np.stack([(o + rnd) >> shift for o in out], -1)


def idct_islow(coef, qt):
    """coef [..., 64] quantised, natural order -> uint8 [..., 8, 8] samples (jidctint.c jpeg_idct_islow)"""
    x = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))
    ws = _idct_1d(np.swapaxes(x, -1, -2), 13 - 2)             # pass 1: columns (last axis = the 8 rows of a column)
    ws = np.swapaxes(ws, -1, -2)
    y = _idct_1d(ws, 13 + 2 + 3)                                # pass 2: rows
    return np.clip(y + 128, 0, 255).astype(np.uint8)


def planes(coef, hdr):
    """-> per component uint8 [blocks_h * 8, blocks_w * 8]"""
    out = []
    for c, (_, h, v, tq) in zip(coef, hdr["comps"]):
        s = idct_islow(c, hdr["qt"][tq])                        # [bh, bw, 8, 8]
        out.append(s.transpose(0, 2, 1, 3).reshape(c.shape[0] * 8, c.shape[1] * 8))
    return out


def _h2_fancy_rows(p, out_w):
    """jdsample.c h2v1_fancy_upsample on every row of p [rows, w] (w = downsampled width) -> [rows, 2 w][:out_w]"""
    p = p.astype(np.int32)
    w = p.shape[1]
    out = np.empty((p.shape[0], 2 * w), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (p * 3 + left + 1) >> 2
    out[:, 1::2] = (p * 3 + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out[:, :out_w].astype(np.uint8)


def _h2v2_fancy(p, out_h, out_w):
    """jdsample.c h2v2_fancy_upsample: p [h, w] (true downsampled size; edge rows replicated as context) -> [2 h, 2 w]"""
    p = p.astype(np.int32)
    h, w = p.shape
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, far in ((0, above), (1, below)):
        cs = p * 3 + far                                        # column sums of the two contributing rows
        last = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        e = (cs * 3 + last + 8) >> 4
        o = (cs * 3 + nxt + 7) >> 4
        e[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        o[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = e
        out[v::2, 1::2] = o
    return out[:out_h, :out_w].astype(np.uint8)


def _color_tables():
    x = np.arange(256, dtype=np.int64) - 128
    half = 1 << 15

    def fix(v):
        return int(v * 65536 + 0.5)
    return ((fix(1.40200) * x + half) >> 16, (fix(1.77200) * x + half) >> 16, -fix(0.71414) * x, -fix(0.34414) * x + half)


def decode(data):
    """JPEG file bytes -> uint8 [H, W, 3] RGB, what PIL.Image.open(...).convert('RGB') returns"""
    data = bytes(data)
    hdr = parse(data)
    W, H = hdr["width"], hdr["height"]
    comps = hdr["comps"]
    pl = planes(decode_coefficients(data, hdr), hdr)
    if len(comps) == 1:
        y = pl[0][:H, :W]
        return np.stack([y, y, y], -1)
    if len(comps) != 3:
        raise Unsupported("%d components" % len(comps))
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    full = []
    for p, (_, h, v, _) in zip(pl, comps):
        dw, dh = -(-W * h // hmax), -(-H * v // vmax)           # true downsampled size (jdmaster.c)
        p = p[:dh, :dw]
        if (h, v) == (hmax, vmax):
            full.append(p[:H, :W])
        elif 2 * h == hmax and v in (vmax, vmax // 2) and dw <= 2:
            # jdsample.c jinit_upsampler: the fancy filters are selected only when downsampled_width > 2; narrower
            # components take plain replication (h2v1_upsample / h2v2_upsample)
            full.append(np.repeat(np.repeat(p, 2, 1), vmax // v, 0)[:H, :W])
        elif 2 * h == hmax and v == vmax:
            full.append(_h2_fancy_rows(p, W)[:H])
        elif 2 * h == hmax and 2 * v == vmax:
            full.append(_h2v2_fancy(p, H, W))
        else:
            raise Unsupported("sampling %dx%d of %dx%d" % (h, v, hmax, vmax))
    y, cb, cr = (f.astype(np.int64) for f in full)
    cr_r, cb_b, cr_g, cb_g = _color_tables()
    r = y + cr_r[cr]
    g = y + ((cb_g[cb] + cr_g[cr]) >> 16)
    b = y + cb_b[cb]
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)
