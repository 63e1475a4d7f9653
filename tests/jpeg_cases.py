"""Encoded test images shared by tests/test_jpeg_cpu.py and tests/test_jpeg_gpu.py: (name, JPEG bytes) pairs made with
Pillow's own encoder at run time (seeded content, so every box sees the same files): the samplings, qualities, sizes and
encoder options the decoder has to get bit-exact -- 4:4:4 / 4:2:2 / 4:2:0, grey, odd and tiny sizes (components narrower
than three samples take libjpeg's replication path), restart intervals, optimised Huffman tables (codes of up to 16 bits),
quality 100 (long codes, large coefficients), a COCO-sized photograph-like image."""
import io

import numpy as np
from PIL import Image


def _picture(rng, h, w, noise=6.0):
    a = rng.rand(h // 4 + 2, w // 4 + 2, 3)
    im = Image.fromarray((a * 255).astype(np.uint8)).resize((w, h), Image.BICUBIC)
    a = np.asarray(im).astype(np.float32) + rng.randn(h, w, 3) * noise
    return Image.fromarray(np.clip(a, 0, 255).astype(np.uint8))


def _enc(im, **kw):
    buf = io.BytesIO()
    im.save(buf, "JPEG", **kw)
    return buf.getvalue()


def cases(big=True):
    rng = np.random.RandomState(0)
    out = []
    for (h, w) in [(16, 16), (37, 48), (64, 64), (50, 30), (90, 71), (33, 17), (8, 8), (1, 1), (129, 255)]:
        for sub in (0, 1, 2):
            for q in (30, 75, 95):
                out.append(("%dx%d sub%d q%d" % (h, w, sub, q), _enc(_picture(rng, h, w), quality=q, subsampling=sub)))
    for (h, w) in [(1, 2), (2, 2), (3, 2), (16, 2), (9, 3), (9, 4), (9, 5), (9, 6), (40, 7), (3, 40)]:
        for sub in (1, 2):
            a = (rng.rand(h, w, 3) * 255).astype(np.uint8)
            out.append(("noise %dx%d sub%d" % (h, w, sub), _enc(Image.fromarray(a), quality=60, subsampling=sub)))
    pic = _picture(rng, 60, 80)
    out.append(("optimised tables", _enc(pic, quality=80, subsampling=2, optimize=True)))
    out.append(("restart every 3 blocks", _enc(pic, quality=80, subsampling=2, restart_marker_blocks=3)))
    out.append(("restart every row", _enc(_picture(rng, 61, 83), quality=70, subsampling=1, restart_marker_rows=1)))
    out.append(("quality 100", _enc(_picture(rng, 40, 40, noise=40.0), quality=100, subsampling=0)))
    out.append(("quality 1", _enc(_picture(rng, 72, 56), quality=1, subsampling=2)))
    out.append(("grey", _enc(_picture(rng, 45, 52).convert("L"), quality=85)))
    out.append(("grey optimised", _enc(_picture(rng, 31, 9).convert("L"), quality=40, optimize=True)))
    if big:
        out.append(("coco-size 480x640 4:2:0", _enc(_picture(rng, 480, 640, noise=12.0), quality=90, subsampling=2)))
        out.append(("coco-size 427x640 4:2:0 optimised", _enc(_picture(rng, 427, 640, noise=20.0), quality=75, subsampling=2,
                                                              optimize=True)))
    return out


def refused():
    """(name, bytes, reason code of objgan_jpeg_parse)"""
    rng = np.random.RandomState(3)
    pic = _picture(rng, 40, 56)
    plain = _enc(pic, quality=80)
    # the same file with its JFIF marker replaced by an Adobe marker that says "no colour transform": libjpeg (and Pillow)
    # then read the three components as R, G, B -- the device path converts YCbCr and must refuse the file
    i = plain.index(b"\xff\xe0")
    n = (plain[i + 2] << 8) | plain[i + 3]
    adobe = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    rgb_coded = plain[:i] + adobe + plain[i + 2 + n:]
    return [("progressive", _enc(pic, quality=80, progressive=True), 2),
            ("adobe transform 0 (RGB-coded)", rgb_coded, 9),
            ("cmyk", _enc(pic.convert("CMYK"), quality=80), 4),
            ("png bytes", b"\x89PNG\r\n\x1a\n" + bytes(64), 1),
            ("truncated header", _enc(pic, quality=80)[:40], 8)]


def pillow(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
