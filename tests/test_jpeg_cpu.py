"""CPU suite of the JPEG path: the numpy oracle is pinned to Pillow (the reference's decoder,
miscc/load.py:141-151) bit for bit, and the host-only entry points of the C-ABI -- objgan_jpeg_parse / objgan_jpeg_plan, no
GPU call -- agree with Pillow's header reading and refuse what the device path does not decode."""
import io

import numpy as np
import pytest
from PIL import Image

import jpeg_cases


def test_jpeg_oracle_is_bit_exact_with_pillow():
    from oracle import jpeg_oracle as J
    n = 0
    for name, data in jpeg_cases.cases(big=False):
        assert np.array_equal(J.decode(data), jpeg_cases.pillow(data)), name
        n += 1
    assert n > 100
    for name, data, _ in jpeg_cases.refused():
        if name in ("progressive", "cmyk", "adobe transform 0 (RGB-coded)"):
            with pytest.raises(J.Unsupported):
                J.decode(data)


def test_jpeg_parse_is_host_only_and_agrees_with_pillow():
    """objgan_jpeg_parse fills the descriptor without touching a GPU: sizes, component count and sampling equal Pillow's
    reading of the same file; the workspace plan is blocks x (128 + 64) bytes; refused files carry their reason."""
    from objgan_hip import ops, _lib
    files = [d for _, d in jpeg_cases.cases(big=False)]
    descs, heads = ops.jpeg_parse(files)
    assert descs.shape == (len(files), _lib.load().objgan_jpeg_desc_bytes())
    for (name, data), h in zip(jpeg_cases.cases(big=False), heads):
        im = Image.open(io.BytesIO(data))
        assert h[8] == 0, (name, h[8])
        assert (h[0], h[1]) == im.size and h[2] == len(im.getbands()), name
        layer = im.layer                                    # [(id, h, v, tq)]
        assert (h[3], h[4]) == (max(l[1] for l in layer), max(l[2] for l in layer)) or h[2] == 1, name
        assert h[5] == -(-im.size[0] // (8 * h[3])) and h[6] == -(-im.size[1] // (8 * h[4])), name
    import ctypes
    n = len(files)
    foffs = np.arange(n, dtype=np.int64) * 65536
    ooffs = np.arange(n, dtype=np.int64) * (1 << 20)
    ws = _lib.load().objgan_jpeg_plan(descs.ctypes.data_as(ctypes.c_void_p), n, foffs.ctypes.data_as(ctypes.c_void_p),
                                       ooffs.ctypes.data_as(ctypes.c_void_p), None)
    assert _lib.load().objgan_jpeg_seg_bytes() == 48
    blocks = 0
    for h in heads:
        per_mcu = (h[3] * h[4] + 2) if h[2] == 3 else 1
        blocks += int(h[5]) * int(h[6]) * per_mcu
    assert ws == blocks * 192
    for name, data, reason in jpeg_cases.refused():
        _, hd = ops.jpeg_parse([data])
        assert hd[0, 8] == reason, (name, hd[0, 8])
    with pytest.raises(_lib.ObjganHipError):
        ops.jpeg_decode_batch(files[:1], "cpu")             # no CPU path
