"""JPEG decode on the device (csrc/jpeg.hip through the C-ABI) against Pillow -- the reference's decoder
(PIL.Image.open(...).convert('RGB'), reference miscc/load.py:141-151) -- and against the numpy oracle: BIT-EXACT, every
byte, for every sampling / size / encoder option of tests/jpeg_cases.py, singly and as one mixed batch, through the
dataset hand-over (TrainDataset(device_jpeg=True) -> prepare_data) and at COCO size."""
import os

import numpy as np
import pytest
import torch

import jpeg_cases
from conftest import ROOT, note

pytestmark = pytest.mark.gpu
DATA = os.path.join(ROOT, "tests", "golden", "data_tiny")


def test_device_jpeg_decode_is_bit_exact_with_pillow(dev):
    from objgan_hip import ops
    from oracle import jpeg_oracle as J
    cases = jpeg_cases.cases(big=True)
    files = [d for _, d in cases]
    # (1) one batch with everything in it: mixed sizes, samplings, table sets
    outs = ops.jpeg_decode(files, dev)
    torch.cuda.synchronize()
    bad = [name for (name, data), o in zip(cases, outs) if not np.array_equal(o.cpu().numpy(), jpeg_cases.pillow(data))]
    assert not bad, bad[:8]
    # (2) singly (batch of one: other launch geometry), small ones also against the oracle
    for name, data in cases[::7]:
        (o,) = ops.jpeg_decode([data], dev)
        got = o.cpu().numpy()
        assert np.array_equal(got, jpeg_cases.pillow(data)), name
        if got.size < 100000:
            assert np.array_equal(got, J.decode(data)), name
    note("device JPEG decode vs Pillow: %d files (4:4:4 / 4:2:2 / 4:2:0 / grey, 1x1 .. 480x640, restart intervals, "
         "optimised tables)" % len(files), "bit-exact")


def test_device_jpeg_entropy_index_gives_the_same_bytes(dev):
    """The first decode of a file (one lane walks the scan) leaves its entropy index -- the decoder state at every MCU-row
    start -- in the cache on the device; the second decode runs one lane per MCU row from that index.  Same state machine
    resumed from the same state: bit-identical output, for every sampling, restart intervals included; a batch that mixes
    indexed and new files works; a stale entry (other file under the same key) is not used."""
    from objgan_hip import ops
    cases = jpeg_cases.cases(big=True)
    files = [d for _, d in cases]
    keys = ["k%d" % i for i in range(len(files))]
    cache = ops.JpegIndexCache()
    first = [o.clone() for o in ops.jpeg_decode(files, dev, cache, keys)]
    assert cache.misses == len(files) and cache.hits == 0 and len(cache.entries) == len(files)
    second = ops.jpeg_decode(files, dev, cache, keys)
    torch.cuda.synchronize()
    assert cache.hits == len(files)
    bad = [name for (name, data), a, b in zip(cases, first, second)
           if not (torch.equal(a, b) and np.array_equal(b.cpu().numpy(), jpeg_cases.pillow(data)))]
    assert not bad, bad[:8]
    # mixed batch: half of the files come with their index, the others are new keys
    keys2 = [k if i % 2 else "new" + k for i, k in enumerate(keys)]
    third = ops.jpeg_decode(files, dev, cache, keys2)
    assert all(torch.equal(a, b) for a, b in zip(first, third))
    # a different file under a cached key: size differs -> the entry is not used (and is replaced)
    swapped = ops.jpeg_decode([files[3]], dev, cache, [keys[10]])
    assert np.array_equal(swapped[0].cpu().numpy(), jpeg_cases.pillow(files[3]))


def test_device_jpeg_refuses_what_it_does_not_decode(dev):
    from objgan_hip import ops
    good = jpeg_cases.cases(big=False)[0][1]
    for name, data, reason in jpeg_cases.refused():
        with pytest.raises(ops.JpegUnsupported) as e:
            ops.jpeg_decode([good, data], dev)
        assert e.value.index == 1 and e.value.reason == reason, name


def test_device_jpeg_ring_refill_and_long_scans(dev):
    """The entropy kernel stages the file through an 8 KB LDS ring refilled in 4 KB halves: files whose scan is many
    windows long, whose scan starts at an unaligned offset (a long COM segment in front), and noise images whose MCUs
    are hundreds of bytes each must decode bit-exactly."""
    import io
    from PIL import Image
    from objgan_hip import ops
    rng = np.random.RandomState(7)
    files = []
    for (h, w, q, sub) in [(256, 256, 98, 0), (200, 333, 100, 2), (123, 457, 95, 1), (512, 512, 85, 2)]:
        a = (rng.rand(h, w, 3) * 255).astype(np.uint8)              # noise: the longest codes, the most bytes per MCU
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, "JPEG", quality=q, subsampling=sub, comment=b"x" * int(rng.randint(1, 5000)))
        files.append(buf.getvalue())
    assert max(len(f) for f in files) > 100000
    outs = ops.jpeg_decode(files, dev)
    for f, o in zip(files, outs):
        assert np.array_equal(o.cpu().numpy(), jpeg_cases.pillow(f))


def test_jpeg_file_handover_through_the_dataset(dev):
    """TrainDataset(device_jpeg=True): the items carry the JPEG file, prepare_data decodes and resizes on the device --
    bit for bit the image tensors of the reference-style host path (PIL decode + PIL resize + ToTensor + Normalize)."""
    import trainDataset
    from miscc.config import cfg
    from torch.utils.data.dataloader import default_collate
    cfg.TREE.BRANCH_NUM = 3
    host = trainDataset.TrainDataset(DATA, "train", base_size=64)
    lean = trainDataset.TrainDataset(DATA, "train", base_size=64, device_jpeg=True, device_masks=True)
    np.random.seed(11)
    want = trainDataset.prepare_data(default_collate([host[i] for i in range(len(host))]))
    np.random.seed(11)
    got = trainDataset.prepare_data(trainDataset.collate_keep_images([lean[i] for i in range(len(lean))]), dev,
                                    num_classes=host.num_classes)
    torch.cuda.synchronize()
    assert got[11] == want[11] and lean.host_decoded == 0
    for b in range(3):
        assert torch.equal(got[0][b].cpu(), want[0][b]), b


def test_device_jpeg_throughput_record(dev):
    """Not a parity test: the record of what the decode costs -- a batch of 16 COCO-sized 4:2:0 files (the bench batch),
    decoded on the device, timed with HIP events on the launch stream; it runs beside the ~130 ms training step."""
    import io
    from PIL import Image
    from objgan_hip import ops
    rng = np.random.RandomState(5)
    files = []
    for i in range(16):
        pic = jpeg_cases._picture(rng, 480, 640, noise=14.0)
        buf = io.BytesIO()
        pic.save(buf, "JPEG", quality=90, subsampling=2)
        files.append(buf.getvalue())
    cache, keys = ops.JpegIndexCache(), list(range(16))
    outs = ops.jpeg_decode(files, dev)                      # warm-up
    torch.cuda.synchronize()
    assert np.array_equal(outs[3].cpu().numpy(), jpeg_cases.pillow(files[3]))

    def timed(reps, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.jpeg_decode(files, dev, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timed(2)                                           # first touch: one lane per file
    ops.jpeg_decode(files, dev, cache=cache, keys=keys)     # (leaves the indexes)
    ms_idx = timed(3, cache=cache, keys=keys)               # every later epoch: one lane per MCU row
    assert cache.hits == 16 * 3
    import time
    t0 = time.perf_counter()
    for f in files:
        jpeg_cases.pillow(f)
    host_ms = 1000.0 * (time.perf_counter() - t0)
    note("device JPEG decode, 16 files of 480x640 4:2:0 q90 (%.0f KB each): ms per batch first touch (one lane per file) / "
         "with the entropy index (one lane per MCU row) / Pillow on one host core" % (sum(len(f) for f in files) / 16e3),
         "%.2f / %.2f / %.1f" % (ms, ms_idx, host_ms))
    assert ms < 400.0 and ms_idx < ms
