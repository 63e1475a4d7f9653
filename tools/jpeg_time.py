"""timing of the device loader path of bench.py --jpeg-input: decode from the entropy index, PIL-bilinear resizes (ms per batch of 16)"""
import io, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "obj-gan_amd"))
from objgan_hip import ops
from PIL import Image
dev = torch.device("cuda:0")
rng = np.random.RandomState(77)
files = []
for i in range(16):
    a = rng.rand(122, 162, 3)
    im = Image.fromarray((a * 255).astype(np.uint8)).resize((640, 480), Image.BICUBIC)
    a = np.clip(np.asarray(im).astype(np.float32) + rng.randn(480, 640, 3) * 14.0, 0, 255).astype(np.uint8)
    buf = io.BytesIO(); Image.fromarray(a).save(buf, "JPEG", quality=90, subsampling=2); files.append(buf.getvalue())
cache = ops.JpegIndexCache()
keys = [(0, j) for j in range(16)]
src, offs, hs, ws = ops.jpeg_decode_batch(files, dev, cache, keys)
sizes = [64, 128, 256]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    h0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); h1 = time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (h1 - h0) * 1e3 / n
print("decode (indexed)  device %.2f ms  host %.2f ms" % t(lambda: ops.jpeg_decode_batch(files, dev, cache, keys)))
print("resize 3 sizes    device %.2f ms  host %.2f ms" % t(lambda: ops.resize_pil_bilinear_device(src, offs, hs, ws, sizes)))
