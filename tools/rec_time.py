"""device time of the record pass (objgan_h2_records) and the maxima pass per tensor size: GB/s of the 8 / 4 bytes per element they move"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "obj-gan_amd"))
from objgan_hip import ops, _lib
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for shape in [(16, 194, 128, 128), (16, 96, 256, 256), (16, 194, 64, 64), (16, 388, 64, 64), (16, 192, 35, 35), (16, 768, 17, 17), (16, 1280, 8, 8), (16, 48, 256, 256), (16, 384, 32, 32)]:
    N, C, H, W = shape
    x = torch.randn(shape, device=dev)
    amax = torch.empty(1024, device=dev)
    rec = torch.empty(N * ((C + 15) // 16 * 16) * H * W, device=dev)
    p = lambda a: a.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    ta = t(lambda: _lib.call("objgan_absmax_partials", p(x), x.numel(), p(amax), s))
    tr = t(lambda: _lib.call("objgan_h2_records", p(x), p(amax), p(rec), N, C, H * W, s))
    print("%-22s maxima %7.1f us %6.0f GB/s | records %7.1f us %6.0f GB/s" % (shape, ta, x.numel() * 4 / ta / 1e3, tr, (x.numel() * 4 + rec.numel() * 4) / tr / 1e3))
