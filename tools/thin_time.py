"""device time of the four-phase thin data gradient (objgan_conv_dgrad_s2_thin) on the discriminator stems.  OG_THIN_VARIANT selected
kernel variants in a development build (LAB 10.10: none faster; the library ignores it now); a run with the variable set compares its
output bit for bit with the tensor a variant-0 run saved."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "obj-gan_amd"))
from objgan_hip import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
var = os.environ.get("OG_THIN_VARIANT", "0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(5)
for (N, Cout, OH, Cin) in [(16, 96, 128, 12), (16, 96, 64, 12), (16, 96, 32, 12), (16, 96, 128, 3), (16, 96, 64, 3)]:
    dy = torch.randn(N, Cout, OH, OH, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05).to(dev)
    dx = torch.empty(N, Cin, 2 * OH, 2 * OH, device=dev)
    nf = lib.objgan_conv_dgrad_s2_thin_floats(Cout, Cin)
    wt = torch.zeros(nf, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    rc = lib.objgan_conv_dgrad_s2_thin(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), wt.data_ptr(), N, Cout, OH, OH, Cin, 0, s)
    assert rc == 1, rc        # (reference-style return code: 1 = ok)
    us = t(lambda: lib.objgan_conv_dgrad_s2_thin(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), wt.data_ptr(), N, Cout, OH, OH, Cin, 1, s))
    mb = (dy.numel() + dx.numel()) * 4 / 1e6
    fl = 2.0 * Cin * Cout * 4 * N * OH * OH * 4
    tag = "/tmp/thin_%d_%d_%d.pt" % (Cout, OH, Cin)
    same = ""
    if var == "0":
        torch.save(dx.cpu(), tag)
    elif os.path.exists(tag):
        same = "bit-identical to variant 0: %s" % torch.equal(torch.load(tag), dx.cpu())
    print("variant %s  dy %dx%dx%dx%d -> %d ch   %7.1f us  %6.0f GB/s algorithmic  %5.1f TFLOP/s  %s" % (var, N, Cout, OH, OH, Cin, us, mb / us * 1e3 / 1e3, fl / us / 1e6, same))
