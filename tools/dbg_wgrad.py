"""debug: where does the bf16 blocked weight gradient put a single product?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "obj-gan_amd"))
import torch
from objgan_hip import ops
dev = torch.device("cuda:0")
ops.set_conv_math("bf16")
def run(N, Cin, Cout, H, W, k, s, p, probes):
    OH = (H + 2 * p - k) // s + 1; OW = (W + 2 * p - k) // s + 1
    for (n, c, h, w, m, oh, ow) in probes:
        x = torch.zeros(N, Cin, H, W, device=dev); x[n, c, h, w] = 1.0
        g = torch.zeros(N, Cout, OH, OW, device=dev); g[n, m, oh, ow] = 1.0
        dw = ops._conv_wgrad(x, g, Cout, k, s, p, 0, False)
        torch.cuda.synchronize()
        nz = torch.nonzero(dw).tolist()
        kh, kw = h - oh * s + p, w - ow * s + p
        exp = [[m, c, kh, kw]] if 0 <= kh < k and 0 <= kw < k else []
        print("probe x[n=%d,c=%d,h=%d,w=%d] dy[m=%d,oh=%d,ow=%d]  expect %s  got %s" % (n, c, h, w, m, oh, ow, exp, nz[:6]), "OK" if nz == exp else "MISMATCH")
run(1, 32, 32, 8, 8, 3, 1, 1, [(0, 0, 0, 0, 0, 0, 0), (0, 5, 3, 4, 7, 3, 4), (0, 5, 3, 4, 7, 2, 4), (0, 17, 6, 1, 20, 6, 2), (0, 31, 7, 7, 31, 7, 7), (0, 9, 4, 4, 3, 5, 5)])
run(2, 40, 64, 8, 8, 3, 1, 1, [(1, 33, 2, 2, 40, 2, 2), (1, 39, 5, 6, 63, 4, 6), (0, 16, 0, 7, 1, 0, 6)])
