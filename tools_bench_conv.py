"""Micro-benchmark of the conv kernels on the hot-path shapes (development aid, GPU only):
python tools_bench_conv.py  -> TFLOP/s (algorithmic) for forward / dgrad / wgrad per shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd")]
import torch
from objgan_hip import ops

SHAPES = [
    # name, N, Cin, H, W, Cout, k, stride, pad, mode, upsample
    ("res1_128 194->388 3x3 refl", 16, 194, 128, 128, 388, 3, 1, 1, "reflect", False),
    ("res2_128 194->194 3x3 refl", 16, 194, 128, 128, 194, 3, 1, 1, "reflect", False),
    ("res1_64  194->388 3x3 refl", 16, 194, 64, 64, 388, 3, 1, 1, "reflect", False),
    ("up_256   194->96 3x3 up", 16, 194, 128, 128, 96, 3, 1, 1, "zeros", True),
    ("hmap_256 80->24 3x3 refl", 16, 80, 256, 256, 24, 3, 1, 1, "reflect", False),
    ("shp_512  80->12 3x3 refl", 16, 80, 512, 512, 12, 3, 1, 1, "reflect", False),
    ("objd_l1  15->96 4x4 s2 @512", 16, 15, 512, 512, 96, 4, 2, 1, "zeros", False),
    ("objd_l2  96->192 4x4 s2 @256", 16, 96, 256, 256, 192, 4, 2, 1, "zeros", False),
    ("objd_l3  192->384 4x4 s2 @128", 16, 192, 128, 128, 384, 4, 2, 1, "zeros", False),
    ("d_l4     384->768 4x4 s2 @32", 16, 384, 32, 32, 768, 4, 2, 1, "zeros", False),
    ("joint    1024->768 3x3 @16", 16, 1024, 16, 16, 768, 3, 1, 1, "zeros", False),
    ("rgb_256  48->3 3x3", 16, 48, 256, 256, 3, 3, 1, 1, "zeros", False),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    dev = torch.device("cuda:0")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, N, Cin, H, W, Cout, k, s, p, mode, up in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(N, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        LH = 2 * H if up else H
        OH = (LH + 2 * p - k) // s + 1
        flops = 2.0 * N * OH * OH * Cout * Cin * k * k
        xr = x.clone().requires_grad_()
        wr = w.clone().requires_grad_()
        y = ops.conv2d(xr, wr, None, s, p, mode, up)
        gy = torch.randn_like(y)
        t_f = timeit(lambda: ops.conv2d(x, w, None, s, p, mode, up))
        xr2 = x.clone().requires_grad_()
        y2 = ops.conv2d(xr2, w, None, s, p, mode, up)
        t_d = timeit(lambda: torch.autograd.grad(y2, xr2, gy, retain_graph=True))
        y3 = ops.conv2d(x, wr, None, s, p, mode, up)
        t_w = timeit(lambda: torch.autograd.grad(y3, wr, gy, retain_graph=True))
        print("%-34s fwd %6.2f ms %6.1f TF | dgrad %6.2f ms %6.1f TF | wgrad %6.2f ms %6.1f TF" % (
            name, t_f * 1e3, flops / t_f / 1e12, t_d * 1e3, flops / t_d / 1e12, t_w * 1e3, flops / t_w / 1e12), flush=True)


if __name__ == "__main__":
    main()
