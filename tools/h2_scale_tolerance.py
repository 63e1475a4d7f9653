"""How tight must the tensor scale of fp16x2 be?  CPU emulation (numpy float16 keeps subnormals, like the fp16 MFMA --
profiles/r04_fp16_subnormal_probe.txt): x * 2^s = h + l, w * 2^t = h' + l', dot products from the three partial
products hh', hl', lh' accumulated in fp32, against fp64 -- with the activation scale chosen k bits BELOW the optimum
(max|x| * 2^s in [2^(14-k), 2^(15-k))), i.e. from an upper bound of the maximum that is 2^k too large.
    python tools/h2_scale_tolerance.py
"""
import numpy as np


def split(v, scale):
    sv = (v * scale).astype(np.float32)
    h = sv.astype(np.float16)
    l = (sv - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def run(K=4608, M=64, N=256, seed=0, heavy_tail=False):
    rng = np.random.RandomState(seed)
    x = rng.standard_normal((K, N)).astype(np.float32)
    if heavy_tail:                       # a few huge activations set the maximum, the bulk sits 2^-8 below
        x *= 2.0 ** -8
        x[rng.randint(0, K, 16), rng.randint(0, N, 16)] = 1.0
    w = (rng.standard_normal((M, K)) * 0.02).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    f32 = (w @ x)                        # plain fp32 evaluation (numpy / BLAS order) for scale
    e32 = np.linalg.norm(f32 - ref) / np.linalg.norm(ref)
    tw = 2.0 ** (14 - np.floor(np.log2(np.abs(w).max())))
    wh, wl = split(w, tw)
    rows = []
    for k in (0, 2, 4, 6, 8, 10, 12, 14):
        sx = 2.0 ** (14 - k - np.floor(np.log2(np.abs(x).max())))
        xh, xl = split(x, sx)
        acc = (wh @ xh).astype(np.float32) + (wh @ xl).astype(np.float32) + (wl @ xh).astype(np.float32)
        y = acc.astype(np.float64) / (sx * tw)
        rows.append((k, np.linalg.norm(y - ref) / np.linalg.norm(ref)))
    return e32, rows


if __name__ == "__main__":
    for name, ht in (("gaussian activations", False), ("bulk 2^-8 below a few maxima", True)):
        e32, rows = run(heavy_tail=ht)
        print("%s: K = 4608, fp32 evaluation vs fp64: %.2e" % (name, e32))
        for k, e in rows:
            print("   scale %2d bits below the optimum: fp16x2 vs fp64 %.2e" % (k, e))
