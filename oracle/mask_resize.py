"""CPU restatement (TEST INFRASTRUCTURE ONLY -- the product never imports oracle/) of the per-box mask resize of the
training loader: reference image_generation/miscc/load.py:160-176 calls `skimage.transform.resize(mask, [s, s])` on every
64 x 64 instance mask for s = 32 (feature scale), 64, 128, 256.

skimage is not installed here (THIRD-PARTY ARITHMETIC): skimage >= 0.19 evaluates `resize` of a 2-D float image with its
defaults (order 1, mode 'reflect', anti_aliasing on when shrinking, clip) as exactly two scipy.ndimage calls -- that is
what `obj-gan_amd/miscc/load.py resize_mask` restates -- and this file restates THOSE two calls operation by operation,
in the order scipy's C code evaluates them (ni_filters.c NI_Correlate1D, symmetric branch; ni_interpolation.c NI_ZoomShift,
order 1, mode 'mirror', grid_mode), in float64 without fused multiply-adds:

    shrink by f = 64 / s > 1:  gaussian_filter(sigma = (f - 1) / 2, truncate 4, mode 'mirror'), axis 0 then axis 1
    zoom:   out[o] <- coordinate c = (o + 0.5) * (64 / s) - 0.5, mirrored into [0, 63], i = floor(c), w = c - i;
            t = 0; t += (x[i][j] * (1 - wy)) * (1 - wx); t += (x[i][j+1] * (1 - wy)) * wx; t += (x[i+1][j] * wy) * (1 - wx); ...
    clip to [min(mask), max(mask)]

Pinned bit for bit against the scipy installed here by tests/test_oracle_cpu.py; the HIP kernel
(csrc/resize_pil.hip mask_resize_kernel) is pinned bit for bit against both on the GPU box.
"""
import numpy as np


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius)[::-1] (symmetric), radius = int(truncate * sigma + 0.5)"""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _mirror(i, n):
    """index extension of mode 'mirror' (d c b | a b c d | c b a), as NI_ExtendLine fills the line buffer"""
    if n == 1:
        return 0
    p = 2 * n - 2
    i = i % p
    return i if i < n else p - i


def correlate1d_symmetric(a, w, axis):
    """NI_Correlate1D, symmetric filter: tmp = x[l] * w[c]; for jj = -c .. -1: tmp += (x[l + jj] + x[l - jj]) * w[jj + c]"""
    a = np.moveaxis(np.asarray(a, np.float64), axis, -1)
    n = a.shape[-1]
    c = len(w) // 2
    out = np.empty_like(a)
    for l in range(n):
        tmp = a[..., l] * w[c]
        for jj in range(-c, 0):
            tmp = tmp + (a[..., _mirror(l + jj, n)] + a[..., _mirror(l - jj, n)]) * w[jj + c]
        out[..., l] = tmp
    return np.moveaxis(out, -1, axis)


def _map_mirror(c, n):
    """map_coordinate(..., NI_EXTEND_MIRROR) of ni_interpolation.c for a coordinate outside [0, n - 1]"""
    if n <= 1:
        return 0.0
    if c < 0:
        sz2 = 2 * n - 2
        c = sz2 * int(-c / sz2) + c
        c = c + sz2 if c <= 1 - n else -c
    elif c > n - 1:
        sz2 = 2 * n - 2
        c -= sz2 * int(c / sz2)
        if c >= n:
            c = sz2 - c
    return c


def zoom_linear(a, size):
    """ndimage.zoom(a, size / n, order=1, mode='mirror', grid_mode=True) for a square 2-D array"""
    a = np.asarray(a, np.float64)
    n = a.shape[0]
    zoom = n / float(size)                       # (grid_mode) input extent / output extent
    idx, wgt = [], []
    for o in range(size):
        c = _map_mirror((o + 0.5) * zoom - 0.5, n)
        i = int(np.floor(c))
        x = c - i
        idx.append((_mirror(i, n), _mirror(i + 1, n)))
        wgt.append((1.0 - x, x))
    out = np.empty((size, size), np.float64)
    for oy in range(size):
        (i0, i1), (wy0, wy1) = idx[oy], wgt[oy]
        for ox in range(size):
            (j0, j1), (wx0, wx1) = idx[ox], wgt[ox]
            t = 0.0
            t += (a[i0, j0] * wy0) * wx0
            t += (a[i0, j1] * wy0) * wx1
            t += (a[i1, j0] * wy1) * wx0
            t += (a[i1, j1] * wy1) * wx1
            out[oy, ox] = t
    return out


def resize_mask(mask, size):
    """-> float64 [size, size]: what obj-gan_amd/miscc/load.py resize_mask (two scipy calls) returns, bit for bit"""
    mask = np.asarray(mask, np.float64)
    n = mask.shape[0]
    assert mask.shape == (n, n)
    if n == size:
        return mask.copy()
    img = mask
    f = n / float(size)
    if f > 1:
        w = gaussian_weights((f - 1) / 2.0)
        img = correlate1d_symmetric(correlate1d_symmetric(img, w, 0), w, 1)
    return np.clip(zoom_linear(img, size), mask.min(), mask.max())
