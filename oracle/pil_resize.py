"""TEST INFRASTRUCTURE (oracle): numpy restatement of Pillow's antialiased bilinear resize for 8-bit RGB images.

The reference resizes every training image to the branch sizes with `transforms.Resize((s, s))(img)` on a PIL
image and normalises it with ToTensor + Normalize(0.5, 0.5) (reference image_generation/miscc/load.py:141-150,
trainDataset.py:60-66).  `transforms.Resize` on a PIL image is `Image.resize(size, BILINEAR)`, i.e. the
third-party library Pillow -- not vendored in the reference, no version pinned by it (its README asks for
"torchvision"); the algorithm restated here is `ImagingResample` of Pillow's `src/libImaging/Resample.c`
(unchanged since Pillow 3.4 for 8-bit images): separable two-pass convolution, horizontal first, triangle
filter stretched by the down-scaling factor (support = max(scale, 1)), coefficients normalised in double,
rounded to 22-bit fixed point, accumulated in int32 with a rounding offset, clipped to uint8 after EACH pass.

Pinned: tests/test_data_cpu.py compares this restatement bit for bit with the Pillow installed in the build
container (12.2.0) on random images (down-, up-scaling, identity, non-square); the gfx950 kernels
(obj-gan_amd/csrc/resize_pil.hip) are compared bit for bit with this file."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coefficients(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the box (0, in_size) and the bilinear
    filter (support 1.0) -> (ksize, bounds [out, 2] int32 (first source index, count), kk [out, ksize] int32)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size)
        n = xmax - xmin
        ww = 0.0
        for x in range(n):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            for x in range(n):
                kk[xx, x] /= ww
        bounds[xx] = (xmin, n)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS))
    return ksize, bounds, np.trunc(fixed).astype(np.int64).astype(np.int32)


def _pass(img, out_size, axis):
    """One 8-bit pass along `axis` of an [H, W, C] uint8 image."""
    in_size = img.shape[axis]
    _, bounds, kk = coefficients(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_rgb8(img, size):
    """Image.resize((size, size), BILINEAR) of an [H, W, 3] uint8 array.  A pass whose size does not change is
    skipped (ImagingResample: need_horizontal / need_vertical)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    if W != size:
        img = _pass(img, size, 1)
    if H != size:
        img = _pass(img, size, 0)
    return img


def to_normalized_chw(img_u8):
    """ToTensor() + Normalize((.5, .5, .5), (.5, .5, .5)) in float32, as torch evaluates it."""
    a = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    return ((a - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)
