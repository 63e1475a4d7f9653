"""Synthetic COCO-shaped minibatch for benchmarks and parity tests.

Shapes and value ranges follow what the reference's `prepare_data` hands to the training step
(reference image_generation/trainDataset.py:79-127, miscc/load.py:141-191, 517-650; contract
tabulated in SURVEY.md section 8d): three image scales in [-1, 1], 80-channel layout maps built
from per-box masks, box slots [x, y, w, h, category, iscrowd] at 64-px scale (x2 / x4 for the
larger scales, /2 for the 32x32 feature map), per-slot box masks, captions of 5..12 words.  The
frozen text encoder is replaced by random word / sentence embeddings (zero beyond each caption's
length, like RNN_ENCODER's padded output).  Everything is drawn from generators seeded with
`seed`, on the CPU, so every machine produces the same batch.
"""
import numpy as np
import torch


def make_batch(batch_size=16, seed=1234, num_classes=80, words_num=12, boxes_num=10,
               base_size=64, branch_num=3, nef=256, glove_dim=50, device=None, rois_dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    B, L, R = batch_size, words_num, boxes_num
    sizes = [base_size * (2 ** i) for i in range(branch_num)]

    imgs = [torch.rand(B, 3, s, s, generator=g) * 2 - 1 for s in sizes]

    cap_lens = np.sort(rng.randint(5, L + 1, size=B))[::-1].copy()
    cap_lens[0] = L
    captions = np.zeros((B, L), np.int64)
    for i in range(B):
        captions[i, :cap_lens[i]] = rng.randint(1, 1000, size=cap_lens[i])
    cap_lens_t = torch.from_numpy(cap_lens.astype(np.int64))
    valid = torch.arange(L).unsqueeze(0) < cap_lens_t.unsqueeze(1)            # B x L
    words_embs = torch.randn(B, nef, L, generator=g) * valid.unsqueeze(1)
    sent_emb = torch.randn(B, nef, generator=g)
    glove_words_embs = torch.randn(B, glove_dim, L, generator=g)
    mask = torch.from_numpy(captions == 0)
    clabels_emb = torch.randn(num_classes, glove_dim, generator=g)

    num_rois = rng.randint(1, R + 1, size=B)
    num_rois[rng.randint(0, B)] = R
    rois0 = np.zeros((B, R, 6), np.float64)
    for i in range(B):
        n = num_rois[i]
        xy = rng.uniform(0, 40, size=(n, 2))
        wh = rng.uniform(4, 24, size=(n, 2))
        wh = np.minimum(wh, base_size - xy)
        rois0[i, :n, 0:2] = xy
        rois0[i, :n, 2:4] = wh
        rois0[i, :n, 4] = rng.randint(0, num_classes, size=n)
    # a few large boxes so that the large-scale object discriminator has work (>= 16 at 32-px scale)
    for i in range(0, B, 3):
        rois0[i, 0, 0:4] = [2.0, 3.0, 40.0 + (i % 5), 36.0]
    rois = []
    for i in range(branch_num):
        r = rois0.copy()
        r[:, :, :4] *= 2 ** i
        rois.append(torch.from_numpy(r).to(rois_dtype))
    fm = rois0.copy()
    fm[:, :, :4] /= 2.0
    fm_rois = torch.from_numpy(fm).to(rois_dtype)

    def box_masks(size):
        """per-slot soft box masks (filled ellipses) B x R x size x size, values in [0, 1]"""
        m = np.zeros((B, R, size, size), np.float32)
        scale = size / float(base_size)
        yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
        for i in range(B):
            for r in range(num_rois[i]):
                x, y, w, h = rois0[i, r, :4] * scale
                cx, cy = x + w / 2.0, y + h / 2.0
                d = ((xx + 0.5 - cx) / max(w / 2.0, 0.5)) ** 2 + ((yy + 0.5 - cy) / max(h / 2.0, 0.5)) ** 2
                m[i, r] = np.clip(1.5 - d, 0.0, 1.0)
        return m

    bt_masks_np = [box_masks(s) for s in sizes]
    fm_bt_masks = torch.from_numpy(box_masks(base_size // 2))
    hmaps = []
    for k, s in enumerate(sizes):
        hm = np.zeros((B, num_classes, s, s), np.float32)
        for i in range(B):
            for r in range(num_rois[i]):
                c = int(rois0[i, r, 4])
                hm[i, c] = np.maximum(hm[i, c], bt_masks_np[k][i, r])
        hmaps.append(torch.from_numpy(hm))
    bt_masks = [torch.from_numpy(m) for m in bt_masks_np]

    batch = {
        "imgs": imgs, "hmaps": hmaps, "rois": rois, "fm_rois": fm_rois,
        "num_rois": torch.from_numpy(num_rois.astype(np.int64)),
        "bt_masks": bt_masks, "fm_bt_masks": fm_bt_masks,
        "captions": torch.from_numpy(captions), "cap_lens": cap_lens_t,
        # GloVe vocabulary ids of the same words (reference trainDataset.py: glove_captions)
        "glove_captions": torch.from_numpy(np.where(captions > 0, (captions * 7) % 400 + 1, 0)),
        "words_embs": words_embs, "sent_emb": sent_emb, "glove_words_embs": glove_words_embs,
        "mask": mask, "clabels_emb": clabels_emb,
        "class_ids": np.arange(B),
        "noise": torch.randn(B, 100, generator=g),
        "ca_eps": torch.randn(B, 100, generator=g),
    }
    if device is not None:
        batch = to_device(batch, device)
    return batch


_HOST_KEYS = ("rois", "fm_rois", "num_rois", "cap_lens")       # small tables the host-side helpers index


def to_device(batch, device):
    from miscc.utils import attach_host

    def mv(v, keep):
        if torch.is_tensor(v):
            d = v.to(device)
            return attach_host(d, v) if keep else d
        if isinstance(v, (list, tuple)):
            return [mv(x, keep) for x in v]
        return v
    return {k: mv(v, k in _HOST_KEYS) for k, v in batch.items()}


def make_shape_inputs(seed=71, B_=2, R=3, nbf=8, S=64, fm=16):
    """Seeded inputs of the shape generator: one-hot box maps (category channel = 1 inside the box),
    their reversed sequence, the 16x16 box crops and the per-box noise [B, R, 4 * nbf]."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    fwd = torch.zeros(B_, R, nbf, S, S)
    fmaps = torch.zeros(B_, R, fm, fm)
    rois = np.zeros((B_, 10, 6))
    num = np.array([R, R - 1])[:B_]
    for b in range(B_):
        for r in range(int(num[b])):
            x, y = rng.randint(0, 36, 2)
            w, h = rng.randint(8, 26, 2)
            c = int(rng.randint(0, nbf))
            fwd[b, r, c, y:y + h, x:x + w] = 1
            fmaps[b, r, y // 4:(y + h) // 4, x // 4:(x + w) // 4] = 1
            rois[b, r] = [x, y, w, h, c, 0]
    bwd = fwd.flip(1).clone()
    z = torch.randn(B_, R, 4 * nbf, generator=g)        # evaluator.py:278-279: noise of 4 * num_classes
    return z, fwd, bwd, fmaps, torch.from_numpy(rois), torch.from_numpy(num.astype(np.int64))
