"""Hot-path helpers of the image generator (the subset of reference
image_generation/miscc/utils.py that the G+D training step calls: :309-329, :365-413,
:445-522).  Visualisation and IS/FID utilities of the reference file are outside the hot-path
scope and are not provided.
"""
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from miscc.config import cfg
from objgan_hip import ops


# ---- host copies of the small box tensors ---------------------------------------------------------
# The reference reads rois / num_rois on the host in several places per step (utils.py:466, 503-505,
# model.py:549, 661): each read of a device tensor is a device->host copy that drains the stream.
# The box tensors are inputs of the step and do not change, so the host copy is taken once per
# tensor (keyed on the tensor object and its version counter) and reused by every caller.
_HOST = []          # [(tensor, version, numpy)], most recent first, at most 8 entries


def attach_host(t_dev, t_cpu):
    """Remember the host copy a device tensor was made from (box tables, box counts: a few hundred bytes
    that the host-side helpers below index).  A `.cpu()` of them in the middle of a step is a full
    host-device sync: the host stops enqueueing until the GPU has drained."""
    if torch.is_tensor(t_dev) and t_dev.is_cuda and torch.is_tensor(t_cpu) and not t_cpu.is_cuda:
        t_dev._og_host = t_cpu.detach().numpy()
    return t_dev


def h2d(t, device):
    """Host array / tensor -> device WITHOUT stalling the enqueueing thread.  A plain `.to(device)` of pageable host
    memory returns only when the copy has executed, i.e. after everything queued in front of it on the stream: every
    small index upload was a full host-device synchronisation (64 per step, r03 host profile: the host could never
    run ahead of the GPU).  Staged through a pinned buffer of torch's caching host allocator, the copy is just
    another stream operation."""
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    elif not torch.is_tensor(t):
        t = torch.as_tensor(t)
    device = torch.device(device)
    if device.type != "cuda" or t.is_cuda:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def take_rows(t, rows):
    """t[rows] (rows: python list of sample indices) that keeps the host copy attached."""
    if torch.is_tensor(t) and t.is_cuda:         # (indexing with a python list uploads it synchronously)
        out = t.index_select(0, h2d(np.asarray(rows, np.int64), t.device))
    else:
        out = t[rows]
    h = getattr(t, "_og_host", None)
    if h is not None and torch.is_tensor(out) and out.is_cuda:
        out._og_host = h[rows]
    return out


def _host(t):
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.detach().numpy()
    h = getattr(t, "_og_host", None)
    if h is not None:
        return h
    for i, (ref, ver, arr) in enumerate(_HOST):
        if ref is t and ver == t._version:
            if i:
                _HOST.insert(0, _HOST.pop(i))
            return arr
    arr = t.detach().cpu().numpy()
    _HOST.insert(0, (t, t._version, arr))
    del _HOST[8:]
    return arr


# ---- ROI blob helpers (host/numpy API of the reference, kept for callers that use it) ----------
def _project_im_rois(im_rois, scales):
    """R x 4 boxes -> (scaled boxes, batch index column); the batch index of row r is
    r // BOXES_NUM (reference utils.py:365-385)."""
    im_rois = im_rois.astype(float, copy=False)
    per_img = cfg.ROI.BOXES_NUM
    levels = np.repeat(np.arange(im_rois.shape[0] // per_img).astype(int), per_img).reshape(-1, 1)
    return im_rois * scales[levels], levels


def _get_rois_blob(im_rois, im_scale_factors):
    """R x 4 boxes -> R x 5 float32 rows [batch_idx, x1, y1, x2, y2] (reference utils.py:387-399)."""
    rois, levels = _project_im_rois(im_rois, im_scale_factors)
    return np.hstack((levels, rois)).astype(np.float32, copy=False)


# ---- masked max over box slots -------------------------------------------------------------------
def pprocess_bt_attns(fmaps, ih, iw, bt_mask):
    """out[b, c, p] = max_r fmaps[b, c, r] * bt_mask[b, r, (c,) p]  (reference utils.py:401-413).

    fmaps  : batch x num x max_num_rois x 1
    bt_mask: batch x max_num_rois x num x ih x iw (the reference's repeated form; an expanded
             view costs nothing) or batch x max_num_rois x ih x iw.
    One fused kernel: the batch x rois x num x ih x iw product is never materialised."""
    return ops.masked_max(fmaps, bt_mask, ih, iw)


# ---- class-permuted layout maps ("wrong shape" negatives) ------------------------------------------
def _class_permutations(C, rois, num_rois):
    """Host part of permute_seg: (perm [B, C] int64 channel map -- identity except for the shuffled class
    channels of the changed samples --, valid_mask).  Draws from python's `random` exactly like the
    reference loop (utils.py:445-462), so a seeded run shuffles identically."""
    rois_np = _host(rois)
    nr = _host(num_rois).tolist()
    B = len(nr)
    perm = np.tile(np.arange(C, dtype=np.int64), (B, 1))
    valid_mask = []
    for b in range(B):
        n = int(nr[b])
        if n == 0:
            continue
        present = [int(c) for c in np.unique(rois_np[b, :n, 4])]
        shuffled = deepcopy(present)
        random.shuffle(shuffled)
        if present != shuffled:
            valid_mask.append(b)
            perm[b, present] = shuffled
    return perm, valid_mask


def permute_seg(seg_conditions, rois, num_rois):
    """Shuffle the class channels present in each sample's layout map (reference utils.py:445-462).
    Returns (new_seg_conditions, valid_mask) with valid_mask the list of changed sample indices.  One
    gather for the whole batch (the reference copies the map and then moves channels sample by sample)."""
    B, C = seg_conditions.size(0), seg_conditions.size(1)
    perm, valid_mask = _class_permutations(C, rois, num_rois)
    dev = seg_conditions.device
    rows = torch.arange(B, device=dev).unsqueeze(1)
    new_seg = seg_conditions[rows, h2d(perm, dev)]
    return new_seg, valid_mask


def draw_class_permutations(seg_conditions, rois, num_rois):
    """The RANDOM part of permuted_valid_seg on its own (host only: python's `random` + the host copies of the box tables).
    The trainer draws the permutations of all five users of a step up front, in the reference's order (reference
    trainer.py:398-443 calls the losses in that order), and hands them to the losses through `draw=`: the ORDER IN WHICH THE
    HOST ISSUES the discriminator jobs is then free of the RNG sequence."""
    return _class_permutations(seg_conditions.size(1), rois, num_rois)


def permuted_valid_seg(seg_conditions, rois, num_rois, draw=None):
    """(permuted layout maps of the CHANGED samples only [len(valid), C, H, W], valid_mask): what the
    discriminator losses consume -- permute_seg(...)[0][valid_mask] without building the unchanged rows.
    draw: the (perm, valid_mask) pair of draw_class_permutations for these arguments (None: drawn here)."""
    C = seg_conditions.size(1)
    perm, valid_mask = draw if draw is not None else _class_permutations(C, rois, num_rois)
    if not valid_mask:
        return None, valid_mask
    dev = seg_conditions.device
    # one upload: [row index | channel map] per changed sample
    both = h2d(np.concatenate([np.asarray(valid_mask, np.int64)[:, None], perm[valid_mask]], axis=1), dev)
    return seg_conditions[both[:, :1], both[:, 1:]], valid_mask


def feat_select(pooled_feat, raw_bt_c_codes, fm_rois, num_rois, is_large_scale=False):
    """Keep the pooled ROI features of real boxes of the requested scale (reference
    utils.py:465-499): drop boxes with w < 1.25 and h < 1.25; small-scale keeps max(w, h) < 16,
    large-scale keeps >= 16.  Returns (features [K, C, h, w], classes [K] int64 (cpu),
    bt_c_codes [K, idf]); empty lists when nothing survives."""
    rois_np = _host(fm_rois)
    nr = _host(num_rois).tolist()
    thr = cfg.ROI.ROI_SIZE_THRS
    sel_b, sel_r, classes = [], [], []
    for b in range(len(nr)):
        keep = []
        for r in range(int(nr[b])):
            w, h = rois_np[b, r, 2], rois_np[b, r, 3]
            if w < 1.25 and h < 1.25:
                continue
            big = max(w, h) >= thr
            if big != bool(is_large_scale):
                continue
            keep.append(r)
        if keep:
            sel_b += [b] * len(keep)
            sel_r += keep
            classes.append(rois_np[b, keep, 4].astype(int))
    if not sel_b:
        return [], [], []
    dev = pooled_feat.device
    cls = np.concatenate(classes).astype(np.int64)
    # one upload for the three index vectors (every small host->device copy stalls the enqueueing thread)
    idx = h2d(np.stack([np.asarray(sel_b, np.int64), np.asarray(sel_r, np.int64), cls]), dev)
    ib, ir = idx[0], idx[1]
    x_code_rois = pooled_feat[ib, ir]
    # NB reference quirk (SURVEY.md trap 6): raw_bt_c_codes is indexed with the batch index of the
    # tensors passed in, also when those are a `valid_mask` subset of the batch.
    if raw_bt_c_codes.device == dev:
        bt_c_codes = raw_bt_c_codes[ib, ir]
    else:
        bt_c_codes = raw_bt_c_codes[ib.to(raw_bt_c_codes.device), ir.to(raw_bt_c_codes.device)]
    classes = torch.from_numpy(cls)
    if dev.type != "cpu":
        classes._og_dev = idx[2]
    return x_code_rois, classes, bt_c_codes


def form_clabels_feat(clabels_emb, rois, num_rois):
    """Per-box class-label embeddings: batch x emb x max_num_roi x 1 (reference utils.py:502-522)."""
    rois_np = _host(rois)
    nr = _host(num_rois).tolist()
    B = rois_np.shape[0]
    max_num_roi = int(np.amax(nr))
    # one index table for the batch (-1 = empty slot), one upload, one gather
    idx = np.full((B, max_num_roi), -1, np.int64)
    for i in range(B):
        n = int(nr[i])
        if n:
            idx[i, :n] = rois_np[i, :n, 4].astype(np.int64)
    idx_dev = h2d(idx, clabels_emb.device)
    feat = clabels_emb[idx_dev.clamp(min=0)] * (idx_dev >= 0).unsqueeze(2).to(clabels_emb.dtype)
    return feat.transpose(1, 2).unsqueeze(3)


# ---- parameter utilities -------------------------------------------------------------------------
def weights_init(m):
    """orthogonal(gain 1) for conv / linear weights, N(1, 0.02) for BatchNorm weights, zero biases
    (reference utils.py:309-319)."""
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.orthogonal_(m.weight.data, 1.0)
    elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data, 1.0)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def copy_G_params(model):
    return deepcopy(list(p.data for p in model.parameters()))


def load_params(model, new_param):
    """Overwrite the parameters in place (reference utils.py: `p.data.copy_(new_p)`, used to swap the
    EMA weights in and out).  Written through the parameter itself, not through `.data`: that keeps
    the values identical and moves the tensors' version counters, which is what the packed-filter
    caches of objgan_hip.ops key on -- a `.data` write would leave stale filter banks in use."""
    with torch.no_grad():
        for p, new_p in zip(model.parameters(), new_param):
            p.copy_(new_p)


def mkdir_p(path):
    import os
    os.makedirs(path, exist_ok=True)


def _split_scores(predictions, num_splits, fn):
    scores = []
    n = predictions.shape[0]
    for i in range(num_splits):
        part = predictions[i * n // num_splits:(i + 1) * n // num_splits, :]
        scores.append(fn(part))
    return np.mean(scores), np.std(scores)


def compute_inception_score(predictions, num_splits=1):
    """exp(mean KL(p(y|x) || p(y))) per split of the class posteriors (reference utils.py:417-428)."""
    return _split_scores(predictions, num_splits, lambda part: np.exp(np.mean(np.sum(
        part * (np.log(part) - np.log(np.expand_dims(np.mean(part, 0), 0))), 1))))


def negative_log_posterior_probability(predictions, num_splits=1):
    """mean -log max_y p(y|x) per split (reference utils.py:431-441)."""
    return _split_scores(predictions, num_splits, lambda part: np.mean(-1. * np.log(np.max(part, 1))))


def _unit_range(mask):
    """(mask - min) / (max - min); a constant map is kept if its value is >= 0.6 and zeroed
    otherwise (reference utils.py:544-549, 563-568) -- branch-free, no host sync."""
    mn, mx = mask.min(), mask.max()
    return torch.where(mx != mn, (mask - mn) / (mx - mn), torch.where(mx < 0.6, mask * 0, mask))


def form_hmaps(raw_masks, num_rois, rois, hmap_size, num_classes):
    """Generated instance masks [B, R, 64, 64] -> layout maps / per-slot masks at every branch
    size and the 32x32 feature scale (reference utils.py:524-584): masks are rescaled to [0, 1],
    merged per category with a running maximum, and resized bilinearly (align_corners=True)."""
    num = _host(num_rois).tolist()
    cats = _host(rois)[:, :, 4]
    B = int(raw_masks.size(0))
    S = hmap_size[-1]
    re_raw_masks = ops.bilinear_resize(raw_masks, S, S).clone()
    raw_gen_hmap = torch.zeros(B, num_classes, S, S, device=raw_masks.device)
    for b in range(B):
        cat_indices = [int(c) for c in cats[b, :int(num[b])].tolist()]
        for count, cat in enumerate(cat_indices):
            tmp = _unit_range(re_raw_masks[b, count])
            re_raw_masks[b, count] = tmp
            raw_gen_hmap[b, cat] = torch.max(tmp, raw_gen_hmap[b, cat])
        for cat in cat_indices:
            raw_gen_hmap[b, cat] = _unit_range(raw_gen_hmap[b, cat])
    gen_hmaps, gen_bt_masks = [], []
    for i in range(cfg.TREE.BRANCH_NUM):
        gen_hmaps.append(ops.bilinear_resize(raw_gen_hmap, hmap_size[i], hmap_size[i]))
        gen_bt_masks.append(ops.bilinear_resize(re_raw_masks, hmap_size[i], hmap_size[i]))
    gen_fm_bt_masks = ops.bilinear_resize(re_raw_masks, hmap_size[0] // 2, hmap_size[0] // 2)
    return gen_hmaps, gen_bt_masks, gen_fm_bt_masks
