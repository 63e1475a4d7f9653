#!/usr/bin/env python
"""Build tests/golden/data_tiny/ (a prepared Obj-GAN data directory with 4 + 2 images) and run the
UNMODIFIED reference loader over it (oracle/ref_data_harness.py): TrainDataset -> default collate
-> prepare_data, plus condGANTrainer.prepare_cat_emb.  What the reference returns is stored in
tests/golden/data_tiny_ref.pt (small tensors in full, the 128^2 / 256^2 maps as strided samples
+ moments).  Needs /root/reference; the committed outputs are what the tests read.

    python tests/golden/make_golden_data.py
"""
import io
import os
import pickle
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_data_harness as H        # noqa: E402

DATA = os.path.join(HERE, "data_tiny")
CATS = [(1, "person"), (2, "bicycle"), (10, "traffic light"), (13, "stop sign"), (3, "car")]
WORDS = ["a", "person", "rides", "bicycle", "near", "the", "traffic", "light", "stop", "sign", "car",
         "on", "street", "red", "two", "people", "and", "dog", "big", "small", "at", "night", "in", "rain"]


def fingerprint(t):
    t = torch.as_tensor(t).double()
    return {"shape": tuple(t.shape), "sum": float(t.sum()), "sq": float((t * t).sum()),
            "sample": t[..., ::8, ::8].float().clone()}


def smooth_image(rng, h, w):
    from PIL import Image
    yy, xx = np.mgrid[0:h, 0:w]
    chans = []
    for c in range(3):
        a, b, ph = rng.uniform(0.05, 0.4, 3)
        chans.append(127 + 100 * np.sin(a * xx + ph) * np.cos(b * yy + c))
    arr = np.clip(np.stack(chans, -1) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="JPEG", quality=90)
    return buf.getvalue()


def make_ann(rng, num_rois, sizes=(64, 128, 256), fmsize=16, R=10):
    rois = [np.zeros((R, 6)) for _ in sizes]
    fm_rois = np.zeros((R, 6))
    if num_rois == 0:
        return {"rois": rois, "fm_rois": fm_rois, "masks": None, "pooled masks": None,
                "bbox maps": None, "bbox fmaps": None, "num_rois": 0}
    raw = np.zeros((num_rois, 6))
    raw[:, 0:2] = rng.uniform(0, 36, (num_rois, 2))
    raw[:, 2:4] = rng.uniform(6, 26, (num_rois, 2))
    raw[:, 4] = rng.randint(0, len(CATS), num_rois)
    for b, s in enumerate(sizes):
        rois[b][:num_rois] = raw
        rois[b][:, :4] *= s / 64.0
    fm_rois[:num_rois] = rois[0][:num_rois]
    fm_rois[:, :4] /= 2.0
    yy, xx = np.mgrid[0:64, 0:64]
    masks, bmaps, bfmaps = [], np.zeros((num_rois, 64, 64)), np.zeros((num_rois, fmsize, fmsize))
    for r in range(num_rois):
        x, y, w, h = raw[r, :4]
        ell = (((xx - (x + w / 2)) / (w / 2)) ** 2 + ((yy - (y + h / 2)) / (h / 2)) ** 2) <= 1.0
        masks.append(np.clip(ell.astype(float) * rng.uniform(0.6, 1.0), 0, 1))
        x0, y0 = min(int(round(x)), 63), min(int(round(y)), 63)
        x1, y1 = min(int(round(x + w)), 63), min(int(round(y + h)), 63)
        bmaps[r, y0:y1, x0:x1] = 1
        fx0, fy0 = min(int(round(x / 4)), fmsize - 1), min(int(round(y / 4)), fmsize - 1)
        fx1, fy1 = min(int(round((x + w) / 4)), fmsize - 1), min(int(round((y + h) / 4)), fmsize - 1)
        bfmaps[r, fy0:fy1, fx0:fx1] = 1
    return {"rois": rois, "fm_rois": fm_rois, "masks": masks, "pooled masks": np.amax(masks, axis=0),
            "bbox maps": bmaps, "bbox fmaps": bfmaps, "num_rois": num_rois}


def build_directory(ns):
    rng = np.random.RandomState(7)
    shutil.rmtree(DATA, ignore_errors=True)
    for sub in ("train", "test", "images"):
        os.makedirs(os.path.join(DATA, sub))
    train = ["COCO_train2014_%012d" % i for i in (9, 25, 30, 34)]
    test = ["COCO_val2014_%012d" % i for i in (42, 73)]
    for split, names in (("train", train), ("test", test)):
        with open(os.path.join(DATA, split, "filenames.pickle"), "wb") as f:
            pickle.dump(names, f, protocol=2)
    with open(os.path.join(DATA, "categories.txt"), "w") as f:
        f.write("".join("%d,%s\n" % c for c in CATS))
    # caption vocabulary (captions.pickle) and the GloVe-side vocabulary (captions_glove.pickle)
    ixtoword = {0: "<end>"}
    ixtoword.update({i + 1: w for i, w in enumerate(WORDS)})
    wordtoix = {w: i for i, w in ixtoword.items()}
    glove_itos = ["<unk>", "<pad>"] + sorted(WORDS)
    glove_stoi = {w: i for i, w in enumerate(glove_itos)}

    def captions(n_imgs):
        caps, gcaps = [], []
        for _ in range(n_imgs * 5):
            n = int(rng.choice([3, 6, 9, 12, 13, 17]))
            words = [WORDS[k] for k in rng.randint(0, len(WORDS), n)]
            caps.append([wordtoix[w] for w in words])
            g = [glove_stoi[w] for w in words]
            gcaps.append(g[:-1] if rng.rand() < 0.25 and n > 3 else g)     # tokenisers disagree sometimes
        return caps, gcaps
    tr_c, tr_g = captions(len(train))
    te_c, te_g = captions(len(test))
    with open(os.path.join(DATA, "captions.pickle"), "wb") as f:
        pickle.dump([tr_c, te_c, ixtoword, wordtoix], f, protocol=2)
    Vocab = ns._stubs["torchtext.vocab"].Vocab
    vocabs = []
    for seed in (1, 2):
        v = Vocab()
        v.itos, v.stoi = list(glove_itos), dict(glove_stoi)
        v.vectors = torch.randn(len(glove_itos), 50, generator=torch.Generator().manual_seed(seed))
        vocabs.append(v)
    with open(os.path.join(DATA, "captions_glove.pickle"), "wb") as f:
        pickle.dump([tr_g, te_g, vocabs[0], vocabs[1]], f, protocol=2)
    # images -> bigfile, written by the REFERENCE's write_imgs
    shapes = [(48, 37), (64, 64), (30, 50), (71, 90), (40, 40), (33, 65)]
    for name, (h, w) in zip(train + test, shapes):
        with open(os.path.join(DATA, "images", name + ".jpg"), "wb") as f:
            f.write(smooth_image(rng, h, w))
    ns.load.write_imgs(DATA, train, os.path.join(DATA, "train_imgs.bigfile"))
    ns.load.write_imgs(DATA, test, os.path.join(DATA, "test_imgs.bigfile"))
    shutil.rmtree(os.path.join(DATA, "images"))
    anns = {name: make_ann(rng, n) for name, n in zip(train, (2, 1, 0, 1))}
    with open(os.path.join(DATA, "train_gt_insanns.pickle"), "wb") as f:
        pickle.dump([anns], f, protocol=2)


def main():
    ns = H.load_reference_data(branch_num=3)
    ns.cfg.CUDA = False
    with H.active(ns):
        build_directory(ns)
        ds = ns.trainDataset.TrainDataset(DATA, "train", base_size=64)
        np.random.seed(11)
        items = [ds[i] for i in range(len(ds))]
        np.random.seed(11)
        from torch.utils.data.dataloader import default_collate
        batch = default_collate([ds[i] for i in range(len(ds))])
        prepared = ns.trainDataset.prepare_data(batch)
        # category embeddings: the reference's own method, bound to a bare namespace
        fake_self = types.SimpleNamespace(glove_emb=ds.glove_embed, cat_labels=ds.cat_labels,
                                          cat_label_lens=ds.cat_label_lens, cats_index_dict=ds.cats_index_dict,
                                          sorted_cat_label_indices=ds.sorted_cat_label_indices)
        clabels = ns.trainer.condGANTrainer.prepare_cat_emb(fake_self).detach()
    out = {
        "filenames": list(ds.filenames), "n_words": ds.n_words, "cats_index_dict": dict(ds.cats_index_dict),
        "cats_dict": dict(ds.cats_dict), "cat_labels": ds.cat_labels, "cat_label_lens": ds.cat_label_lens,
        "sorted_cat_label_indices": ds.sorted_cat_label_indices, "class_id": np.asarray(ds.class_id),
        "img_bytes_len": [len(b) for b in ds.img_bytes], "img_bytes_first16": [bytes(b[:16]) for b in ds.img_bytes],
        "clabels_emb": clabels, "items": [], "prepared": {},
    }
    for it in items:
        imgs, caps, gcaps, cap_len, hmaps, rois, fm_rois, num_rois, bt_masks, fm_bt_masks, cls_id, key = it
        out["items"].append({
            "img64": imgs[0], "img128": fingerprint(imgs[1]), "img256": fingerprint(imgs[2]),
            "caps": torch.as_tensor(caps), "glove_caps": torch.as_tensor(gcaps), "cap_len": int(cap_len),
            "hmap64": torch.as_tensor(hmaps[0]).float(), "hmap128": fingerprint(hmaps[1]), "hmap256": fingerprint(hmaps[2]),
            "rois": [torch.as_tensor(r) for r in rois], "fm_rois": torch.as_tensor(fm_rois), "num_rois": int(num_rois),
            "bt_mask64": torch.as_tensor(bt_masks[0])[:max(int(num_rois), 1)].float(),
            "bt_mask64_fp": fingerprint(bt_masks[0]), "bt_mask256": fingerprint(bt_masks[2]),
            "fm_bt_masks": torch.as_tensor(fm_bt_masks)[:max(int(num_rois), 1)].float(),
            "fm_bt_masks_fp": fingerprint(fm_bt_masks), "cls_id": int(cls_id), "key": key})
    (p_imgs, p_caps, p_gcaps, p_lens, p_hmaps, p_rois, p_fm_rois, p_num, p_bt, p_fmbt, p_cls, p_keys) = prepared
    out["prepared"] = {"img64": fingerprint(p_imgs[0]), "captions": p_caps, "glove_captions": p_gcaps, "cap_lens": p_lens,
                       "hmap64": fingerprint(p_hmaps[0]), "hmap256": fingerprint(p_hmaps[2]), "rois": p_rois, "fm_rois": p_fm_rois,
                       "num_rois": p_num, "bt_mask64": fingerprint(p_bt[0]), "fm_bt_masks": fingerprint(p_fmbt),
                       "class_ids": np.asarray(p_cls), "keys": list(p_keys),
                       "dtypes": {"hmaps": str(p_hmaps[0].dtype), "rois": str(p_rois[0].dtype),
                                  "bt_masks": str(p_bt[0].dtype), "captions": str(p_caps.dtype)}}
    torch.save(out, os.path.join(HERE, "data_tiny_ref.pt"))
    sz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(DATA) for f in fs)
    print("data_tiny: %d KB, golden: %d KB" % (sz // 1024, os.path.getsize(os.path.join(HERE, "data_tiny_ref.pt")) // 1024))


if __name__ == "__main__":
    main()
