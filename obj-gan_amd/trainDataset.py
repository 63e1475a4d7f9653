"""Training dataset over a prepared Obj-GAN data directory and the batch hand-over to the
training step (reference image_generation/trainDataset.py): same class, same item tuple, same
`prepare_data` return list, plus `batch_dict` -- the dictionary `condGANTrainer.train_step` takes.
"""
import os

import numpy as np
import numpy.random as random
import torch
import torch.utils.data as data

from miscc.config import cfg
from miscc.utils import attach_host, _host
from miscc.load import (load_filenames, load_text_data, load_glove_emb, load_cat_label, load_class_id,
                        load_cats, load_imgs_data, load_anns_data, get_imgs, decode_rgb, get_caption,
                        get_hmaps_rois)
from objgan_hip import ops


class TrainDataset(data.Dataset):
    """`device_hmaps=True` is the lean hand-over: items carry empty layout maps and `prepare_data`
    rebuilds them on the device from the per-slot box masks (`hmaps_from_box_masks`).  The 80-channel
    float64 maps are 55 MB per sample on the host and 27 MB over PCIe, the ten box masks they are
    sums of are 3.4 MB.
    `device_masks=True` goes one step further: no `skimage.transform.resize` in the loader workers at all (33 calls
    per sample in the reference, load.py:160-176) -- the items carry the ten raw 64 x 64 instance masks (0.3 MB) and
    `prepare_data` resizes them to 32 / 128 / 256 on the device (`ops.resize_masks`, bit for bit scipy's arithmetic).

    Attributes follow the reference class (trainer and evaluator read them): filenames, captions,
    ixtoword / wordtoix / n_words, glove_captions / glove_ixtoword / glove_wordtoix / glove_embed,
    cat_labels / cat_label_lens / sorted_cat_label_indices, class_id, cats_dict / cats_index_dict,
    img_bytes, insanns_dict."""

    def __init__(self, data_dir, split='train', base_size=64, device_hmaps=False, device_imgs=False, device_masks=False,
                 device_jpeg=False):
        # `device_jpeg=True`: the items carry the JPEG FILE (1-D uint8 tensor) and `prepare_data` decodes it on the device
        # (ops.jpeg_decode_batch, bit for bit Pillow) before the device resize -- no decode, no resize in the loader
        # workers, only the file bytes cross PCIe.  A file the device path refuses (progressive, CMYK: objgan_jpeg_parse,
        # host-only) is decoded by Pillow in the worker and travels as the decoded image; `host_decoded` counts those.
        self.device_jpeg = device_jpeg
        device_imgs = device_imgs or device_jpeg
        self.host_decoded = 0
        self.device_masks = device_masks
        self.device_hmaps = device_hmaps or device_masks      # the maps are sums of the masks: rebuilt behind them
        self.device_imgs = device_imgs
        self.data_dir = data_dir
        self.embeddings_num = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.imsize = [base_size * (2 ** b) for b in range(cfg.TREE.BRANCH_NUM)]
        self.fmsize = cfg.ROI.FM_SIZE
        names = {s: load_filenames(data_dir, s) for s in ('train', 'test')}
        text = load_text_data(data_dir, split, names['train'], names['test'])
        self.filenames, self.captions, self.ixtoword, self.wordtoix, self.n_words = text
        glove = load_glove_emb(data_dir, split, names['train'], names['test'])
        self.glove_captions, self.glove_ixtoword, self.glove_wordtoix, self.glove_embed = glove
        # third entry: the permutation from length-sorted rows back to categories.txt order
        self.cat_labels, self.cat_label_lens, self.sorted_cat_label_indices = \
            load_cat_label(data_dir, self.glove_wordtoix)
        self.number_example = len(self.filenames)
        self.class_id = load_class_id(os.path.join(data_dir, split), self.number_example)
        self.cats_dict, self.cats_index_dict = load_cats(data_dir, self.wordtoix)
        self.num_classes = len(self.cats_index_dict)
        self.img_bytes = load_imgs_data(data_dir, split, self.filenames)
        self.insanns_dict = load_anns_data(data_dir, split, '_gt_insanns.pickle', 'gt', self.filenames,
                                           self.imsize, self.fmsize, self.cats_index_dict)

    def __len__(self):
        return self.number_example

    def __getitem__(self, index):
        """-> (imgs[3], caption ids [12, 1], GloVe ids [12, 1], length, hmaps[3], rois[3], fm_rois,
        num_rois, bt_masks[3], fm_bt_masks, class id, key) -- the reference item tuple."""
        key = self.filenames[index]
        maps = get_hmaps_rois(self.insanns_dict[key], self.imsize, self.fmsize, self.cats_index_dict,
                              with_hmaps=not self.device_hmaps, with_resized_masks=not self.device_masks)
        hmaps, rois, fm_rois, num_rois, bt_masks, fm_bt_masks = maps[0], maps[4], maps[5], maps[6], maps[7], maps[8]
        pick = index * self.embeddings_num + random.randint(0, self.embeddings_num)    # one of the image's captions
        caps, glove_caps, cap_len = get_caption(self.captions, self.glove_captions, pick)
        if self.device_jpeg:
            raw = self.img_bytes[index]
            if ops.jpeg_parse([raw])[1][0, 8] == 0:
                imgs = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            else:
                imgs = decode_rgb(raw)
                self.host_decoded += 1
        else:
            imgs = decode_rgb(self.img_bytes[index]) if self.device_imgs else get_imgs(self.img_bytes[index], self.imsize)
        return (imgs, caps, glove_caps, cap_len, hmaps, rois, fm_rois,
                num_rois, bt_masks, fm_bt_masks, self.class_id[index], key)


def hmaps_from_box_masks(bt_masks, rois0, num_classes):
    """Layout maps [B, num_classes, S, S] from the per-slot box masks [B, R, S, S] and the slots'
    categories (column 4 of the 64-px boxes): hmap[b, cat(b, r)] += mask[b, r], accumulated in
    float64 like the reference's numpy loop (load.py:166-176); empty slots hold zero masks."""
    B, R, S, _ = bt_masks.shape
    cat = rois0[:, :, 4].to(torch.long)
    idx = (torch.arange(B, device=bt_masks.device).unsqueeze(1) * num_classes + cat).reshape(-1)
    out = torch.zeros((B * num_classes, S, S), dtype=torch.float64, device=bt_masks.device)
    out.index_add_(0, idx, bt_masks.reshape(B * R, S, S).to(torch.float64))
    return out.view(B, num_classes, S, S).to(torch.float32)


_JPEG_INDEX = []


def _jpeg_index_cache():
    if not _JPEG_INDEX:
        _JPEG_INDEX.append(ops.JpegIndexCache() if hasattr(ops, "JpegIndexCache") else None)
    return _JPEG_INDEX[0]


def prepare_data(data, device=None, num_classes=None):
    """Collated loader output -> the reference's 12-item list: every per-sample tensor reordered by
    caption length (descending; `torch.sort` like the reference, so ties fall the same way), maps as
    float32, tensors moved to `device` (None keeps them where they are: the reference's cfg.CUDA False
    branch).  Reference trainDataset.py:79-127."""
    (imgs, captions, glove_captions, cap_lens, hmaps, rois, fm_rois, num_rois, bt_masks, fm_bt_masks,
     class_ids, keys) = data
    lens_sorted, order = torch.sort(cap_lens, 0, True)

    def put(t):
        return t if device is None else t.to(device, non_blocking=True)

    def take(t):
        return put(t[order])

    def take_small(t):      # box tables / counts: the host copy stays attached (miscc.utils._host)
        h = t[order]
        return h if device is None else attach_host(h.to(device, non_blocking=True), h)
    branches = range(len(bt_masks))
    if len(imgs) and torch.is_tensor(imgs[0]) and imgs[0].dtype == torch.uint8:
        # `device_imgs` hand-over: one decoded [H, W, 3] image per sample (collate_keep_images)
        if device is None:
            raise ValueError("prepare_data: decoded images are resized on the device; pass `device`")
        picked = [imgs[i] for i in order.tolist()]
        if any(im.dim() == 1 for im in picked):
            # `device_jpeg` hand-over: JPEG files (and the odd host-decoded image) -> decoded on the device -> resized there
            # (the entropy index of every file decoded so far lives on the device: from the second epoch on a file is decoded
            #  by one lane per MCU row instead of one lane per file)
            src, offs, hs, ws = ops.images_to_device(picked, device, _jpeg_index_cache(), [keys[i] for i in order.tolist()])
            out_imgs = ops.resize_pil_bilinear_device(src, offs, hs, ws, [bt_masks[b].shape[-1] for b in branches])
        else:
            out_imgs = ops.resize_pil_bilinear(picked, [bt_masks[b].shape[-1] for b in branches], device)
    else:
        out_imgs = [take(imgs[b]) for b in branches]
    out_rois = [take_small(rois[b]) for b in branches]
    out_masks, out_hmaps = [], []
    fm_masks = None
    if fm_bt_masks.dim() >= 2 and fm_bt_masks.shape[1] == 0 and bt_masks[0].shape[1] > 0:
        # `device_masks` hand-over: bt_masks[0] holds the raw 64 x 64 instance masks, everything else is made here
        # (detected on the feature-map masks, which are ALWAYS empty in that mode -- also with one branch, where no
        # bt_masks[1] exists to look at)
        if device is None:
            raise ValueError("prepare_data: raw instance masks are resized on the device; pass `device`")
        raw = take(bt_masks[0]).to(torch.float64)
        sizes = [fm_bt_masks.shape[-1]] + [bt_masks[b].shape[-1] for b in branches[1:]]
        resized = ops.resize_masks(raw, sizes)
        fm_masks, dev_masks = resized[0], [raw] + resized[1:]
    else:
        dev_masks = [take(bt_masks[b]) for b in branches]
    for b in branches:
        masks = dev_masks[b]
        if hmaps[b].numel() == 0:           # lean hand-over: rebuild on the device, before the float32 cast
            if num_classes is None:
                raise ValueError("prepare_data: num_classes is needed to rebuild the layout maps")
            out_hmaps.append(hmaps_from_box_masks(masks, out_rois[0], num_classes))
        else:
            out_hmaps.append(take(hmaps[b].float()))
        out_masks.append(masks.float())
    order_list = order.tolist()
    lens_out = lens_sorted if device is None else attach_host(put(lens_sorted), lens_sorted)
    return [out_imgs, take(captions).squeeze(), take(glove_captions).squeeze(), lens_out, out_hmaps,
            out_rois, take_small(fm_rois), take_small(num_rois), out_masks,
            (take(fm_bt_masks) if fm_masks is None else fm_masks).float(),
            class_ids[order].numpy(), [keys[i] for i in order_list]]


def batch_dict(prepared, clabels_emb):
    """The prepared list as the dictionary `condGANTrainer.train_step` reads (the frozen caption
    encoder and the GloVe table are applied inside the step, reference trainer.py:367-383)."""
    imgs, captions, glove_captions, cap_lens, hmaps, rois, fm_rois, num_rois, bt_masks, \
        fm_bt_masks, class_ids, keys = prepared
    return {"imgs": imgs, "captions": captions, "glove_captions": glove_captions, "cap_lens": cap_lens,
            "max_len": int(_host(cap_lens).max()), "hmaps": hmaps, "rois": rois, "fm_rois": fm_rois,
            "num_rois": num_rois, "bt_masks": bt_masks, "fm_bt_masks": fm_bt_masks,
            "class_ids": class_ids, "keys": keys, "clabels_emb": clabels_emb}


def collate_keep_images(samples):
    """default_collate for every field but the first: decoded images have different sizes and stay a
    list of uint8 [H, W, 3] tensors (the `device_imgs` hand-over)."""
    rest = data.default_collate([s[1:] for s in samples])
    return [[s[0] for s in samples]] + list(rest)


def build_loader(dataset, batch_size, workers=0, rank=0, world=1, seed=0, shuffle=True):
    """The reference's loader settings (main.py: shuffle, drop_last) with one shard per rank: under
    DDP every process draws a disjoint 1/world of each epoch's permutation (DistributedSampler),
    which replaces the reference's single loader feeding nn.DataParallel."""
    sampler = None
    if world > 1:
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank,
                                                      shuffle=shuffle, seed=seed, drop_last=True)
    return data.DataLoader(dataset, batch_size=batch_size, drop_last=True,
                           shuffle=(shuffle and sampler is None), sampler=sampler,
                           num_workers=int(workers), pin_memory=True, persistent_workers=int(workers) > 0,
                           collate_fn=collate_keep_images if getattr(dataset, "device_imgs", False) else None)
