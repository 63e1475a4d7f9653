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
from miscc.load import (load_filenames, load_text_data, load_glove_emb, load_cat_label, load_class_id,
                        load_cats, load_imgs_data, load_anns_data, get_imgs, get_caption, get_hmaps_rois)


class TrainDataset(data.Dataset):
    """`device_hmaps=True` is the lean hand-over: items carry empty layout maps and `prepare_data`
    rebuilds them on the device from the per-slot box masks (`hmaps_from_box_masks`).  The 80-channel
    float64 maps are 55 MB per sample on the host and 27 MB over PCIe, the ten box masks they are
    sums of are 3.4 MB."""

    def __init__(self, data_dir, split='train', base_size=64, device_hmaps=False):
        self.device_hmaps = device_hmaps
        self.embeddings_num = cfg.TEXT.CAPTIONS_PER_IMAGE
        self.imsize = []
        for _ in range(cfg.TREE.BRANCH_NUM):
            self.imsize.append(base_size)
            base_size = base_size * 2
        self.fmsize = cfg.ROI.FM_SIZE
        self.data_dir = data_dir
        split_dir = os.path.join(data_dir, split)

        train_names = load_filenames(data_dir, 'train')
        test_names = load_filenames(data_dir, 'test')
        self.filenames, self.captions, self.ixtoword, self.wordtoix, self.n_words = \
            load_text_data(data_dir, split, train_names, test_names)
        self.glove_captions, self.glove_ixtoword, self.glove_wordtoix, self.glove_embed = \
            load_glove_emb(data_dir, split, train_names, test_names)
        self.cat_labels, self.cat_label_lens, self.sorted_cat_label_indices = \
            load_cat_label(data_dir, self.glove_wordtoix)
        self.class_id = load_class_id(split_dir, len(self.filenames))
        self.number_example = len(self.filenames)
        self.cats_dict, self.cats_index_dict = load_cats(data_dir, self.wordtoix)
        self.num_classes = len(self.cats_index_dict)
        self.img_bytes = load_imgs_data(data_dir, split, self.filenames)
        self.insanns_dict = load_anns_data(data_dir, split, '_gt_insanns.pickle', 'gt', self.filenames,
                                           self.imsize, self.fmsize, self.cats_index_dict)

    def __getitem__(self, index):
        key = self.filenames[index]
        cls_id = self.class_id[index]
        imgs = get_imgs(self.img_bytes[index], self.imsize)
        hmaps, _, _, _, rois, fm_rois, num_rois, bt_masks, fm_bt_masks = get_hmaps_rois(
            self.insanns_dict[key], self.imsize, self.fmsize, self.cats_index_dict,
            with_hmaps=not self.device_hmaps)
        sent_ix = random.randint(0, self.embeddings_num)          # one of the image's captions
        new_sent_ix = index * self.embeddings_num + sent_ix
        caps, glove_caps, cap_len = get_caption(self.captions, self.glove_captions, new_sent_ix)
        return imgs, caps, glove_caps, cap_len, hmaps, rois, fm_rois, num_rois, \
            bt_masks, fm_bt_masks, cls_id, key

    def __len__(self):
        return len(self.filenames)


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


def prepare_data(data, device=None, num_classes=None):
    """Collated loader output -> the reference's 12-item list: everything sorted by caption
    length (descending, torch.sort like the reference), float32 maps, tensors on `device`
    (reference trainDataset.py:79-127; `device=None` keeps them where they are, the reference's
    cfg.CUDA False branch)."""
    imgs, captions, glove_captions, captions_lens, hmaps, rois, fm_rois, \
        num_rois, bt_masks, fm_bt_masks, class_ids, keys = data
    sorted_cap_lens, sorted_cap_indices = torch.sort(captions_lens, 0, True)
    mv = (lambda t: t.to(device, non_blocking=True)) if device is not None else (lambda t: t)

    num_rois = num_rois[sorted_cap_indices]
    real_hmaps, real_imgs, real_bt_masks, real_rois = [], [], [], []
    for i in range(len(imgs)):
        real_imgs.append(mv(imgs[i][sorted_cap_indices]))
        real_rois.append(mv(rois[i][sorted_cap_indices]))
        bt = mv(bt_masks[i][sorted_cap_indices])
        if hmaps[i].numel() == 0:           # lean hand-over: rebuild on the device, before the float32 cast
            if num_classes is None:
                raise ValueError("prepare_data: num_classes is needed to rebuild the layout maps")
            real_hmaps.append(hmaps_from_box_masks(bt, mv(rois[0][sorted_cap_indices]), num_classes))
        else:
            real_hmaps.append(mv(hmaps[i][sorted_cap_indices].float()))
        real_bt_masks.append(bt.float())
    fm_rois = mv(fm_rois[sorted_cap_indices])
    fm_bt_masks = mv(fm_bt_masks[sorted_cap_indices].float())
    captions = mv(captions[sorted_cap_indices].squeeze())
    glove_captions = mv(glove_captions[sorted_cap_indices].squeeze())
    class_ids = class_ids[sorted_cap_indices].numpy()
    keys = [keys[i] for i in sorted_cap_indices.numpy()]
    return [real_imgs, captions, glove_captions, mv(sorted_cap_lens), real_hmaps, real_rois,
            fm_rois, mv(num_rois), real_bt_masks, fm_bt_masks, class_ids, keys]


def batch_dict(prepared, clabels_emb):
    """The prepared list as the dictionary `condGANTrainer.train_step` reads (the frozen caption
    encoder and the GloVe table are applied inside the step, reference trainer.py:367-383)."""
    imgs, captions, glove_captions, cap_lens, hmaps, rois, fm_rois, num_rois, bt_masks, \
        fm_bt_masks, class_ids, keys = prepared
    return {"imgs": imgs, "captions": captions, "glove_captions": glove_captions, "cap_lens": cap_lens,
            "max_len": int(torch.max(cap_lens)), "hmaps": hmaps, "rois": rois, "fm_rois": fm_rois,
            "num_rois": num_rois, "bt_masks": bt_masks, "fm_bt_masks": fm_bt_masks,
            "class_ids": class_ids, "keys": keys, "clabels_emb": clabels_emb}


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
                           num_workers=int(workers), pin_memory=True, persistent_workers=int(workers) > 0)
