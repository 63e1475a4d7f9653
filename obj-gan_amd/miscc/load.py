"""Host-side readers of the reference's on-disk training data (reference image_generation/
miscc/load.py): the pickles and the image bigfile a prepared Obj-GAN data directory holds, and
the per-sample assembly of images, captions, layout maps and box slots.

Same function names, arguments and return values as the reference, restricted to what the
training path reads:

  <data_dir>/train/filenames.pickle, test/filenames.pickle   list of image keys
  <data_dir>/captions.pickle          [train_captions, test_captions, ixtoword, wordtoix]
  <data_dir>/captions_glove.pickle    [train_captions, test_captions, train_vocab, test_vocab]
                                      (the vocabs are torchtext Vocab objects: itos, stoi, vectors)
  <data_dir>/categories.txt           "<coco id>,<name>" per line
  <data_dir>/<split>_imgs.bigfile     for every key: int32 byte count + the JPEG file
  <data_dir>/<split>_gt_insanns.pickle [ {key: {'rois', 'fm_rois', 'masks', 'bbox maps',
                                               'bbox fmaps', 'num_rois', ...}} ]
  <data_dir>/<split>/class_info.pickle optional class ids

Building those files from raw COCO (captions via nltk / spacy / torchtext, masks via
pycocotools) is preprocessing, not the training path: where a prepared file is missing the
readers below raise instead of trying.  Third-party arithmetic the reference delegates to
packages that are absent here is restated and named where it happens (PIL bilinear resize for
torchvision.transforms.Resize, scipy.ndimage for skimage.transform.resize): parity of those two
is against the restatement, not against the packages (LAB.md section 6, "data path").
"""
import io
import mmap
import os
import pickle
import re
import struct
import types

import numpy as np
import torch
import torch.nn as nn

from miscc.config import cfg

_WORD = re.compile(r'\w+', re.UNICODE | re.MULTILINE | re.DOTALL)   # nltk RegexpTokenizer(r'\w+')


# ---------------------------------------------------------------------------------------------
# image bigfile (reference load.py:73-95)
# ---------------------------------------------------------------------------------------------
def write_imgs(data_dir, filenames, filepath):
    """Concatenate <data_dir>/images/<key>.jpg into the bigfile: native int32 length + bytes."""
    with open(filepath, 'wb') as wfid:
        for name in filenames:
            with open('%s/images/%s.jpg' % (data_dir, name), 'rb') as fid:
                img_bytes = fid.read()
            wfid.write(struct.pack('i', len(img_bytes)))
            wfid.write(img_bytes)


class BigFile(object):
    """The bigfile as an indexable sequence of byte strings, memory-mapped: the reference reads
    the whole file into a python list (13 GB for COCO train); here only the offset table lives in
    memory and every worker process shares the page cache."""

    def __init__(self, filepath, count):
        self._f = open(filepath, 'rb')
        size = os.fstat(self._f.fileno()).st_size
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ) if size else b''
        offs = np.zeros((count, 2), np.int64)
        pos = 0
        for i in range(count):
            if pos + 4 > size:
                raise ValueError("%s: truncated after %d of %d images" % (filepath, i, count))
            n = struct.unpack_from('i', self._mm, pos)[0]
            if n < 0 or pos + 4 + n > size:
                raise ValueError("%s: bad length %d for image %d" % (filepath, n, i))
            offs[i] = (pos + 4, n)
            pos += 4 + n
        self._offs = offs

    def __len__(self):
        return len(self._offs)

    def __getitem__(self, i):
        start, n = self._offs[i]
        return bytes(self._mm[start:start + n])


def read_imgs(data_dir, filenames, filepath):
    return BigFile(filepath, len(filenames))


def load_imgs_data(data_dir, split, filenames):
    filepath = os.path.join(data_dir, '%s_imgs.bigfile' % split)
    if not os.path.isfile(filepath):
        write_imgs(data_dir, filenames, filepath)
    return read_imgs(data_dir, filenames, filepath)


# ---------------------------------------------------------------------------------------------
# pickles
# ---------------------------------------------------------------------------------------------
class Vocab(object):
    """Stand-in for torchtext.vocab.Vocab when captions_glove.pickle is read without torchtext:
    the three attributes the training path uses are itos, stoi and vectors."""

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __len__(self):
        return len(self.itos)


def _default_unk_index():
    return 0


class _DataUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith('torchtext'):
            if name == 'Vocab':
                return Vocab
            if name == '_default_unk_index':
                return _default_unk_index
        return super().find_class(module, name)


def _load_pickle(filepath, encoding='ASCII'):
    with open(filepath, 'rb') as f:
        return _DataUnpickler(f, encoding=encoding).load()


def load_filenames(data_dir, split):
    filepath = '%s/%s/filenames.pickle' % (data_dir, split)
    if os.path.isfile(filepath):
        return _load_pickle(filepath)
    return []


def load_text_data(data_dir, split, train_names, test_names):
    filepath = os.path.join(data_dir, 'captions.pickle')
    if not os.path.isfile(filepath):
        raise FileNotFoundError("%s: build it with the reference's preprocessing" % filepath)
    x = _load_pickle(filepath)
    train_captions, test_captions, ixtoword, wordtoix = x[0], x[1], x[2], x[3]
    n_words = len(ixtoword)
    if split == 'train':
        return train_names, train_captions, ixtoword, wordtoix, n_words
    return test_names, test_captions, ixtoword, wordtoix, n_words


def load_glove_emb(data_dir, split, train_names, test_names):
    filepath = os.path.join(data_dir, 'captions_glove.pickle')
    if not os.path.isfile(filepath):
        raise FileNotFoundError("%s: build it with the reference's preprocessing" % filepath)
    x = _load_pickle(filepath)
    captions, vocab = (x[0], x[2]) if split == 'train' else (x[1], x[3])
    glove_embed = nn.Embedding(len(vocab.itos), cfg.TEXT.GLOVE_EMBEDDING_DIM)
    glove_embed.weight.data.copy_(torch.as_tensor(vocab.vectors))
    return captions, vocab.itos, vocab.stoi, glove_embed


def _tokens(text):
    out = []
    for t in _WORD.findall(text.lower()):
        t = t.encode('ascii', 'ignore').decode('ascii')
        if len(t) > 0:
            out.append(t)
    return out


def _category_lines(data_dir):
    with open('%s/categories.txt' % data_dir, "r") as f:
        return [ln for ln in f.read().split('\n') if len(ln) > 0]


def load_cats(data_dir, wordtoix):
    """-> (cats_dict: row -> caption-vocabulary ids of the name, cats_index_dict: coco id -> row)"""
    cats_dict, cats_index_dict = {}, {}
    for i, line in enumerate(_category_lines(data_dir)):
        cid, name = line.split(',', 1)
        cats_dict[i] = [wordtoix[t] for t in _tokens(name)]
        cats_index_dict[int(cid)] = i
    return cats_dict, cats_index_dict


def load_cat_label(data_dir, glove_wordtoix, device=None):
    """GloVe ids of the category names, sorted by name length (descending), plus the permutation
    back to file order (reference load.py:307-354)."""
    cat_labels = []
    for line in _category_lines(data_dir):
        raw = _WORD.findall(line.replace("\ufffd\ufffd", " ").lower())[1:]    # drop the leading id
        tokens = [t for t in (w.encode('ascii', 'ignore').decode('ascii') for w in raw) if len(t) > 0]
        if len(tokens) == 0:
            continue
        cat_labels.append([glove_wordtoix[t] for t in tokens])
    lens = [len(c) for c in cat_labels]
    arr = np.zeros((len(cat_labels), max(lens)), dtype=np.int64)
    for i, c in enumerate(cat_labels):
        arr[i, :len(c)] = c
    cat_labels = torch.from_numpy(arr)
    cat_label_lens = torch.LongTensor(lens)
    sorted_lens, sorted_idx = torch.sort(cat_label_lens, 0, True)
    sorted_labels = cat_labels[sorted_idx]
    _, resorted_idx = torch.sort(sorted_idx, 0, False)
    if device is not None:
        sorted_labels, sorted_lens, resorted_idx = (t.to(device) for t in (sorted_labels, sorted_lens, resorted_idx))
    return sorted_labels, sorted_lens, resorted_idx


def load_class_id(data_dir, total_num):
    path = data_dir + '/class_info.pickle'
    if os.path.isfile(path):
        return _load_pickle(path)
    return np.arange(total_num)


def load_anns_data(data_dir, split, postfix, ann_type, filenames, imsize, fmsize, cats_index_dict):
    filepath = os.path.join(data_dir, '%s%s' % (split, postfix))
    if not os.path.isfile(filepath):
        raise FileNotFoundError("%s: build it with the reference's preprocessing (pycocotools)" % filepath)
    return _load_pickle(filepath, encoding='latin1')[0]


# ---------------------------------------------------------------------------------------------
# per-sample assembly
# ---------------------------------------------------------------------------------------------
def get_caption(captions, glove_captions, sent_ix):
    """Caption ids and their GloVe ids, truncated to the shorter of the two, zero-padded to
    WORDS_NUM; longer captions keep WORDS_NUM randomly chosen words in order (np.random, like
    the reference load.py:114-140)."""
    sent_caption = np.asarray(captions[sent_ix]).astype('int64')
    sent_glove_caption = np.asarray(glove_captions[sent_ix]).astype('int64')
    n = min(len(sent_caption), len(sent_glove_caption))
    sent_caption, sent_glove_caption = sent_caption[:n], sent_glove_caption[:n]
    W = cfg.TEXT.WORDS_NUM
    x = np.zeros((W, 1), dtype='int64')
    glove_x = np.zeros((W, 1), dtype='int64')
    x_len = n
    if n <= W:
        x[:n, 0] = sent_caption
        glove_x[:n, 0] = sent_glove_caption
    else:
        ix = list(np.arange(n))
        np.random.shuffle(ix)
        ix = np.sort(ix[:W])
        x[:, 0] = sent_caption[ix]
        glove_x[:, 0] = sent_glove_caption[ix]
        x_len = W
    return x, glove_x, x_len


def _normalize_to_tensor(img):
    """transforms.ToTensor() + Normalize((.5,.5,.5), (.5,.5,.5)) of an RGB PIL image."""
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div_(255)
    return t.sub_(0.5).div_(0.5)


def decode_rgb(img_bytes):
    """The decoded image as a uint8 [H, W, 3] tensor: the `device_imgs` hand-over of trainDataset (the
    branch sizes are produced on the device, objgan_hip.ops.resize_pil_bilinear)."""
    from PIL import Image
    img = Image.open(io.BytesIO(img_bytes)).convert('RGB')
    return torch.from_numpy(np.array(img, dtype=np.uint8))


def get_imgs(img_bytes, imsize, normalize=None):
    """Decode and resize to every branch size (torchvision Resize((s, s)) on a PIL image is PIL's
    bilinear resize with its built-in antialiasing)."""
    from PIL import Image
    img = Image.open(io.BytesIO(img_bytes)).convert('RGB')
    norm = normalize if normalize is not None else _normalize_to_tensor
    ret = []
    for i in range(cfg.TREE.BRANCH_NUM):
        ret.append(norm(img.resize((imsize[i], imsize[i]), Image.BILINEAR)))
    return ret


def resize_mask(mask, size):
    """skimage.transform.resize(mask, [size, size]) for a 2-D float image with the package's
    defaults (order 1, mode 'reflect', anti-aliasing Gaussian of sigma (s-1)/2 when shrinking by
    s, clip to the input range), restated on scipy.ndimage -- skimage itself is built on the same
    two calls."""
    from scipy import ndimage as ndi
    mask = np.asarray(mask, dtype=np.float64)
    if mask.shape == (size, size):
        return mask.copy()
    factors = (mask.shape[0] / float(size), mask.shape[1] / float(size))
    img = mask
    if max(factors) > 1:
        sigma = tuple(max(0.0, (f - 1) / 2.0) for f in factors)
        img = ndi.gaussian_filter(img, sigma, cval=0, mode='mirror')
    out = ndi.zoom(img, (size / float(mask.shape[0]), size / float(mask.shape[1])), order=1,
                   mode='mirror', cval=0, grid_mode=True)
    return np.clip(out, mask.min(), mask.max())


def get_hmaps_rois(anno_dict, hmap_size, fmap_size, cats_index_dict, with_hmaps=True, with_resized_masks=True):
    """Layout maps (per-category sums of the instance masks), per-slot box masks at every branch
    size and at the 32x32 feature scale, and the box-map tensors of the shape generator
    (reference load.py:152-191).
    with_resized_masks=False (the `device_masks` hand-over of TrainDataset): no host resize at all -- bt_masks[0] carries
    the raw instance masks (the 64-px branch mask IS the raw mask: resizing 64 -> 64 is the identity), the other mask
    arrays are empty ([0, S, S]) and `prepare_data` produces them on the device (objgan_hip.ops.resize_masks)."""
    rois = anno_dict['rois']
    fm_rois = anno_dict['fm_rois']
    raw_masks = anno_dict['masks']
    num_rois = anno_dict['num_rois']
    nb, ncat, R = cfg.TREE.BRANCH_NUM, len(cats_index_dict), cfg.ROI.BOXES_NUM
    hmaps = [np.zeros((ncat, hmap_size[b], hmap_size[b]) if with_hmaps else (0,)) for b in range(nb)]
    if not with_resized_masks:
        if with_hmaps:
            raise ValueError("get_hmaps_rois: the device mask hand-over rebuilds the layout maps on the device too")
        bt_masks = [np.zeros((R, hmap_size[0], hmap_size[0]))] + [np.zeros((0, hmap_size[b], hmap_size[b])) for b in range(1, nb)]
        fm_bt_masks = np.zeros((0, hmap_size[0] // 2, hmap_size[0] // 2))
        for r in range(num_rois):
            m = np.asarray(raw_masks[r], dtype=np.float64)
            if m.shape != (hmap_size[0], hmap_size[0]):
                raise ValueError("get_hmaps_rois: raw instance masks must be %d x %d for the device hand-over, got %s"
                                 % (hmap_size[0], hmap_size[0], m.shape))
            bt_masks[0][r] = m
    else:
        bt_masks = [np.zeros((R, hmap_size[b], hmap_size[b])) for b in range(nb)]
        fm_bt_masks = np.zeros((R, hmap_size[0] // 2, hmap_size[0] // 2))
        for r in range(num_rois):
            mask = raw_masks[r]
            cat = int(rois[0][r, 4])
            fm_bt_masks[r] = resize_mask(mask, hmap_size[0] // 2)
            for b in range(nb):
                re_mask = resize_mask(mask, hmap_size[b])
                bt_masks[b][r] = re_mask
                if with_hmaps:
                    hmaps[b][cat] += re_mask
    bbox_maps_fwd = np.zeros((R, ncat, hmap_size[0], hmap_size[0]))
    bbox_maps_bwd = np.zeros((R, ncat, hmap_size[0], hmap_size[0]))
    bbox_fmaps = np.zeros((R, fmap_size, fmap_size))
    if num_rois > 0:
        for r in range(num_rois):
            bbox_maps_fwd[r, int(rois[0][r, 4])] = anno_dict['bbox maps'][r]
        bbox_maps_bwd = bbox_maps_fwd[::-1].copy()
        bbox_fmaps[:num_rois] = anno_dict['bbox fmaps']
    return hmaps, bbox_maps_fwd, bbox_maps_bwd, bbox_fmaps, rois, fm_rois, num_rois, bt_masks, fm_bt_masks


def install_torchtext_stub():
    """Make `torchtext.vocab.Vocab` resolvable for plain pickle.load callers (tools, tests)."""
    import sys
    if 'torchtext' not in sys.modules:
        tt = types.ModuleType('torchtext')
        ttv = types.ModuleType('torchtext.vocab')
        ttv.Vocab = Vocab
        ttv._default_unk_index = _default_unk_index
        tt.vocab = ttv
        sys.modules['torchtext'] = tt
        sys.modules['torchtext.vocab'] = ttv
