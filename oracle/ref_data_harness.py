"""ORACLE (test infrastructure): the UNMODIFIED reference data path -- miscc/load.py and
trainDataset.py -- imported from /root/reference on CPU.

The reference loader pulls in packages that are not installed here.  They are replaced by the
stubs below; each stub either restates a few lines of well-defined behaviour or fails loudly:

  nltk.tokenize.RegexpTokenizer(r'\\w+')   re.findall with nltk's flags (UNICODE | MULTILINE | DOTALL)
  torchvision.transforms                   Compose / ToTensor / Normalize / Resize on PIL images
                                           (Resize = PIL bilinear, what torchvision calls)
  skimage.transform.resize                 the scipy.ndimage restatement in obj-gan_amd/miscc/load.py
                                           (THIRD-PARTY ARITHMETIC, PARITY UNPINNED: skimage is absent, so
                                           mask resizing is compared restatement-against-restatement)
  torchtext, spacy, pycocotools, skimage.io   only used by the preprocessing that builds the pickles:
                                           placeholders that raise when touched
  torchtext.vocab.Vocab                    plain attribute holder, so that captions_glove.pickle round-trips

Only usable where /root/reference exists (the build container).  tests/golden/make_golden_data.py
uses it to run the reference loader over the tiny data directory under tests/golden/data_tiny/ and
to store what it returns; tests compare the product loader against that file.
"""
import os
import re
import sys
import types

import numpy as np
import torch

from . import ref_harness

REF_ROOT = ref_harness.REF_ROOT


def _raising(name):
    def f(*a, **k):
        raise RuntimeError("%s is preprocessing-only and stubbed in the oracle harness" % name)
    return f


class _RegexpTokenizer(object):
    def __init__(self, pattern):
        self._re = re.compile(pattern, re.UNICODE | re.MULTILINE | re.DOTALL)

    def tokenize(self, text):
        return self._re.findall(text)


def _transforms_module():
    from PIL import Image
    m = types.ModuleType("torchvision.transforms")

    class Compose(object):
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToTensor(object):
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std

    class Resize(object):
        def __init__(self, size):
            self.size = size

        def __call__(self, img):
            return img.resize((self.size[1], self.size[0]), Image.BILINEAR)

    m.Compose, m.ToTensor, m.Normalize, m.Resize = Compose, ToTensor, Normalize, Resize
    return m


_LOADED = None


def load_reference_data(branch_num=3):
    """-> namespace with the reference modules .load (miscc/load.py), .trainDataset, .trainer, .cfg"""
    global _LOADED
    if _LOADED is not None:
        _LOADED.cfg.TREE.BRANCH_NUM = branch_num
        return _LOADED
    ns = ref_harness.load_reference(branch_num=branch_num)
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "obj-gan_amd")
    if pkg not in sys.path:
        sys.path.append(pkg)
    from miscc import load as product_load           # the scipy restatement of skimage's resize

    stubs = {}
    nltk = types.ModuleType("nltk")
    nltk_tok = types.ModuleType("nltk.tokenize")
    nltk_tok.RegexpTokenizer = _RegexpTokenizer
    nltk.tokenize = nltk_tok
    stubs.update({"nltk": nltk, "nltk.tokenize": nltk_tok})
    coco_pkg = types.ModuleType("pycocotools")
    coco_mod = types.ModuleType("pycocotools.coco")
    coco_mod.COCO = _raising("pycocotools.coco.COCO")
    coco_pkg.coco = coco_mod
    stubs.update({"pycocotools": coco_pkg, "pycocotools.coco": coco_mod})
    tt = types.ModuleType("torchtext")
    ttv = types.ModuleType("torchtext.vocab")

    class Vocab(object):                 # pickles as torchtext.vocab.Vocab, like the released files
        def __len__(self):
            return len(self.itos)
    Vocab.__module__ = "torchtext.vocab"
    Vocab.__qualname__ = "Vocab"
    ttv.Vocab = Vocab
    ttv._default_unk_index = product_load._default_unk_index
    ttd = types.ModuleType("torchtext.data")
    ttd.Field = _raising("torchtext.data.Field")
    ttd.TabularDataset = _raising("torchtext.data.TabularDataset")
    tt.vocab, tt.data = ttv, ttd
    stubs.update({"torchtext": tt, "torchtext.vocab": ttv, "torchtext.data": ttd})
    spacy = types.ModuleType("spacy")
    spacy.load = lambda name: types.SimpleNamespace(tokenizer=_raising("spacy tokenizer"))
    stubs["spacy"] = spacy
    sk = types.ModuleType("skimage")
    skt = types.ModuleType("skimage.transform")
    skt.resize = lambda image, output_shape, **kw: product_load.resize_mask(image, int(output_shape[0]))
    skio = types.ModuleType("skimage.io")
    skio.imread = _raising("skimage.io.imread")
    sk.transform, sk.io = skt, skio
    stubs.update({"skimage": sk, "skimage.transform": skt, "skimage.io": skio})
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    tvt = _transforms_module()
    tv.transforms = tvt
    stubs.update({"torchvision": tv, "torchvision.transforms": tvt})

    names = ("miscc", "miscc.config", "miscc.utils", "miscc.load", "miscc.losses", "trainDataset", "trainer",
             "model", "GlobalAttention",
             "models", "models.roi_align", "models.roi_align.modules", "models.roi_align.modules.roi_align")
    saved = {n: sys.modules.pop(n) for n in list(names) + list(stubs) if n in sys.modules}
    saved_path = list(sys.path)
    try:
        sys.modules.update(stubs)
        # the reference modules loaded by ref_harness (same cfg object, same miscc.utils)
        for n, mod in ns._modules.items():
            if mod is not None and n in names:
                sys.modules[n] = mod
        sys.path = [REF_ROOT] + [p for p in sys.path if os.path.abspath(p) != os.path.abspath(pkg)]
        import miscc.load as ref_load
        import trainDataset as ref_ds
        sys.modules.pop("trainer", None)
        import trainer as ref_trainer
        out = types.SimpleNamespace(load=ref_load, trainDataset=ref_ds, trainer=ref_trainer, cfg=ns.cfg,
                                    _stubs=stubs)
    finally:
        sys.path = saved_path
        for n in list(names) + list(stubs):
            sys.modules.pop(n, None)
        sys.modules.update(saved)
    _LOADED = out
    return out


class active(object):
    """`with active(ns):` -- the stub packages visible again (the reference unpickles
    torchtext.vocab.Vocab objects at run time, not only at import time)."""

    def __init__(self, ns):
        self.ns = ns

    def __enter__(self):
        self.saved = {n: sys.modules.get(n) for n in self.ns._stubs}
        sys.modules.update(self.ns._stubs)
        return self.ns

    def __exit__(self, *exc):
        for n, m in self.saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
        return False
