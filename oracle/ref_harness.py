"""ORACLE (test infrastructure): run the UNMODIFIED reference on CPU.

Imports /root/reference/image_generation/{model,GlobalAttention,miscc/*}.py as they are (nothing
is copied into this repo) through the small shim set the reference needs on PyTorch 2.x / numpy 2
(SURVEY.md section 8c): stub packages for the three missing imports, numpy aliases, the ROIAlign
extension replaced by the reference's own C loop (oracle/_ref), ByteTensor masks, the
`Variable(...).data.resize_` idiom, and CUDA copy semantics for the rois (trap 4).

Only usable where /root/reference exists (the build container).  It is what pins the restatements
in oracle/torch_ref.py and generates tests/golden/* (tests/golden/make_golden.py); the GPU box
never imports this module.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF_ROOT = os.environ.get("OBJGAN_REFERENCE", "/root/reference/image_generation")


def available():
    return os.path.isdir(REF_ROOT)


class _AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            if isinstance(v, dict):
                self[key] = _AttrDict(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _install_stub_packages(inception_factory):
    easydict = types.ModuleType("easydict")
    easydict.EasyDict = _AttrDict
    sys.modules["easydict"] = easydict
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.inception_v3 = inception_factory
    tv.models = tvm
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.transforms": tvt})
    sk = types.ModuleType("skimage")
    skt = types.ModuleType("skimage.transform")
    skt.resize = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("skimage stub"))
    sk.transform = skt
    sys.modules.update({"skimage": sk, "skimage.transform": skt})


class _RefRoIAlignAvg(torch.nn.Module):
    """models/roi_align/modules/roi_align.py:18-29 with the extension call replaced by the
    reference's own ROIAlignForwardCpu (oracle/_ref) and the CUDA-kernel backward semantics."""

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super().__init__()
        self.ah, self.aw, self.scale = int(aligned_height), int(aligned_width), float(spatial_scale)

    def forward(self, features, rois):
        return F.avg_pool2d(_RefRoiFn.apply(features, rois, self.ah + 1, self.aw + 1, self.scale),
                            kernel_size=2, stride=1)


class _RefRoiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, ah, aw, scale):
        from . import roi
        out = roi.reference_forward(features.detach().numpy(), rois.detach().numpy(), ah, aw, scale)
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(features.shape), scale)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        from . import roi
        (rois,) = ctx.saved_tensors
        shape, scale = ctx.meta
        return torch.from_numpy(roi.backward(g.contiguous().numpy(), rois.numpy(), shape, scale)), None, None, None, None


_LOADED = None


def load_reference(branch_num=3, batch_size=4):
    """-> namespace with the reference modules: .model, .GlobalAttention, .losses, .utils, .cfg"""
    global _LOADED
    if _LOADED is not None:
        _LOADED.cfg.TREE.BRANCH_NUM = branch_num
        _LOADED.cfg.TRAIN.BATCH_SIZE = batch_size
        return _LOADED
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "obj-gan_amd")
    if pkg not in sys.path:
        sys.path.append(pkg)
    from oracle import torch_encoders    # the plain-PyTorch Inception-v3 stand-in (no torchvision here)
    _install_stub_packages(torch_encoders.inception_v3)
    np.int = int
    np.float = float
    # the product package uses the same top-level module names as the reference: load the
    # reference under a private import context and restore sys.modules afterwards
    names = ("model", "GlobalAttention", "miscc", "miscc.config", "miscc.utils", "miscc.losses",
             "models", "models.roi_align", "models.roi_align.modules", "models.roi_align.modules.roi_align",
             "trainer")
    saved = {n: sys.modules.pop(n) for n in names if n in sys.modules}
    saved_path = list(sys.path)
    try:
        sys.path = [REF_ROOT] + [p for p in sys.path if os.path.abspath(p) != os.path.abspath(pkg)]
        for n in ("models", "models.roi_align", "models.roi_align.modules"):
            sys.modules[n] = types.ModuleType(n)
        shim = types.ModuleType("models.roi_align.modules.roi_align")
        shim.RoIAlignAvg = _RefRoIAlignAvg
        sys.modules["models.roi_align.modules.roi_align"] = shim
        from miscc.config import cfg
        cfg.CUDA = False
        cfg.GPU_IDS = [-1]
        cfg.TRAIN.BATCH_SIZE = batch_size
        cfg.TREE.BRANCH_NUM = branch_num
        import model
        import GlobalAttention
        from miscc import losses, utils

        class _SelfData(torch.Tensor):
            @property
            def data(self):
                return self

        model.Variable = lambda t, requires_grad=False: (
            t.as_subclass(_SelfData) if (t.dim() == 1 and t.numel() == 1) else t)
        for cls in (model.OBJ_SS_D_NET, model.OBJ_LS_D_NET):
            orig = cls.forward

            def fwd(self, x, s, fm_rois, num_rois, img_size=512, _o=orig):
                return _o(self, x, s, fm_rois.clone(), num_rois, img_size)
            cls.forward = fwd

        class _TorchProxy(object):
            def __getattr__(self, k):
                return getattr(torch, k)

            @staticmethod
            def ByteTensor(a):
                return torch.from_numpy(np.asarray(a)).bool()
        losses.torch = _TorchProxy()
        ns = types.SimpleNamespace(model=model, GlobalAttention=GlobalAttention, losses=losses,
                                   utils=utils, cfg=cfg)
        ns._modules = {n: sys.modules.get(n) for n in names}
    finally:
        sys.path = saved_path
        for n in names:
            sys.modules.pop(n, None)
        sys.modules.update(saved)
    _LOADED = ns
    return ns


def seeded_state_(module, seed, scale=None):
    """Fill a module's parameters/buffers deterministically BY KEY (sorted), so that the reference
    module and the product module -- which share state-dict keys and shapes -- get identical weights
    without copying tensors between them.  Conv/linear weights ~ N(0, 1/fan_in) * gain, BN weights
    ~ N(1, 0.02), biases ~ N(0, 0.02); running stats keep their defaults."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd.keys()):
            t = sd[key]
            if not t.dtype.is_floating_point or "running_" in key:
                continue
            if t.dim() > 1:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (1.5 / fan_in) ** 0.5)
            elif key.endswith("weight"):
                t.copy_(1.0 + 0.02 * torch.randn(t.shape, generator=g))
            else:
                t.copy_(0.02 * torch.randn(t.shape, generator=g))
    return module
