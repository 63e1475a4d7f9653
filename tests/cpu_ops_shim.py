"""TEST INFRASTRUCTURE: the `objgan_hip.ops` API with every kernel replaced by its CPU definition
(oracle/torch_ref.py, oracle/roi.py, torch.nn.functional).  Installed by tests only
(`install(monkeypatch)`), it lets the host side of the product -- model.py, GlobalAttention.py,
miscc/losses.py, miscc/utils.py, trainer.py with its flat arenas -- run on CPU tensors, so that the
wiring around the kernels is checked against the oracle without a GPU.  The product never imports
this module; on a GPU box the same host code runs on libobjgan_hip.so (and refuses CPU tensors)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import torch_ref as tr, roi as oroi

conv2d = tr.conv2d
norm_act = tr.norm_act
attn_general = tr.attn_general
attn_bu = tr.attn_bu
masked_max = tr.masked_max
bilinear_resize = tr.bilinear_resize
avgpool2s1 = tr.avgpool2s1


def _act(y, act):
    if act in (None, "none"):
        return y
    if act == "lrelu":
        return F.leaky_relu(y, 0.2)
    if act == "relu":
        return F.relu(y)
    return torch.tanh(y) if act == "tanh" else torch.sigmoid(y)


def conv2d_cat(x1, x2, w, stride=1, pad=0, act=None):
    return tr.conv2d(torch.cat([x1, x2], dim=1), w, None, stride, pad, "zeros", False, act)


def conv2d_frozen(x, w, bias=None, stride=1, pad=(0, 0), act=None):
    return _act(F.conv2d(x, w, bias, stride, pad), act)


def linear(x, w, bias=None, act=None):
    return _act(F.linear(x, w, bias), act)


def norm_act_eval(x, gamma, beta, running_mean, running_var, mode=None, eps=1e-5):
    y = F.batch_norm(x, running_mean, running_var, gamma, beta, False, 0.0, eps)
    if mode == "glu":
        return tr.glu(y)
    return F.leaky_relu(y, 0.2) if mode == "lrelu" else y


def softmax_strided(x, dim, scale=1.0, lens=None, rowvalid=None):
    d = dim % x.dim()
    outer = int(np.prod(x.shape[:d])) if d else 1
    n = x.shape[d]
    xs = (scale * x).reshape(outer, n, -1)
    keep = torch.ones(outer, n, 1, dtype=torch.bool)
    if lens is not None:
        span = lens.to(torch.long).clamp(max=n)[torch.arange(outer) % lens.numel()]
        keep = torch.arange(n).view(1, n, 1) < span.view(outer, 1, 1)
    y = torch.softmax(xs.masked_fill(~keep, float("-inf")), dim=1)
    y = torch.where(keep, y, torch.zeros_like(y))
    if rowvalid is not None:
        y = y * rowvalid.reshape(outer, 1, 1).to(y.dtype)
    return y.reshape(x.shape)


class _RoiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rois, ah, aw, scale):
        out = oroi.forward(features.detach().numpy(), rois.detach().numpy(), ah, aw, scale)
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(features.shape), scale)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, scale = ctx.meta
        return torch.from_numpy(oroi.backward(g.contiguous().numpy(), rois.numpy(), shape, scale)), None, None, None, None


def roi_align(features, rois, aligned_height, aligned_width, spatial_scale):
    return _RoiFn.apply(features, rois, int(aligned_height), int(aligned_width), float(spatial_scale))


def lstm_bidir_forward(table, captions, lens, wt_ih, wt_hh, b_ih, b_hh, max_len):
    """The fused embedding + bidirectional LSTM kernel's contract on the oracle's cell loops
    (wt_* are the transposed [2][I or H][4H] copies the product hands to the kernel)."""
    from oracle import torch_model as tm
    sd = {"encoder.weight": table}
    for d, suffix in enumerate(("", "_reverse")):
        sd["rnn.weight_ih_l0" + suffix] = wt_ih[d].t()
        sd["rnn.weight_hh_l0" + suffix] = wt_hh[d].t()
        sd["rnn.bias_ih_l0" + suffix] = b_ih[d]
        sd["rnn.bias_hh_l0" + suffix] = b_hh[d]
    return tm.rnn_encoder_forward(sd, captions, lens, int(max_len))


def lift_stem_conv(seg, w, bias, size):
    """definition: the reference formulation (model.py:1217-1226)"""
    up = F.interpolate(seg, size=(size, size), mode="bilinear", align_corners=True)
    return F.conv2d(F.pad(up, (1, 1, 1, 1), mode="reflect"), w, bias)


def bmm(A, B):
    return torch.bmm(A, B)


def bce_const(prob, target):
    return F.binary_cross_entropy(prob, torch.full_like(prob, float(target)))


def repack_arena(epoch_cell):
    return 0


def max_pool2d(x, kernel_size, stride):
    return F.max_pool2d(x, kernel_size, stride)


def avg_pool2d(x, kernel_size, stride=None, padding=0):
    return F.avg_pool2d(x, kernel_size, stride, padding)


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, n=None):
    n = p.numel() if n is None else int(n)
    pn, mn, vn = tr.adam_step(p[:n], g[:n] * grad_scale, m[:n], v[:n], lr, beta1, beta2, eps, step)
    p[:n].copy_(pn)
    m[:n].copy_(mn)
    v[:n].copy_(vn)


def adam_step_gated_(p, g, m, v, lr, beta1, beta2, eps, state, flag, coef, grad_scale=1.0, n=None):
    """CPU definition of objgan_adam_step_gated: the update (and the step counter in `state`) only
    moves when the flag is positive."""
    if float(flag.reshape(-1)[0]) <= 0:
        coef[0] = 0.0
        return
    state[0] += 1.0
    state[1] *= beta1
    state[2] *= beta2
    coef[0] = 1.0
    if grad_scale < 0:                       # divide by the flag (ranks that contributed a gradient)
        grad_scale = 1.0 / float(flag.reshape(-1)[0])
    adam_step_(p, g, m, v, lr, beta1, beta2, eps, int(state[0].item()), grad_scale=grad_scale, n=n)


def ema_update_(avg, p, decay):
    avg.mul_(decay).add_(p, alpha=1.0 - decay)


def resize_pil_bilinear(images, sizes, device):
    from oracle import pil_resize as pr
    arrs = [np.asarray(im.numpy() if torch.is_tensor(im) else im, dtype=np.uint8) for im in images]
    return [torch.from_numpy(np.stack([pr.to_normalized_chw(pr.resize_rgb8(a, int(S))) for a in arrs])).to(device)
            for S in sizes]


def jpeg_parse(files):
    """the real host-only parser of the C-ABI (no GPU call)"""
    from objgan_hip import ops as real_ops
    return real_ops.jpeg_parse(files)


def images_to_device(items, device, cache=None, keys=None):
    """CPU definition of the `device_jpeg` hand-over: JPEG files through the numpy oracle (bit for bit Pillow), host-decoded
    images as they are -> (list of uint8 [H, W, 3] arrays, None, heights, widths)"""
    from oracle import jpeg_oracle as J
    arrs = []
    for it in items:
        a = it.numpy() if torch.is_tensor(it) else np.asarray(it)
        arrs.append(J.decode(a.tobytes()) if a.ndim == 1 else np.asarray(a, np.uint8))
    return arrs, None, [a.shape[0] for a in arrs], [a.shape[1] for a in arrs]


def resize_pil_bilinear_device(src, offs, hs, ws, sizes):
    return resize_pil_bilinear(src, sizes, "cpu")


def resize_masks(masks, sizes):
    from oracle import mask_resize as mr
    m = masks.detach().cpu().numpy().astype(np.float64)
    flat = m.reshape((-1,) + m.shape[-2:])
    return [torch.from_numpy(np.stack([mr.resize_mask(x, int(S)) for x in flat]).reshape(m.shape[:-2] + (int(S), int(S))))
            .to(masks.device) for S in sizes]


class direct_wgrad_scope(object):
    """the product's switch for weight gradients straight into the arena: nothing to switch on the CPU definitions
    (autograd accumulates)"""

    def __init__(self, on=True):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def set_conv_math(mode):
    pass


def get_conv_math():
    return "fp32"


API = ("conv2d", "conv2d_cat", "conv2d_frozen", "linear", "norm_act", "norm_act_eval", "attn_general", "attn_bu", "masked_max",
       "softmax_strided", "roi_align", "avgpool2s1", "bilinear_resize", "lstm_bidir_forward", "adam_step_",
       "adam_step_gated_", "ema_update_", "max_pool2d", "avg_pool2d", "lift_stem_conv", "bmm", "repack_arena", "bce_const",
       "resize_pil_bilinear", "resize_masks", "jpeg_parse", "images_to_device", "resize_pil_bilinear_device")


def install(monkeypatch):
    """Point the `ops` name of every host module at this shim (and undo it after the test)."""
    import sys
    import types
    import GlobalAttention
    import model
    import trainer
    import encoders
    import trainDataset
    from miscc import losses, utils
    shim = types.SimpleNamespace(**{k: globals()[k] for k in API})
    shim.set_conv_math, shim.get_conv_math = set_conv_math, get_conv_math
    shim.direct_wgrad_scope = direct_wgrad_scope
    for mod in (model, GlobalAttention, trainer, losses, utils, encoders, trainDataset):
        if hasattr(mod, "ops"):
            monkeypatch.setattr(mod, "ops", shim)
    for name in ("models.roi_align.modules.roi_align", "models.roi_align.functions.roi_align"):
        m = sys.modules.get(name)
        if m is not None and hasattr(m, "ops"):
            monkeypatch.setattr(m, "ops", shim)
    monkeypatch.setattr(utils, "_HOST", [])
    return shim


class _Patch(object):
    """monkeypatch.setattr without pytest (spawned worker processes)."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def install_plain():
    return install(_Patch())
