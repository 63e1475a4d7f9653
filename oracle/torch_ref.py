"""ORACLE (test infrastructure): plain-PyTorch CPU fp32 restatement of the hot-path operators.

Each function restates what the reference computes at the cited lines, written functionally
(tensors in, tensors out) so the same code checks both single kernels and whole blocks.  The
restatement is itself pinned against the unmodified reference modules in
tests/test_oracle_cpu.py (this container) and through tests/golden/*.pt on the GPU box.
"""
import torch
import torch.nn.functional as F


# ---- convolution variants (reference model.py:30-60, 63-81, 599-603, 999-1017) -------------------
def conv2d(x, w, bias=None, stride=1, pad=0, pad_mode="zeros", upsample=False, act=None):
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if pad_mode == "reflect":
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
        y = F.conv2d(x, w, bias, stride=stride, padding=0)
    else:
        y = F.conv2d(x, w, bias, stride=stride, padding=pad)
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2)
    elif act == "tanh":
        y = torch.tanh(y)
    elif act == "sigmoid":
        y = torch.sigmoid(y)
    return y


def glu(x):
    nc = x.size(1) // 2
    return x[:, :nc] * torch.sigmoid(x[:, nc:])


def norm_act(x, gamma=None, beta=None, residual=None, running_mean=None, running_var=None,
             per_channel=False, mode=None, eps=1e-5, momentum=0.1):
    """BatchNorm (batch statistics, running stats updated in place) or InstanceNorm(affine=False),
    then GLU / LeakyReLU(0.2), then + residual."""
    if per_channel:
        shape = x.shape
        x2 = x.reshape(shape[0], shape[1], -1)
        y = F.batch_norm(x2, running_mean, running_var, gamma, beta, True, momentum, eps).reshape(shape)
    else:
        y = F.instance_norm(x, eps=eps)
    if mode == "glu":
        y = glu(y)
    elif mode == "lrelu":
        y = F.leaky_relu(y, 0.2)
    if residual is not None:
        y = y + residual
    return y


# ---- attention (reference GlobalAttention.py) -----------------------------------------------------
def attn_general(x, src, mask=None):
    """GlobalAttentionGeneral.forward after the 1x1 projection (GlobalAttention.py:95-120),
    INCLUDING the reference's mask tiling: mask.repeat(queryL, 1) applied to rows ordered (b, q)."""
    B, idf, ih, iw = x.shape
    Q = ih * iw
    L = src.shape[2]
    target_t = x.reshape(B, idf, Q).transpose(1, 2)                 # B x Q x idf
    attn = torch.bmm(target_t, src).reshape(B * Q, L)
    if mask is not None:
        attn = attn.masked_fill(mask.bool().repeat(Q, 1), -float("inf"))
    attn = torch.softmax(attn, dim=-1).reshape(B, Q, L).transpose(1, 2)   # B x L x Q
    wc = torch.bmm(src, attn)
    return wc.reshape(B, idf, ih, iw), attn.reshape(B, L, ih, iw)


def attn_bu(tgt, ctx1, src, mask=None, normalize=True, eps=1e-8):
    """GlobalBUAttentionGeneral.forward after the 1x1 projection (GlobalAttention.py:153-179)."""
    B, d2, ih, iw = tgt.shape
    R = ih * iw
    L = src.shape[2]
    target_t = tgt.reshape(B, d2, R).transpose(1, 2)                # B x R x d2
    attn = torch.bmm(target_t, ctx1)                                # B x R x L
    if normalize:
        nt = torch.norm(target_t, 2, dim=2, keepdim=True)
        nc = torch.norm(ctx1, 2, dim=1, keepdim=True)
        attn = attn / (nt * nc).clamp(min=eps)
    attn = attn.reshape(B * R, L)
    if mask is not None:
        attn = attn.masked_fill(mask.bool().repeat(R, 1), -float("inf"))
    attn = torch.softmax(attn, dim=-1).reshape(B, R, L).transpose(1, 2)   # B x L x R
    wc = torch.bmm(src, attn)
    return wc.reshape(B, -1, ih, iw), attn.reshape(B, L, ih, iw)


def masked_max(f, m, ih, iw):
    """pprocess_bt_attns (miscc/utils.py:401-413) with a [B, R, ih, iw] mask shared by all channels."""
    B, num, R = f.shape[0], f.shape[1], f.shape[2]
    prod = f.reshape(B, num, R, 1, 1) * m.reshape(B, 1, R, ih, iw)
    return prod.max(dim=2)[0]


def func_attention(query, context, gamma1):
    """DAMSM attention (GlobalAttention.py:32-70)."""
    B, ndf, queryL = query.shape
    ih, iw = context.shape[2], context.shape[3]
    S = ih * iw
    ctx = context.reshape(B, ndf, S)
    attn = torch.bmm(ctx.transpose(1, 2), query)                    # B x S x queryL
    attn = torch.softmax(attn.reshape(B * S, queryL), dim=-1).reshape(B, S, queryL)
    attn = attn.transpose(1, 2).reshape(B * queryL, S) * gamma1
    attn = torch.softmax(attn, dim=-1).reshape(B, queryL, S)
    wc = torch.bmm(ctx, attn.transpose(1, 2))
    return wc, attn.reshape(B, queryL, ih, iw)


# ---- resize / pooling / optimiser ------------------------------------------------------------------
def bilinear_resize(x, oh, ow):
    return F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True)


def avgpool2s1(x):
    return F.avg_pool2d(x, kernel_size=2, stride=1)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    """torch.optim.Adam single-tensor update (no weight decay / amsgrad); returns new (p, m, v)."""
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
