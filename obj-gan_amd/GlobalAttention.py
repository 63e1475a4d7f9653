"""Word- and object-driven attention of the image generator, on MI355X kernels.

Drop-in for the reference module of the same name (reference
image_generation/GlobalAttention.py:26-181): `func_attention`, `GlobalAttentionGeneral`,
`GlobalBUAttentionGeneral`, same constructor arguments, `applyMask`, forward signatures, return
tuples and state-dict keys (`conv_context.weight`).  The bmm -> mask -> softmax -> bmm chains are
single fused HIP kernels (objgan_hip/ops.py, csrc/attention.hip); the 1x1 context projection
runs on the MFMA implicit-GEMM kernel.
"""
import torch
import torch.nn as nn

from miscc.config import cfg
from objgan_hip import ops


def conv1x1(in_planes, out_planes):
    """1x1 convolution, no bias (parameter holder; executed by ops.conv2d)."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=False)


def func_attention(query, context, gamma1):
    """DAMSM attention, AttnGAN Eq. 7-9 (reference GlobalAttention.py:32-70).

    query   : batch x ndf x queryL
    context : batch x ndf x ih x iw
    returns (weightedContext batch x ndf x queryL, attn batch x queryL x ih x iw)

    Layouts are chosen so that no transpose is ever materialised:
      scores[b, q, s] = sum_c query[b, c, q] * context[b, c, s]
      a1 = softmax over q (the words)     a2 = softmax over s of gamma1 * a1
      weightedContext[b, c, q] = sum_s context[b, c, s] * a2[b, q, s]
    """
    B, ndf, queryL = query.shape
    ih, iw = context.shape[2], context.shape[3]
    S = ih * iw
    wcs, attns = [], []
    for b in range(B):
        # queries of sample b act as a [queryL x ndf] 1x1 filter bank over its own context map
        wq = query[b].t().reshape(queryL, ndf, 1, 1)
        scores = ops.conv2d(context[b:b + 1], wq)                       # 1 x queryL x ih x iw
        a1 = ops.softmax_strided(scores.reshape(1, queryL, S), 1)       # over the words
        a2 = ops.softmax_strided(a1, 2, scale=float(gamma1))            # over the regions
        # weightedContext = context[b] (ndf x S) @ a2^T (S x queryL): 1x1 conv with filter bank
        # context[b] over the "image" a2^T laid out as [1, S, queryL, 1]
        wctx = context[b].reshape(ndf, S, 1, 1)
        wc = ops.conv2d(a2.transpose(1, 2).reshape(1, S, queryL, 1), wctx)  # 1 x ndf x queryL x 1
        wcs.append(wc.reshape(1, ndf, queryL))
        attns.append(a2.reshape(1, queryL, ih, iw))
    return torch.cat(wcs, 0), torch.cat(attns, 0)


class GlobalAttentionGeneral(nn.Module):
    """Word -> pixel attention (reference GlobalAttention.py:73-122)."""

    def __init__(self, idf, cdf):
        super(GlobalAttentionGeneral, self).__init__()
        self.conv_context = conv1x1(cdf, idf)
        self.sm = nn.Softmax(dim=-1)      # kept for attribute parity; the softmax is fused
        self.mask = None

    def applyMask(self, mask):
        self.mask = mask  # batch x sourceL

    def forward(self, input, context):
        """input: batch x idf x ih x iw;  context: batch x cdf x sourceL."""
        # sourceT = conv_context(context): batch x idf x sourceL
        sourceT = ops.conv2d(context.unsqueeze(3), self.conv_context.weight).squeeze(3)
        # scores + (mis-tiled, see csrc/attention.hip) mask + softmax + weighted sum, one kernel
        weightedContext, attn = ops.attn_general(input, sourceT, self.mask)
        return weightedContext, attn


class GlobalBUAttentionGeneral(nn.Module):
    """Word -> object ("bottom-up") attention (reference GlobalAttention.py:125-181)."""

    def __init__(self, idf, cdf):
        super(GlobalBUAttentionGeneral, self).__init__()
        self.conv_context = conv1x1(cdf, idf)
        self.sm = nn.Softmax(dim=-1)
        self.mask = None
        self.eps = 1e-8

    def applyMask(self, mask):
        self.mask = mask  # batch x sourceL

    def forward(self, input, context1, context2):
        """input: batch x idf2 x ih x iw (label features); context1: batch x idf2 x sourceL
        (GloVe words); context2: batch x cdf x sourceL (word embeddings)."""
        sourceT = ops.conv2d(context2.unsqueeze(3), self.conv_context.weight).squeeze(3)
        weightedContext, attn = ops.attn_bu(input, context1, sourceT, self.mask,
                                            normalize=bool(cfg.TRAIN.BUATTN_NORM), eps=self.eps)
        return weightedContext, attn
