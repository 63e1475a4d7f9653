"""Investigation: per-operator error of the bf16x3 and fp32 conv modes against fp64 on the ACTUAL tensors of the
ShpD128 loss (forward, data gradient, weight gradient of every convolution).  (GPU box)"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F
import model as M
import synth_batch
from oracle import ref_harness as rh
from miscc.config import cfg
from miscc.losses import shpD_loss
from miscc.utils import form_clabels_feat
from objgan_hip import ops
dev = torch.device("cuda:0")
gold = torch.load(os.path.join(ROOT, "tests", "golden", "step_b2.pt"), weights_only=False)
cfg.TREE.BRANCH_NUM = 3; cfg.TRAIN.BATCH_SIZE = 2
s = gold["seeds"]
G = rh.seeded_state_(M.G_NET(80), s["G"]).to(dev).train()
net = rh.seeded_state_(M.SHP_D_NET128(80), s["shp"] + 1).to(dev).train()
b = synth_batch.make_batch(2, seed=s["batch"], device=dev)
G.ca_net.fixed_eps = b["ca_eps"]
cl = form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
with torch.no_grad():
    fake = G(b["noise"], b["sent_emb"], b["words_embs"], b["glove_words_embs"], cl, b["mask"], b["hmaps"],
             b["rois"], b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"], int(b["num_rois"].max()))[0]
calls = []
orig = (ops._conv_fwd, ops._conv_dgrad, ops._conv_wgrad)
def rec_fwd(x, w, bias, stride, pad, refl, upsample, act):
    y = orig[0](x, w, bias, stride, pad, refl, upsample, act)
    if act in (None, "none") and not upsample:
        calls.append(("fwd", x.detach().clone(), w.detach().clone(), None, (stride, pad, refl), y.detach().clone()))
    return y
def rec_dgrad(g, w, N, Cin, H, W, stride, pad, refl, upsample, cacheable=True):
    dx = orig[1](g, w, N, Cin, H, W, stride, pad, refl, upsample, cacheable)
    if not upsample:
        calls.append(("dgrad", g.detach().clone(), w.detach().clone(), (N, Cin, H, W), (stride, pad, refl), dx.detach().clone()))
    return dx
def rec_wgrad(x, g, Cout, k, stride, pad, refl, upsample):
    dw = orig[2](x, g, Cout, k, stride, pad, refl, upsample)
    if not upsample:
        calls.append(("wgrad", x.detach().clone(), g.detach().clone(), (Cout, k), (stride, pad, refl), dw.detach().clone()))
    return dw
ops._conv_fwd, ops._conv_dgrad, ops._conv_wgrad = rec_fwd, rec_dgrad, rec_wgrad
net.zero_grad(); random.seed(101)
e = shpD_loss(net, b["imgs"][1], fake[1], b["hmaps"][1], b["rois"][1], b["num_rois"])
e.backward(); torch.cuda.synchronize()
ops._conv_fwd, ops._conv_dgrad, ops._conv_wgrad = orig
def rel(a, t):
    a = a.double().cpu(); return float((a - t).norm() / (t.norm() + 1e-300))
print("%d conv calls recorded" % len(calls))
for kind, a, bb, shp, (stride, pad, refl), got in calls:
    A, Bc = a.double().cpu(), bb.double().cpu()
    padmode = "reflect" if refl else "zeros"
    if kind == "fwd":
        xin = F.pad(A, (pad,) * 4, mode="reflect") if refl else A
        truth = F.conv2d(xin, Bc, None, stride, 0 if refl else pad)
        redo = lambda: orig[0](a, bb, None, stride, pad, refl, False, None)
        desc = "x%s w%s" % (tuple(a.shape), tuple(bb.shape))
    elif kind == "dgrad":
        N, Cin, H, W = shp
        if refl:
            xin = torch.zeros(N, Cin, H, W, dtype=torch.double, requires_grad=True)
            y = F.conv2d(F.pad(xin, (pad,) * 4, mode="reflect"), Bc, None, stride, 0)
            truth, = torch.autograd.grad(y, xin, A)
        else:
            truth = torch.nn.grad.conv2d_input((N, Cin, H, W), Bc, A, stride, pad)
        redo = lambda: orig[1](a, bb, N, Cin, H, W, stride, pad, refl, False, False)
        desc = "dy%s w%s" % (tuple(a.shape), tuple(bb.shape))
    else:
        Cout, k = shp
        xin = F.pad(A, (pad,) * 4, mode="reflect") if refl else A
        truth = torch.nn.grad.conv2d_weight(xin, (Cout, A.shape[1], k, k), Bc, stride, 0 if refl else pad)
        redo = lambda: orig[2](a, bb, Cout, k, stride, pad, refl, False)
        desc = "x%s dy%s" % (tuple(a.shape), tuple(bb.shape))
    errs = {}
    for math in ("fp32", "bf16x3"):
        ops.set_conv_math(math)
        r = redo(); torch.cuda.synchronize()
        errs[math] = rel(r, truth)
    ops.set_conv_math("bf16x3")
    print("%-5s s%d p%d %-7s %-46s fp32 %.2e  bf16x3 %.2e  (recorded %.2e)%s" % (kind, stride, pad, padmode, desc, errs["fp32"], errs["bf16x3"],
          rel(got, truth), "   <<<<" if errs["bf16x3"] > 3 * errs["fp32"] + 1e-7 else ""))
