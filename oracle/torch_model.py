"""ORACLE (test infrastructure): functional plain-PyTorch CPU fp32 restatement of the whole
image_generation training step -- generator, discriminators, losses, one optimisation iteration.

Networks are pure functions of a state dict (`sd`: key -> tensor, same keys as the reference
modules' state_dict()), so the very same weights can be pushed through the reference modules
(oracle/ref_harness.py), through this restatement and through the MI355X modules.  Each function
cites the reference lines it restates.  Pinned against the unmodified reference in
tests/test_oracle_cpu.py; used by the GPU parity tests at sizes the golden fixtures do not cover
and by bench.py's cpu_baseline leg (kind "port").
"""
import random
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

from . import torch_ref as T
from . import roi as _roi

CFG = dict(GF_DIM=48, DF_DIM=96, Z_DIM=100, CONDITION_DIM=100, EMBEDDING_DIM=256, GLOVE_DIM=50,
           GLB_R_NUM=7, LOCAL_R_NUM=3, BOXES_NUM=10, ROI_BASE_SIZE=5, ROI_SIZE_THRS=16.0,
           GAMMA1=4.0, GAMMA2=5.0, GAMMA3=10.0, DAMSM_LAMBDA=100.0, TXT_LAMBDA=0.1, SHP_LAMBDA=1.0,
           OBJ_LAMBDA=0.1, UNCOND_LAMBDA=1.0)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


_BN_EVAL = [False]


class bn_eval(object):
    """with bn_eval(): BatchNorm layers use their running statistics (module.eval() semantics)."""

    def __enter__(self):
        _BN_EVAL[0] = True

    def __exit__(self, *exc):
        _BN_EVAL[0] = False


def _bn(x, sd, p, mode):
    """BatchNorm in train mode (batch statistics, running stats updated in place) + activation;
    under bn_eval(): running statistics."""
    if _BN_EVAL[0]:
        shape = x.shape
        y = torch.nn.functional.batch_norm(x.reshape(shape[0], shape[1], -1), sd[p + "running_mean"],
                                           sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False,
                                           0.0, 1e-5).reshape(shape)
        if mode == "glu":
            return T.glu(y)
        return torch.nn.functional.leaky_relu(y, 0.2) if mode == "lrelu" else y
    if sd.get(p + "num_batches_tracked") is not None:
        sd[p + "num_batches_tracked"] += 1
    return T.norm_act(x, sd[p + "weight"], sd[p + "bias"], None, sd[p + "running_mean"],
                      sd[p + "running_var"], True, mode)


# ---- generator blocks (reference model.py:43-81, 455-518, 589-617, 708-719) ------------------------
def up_block(x, sd, p):
    return _bn(T.conv2d(x, sd[p + "1.weight"], None, 1, 1, "zeros", upsample=True), sd, p + "2.", "glu")


def res_block(x, sd, p):
    y = T.conv2d(x, sd[p + "block.1.weight"], None, 1, 1, "reflect")
    y = T.norm_act(y, mode="glu")
    y = T.conv2d(y, sd[p + "block.5.weight"], None, 1, 1, "reflect")
    return T.norm_act(y, residual=x)


def shape_stem(x, sd, p):
    """[ReflectionPad, Conv(k3, bias), InstanceNorm, LeakyReLU] (model.py:599-603, 1119-1123)."""
    return T.norm_act(T.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], 1, 1, "reflect"), mode="lrelu")


def g_hmap(x, sd, p):
    y = shape_stem(x, sd, p + "conv3x3.")
    return T.conv2d(y, sd[p + "downsample1.0.weight"], None, 2, 1, "zeros", act="lrelu")


def ca_net(sent_emb, sd, p, eps):
    x = T.glu(F.linear(sent_emb, sd[p + "fc.weight"], sd[p + "fc.bias"]))
    c = CFG["CONDITION_DIM"]
    mu, logvar = x[:, :c], x[:, c:]
    return eps * (logvar * 0.5).exp() + mu, mu, logvar


def init_stage_sent(z, c_code, sd, p):
    h = F.linear(torch.cat((c_code, z), 1), sd[p + "fc.0.weight"])
    h = _bn(h.reshape(h.shape[0], h.shape[1], 1, 1), sd, p + "fc.1.", "glu").reshape(h.shape[0], -1)
    h = h.view(-1, CFG["GF_DIM"] * 4, 8, 8)
    return up_block(up_block(h, sd, p + "upsample1."), sd, p + "upsample2.")


def _bottom_up(sd, p, word_embs, glove, slabels_feat, mask, bt_mask, max_num_roi, ih, iw):
    slabels_feat = slabels_feat[:, :, :max_num_roi]
    src = T.conv2d(word_embs.unsqueeze(3), sd[p + "bt_att.conv_context.weight"]).squeeze(3)
    raw_c, raw_att = T.attn_bu(slabels_feat, glove, src, mask, True, 1e-8)
    bt_mask = bt_mask[:, :max_num_roi]
    return (raw_c, T.masked_max(raw_c, bt_mask, ih, iw), T.masked_max(raw_att, bt_mask, ih, iw),
            T.masked_max(slabels_feat, bt_mask, ih, iw))


def init_stage_main(sd, p, h_hmap, h_sent, word_embs, glove, slabels_feat, mask, bt_mask, max_num_roi):
    ih, iw = h_hmap.shape[2], h_hmap.shape[3]
    _, bt_c, _, bt_sl = _bottom_up(sd, p, word_embs, glove, slabels_feat, mask, bt_mask, max_num_roi, ih, iw)
    h = torch.cat((h_hmap, h_sent, bt_c, bt_sl), 1)
    for i in range(CFG["GLB_R_NUM"]):
        h = res_block(h, sd, p + "residual.%d." % i)
    return up_block(h, sd, p + "upsample.")


def next_stage_main(sd, p, h_code, h_hmap, word_embs, glove, slabels_feat, mask, bt_mask, max_num_roi,
                    glb_max_num_roi):
    B, idf, ih, iw = h_code.shape
    src = T.conv2d(word_embs.unsqueeze(3), sd[p + "att.conv_context.weight"]).squeeze(3)
    c_code, att = T.attn_general(h_code, src, mask)
    raw, bt_c, bt_att, bt_sl = _bottom_up(sd, p, word_embs, glove, slabels_feat, mask, bt_mask,
                                          max_num_roi, ih, iw)
    raw_full = torch.zeros(B, idf, glb_max_num_roi, 1, dtype=h_code.dtype)
    raw_full = torch.cat((raw, raw_full[:, :, max_num_roi:]), 2)
    h = torch.cat((h_code + h_hmap, c_code, bt_c, bt_sl), 1)
    for i in range(CFG["LOCAL_R_NUM"]):
        h = res_block(h, sd, p + "residual.%d." % i)
    out = up_block(h, sd, p + "upsample.")
    return out, raw_full.transpose(1, 2).squeeze(-1), att, bt_att


def get_image(h, sd, p):
    return T.conv2d(h, sd[p + "img.0.weight"], None, 1, 1, "zeros", act="tanh")


def g_net(sd, z, sent_emb, word_embs, glove, slabels_feat, mask, hmaps, rois, fm_rois, num_rois,
          bt_masks, fm_bt_masks, glb_max_num_roi, ca_eps, branch_num=3):
    """G_NET.forward (model.py:747-795)."""
    fake, bt_codes, atts, bt_atts = [], [], [], []
    c_code, mu, logvar = ca_net(sent_emb, sd, "ca_net.", ca_eps)
    mx = int(num_rois.max())
    h = init_stage_main(sd, "h_net1_main.", g_hmap(hmaps[0], sd, "h_net1_hmap."),
                        init_stage_sent(z, c_code, sd, "h_net1_sent."), word_embs, glove, slabels_feat,
                        mask, fm_bt_masks, mx)
    fake.append(get_image(h, sd, "img_net1."))
    for s in range(1, branch_num):
        k = s + 1
        h, bt_c, att, bt_att = next_stage_main(sd, "h_net%d_main." % k, h,
                                               g_hmap(hmaps[s], sd, "h_net%d_hmap." % k), word_embs,
                                               glove, slabels_feat, mask, bt_masks[s - 1], mx,
                                               glb_max_num_roi)
        fake.append(get_image(h, sd, "img_net%d." % k))
        bt_codes.append(bt_c)
        atts.append(att)
        bt_atts.append(bt_att)
    return fake, bt_codes, atts, bt_atts, mu, logvar


# ---- discriminators (reference model.py:989-1312) ---------------------------------------------------
def encoder(x, sd, p, n_layer):
    x = T.conv2d(x, sd[p + "0.weight"], None, 2, 1, "zeros", act="lrelu")
    for n in range(1, n_layer):
        i = 2 + 3 * (n - 1)
        x = _bn(T.conv2d(x, sd[p + "%d.weight" % i], None, 2, 1, "zeros"), sd, p + "%d." % (i + 1), "lrelu")
    return x


def head(h, sd, p, c_code=None):
    """D_GET_LOGITS.forward (model.py:1035-1048); p ends with 'COND_DNET.' or 'UNCOND_DNET.'."""
    if c_code is not None:
        c = c_code.view(c_code.shape[0], -1, 1, 1).expand(-1, -1, h.shape[2], h.shape[3])
        h = _bn(T.conv2d(torch.cat((h, c), 1), sd[p + "jointConv.0.weight"], None, 1, 1), sd,
                p + "jointConv.1.", "lrelu")
    return T.conv2d(h, sd[p + "outlogits.0.weight"], sd[p + "outlogits.0.bias"], 2, 0, "zeros", act="sigmoid")


def pat_d(sd, x):
    return encoder(x, sd, "img_code.", 4)


def shp_d(sd, x, seg):
    return encoder(torch.cat([x, shape_stem(seg, sd, "shp_code.")], 1), sd, "img_code.", 4)


def rois_blob(fm_rois):
    B, R = fm_rois.shape[0], fm_rois.shape[1]
    box = fm_rois[:, :, :4].double()
    idx = torch.arange(B, dtype=torch.float64).view(B, 1, 1).expand(B, R, 1)
    return torch.cat((idx, box[:, :, :2], box[:, :, :2] + box[:, :, 2:4]), 2).reshape(B * R, 5).float()


class _RoiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, ah, aw, scale):
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(feat.shape), scale)
        return torch.from_numpy(_roi.forward(feat.detach().numpy(), rois.numpy(), ah, aw, scale))

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, scale = ctx.meta
        return torch.from_numpy(_roi.backward(g.contiguous().numpy(), rois.numpy(), shape, scale)), None, None, None, None


def obj_d(sd, x, seg, fm_rois, n_layer, img_size=512):
    """OBJ_SS/LS_D_NET.forward (model.py:1212-1246): lift to 512^2, encode, ROIAlignAvg(5,5,1/16)
    over ALL box slots, 4x4 conv + LeakyReLU."""
    x = T.bilinear_resize(x, img_size, img_size)
    seg = T.bilinear_resize(seg, img_size, img_size)
    code = encoder(torch.cat([x, shape_stem(seg, sd, "shp_code.")], 1), sd, "img_code.", n_layer)
    r = CFG["ROI_BASE_SIZE"]
    pooled = T.avgpool2s1(_RoiFn.apply(code, rois_blob(fm_rois), r + 1, r + 1, 1.0 / 16.0))
    pooled = T.conv2d(pooled, sd["roi_code.0.weight"], sd["roi_code.0.bias"], 1, 1, "zeros", act="lrelu")
    return pooled.view(fm_rois.shape[0], CFG["BOXES_NUM"], pooled.shape[1], pooled.shape[2], pooled.shape[3])


# ---- helpers restated from miscc/utils.py:445-522 ----------------------------------------------------
def permute_seg(seg, rois, num_rois):
    new = seg.clone()
    valid = []
    r = rois.numpy()
    for b in range(seg.shape[0]):
        n = int(num_rois[b])
        if n == 0:
            continue
        present = [int(c) for c in np.unique(r[b, :n, 4])]
        shuf = deepcopy(present)
        random.shuffle(shuf)
        if present != shuf:
            valid.append(b)
            new[b, present] = seg[b, shuf]
    return new, valid


def feat_select(pooled, raw_bt, fm_rois, num_rois, large):
    r = fm_rois.numpy()
    ib, ir, cls = [], [], []
    for b in range(len(num_rois)):
        for k in range(int(num_rois[b])):
            w, h = r[b, k, 2], r[b, k, 3]
            if (w < 1.25 and h < 1.25) or ((max(w, h) >= CFG["ROI_SIZE_THRS"]) != bool(large)):
                continue
            ib.append(b); ir.append(k); cls.append(int(r[b, k, 4]))
    if not ib:
        return [], [], []
    ib, ir = torch.tensor(ib), torch.tensor(ir)
    return pooled[ib, ir], torch.tensor(cls), raw_bt[ib, ir]


def form_clabels_feat(clabels_emb, rois, num_rois):
    B = rois.shape[0]
    mx = int(num_rois.max())
    out = torch.zeros(B, mx, clabels_emb.shape[1])
    for i in range(B):
        n = int(num_rois[i])
        if n:
            out[i, :n] = clabels_emb[rois[i, :n, 4].long()]
    return out.transpose(1, 2).unsqueeze(3)


# ---- losses (reference miscc/losses.py) ---------------------------------------------------------------
def _bce(p, t):
    return F.binary_cross_entropy(p, torch.full_like(p, float(t)))


def pat_d_loss(sd, real, fake, cond):
    rf, ff = pat_d(sd, real), pat_d(sd, fake.detach())
    B = rf.shape[0]
    cr, cf = _bce(head(rf, sd, "COND_DNET.", cond), 1), _bce(head(ff, sd, "COND_DNET.", cond), 0)
    cw = _bce(head(rf[:B - 1], sd, "COND_DNET.", cond[1:B]), 0)
    re, fe = _bce(head(rf, sd, "UNCOND_DNET."), 1), _bce(head(ff, sd, "UNCOND_DNET."), 0)
    lu, lt = CFG["UNCOND_LAMBDA"], CFG["TXT_LAMBDA"]
    return (re * lu + cr * lt) / 2. + (fe * lu + (cf + cw) * lt) / 3.


def shp_d_loss(sd, real, fake, seg, rois, num_rois):
    rf, ff = shp_d(sd, real, seg), shp_d(sd, fake.detach(), seg)
    fseg, valid = permute_seg(seg, rois, num_rois)
    err = _bce(head(rf, sd, "UNCOND_DNET."), 1)
    fe = _bce(head(ff, sd, "UNCOND_DNET."), 0)
    if valid:
        we = _bce(head(shp_d(sd, real[valid], fseg[valid]), sd, "UNCOND_DNET."), 0)
        return err + (fe + we) / 2.
    return err + fe


def obj_d_loss(sd, n_layer, real, fake, seg, class_table, raw_bt, fm_rois, num_rois, large):
    rfeat, cls, btc = feat_select(obj_d(sd, real, seg, fm_rois, n_layer), raw_bt, fm_rois, num_rois, large)
    ffeat, _, _ = feat_select(obj_d(sd, fake.detach(), seg, fm_rois, n_layer), raw_bt, fm_rois, num_rois, large)
    fseg, valid = permute_seg(seg, fm_rois, num_rois)
    cls2 = []
    if valid:
        f2, cls2, btc2 = feat_select(obj_d(sd, real[valid], fseg[valid], fm_rois[valid], n_layer), raw_bt,
                                     fm_rois[valid], num_rois[valid], large)
    K = len(cls)
    if K == 0:
        return 0
    cond = torch.cat((class_table[cls], btc), 1)
    cr = _bce(head(rfeat, sd, "COND_DNET.", cond), 1)
    cf = _bce(head(ffeat, sd, "COND_DNET.", cond), 0)
    extra, n_extra = 0, 0
    if K > 1:
        extra = extra + _bce(head(rfeat[:K - 1], sd, "COND_DNET.", cond[1:K]), 0)
    if valid and len(cls2) > 0:
        cond2 = torch.cat((class_table[cls[:len(cls2)]], btc2), 1)      # losses.py:312-313 quirk
        extra = extra + _bce(head(f2, sd, "COND_DNET.", cond2), 0)
        n_extra = 1
    re, fe = _bce(head(rfeat, sd, "UNCOND_DNET."), 1), _bce(head(ffeat, sd, "UNCOND_DNET."), 0)
    return (re + cr) / 2. + (fe + cf + extra) / (3. + n_extra)


def words_loss(regions, words_emb, labels, cap_lens, class_ids):
    """losses.py:77-159, caption by caption like the reference."""
    B = regions.shape[0]
    sims = []
    for i in range(B):
        n = int(cap_lens[i])
        word = words_emb[i, :, :n].unsqueeze(0).repeat(B, 1, 1)
        wc, _ = T.func_attention(word, regions, CFG["GAMMA1"])
        w = word.transpose(1, 2).reshape(B * n, -1)
        c = wc.transpose(1, 2).reshape(B * n, -1)
        cos = (w * c).sum(1) / (w.norm(2, 1) * c.norm(2, 1)).clamp(min=1e-8)
        sims.append(torch.log((cos.view(B, n) * CFG["GAMMA2"]).exp().sum(1, keepdim=True)))
    s = torch.cat(sims, 1) * CFG["GAMMA3"]
    s = _mask_same_class(s, class_ids)
    return F.cross_entropy(s, labels), F.cross_entropy(s.t(), labels)


def _mask_same_class(s, class_ids):
    if class_ids is None:
        return s
    ids = np.asarray(class_ids)
    m = ids.reshape(-1, 1) == ids.reshape(1, -1)
    np.fill_diagonal(m, False)
    return s.masked_fill(torch.from_numpy(m), -float("inf"))


def sent_loss(cnn_code, rnn_code, labels, class_ids, eps=1e-8):
    n0 = cnn_code.norm(2, 1, keepdim=True) @ rnn_code.norm(2, 1, keepdim=True).t()
    s = (cnn_code @ rnn_code.t()) / n0.clamp(min=eps) * CFG["GAMMA3"]
    s = _mask_same_class(s, class_ids)
    return F.cross_entropy(s, labels), F.cross_entropy(s.t(), labels)


def kl_loss(mu, logvar):
    return torch.mean(1 + logvar - mu.pow(2) - logvar.exp()) * (-0.5)


def g_loss(sds, image_encoder, fake, hmaps, words_embs, sent_emb, class_table, raw_bt, labels,
           cap_lens, class_ids, rois, fm_rois, num_rois, use_obj=True):
    """G_loss (losses.py:364-529). sds: dict with 'pat' [3], 'shp' [3], 'objss', 'objls' state dicts."""
    total = 0
    parts = {}
    n = len(sds["pat"])
    for i in range(n):
        f = pat_d(sds["pat"][i], fake[i])
        pg = (_bce(head(f, sds["pat"][i], "UNCOND_DNET."), 1) * CFG["UNCOND_LAMBDA"] +
              _bce(head(f, sds["pat"][i], "COND_DNET.", sent_emb), 1) * CFG["TXT_LAMBDA"])
        sg = _bce(head(shp_d(sds["shp"][i], fake[i], hmaps[i]), sds["shp"][i], "UNCOND_DNET."), 1) * CFG["SHP_LAMBDA"]
        total = total + pg + sg
        parts["pat_g%d" % i], parts["shp_g%d" % i] = pg, sg
        if i == n - 1 and image_encoder is not None:
            regions, code = image_encoder(fake[i])
            w0, w1 = words_loss(regions, words_embs, labels, cap_lens, class_ids)
            s0, s1 = sent_loss(code, sent_emb, labels, class_ids)
            parts["w_loss"], parts["s_loss"] = (w0 + w1) * CFG["DAMSM_LAMBDA"], (s0 + s1) * CFG["DAMSM_LAMBDA"]
            total = total + parts["w_loss"] + parts["s_loss"]
    if use_obj:
        for tag, key, nl, r, large in (("objss_g", "objss", 3, rois[0], False), ("objls_g", "objls", 4, fm_rois, True)):
            sd = sds[key]
            feats, cls, btc = feat_select(obj_d(sd, fake[-1], hmaps[-1], r, nl), raw_bt, r, num_rois, large)
            if len(cls) == 0:
                continue
            cond = torch.cat((class_table[cls], btc), 1)
            term = (_bce(head(feats, sd, "COND_DNET.", cond), 1) + _bce(head(feats, sd, "UNCOND_DNET."), 1)) * CFG["OBJ_LAMBDA"]
            parts[tag] = term
            total = total + term
    return total, parts


# ---- one full training iteration (reference trainer.py:357-462) -----------------------------------------
def params_of(sd):
    """leaf tensors that the optimiser updates (everything except BN running statistics)."""
    return [v for k, v in sorted(sd.items()) if v.dtype.is_floating_point and "running_" not in k]


def train_step(sds, opts, ema, batch, image_encoder=None, use_obj=True, lr=2e-4):
    """sds: {'G', 'pat': [..], 'shp': [..], 'objss', 'objls'} state dicts whose float tensors have
    requires_grad=True; opts: matching torch.optim.Adam instances; ema: list of tensors (EMA of G).
    Follows the reference update order: PatD0..2, ShpD0..2, ObjSSD, ObjLSD, G, EMA."""
    b = batch
    n = len(sds["pat"])
    out = {}
    cl = form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    glb = int(b["num_rois"].max())
    fake, bt_codes, _, _, mu, logvar = g_net(sds["G"], b["noise"], b["sent_emb"], b["words_embs"],
                                             b["glove_words_embs"], cl, b["mask"], b["hmaps"], b["rois"],
                                             b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"],
                                             glb, b["ca_eps"], branch_num=n)
    bt = [c.detach() for c in bt_codes]
    for i in range(n):
        opts["pat"][i].zero_grad()
        e = pat_d_loss(sds["pat"][i], b["imgs"][i], fake[i], b["sent_emb"])
        e.backward(); opts["pat"][i].step(); out["errPatD%d" % i] = e.detach()
    for i in range(n):
        opts["shp"][i].zero_grad()
        e = shp_d_loss(sds["shp"][i], b["imgs"][i], fake[i], b["hmaps"][i], b["rois"][i], b["num_rois"])
        e.backward(); opts["shp"][i].step(); out["errShpD%d" % i] = e.detach()
    if use_obj:
        for tag, key, nl, r, large in (("errObjSSD", "objss", 3, b["rois"][0], False),
                                       ("errObjLSD", "objls", 4, b["fm_rois"], True)):
            opts[key].zero_grad()
            e = obj_d_loss(sds[key], nl, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], bt[-1],
                           r, b["num_rois"], large)
            if float(e) > 0:
                e.backward(); opts[key].step(); out[tag] = e.detach()
    opts["G"].zero_grad()
    labels = torch.arange(b["imgs"][0].shape[0])
    total, parts = g_loss(sds, image_encoder, fake, b["hmaps"], b["words_embs"], b["sent_emb"],
                          b["clabels_emb"], bt[-1] if bt else None, labels, b["cap_lens"], b["class_ids"],
                          b["rois"], b["fm_rois"], b["num_rois"], use_obj=use_obj)
    kl = kl_loss(mu, logvar)
    total = total + kl
    total.backward()
    opts["G"].step()
    with torch.no_grad():
        for p, a in zip(params_of(sds["G"]), ema):
            a.mul_(0.999).add_(p, alpha=0.001)
    out.update(errG=total.detach(), kl=kl.detach(), fake_imgs=[f.detach() for f in fake])
    out.update({k: v.detach() for k, v in parts.items()})
    return out


# ---------------------------------------------------------------------------------------------
# frozen text encoder (reference model.py:85-179)
# ---------------------------------------------------------------------------------------------
def rnn_encoder_forward(sd, captions, cap_lens, max_len):
    """RNN_ENCODER.forward restated with explicit LSTM cell loops: Embedding -> bidirectional LSTM over
    the first cap_lens[b] words of each caption (packed-sequence semantics, reference model.py:152-161)
    -> words_emb [B, 2H, max_len] zero past each length (:164-167, 139-146), sent_emb = final hidden
    states of both directions concatenated (:169-175).  Gate order i, f, g, o (PyTorch nn.LSTM)."""
    emb = sd["encoder.weight"][captions]                       # B x L x I   (eval: dropout = identity)
    B, L, _ = emb.shape
    H = sd["rnn.weight_hh_l0"].shape[1]
    words = torch.zeros(B, 2 * H, max_len, dtype=emb.dtype)
    sent = torch.zeros(B, 2 * H, dtype=emb.dtype)
    for d, suffix in enumerate(("", "_reverse")):
        w_ih, w_hh = sd["rnn.weight_ih_l0" + suffix], sd["rnn.weight_hh_l0" + suffix]
        bias = sd["rnn.bias_ih_l0" + suffix] + sd["rnn.bias_hh_l0" + suffix]
        for b in range(B):
            n = int(cap_lens[b])
            h = torch.zeros(H, dtype=emb.dtype)
            c = torch.zeros(H, dtype=emb.dtype)
            steps = range(n) if d == 0 else range(n - 1, -1, -1)
            for t in steps:
                g = w_ih @ emb[b, t] + w_hh @ h + bias
                i, f, gg, o = g[:H].sigmoid(), g[H:2 * H].sigmoid(), g[2 * H:3 * H].tanh(), g[3 * H:].sigmoid()
                c = f * c + i * gg
                h = o * c.tanh()
                if t < max_len:
                    words[b, d * H:(d + 1) * H, t] = h
            sent[b, d * H:(d + 1) * H] = h
    return words, sent
