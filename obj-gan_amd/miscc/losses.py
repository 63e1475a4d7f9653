"""Adversarial, DAMSM and KL losses of the image generator (same functions, arguments and return
values as reference image_generation/miscc/losses.py:13-537).

The network bodies and heads they call are the MI355X modules of model.py.  The DAMSM word loss
evaluates all batch x batch (image, caption) pairs at once -- one MFMA 1x1-convolution for every
region/word score, two fused strided-softmax kernels -- instead of the reference's python loop
over captions (losses.py:87-127).  What remains in PyTorch here is scalar glue on tensors of at
most a few hundred kilobytes (BCE on B x 1 x k x k probabilities, cosine / log-sum-exp on
B x B x L values, index selects).
"""
import numpy as np
import torch
import torch.nn.functional as F

from miscc.config import cfg
from miscc.utils import permute_seg, permuted_valid_seg, feat_select, take_rows, h2d  # noqa: F401
from GlobalAttention import func_attention  # noqa: F401  (API parity)
from objgan_hip import ops


def _net(m):
    """The reference reaches the heads through `.module` when nn.DataParallel wraps the network."""
    return m.module if hasattr(m, "module") else m


def _bce(prob, target_value):
    """nn.BCELoss()(prob, labels) with constant labels (reference losses.py:182-204)"""
    return ops.bce_const(prob, float(target_value))


def _class_mask(class_ids, batch_size, device):
    """masks[i, j] = 1 where sample j has the class of sample i (j != i): mismatched pairs that are
    not true negatives (reference losses.py:28-36, 95-98, 130-136)."""
    if class_ids is None:
        return None
    ids = np.asarray(class_ids)
    m = (ids.reshape(-1, 1) == ids.reshape(1, -1))
    np.fill_diagonal(m, False)
    return h2d(np.ascontiguousarray(m[:batch_size, :batch_size]), device)


# ################## Loss for matching text-image ###################
def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """Returns cosine similarity between x1 and x2, computed along dim."""
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return (w12 / (w1 * w2).clamp(min=eps)).squeeze()


def sent_loss(cnn_code, rnn_code, labels, class_ids, batch_size, eps=1e-8, top1=True,
              is_training=True):
    masks = _class_mask(class_ids, batch_size, cnn_code.device)
    if cnn_code.dim() == 2:
        cnn_code = cnn_code.unsqueeze(0)
        rnn_code = rnn_code.unsqueeze(0)
    cnn_norm = torch.norm(cnn_code, 2, dim=2, keepdim=True)
    rnn_norm = torch.norm(rnn_code, 2, dim=2, keepdim=True)
    scores0 = torch.bmm(cnn_code, rnn_code.transpose(1, 2))
    norm0 = torch.bmm(cnn_norm, rnn_norm.transpose(1, 2))
    scores0 = (scores0 / norm0.clamp(min=eps) * cfg.TRAIN.SMOOTH.GAMMA3).squeeze()
    if masks is not None:
        scores0 = scores0.masked_fill(masks, -float('inf'))
    scores1 = scores0.transpose(0, 1)
    if labels is None:
        return None, None, None
    loss0 = F.cross_entropy(scores0, labels)
    loss1 = F.cross_entropy(scores1, labels)
    if top1:
        correct = (scores0.argmax(1) == labels).sum().item() + (scores1.argmax(1) == labels).sum().item()
        accuracy = (100. * correct) / (batch_size * 2.)
    else:
        accuracy = scores0
    return loss0, loss1, accuracy


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, batch_size, top1=True,
               is_training=True, need_att_maps=True):
    """
        words_emb(query): batch x nef x seq_len
        img_features(context): batch x nef x 17 x 17
    similarities[b, i] = log sum_l exp(gamma2 * cos(word_i[:, l], weightedContext_{b,i}[:, l])).
    """
    B = batch_size
    nef, L = words_emb.size(1), words_emb.size(2)
    ih, iw = img_features.size(2), img_features.size(3)
    S = ih * iw
    dev = img_features.device
    lens = cap_lens.detach().to(torch.int32)
    lens_dev = lens.to(dev).clamp(max=L)

    # Eq. (7): every region x every word of every caption in one 1x1 convolution
    bank = words_emb.permute(0, 2, 1).reshape(B * L, nef, 1, 1)             # row i*L + l
    scores = ops.conv2d(img_features, bank)                                  # B x (B*L) x ih x iw
    scores = scores.reshape(B, B, L, S)
    # Eq. (8): softmax over the words of caption i (first cap_lens[i] only), per region
    a1 = ops.softmax_strided(scores, 2, 1.0, lens=lens_dev)
    # Eq. (9): softmax over regions of gamma1 * a1; rows of padded words are zeroed
    word_ok = (torch.arange(L, device=dev).view(1, 1, L) < lens_dev.view(1, B, 1)).expand(B, B, L)
    a2 = ops.softmax_strided(a1, 3, float(cfg.TRAIN.SMOOTH.GAMMA1),
                             rowvalid=word_ok.reshape(-1).to(torch.uint8).contiguous())
    # weightedContext[b, :, i, l] = sum_s context[b, :, s] * a2[b, i, l, s]: B small products, one launch
    wc = ops.bmm(img_features.reshape(B, nef, S), a2.reshape(B, B * L, S).transpose(1, 2)).reshape(B, nef, B, L)
    word = words_emb.permute(1, 0, 2).unsqueeze(0)                           # 1 x nef x B x L
    w12 = (word * wc).sum(1)
    denom = (torch.norm(word, 2, 1) * torch.norm(wc, 2, 1)).clamp(min=1e-8)
    row_sim = (w12 / denom * cfg.TRAIN.SMOOTH.GAMMA2).exp()                  # Eq. (10), B x B x L
    row_sim = (row_sim * word_ok.to(row_sim.dtype)).sum(2)
    similarities = torch.log(row_sim) * cfg.TRAIN.SMOOTH.GAMMA3              # [image b, caption i]

    att_maps = None
    if need_att_maps:        # per-caption maps for the visualisations: needs the lengths on the host (a sync)
        lens_list = lens.cpu().tolist()
        att_maps = [a2[i, i, :lens_list[i]].reshape(1, -1, ih, iw).detach() for i in range(B)]
    masks = _class_mask(class_ids, B, dev)
    if masks is not None:
        similarities = similarities.masked_fill(masks, -float('inf'))
    similarities1 = similarities.transpose(0, 1)
    if labels is None:
        return None, None, att_maps, None
    loss0 = F.cross_entropy(similarities, labels)
    loss1 = F.cross_entropy(similarities1, labels)
    if top1:
        correct = ((similarities.argmax(1) == labels).sum().item() +
                   (similarities1.argmax(1) == labels).sum().item())
        accuracy = (100. * correct) / (B * 2.)
    else:
        accuracy = [F.softmax(similarities, dim=1), F.softmax(similarities1, dim=1)]
    return loss0, loss1, att_maps, accuracy


# ################## Loss for G and Ds ##############################
# The passes of a discriminator update over the REAL images do not depend on the generator: `*_real` evaluates them on their
# own (the trainer issues them on the discriminators' side streams while the generator's forward pass runs on the main
# stream) and the loss takes the result through `real=`.  Same operators in the same per-network order (real, fake, wrong:
# the BatchNorm running statistics see the same sequence), same arithmetic: the losses are bit-identical either way.
def patD_real(netPatD, real_imgs):
    return {"real_features": netPatD(real_imgs)}


def shpD_real(netShpD, real_imgs, seg_conditions):
    s_code = _net(netShpD).encode_seg(seg_conditions)
    return {"s_code": s_code, "real_features": netShpD(real_imgs, seg_conditions, s_code=s_code)}


def objD_real(netObjD, real_imgs, seg_conditions, fm_rois, num_rois):
    s_code = _net(netObjD).encode_seg(seg_conditions)
    return {"s_code": s_code, "real_pooled": netObjD(real_imgs, seg_conditions, fm_rois, num_rois, s_code=s_code)}


def patD_loss(netPatD, real_imgs, fake_imgs, conditions, real=None):
    net = _net(netPatD)
    real_features = real["real_features"] if real is not None else netPatD(real_imgs)
    fake_features = netPatD(fake_imgs.detach())
    B = real_features.size(0)
    lam_u, lam_t = cfg.TRAIN.SMOOTH.UNCOND_LAMBDA, cfg.TRAIN.SMOOTH.TXT_LAMBDA
    cond_real = _bce(net.COND_DNET(real_features, conditions), 1)
    cond_fake = _bce(net.COND_DNET(fake_features, conditions), 0)
    # "wrong pair": real image i with the sentence of sample i+1
    cond_wrong = _bce(net.COND_DNET(real_features[:B - 1], conditions[1:B]), 0)
    if net.UNCOND_DNET is not None:
        real_err = _bce(net.UNCOND_DNET(real_features), 1)
        fake_err = _bce(net.UNCOND_DNET(fake_features), 0)
        return ((real_err * lam_u + cond_real * lam_t) / 2. +
                (fake_err * lam_u + (cond_fake + cond_wrong) * lam_t) / 3.)
    return (cond_real + (cond_fake + cond_wrong) / 2.) * lam_t


def shpD_loss(netShpD, real_imgs, fake_imgs, seg_conditions, rois, num_rois, real=None, draw=None):
    net = _net(netShpD)
    # the encoded layout map is the same tensor in the real and the fake pass: evaluate it once
    if real is None:
        real = shpD_real(netShpD, real_imgs, seg_conditions)
    s_code, real_features = real["s_code"], real["real_features"]
    fake_features = netShpD(fake_imgs.detach(), seg_conditions, s_code=s_code)
    wrong_seg, valid = permuted_valid_seg(seg_conditions, rois, num_rois, draw=draw)
    errD = _bce(net.UNCOND_DNET(real_features), 1)
    fake_err = _bce(net.UNCOND_DNET(fake_features), 0)
    if len(valid) > 0:
        wrong_features = netShpD(take_rows(real_imgs, valid), wrong_seg)
        wrong_err = _bce(net.UNCOND_DNET(wrong_features), 0)
        return errD + (fake_err + wrong_err) / 2.
    return errD + fake_err


def _obj_conditions(class_table, classes, bt_c_codes, count=None):
    """[class embedding | bottom-up context] per selected box."""
    idx = getattr(classes, "_og_dev", None)          # device copy made by feat_select (one upload for all indices)
    if idx is None:
        idx = h2d(classes, class_table.device)
    if count is not None:
        idx = idx[:count]
    emb = class_table[idx]
    return torch.cat((emb, bt_c_codes), dim=1)


def objD_loss(netObjD, real_imgs, fake_imgs, seg_conditions, raw_conditions, raw_bt_c_codes,
              fm_rois, num_rois, is_large_scale=False, real=None, draw=None):
    net = _net(netObjD)
    if real is None:
        real = objD_real(netObjD, real_imgs, seg_conditions, fm_rois, num_rois)
    s_code, real_pooled = real["s_code"], real["real_pooled"]      # (the encoded layout: shared by the real and the fake pass)
    real_features, classes, bt_c_codes = feat_select(real_pooled, raw_bt_c_codes, fm_rois, num_rois,
                                                     is_large_scale=is_large_scale)
    fake_pooled = netObjD(fake_imgs.detach(), seg_conditions, fm_rois, num_rois, s_code=s_code)
    fake_features, _, _ = feat_select(fake_pooled, raw_bt_c_codes, fm_rois, num_rois,
                                      is_large_scale=is_large_scale)
    wrong_seg, valid = permuted_valid_seg(seg_conditions, fm_rois, num_rois, draw=draw)
    classes2 = []
    if len(valid) > 0:
        rois_v, num_v = take_rows(fm_rois, valid), take_rows(num_rois, valid)      # host copies stay attached
        pooled2 = netObjD(take_rows(real_imgs, valid), wrong_seg, rois_v, num_v)
        fake_features2, classes2, bt_c_codes2 = feat_select(pooled2, raw_bt_c_codes, rois_v,
                                                            num_v,
                                                            is_large_scale=is_large_scale)
    K = len(classes)
    if K == 0:
        return 0
    conditions = _obj_conditions(raw_conditions, classes, bt_c_codes)
    cond_real = _bce(net.COND_DNET(real_features, conditions), 1)
    cond_fake = _bce(net.COND_DNET(fake_features, conditions), 0)
    extra = 0
    n_extra = 0
    if K > 1:
        extra = extra + _bce(net.COND_DNET(real_features[:K - 1], conditions[1:K]), 0)
    if len(valid) > 0 and len(classes2) > 0:
        # reference quirk (losses.py:312-313): the class table is indexed with `classes`, the
        # class list of the FULL batch, truncated to len(classes2)
        conditions2 = _obj_conditions(raw_conditions, classes, bt_c_codes2, count=len(classes2))
        extra = extra + _bce(net.COND_DNET(fake_features2, conditions2), 0)
        n_extra = 1
    if net.UNCOND_DNET is not None:
        real_err = _bce(net.UNCOND_DNET(real_features), 1)
        fake_err = _bce(net.UNCOND_DNET(fake_features), 0)
        return (real_err + cond_real) / 2. + (fake_err + cond_fake + extra) / (3. + n_extra)
    return cond_real + (cond_fake + extra) / (2. + n_extra)


def _obj_g_term(netObjD, fake_img, seg, slabels_emb, raw_bt_c_codes, rois, num_rois, large):
    net = _net(netObjD)
    pooled = netObjD(fake_img, seg, rois, num_rois)
    feats, classes, bt_c_codes = feat_select(pooled, raw_bt_c_codes, rois, num_rois,
                                             is_large_scale=large)
    if len(classes) == 0:
        return 0
    conditions = _obj_conditions(slabels_emb, classes, bt_c_codes)
    err = _bce(net.COND_DNET(feats, conditions), 1)
    if net.UNCOND_DNET is not None:
        err = err + _bce(net.UNCOND_DNET(feats), 1)
    return err * cfg.TRAIN.SMOOTH.OBJ_LAMBDA


def G_loss(netsPatD, netsShpD, netObjSSD, netObjLSD, image_encoder, fake_imgs, seg_conditions,
           words_embs, sent_emb, slabels_emb, raw_bt_c_codes, match_labels, cap_lens, class_ids,
           rois, fm_rois, num_rois, quiet=False, use_obj=True, streams=None, damsm_pre=None):
    """quiet=True skips the log string, the DAMSM accuracies and attention maps (every `.item()` / `.cpu()` in
    them is a device->host sync).
    streams (quiet only): HIP streams the nine terms -- DAMSM first, then the discriminators in the order below --
    are spread over; their backward nodes run on the same streams (autograd replays a node on the stream of its
    forward).  (Round 2 measured no gain from side streams: the host was stalling on index uploads then and never
    ran ahead of the device.  Round 3, host 130 ms ahead per step: 205.6 -> 190.0 ms with three streams.)
    use_obj=False leaves the two object-discriminator terms out (BASELINE.json configs 1-3: the
    reference has no such switch, its stage-1 / no-ObjD runs are harness compositions)."""
    if streams and quiet:
        return _g_loss_streams(netsPatD, netsShpD, netObjSSD, netObjLSD, image_encoder, fake_imgs, seg_conditions,
                               words_embs, sent_emb, slabels_emb, raw_bt_c_codes, match_labels, cap_lens, class_ids,
                               rois, fm_rois, num_rois, use_obj, streams, damsm_pre=damsm_pre), ''
    numDs = len(netsPatD)
    batch_size = fake_imgs[0].size(0)
    logs = ''
    errG_total = 0
    sm = cfg.TRAIN.SMOOTH
    for i in range(numDs):
        pat = _net(netsPatD[i])
        features = netsPatD[i](fake_imgs[i])
        pat_g_loss = _bce(pat.COND_DNET(features, sent_emb), 1)
        if pat.UNCOND_DNET is not None:
            pat_g_loss = _bce(pat.UNCOND_DNET(features), 1) * sm.UNCOND_LAMBDA + pat_g_loss * sm.TXT_LAMBDA
        errG_total = errG_total + pat_g_loss
        if not quiet:
            logs += 'pat_g_loss %d: %.2f ' % (i, pat_g_loss.item())

        shp = _net(netsShpD[i])
        features = netsShpD[i](fake_imgs[i], seg_conditions[i])
        shp_g_loss = _bce(shp.UNCOND_DNET(features), 1) * sm.SHP_LAMBDA
        errG_total = errG_total + shp_g_loss
        if not quiet:
            logs += 'shp_g_loss%d: %.2f ' % (i, shp_g_loss.item())

        if i == (numDs - 1):        # DAMSM ranking loss on the full-resolution image
            region_features, cnn_code = image_encoder(fake_imgs[i])
            w_loss0, w_loss1, _, _ = words_loss(region_features, words_embs, match_labels, cap_lens,
                                                class_ids, batch_size, top1=not quiet, need_att_maps=not quiet)
            w_loss = (w_loss0 + w_loss1) * sm.DAMSM_LAMBDA
            s_loss0, s_loss1, _ = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size,
                                            top1=not quiet)
            s_loss = (s_loss0 + s_loss1) * sm.DAMSM_LAMBDA
            errG_total = errG_total + w_loss + s_loss
            if not quiet:
                logs += 'w_loss: %.2f s_loss: %.2f ' % (w_loss.item(), s_loss.item())

    if not use_obj:
        return errG_total, logs
    objss_g_loss = _obj_g_term(netObjSSD, fake_imgs[-1], seg_conditions[-1], slabels_emb,
                               raw_bt_c_codes, rois, num_rois, False)
    objls_g_loss = _obj_g_term(netObjLSD, fake_imgs[-1], seg_conditions[-1], slabels_emb,
                               raw_bt_c_codes, fm_rois, num_rois, True)
    # reference: `if float(loss) > 0` -- the term is the python int 0 when no box of that scale
    # exists and a strictly positive BCE otherwise, so the tensor test is equivalent (and sync-free)
    for tag, term in (('objss_g_loss', objss_g_loss), ('objls_g_loss', objls_g_loss)):
        if torch.is_tensor(term) if quiet else float(term) > 0:
            if not quiet:
                logs += '%s: %.2f ' % (tag, term.item())
            errG_total = errG_total + term
    return errG_total, logs


def damsm_term(image_encoder, fake_img, words_embs, sent_emb, match_labels, cap_lens, class_ids):
    """the DAMSM term of G_loss on the last stage's image -> (word part, sentence part), both x DAMSM_LAMBDA
    (reference miscc/losses.py:421-432)"""
    batch_size = fake_img.size(0)
    region_features, cnn_code = image_encoder(fake_img)
    w0, w1, _, _ = words_loss(region_features, words_embs, match_labels, cap_lens, class_ids, batch_size,
                              top1=False, need_att_maps=False)
    s0, s1, _ = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size, top1=False)
    lam = cfg.TRAIN.SMOOTH.DAMSM_LAMBDA
    return (w0 + w1) * lam, (s0 + s1) * lam


def _g_loss_streams(netsPatD, netsShpD, netObjSSD, netObjLSD, image_encoder, fake_imgs, seg_conditions,
                    words_embs, sent_emb, slabels_emb, raw_bt_c_codes, match_labels, cap_lens, class_ids,
                    rois, fm_rois, num_rois, use_obj, streams, damsm_pre=None):
    """The terms of G_loss (same arithmetic, same summation order) issued round-robin on `streams`.
    damsm_pre = (terms, stream): the DAMSM term was already issued on `stream` (trainer: beside the discriminator
    updates, it does not depend on them); this function joins that stream instead of computing the term."""
    numDs = len(netsPatD)
    batch_size = fake_imgs[0].size(0)
    sm = cfg.TRAIN.SMOOTH
    main = torch.cuda.current_stream()
    for s_ in streams:
        s_.wait_stream(main)
    jobs = []

    def damsm():
        return damsm_term(image_encoder, fake_imgs[numDs - 1], words_embs, sent_emb, match_labels, cap_lens, class_ids)

    def pat(i):
        p = _net(netsPatD[i])
        features = netsPatD[i](fake_imgs[i])
        loss = _bce(p.COND_DNET(features, sent_emb), 1)
        if p.UNCOND_DNET is not None:
            loss = _bce(p.UNCOND_DNET(features), 1) * sm.UNCOND_LAMBDA + loss * sm.TXT_LAMBDA
        return loss

    def shp(i):
        n = _net(netsShpD[i])
        return _bce(n.UNCOND_DNET(netsShpD[i](fake_imgs[i], seg_conditions[i])), 1) * sm.SHP_LAMBDA

    # the longest chain of small launches first: it overlaps everything
    order = [("damsm", damsm)] if damsm_pre is None else []
    for i in range(numDs):
        order += [("pat%d" % i, lambda i=i: pat(i)), ("shp%d" % i, lambda i=i: shp(i))]
    if use_obj:
        order += [("objss", lambda: _obj_g_term(netObjSSD, fake_imgs[-1], seg_conditions[-1], slabels_emb,
                                                raw_bt_c_codes, rois, num_rois, False)),
                  ("objls", lambda: _obj_g_term(netObjLSD, fake_imgs[-1], seg_conditions[-1], slabels_emb,
                                                raw_bt_c_codes, fm_rois, num_rois, True))]
    terms = {}
    if damsm_pre is not None:
        terms["damsm"] = damsm_pre[0]
    for j, (name, fn) in enumerate(order):
        with torch.cuda.stream(streams[j % len(streams)]):
            terms[name] = fn()
    for s_ in streams:
        main.wait_stream(s_)
    if damsm_pre is not None:
        main.wait_stream(damsm_pre[1])
    # the reference's summation order: pat_i, shp_i (i ascending), DAMSM after the last pair, then the object terms
    total = 0
    for i in range(numDs):
        total = total + terms["pat%d" % i]
        total = total + terms["shp%d" % i]
    total = total + terms["damsm"][0] + terms["damsm"][1]
    for name in ("objss", "objls"):
        if name in terms and torch.is_tensor(terms[name]):
            total = total + terms[name]
    return total


##################################################################
def KL_loss(mu, logvar):
    # -0.5 * mean(1 + log(sigma^2) - mu^2 - sigma^2)
    return torch.mean(1 + logvar - mu.pow(2) - logvar.exp()) * (-0.5)
