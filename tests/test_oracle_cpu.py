"""CPU tests (no GPU): the oracle is pinned against the reference and its golden vectors, the
C-ABI library loads and exports every declared symbol, and the host-side logic behaves."""
import os
import random
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2

GOLD = os.path.join(ROOT, "tests", "golden")


def _golden():
    return torch.load(os.path.join(GOLD, "step_b2.pt"), weights_only=False)


def _seeded_sd(keys_like, seed):
    """state dict filled BY KEY exactly like oracle.ref_harness.seeded_state_ (needs the module's
    keys/shapes: taken from the product modules, whose state-dict layout equals the reference's)."""
    from oracle import ref_harness as rh
    rh.seeded_state_(keys_like, seed)
    return {k: v.clone() for k, v in keys_like.state_dict().items()}


def _oracle_nets(g):
    """oracle state dicts for every network, seeded like tests/golden/make_golden.py"""
    import model as M
    s = g["seeds"]
    sds = {"G": _seeded_sd(M.G_NET(80), s["G"]),
           "pat": [_seeded_sd(c(), s["pat"] + i) for i, c in enumerate((M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256))],
           "shp": [_seeded_sd(c(80), s["shp"] + i) for i, c in enumerate((M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256))],
           "objss": _seeded_sd(M.OBJ_SS_D_NET(80), s["objss"]),
           "objls": _seeded_sd(M.OBJ_LS_D_NET(80), s["objls"])}
    return sds


def _req(sd):
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    return sd


def test_roi_oracle_matches_reference_golden_vectors():
    from oracle import roi
    z = np.load(os.path.join(GOLD, "roi_align_ref.npz"))
    got = roi.forward(z["feat"], z["rois"], 6, 6, 1 / 16.0)
    assert np.array_equal(got.view(np.uint32), z["out"].view(np.uint32))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_roi_oracle_bit_exact_with_reference_c_loop():
    from oracle import roi
    roi.build()
    rng = np.random.RandomState(3)
    for trial in range(12):
        H, W = rng.randint(4, 40), rng.randint(4, 40)
        feat = rng.randn(3, 4, H, W).astype(np.float32)
        rois = np.zeros((24, 5), np.float32)
        rois[:, 0] = rng.randint(0, 3, 24)
        sc = [1 / 16., 1.0, 0.5][trial % 3]
        xy = rng.uniform(-5, W / sc, (24, 2)); wh = rng.uniform(0, W / sc, (24, 2))
        rois[:, 1:3] = xy; rois[:, 3:5] = xy + wh
        if trial % 2 == 0:
            rois[:, 1:] = np.round(rois[:, 1:])
        ah, aw = [(6, 6), (2, 2), (7, 5)][trial % 3]
        a = roi.forward(feat, rois, ah, aw, sc)
        b = roi.reference_forward(feat, rois, ah, aw, sc)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), trial


def test_roi_backward_oracle_is_adjoint_of_forward():
    """the restated CUDA backward is the transpose of the (linear in features) forward"""
    from oracle import roi
    rng = np.random.RandomState(9)
    feat = rng.randn(2, 3, 16, 16).astype(np.float32)
    rois = np.array([[0, 3, 4, 100, 90], [1, 0, 0, 255, 255], [1, 17.5, 30.25, 60, 80]], np.float32)
    g = rng.randn(3, 3, 6, 6).astype(np.float32)
    lhs = float((roi.forward(feat, rois, 6, 6, 1 / 16.) * g).sum())
    rhs = float((roi.backward(g, rois, feat.shape, 1 / 16.) * feat).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_oracle_generator_matches_reference_golden():
    from oracle import torch_model as tm
    import synth_batch
    g = _golden()
    b = synth_batch.make_batch(g["B"], seed=g["seeds"]["batch"])
    sds = _oracle_nets(g)
    cl = tm.form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    with torch.no_grad():
        fake, bt, atts, bt_atts, mu, logvar = tm.g_net(
            sds["G"], b["noise"], b["sent_emb"], b["words_embs"], b["glove_words_embs"], cl, b["mask"],
            b["hmaps"], b["rois"], b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"],
            int(b["num_rois"].max()), b["ca_eps"])
    assert rel_l2(fake[0], g["fake64"]) < 1e-5
    assert rel_l2(fake[1], g["fake128"]) < 1e-5
    assert rel_l2(fake[2][:, :, ::2, ::2], g["fake256_s2"]) < 1e-5
    assert rel_l2(bt[1], g["bt_c_codes"][1]) < 1e-5
    assert rel_l2(atts[1][:, :, ::4, ::4], g["att128_s4"]) < 1e-5
    assert rel_l2(bt_atts[0][:, :, ::2, ::2], g["bt_att64_s2"]) < 1e-5
    assert rel_l2(mu, g["mu"]) < 1e-6


def test_oracle_losses_match_reference_golden():
    from oracle import torch_model as tm
    import synth_batch
    from oracle import torch_encoders as encoders
    from oracle import ref_harness as rh
    g = _golden()
    s = g["seeds"]
    b = synth_batch.make_batch(g["B"], seed=s["batch"])
    sds = _oracle_nets(g)
    for k in ("pat", "shp"):
        sds[k] = [_req(x) for x in sds[k]]
    _req(sds["objss"]); _req(sds["objls"]); _req(sds["G"])
    cl = tm.form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    fake, bt, _, _, mu, logvar = tm.g_net(sds["G"], b["noise"], b["sent_emb"], b["words_embs"],
                                          b["glove_words_embs"], cl, b["mask"], b["hmaps"], b["rois"],
                                          b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"],
                                          int(b["num_rois"].max()), b["ca_eps"])
    btd = [c.detach() for c in bt]

    def gnorms(sd):
        return {k: (v.grad.norm().item() if v.grad is not None else 0.0) for k, v in sd.items() if v.requires_grad}

    def check_grads(sd, want, tol=2e-3):
        got = gnorms(sd)
        for k, w in want.items():
            assert abs(got[k] - w) <= tol * max(w, 1e-6) + 1e-7, (k, got[k], w)

    for i in range(3):
        e = tm.pat_d_loss(sds["pat"][i], b["imgs"][i], fake[i], b["sent_emb"])
        e.backward()
        assert abs(e.item() - g["errPatD%d" % i]) < 1e-5
        check_grads(sds["pat"][i], g["gradPatD%d" % i])
    for i in range(3):
        random.seed(100 + i)
        e = tm.shp_d_loss(sds["shp"][i], b["imgs"][i], fake[i], b["hmaps"][i], b["rois"][i], b["num_rois"])
        e.backward()
        assert abs(e.item() - g["errShpD%d" % i]) < 1e-5
        check_grads(sds["shp"][i], g["gradShpD%d" % i])
    random.seed(200)
    e = tm.obj_d_loss(sds["objss"], 3, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], btd[-1],
                      b["rois"][0], b["num_rois"], False)
    e.backward()
    assert abs(e.item() - g["errObjSSD"]) < 1e-5
    check_grads(sds["objss"], g["gradObjSSD"])
    random.seed(201)
    e = tm.obj_d_loss(sds["objls"], 4, b["imgs"][-1], fake[-1], b["hmaps"][-1], b["clabels_emb"], btd[-1],
                      b["fm_rois"], b["num_rois"], True)
    e.backward()
    assert abs(e.item() - g["errObjLSD"]) < 1e-5
    check_grads(sds["objls"], g["gradObjLSD"])

    enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), s["inception"]))
    rh.seeded_state_(enc.emb_features, s["enc_proj"]); rh.seeded_state_(enc.emb_cnn_code, s["enc_proj"] + 1)
    enc.eval()
    for k in ("pat", "shp"):
        for sd in sds[k]:
            for v in sd.values():
                v.grad = None
    labels = torch.arange(g["B"])
    total, parts = tm.g_loss(sds, enc, fake, b["hmaps"], b["words_embs"], b["sent_emb"], b["clabels_emb"],
                             btd[-1], labels, b["cap_lens"], b["class_ids"], b["rois"], b["fm_rois"],
                             b["num_rois"])
    kl = tm.kl_loss(mu, logvar)
    (total + kl).backward(retain_graph=True)
    assert abs((total + kl).item() - g["errG"]) < 1e-3 * g["errG"]
    assert abs(kl.item() - g["kl"]) < 1e-6
    assert abs(parts["w_loss"].item() / 100 - g["w_loss"]) < 1e-4
    assert abs(parts["s_loss"].item() / 100 - g["s_loss"]) < 1e-4
    # through the (ill-conditioned, see tests/test_modules_gpu.py::_ConstEncoder) Inception gradient
    # only a loose bound holds even between two CPU runs of the same arithmetic
    check_grads(sds["G"], g["gradG"], tol=5e-2)
    # constant-encoder variant: tight
    for v in sds["G"].values():
        v.grad = None
    for k in ("pat", "shp"):
        for sd in sds[k]:
            for v in sd.values():
                v.grad = None
    with torch.no_grad():
        regions, code = enc(fake[2].detach())
    assert rel_l2(regions[:, ::8], g["regions_s"]) < 1e-4
    total2, _ = tm.g_loss(sds, lambda x: (regions, code), fake, b["hmaps"], b["words_embs"], b["sent_emb"],
                          b["clabels_emb"], btd[-1], labels, b["cap_lens"], b["class_ids"], b["rois"],
                          b["fm_rois"], b["num_rois"])
    (total2 + tm.kl_loss(mu, logvar)).backward()
    check_grads(sds["G"], g["gradG_constenc"], tol=2e-3)
    assert rel_l2(sds["G"]["img_net3.img.0.weight"].grad, g["gradG_constenc_img3_w"]) < 1e-4
    assert rel_l2(sds["G"]["h_net3_main.att.conv_context.weight"].grad, g["gradG_constenc_att_ctx_w"]) < 1e-4
    assert rel_l2(sds["G"]["ca_net.fc.weight"].grad, g["gradG_constenc_ca_fc_w"]) < 1e-4


def test_c_abi_library_exports_every_declared_symbol():
    """include/objgan_hip.h <-> libobjgan_hip.so <-> ctypes table agree (no compute calls)."""
    from objgan_hip import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "objgan_hip.h")).read()
    declared = set(re.findall(r"\b(objgan_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    table = set(_lib.SIGNATURES) | set(_lib.LONG_RETURN)
    assert declared == table, declared ^ table
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.objgan_conv_packed_floats(388, 194, 9) == 512 * 208 * 9 * 3 // 2 + 1024   # 6-byte bf16x3 bank + |w| maxima


def test_workspace_queries_are_host_only_and_consistent():
    """The *_ws_floats entry points plan on the host (no HIP call): they must answer without a GPU, and the answers must
    cover what the kernels put there -- split partials, and in the bf16 mode the bf16 copies of the operands."""
    from objgan_hip import _lib
    lib = _lib.load()
    up = lambda n, m: (n + m - 1) // m * m
    # forward 192 -> 384, 4x4 stride 2 at 128^2, N = 16: a large unsplit launch
    geo = (16, 192, 128, 128, 0, 0, 384, 192, 16, 0, 16, 64, 64, 2, 64, 64, 1, 1, 0, 0)
    assert lib.objgan_conv_igemm_ws_floats(*geo, 2, 0) == 0                    # bf16x3: nothing to reduce
    blocked = up(16 * 128 * 128 * up(192, 16) // 2, 4)
    assert lib.objgan_conv_igemm_ws_floats(*geo, 1, 0) == blocked              # bf16: the channel-blocked copy of x
    # a discriminator head: 8x8 maps, K = 16 * 768 -- split along K in every mode, slots are whole outputs
    head = (16, 768, 8, 8, 0, 0, 1536, 768, 16, 0, 16, 4, 4, 2, 4, 4, 1, 1, 0, 0)
    n2 = lib.objgan_conv_igemm_ws_floats(*head, 2, 0)
    assert n2 > 0 and n2 % (16 * 1536 * 4 * 4) == 0
    n1 = lib.objgan_conv_igemm_ws_floats(*head, 1, 0)
    assert n1 >= up(16 * 8 * 8 * 768 // 2, 4) + 16 * 1536 * 4 * 4
    # stride-2 data gradient in phases: only the bf16 mode needs a workspace (the copy of dY)
    assert lib.objgan_conv_dgrad_s2_phases_ws_floats(16, 384, 64, 64, 2) == 0
    assert lib.objgan_conv_dgrad_s2_phases_ws_floats(16, 384, 64, 64, 1) == up(16 * 64 * 64 * 384 // 2, 4)
    # weight gradient: split partials; bf16 mode adds the blocked copy of x and the bf16 copy of dy
    wg = (16, 192, 128, 128, 0, 0, 384, 64, 64, 4, 2, 1)
    w2, w1 = lib.objgan_conv_wgrad_ws_floats(*wg, 2), lib.objgan_conv_wgrad_ws_floats(*wg, 1)
    assert w2 > 0 and w2 % (384 * 192 * 16) == 0
    assert w1 >= blocked + 16 * 384 * 64 * 64 // 2
    # ROIAlign backward, ordered: per image [header | touched pixels | anchor starts | sorted samples | geometry]
    r = lib.objgan_roi_align_backward_ws_floats(16, 160, 384, 64, 64, 6, 6)
    assert r > 0 and r % 16 == 0
    assert lib.objgan_roi_align_backward_ws_floats(32, 320, 384, 64, 64, 6, 6) > 0         # B = 32: still the table path
    assert lib.objgan_roi_align_backward_ws_floats(16, 600, 384, 64, 64, 6, 6) == 0        # > 512 rois: scatter path
    assert lib.objgan_norm_ws_floats(16, 96, 128 * 128, 1) > 2 * 96                         # totals + partial slots
    assert lib.objgan_channel_sum_ws_floats(16, 96, 128 * 128) >= 0


def test_state_dict_keys_match_reference_contract():
    import model as M
    keys = set(M.G_NET(80).state_dict().keys())
    for k in ("ca_net.fc.weight", "h_net1_sent.fc.0.weight", "h_net1_sent.fc.1.running_mean",
              "h_net1_sent.upsample1.1.weight", "h_net1_sent.upsample2.2.bias",
              "h_net1_hmap.conv3x3.1.bias", "h_net2_hmap.downsample1.0.weight",
              "h_net1_main.bt_att.conv_context.weight", "h_net3_main.att.conv_context.weight",
              "h_net2_main.residual.2.block.1.weight", "h_net1_main.residual.6.block.5.weight",
              "h_net3_main.upsample.1.weight", "img_net3.img.0.weight"):
        assert k in keys, k
    assert sum(p.numel() for p in M.G_NET(80).parameters()) == 19344452
    d = M.OBJ_LS_D_NET(80).state_dict()
    for k in ("img_code.0.weight", "img_code.9.running_var", "shp_code.1.bias", "roi_code.0.weight",
              "COND_DNET.jointConv.0.weight", "COND_DNET.outlogits.0.bias", "UNCOND_DNET.outlogits.0.weight"):
        assert k in d, k
    assert sum(p.numel() for p in M.PAT_D_NET256().parameters()) == 13304450
    assert sum(p.numel() for p in M.SHP_D_NET64(80).parameters()) == 6239821
    assert sum(p.numel() for p in M.OBJ_SS_D_NET(80).parameters()) == 5545934
    assert sum(p.numel() for p in M.OBJ_LS_D_NET(80).parameters()) == 12625358


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_reference_and_product_state_dicts_are_identical_in_layout():
    from oracle import ref_harness as rh
    import model as M
    ref = rh.load_reference(3, 2)
    for mine, theirs in ((M.G_NET(80), ref.model.G_NET(80)), (M.PAT_D_NET64(), ref.model.PAT_D_NET64()),
                         (M.SHP_D_NET256(80), ref.model.SHP_D_NET256(80)),
                         (M.OBJ_SS_D_NET(80), ref.model.OBJ_SS_D_NET(80)),
                         (M.OBJ_LS_D_NET(80), ref.model.OBJ_LS_D_NET(80))):
        a = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
        assert a == b


def test_product_path_refuses_to_run_without_gpu():
    from objgan_hip import ops, ObjganHipError
    with pytest.raises(ObjganHipError):
        ops.roi_align(torch.zeros(1, 2, 8, 8), torch.zeros(1, 5), 6, 6, 1.0)
    with pytest.raises(ObjganHipError):
        ops.norm_act(torch.zeros(1, 2, 4, 4))


def test_feat_select_and_permute_seg_host_logic():
    from miscc.utils import feat_select, permute_seg
    from oracle import torch_model as tm
    import synth_batch
    b = synth_batch.make_batch(4, seed=3)
    pooled = torch.randn(4, 10, 8, 4, 4)
    raw = torch.randn(4, 10, 48)
    for large in (False, True):
        f1, c1, b1 = feat_select(pooled, raw, b["fm_rois"], b["num_rois"], large)
        f2, c2, b2 = tm.feat_select(pooled, raw, b["fm_rois"], b["num_rois"], large)
        assert torch.equal(f1, f2) and torch.equal(c1, c2) and torch.equal(b1, b2)
    random.seed(5)
    s1, v1 = permute_seg(b["hmaps"][0], b["rois"][0], b["num_rois"])
    random.seed(5)
    s2, v2 = tm.permute_seg(b["hmaps"][0], b["rois"][0], b["num_rois"])
    assert v1 == v2 and torch.equal(s1, s2)


def test_oracle_rnn_encoder_matches_reference_golden():
    """oracle restatement of RNN_ENCODER.forward vs the output of the reference class itself
    (tests/golden/rnn_encoder_ref.pt, written by make_golden.py --rnn)."""
    from oracle import torch_model as tm
    from oracle import ref_harness as rh
    import model as M
    import synth_batch
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "rnn_encoder_ref.pt"))
    b = synth_batch.make_batch(gold["B"], seed=1234)
    enc = rh.seeded_state_(M.RNN_ENCODER(gold["ntoken"], nhidden=256), gold["seed"])
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    assert set(sd) == {"encoder.weight", "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0",
                       "rnn.bias_hh_l0", "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse",
                       "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse"}
    words, sent = tm.rnn_encoder_forward(sd, b["captions"], b["cap_lens"], 12)
    assert rel_l2(words, gold["words_emb"]) < 1e-5
    assert rel_l2(sent, gold["sent_emb"]) < 1e-5
    lens = b["cap_lens"].tolist()
    for i, n in enumerate(lens):
        assert float(words[i, :, n:].abs().sum()) == 0.0
