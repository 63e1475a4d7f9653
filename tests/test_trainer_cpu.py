"""The product's HOST code (model.py wiring, losses, roi / feature selection helpers, flat-arena
trainer with its update order, EMA, use_obj switch) executed on CPU with every kernel replaced by its
CPU definition (tests/cpu_ops_shim.py) and compared with the oracle's train_step on the same seeded
inputs and weights.  What this cannot see -- the kernels themselves -- is what the `-m gpu` tests
cover; what it does see on every CPU-only round is everything around them."""
import random

import pytest
import torch

import cpu_ops_shim
from conftest import rel_l2


class _ConstEncoder(object):
    """Constant image encoder: the DAMSM terms get fixed region features (see tests/test_modules_gpu.py)."""

    def __init__(self, regions, code):
        self.regions, self.code = regions.detach(), code.detach()

    def __call__(self, x):
        return self.regions, self.code

    def parameters(self):
        return []

    def eval(self):
        return self


def _sd_of(m):
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running_" not in k:
            v.requires_grad_(True)
    return sd


@pytest.mark.parametrize("branch_num,B,use_obj", [(1, 2, False), (2, 2, True)])
def test_product_trainer_host_logic_matches_oracle(monkeypatch, branch_num, B, use_obj):
    import model as M
    import synth_batch
    import trainer as T
    from oracle import ref_harness as rh, torch_model as tm
    from miscc.config import cfg
    cpu_ops_shim.install(monkeypatch)
    monkeypatch.setattr(cfg.TREE, "BRANCH_NUM", branch_num)
    monkeypatch.setattr(cfg.TRAIN, "BATCH_SIZE", B)
    monkeypatch.setattr(cfg.TRAIN, "NET_G", '')
    torch.set_num_threads(8)
    dev = torch.device("cpu")

    class DS(object):
        num_classes = 80
    ds = DS()
    g0 = torch.Generator().manual_seed(321)
    regions_c, code_c = torch.randn(B, 256, 17, 17, generator=g0), torch.randn(B, 256, generator=g0)
    ds.image_encoder = _ConstEncoder(regions_c, code_c)
    tr = T.condGANTrainer('', None, ds, device=dev)
    tr.batch_size = B
    assert tr.use_obj == (branch_num >= 2)
    pat_cls = (M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256)[:branch_num]
    shp_cls = (M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256)[:branch_num]
    nets = [None, ds.image_encoder, rh.seeded_state_(M.G_NET(80), 161),
            [rh.seeded_state_(c(), 162 + i) for i, c in enumerate(pat_cls)],
            [rh.seeded_state_(c(80), 165 + i) for i, c in enumerate(shp_cls)],
            rh.seeded_state_(M.OBJ_SS_D_NET(80), 168), rh.seeded_state_(M.OBJ_LS_D_NET(80), 169), 0]
    sds = {"G": _sd_of(nets[2]), "pat": [_sd_of(m) for m in nets[3]], "shp": [_sd_of(m) for m in nets[4]],
           "objss": _sd_of(nets[5]), "objls": _sd_of(nets[6])}
    adam = lambda sd: torch.optim.Adam(tm.params_of(sd), lr=2e-4, betas=(0.5, 0.999))   # noqa: E731
    opts = {"G": adam(sds["G"]), "pat": [adam(s) for s in sds["pat"]], "shp": [adam(s) for s in sds["shp"]],
            "objss": adam(sds["objss"]), "objls": adam(sds["objls"])}
    ema = [p.detach().clone() for p in tm.params_of(sds["G"])]
    for m in [nets[2], nets[5], nets[6]] + nets[3] + nets[4]:
        m.train()
    tr.build_models = lambda: nets
    tr.setup()
    b = synth_batch.make_batch(B, seed=78, branch_num=branch_num)
    b2 = synth_batch.make_batch(B, seed=78, branch_num=branch_num)      # independent copy for the product
    tr.netG.ca_net.fixed_eps = b2["ca_eps"]

    random.seed(9)
    want = tm.train_step(sds, opts, ema, b, image_encoder=_ConstEncoder(regions_c, code_c), use_obj=use_obj)
    random.seed(9)
    got = tr.train_step(b2, noise=b2["noise"])

    keys = ["errPatD%d" % i for i in range(branch_num)] + ["errShpD%d" % i for i in range(branch_num)] + ["errG", "kl"]
    if use_obj:
        keys += [k for k in ("errObjSSD", "errObjLSD") if k in want]
    for k in keys:
        assert k in got, k
        assert abs(got[k].item() - want[k].item()) < 1e-4 * abs(want[k].item()) + 1e-6, (k, got[k].item(), want[k].item())
    if not use_obj:
        assert "errObjSSD" not in got and "errObjLSD" not in got
    assert len(got["fake_imgs"]) == branch_num
    assert rel_l2(got["fake_imgs"][-1], want["fake_imgs"][-1]) < 1e-5
    # parameters after the nine (or three) Adam steps, and the EMA buffer
    for name, module, sd in [("G", tr.netG, sds["G"]), ("pat0", tr.netsPatD[0], sds["pat"][0]),
                             ("shp0", tr.netsShpD[0], sds["shp"][0])]:
        num = den = 0.0
        for k, p in module.named_parameters():
            num += float((p.detach() - sd[k].detach()).double().pow(2).sum())
            den += float(sd[k].detach().double().pow(2).sum())
        # first Adam step = lr * g / (|g| + eps): elements whose gradient is at rounding level flip by
        # 2 * lr when the summation order differs (the product evaluates the layout stem once for the
        # real and the fake pass), hence 1e-4 on the parameters while losses and images agree to 1e-5
        assert (num / den) ** 0.5 < 1e-4, (name, (num / den) ** 0.5)
    order = [k for k, _ in tr.netG.named_parameters()]
    ema_sorted = {k: a for k, a in zip(sorted(k for k in sds["G"] if sds["G"][k].requires_grad), ema)}
    assert rel_l2(tr.avg_param_G, torch.cat([ema_sorted[k].reshape(-1) for k in order])) < 1e-6


def test_real_data_directory_through_the_training_step(monkeypatch):
    """tests/golden/data_tiny -> TrainDataset -> default collate -> prepare_data -> batch_dict ->
    condGANTrainer.train_step with the frozen caption encoder and the GloVe table (stage-1 tree):
    the whole real-data contract of reference trainer.py:357-472 on CPU, against the oracle step fed
    with the embeddings the oracle's own RNN_ENCODER restatement computes."""
    import os
    import numpy as np
    from torch.utils.data.dataloader import default_collate
    from conftest import ROOT
    import model as M
    import trainDataset
    import trainer as T
    from oracle import ref_harness as rh, torch_model as tm
    from miscc.config import cfg
    cpu_ops_shim.install(monkeypatch)
    monkeypatch.setattr(cfg.TREE, "BRANCH_NUM", 1)
    monkeypatch.setattr(cfg.TRAIN, "BATCH_SIZE", 2)
    monkeypatch.setattr(cfg.TRAIN, "NET_G", '')
    torch.set_num_threads(8)
    ds = trainDataset.TrainDataset(os.path.join(ROOT, "tests", "golden", "data_tiny"), "train", base_size=64,
                                   device_hmaps=True)
    nc = ds.num_classes
    B = 2
    g0 = torch.Generator().manual_seed(11)
    regions_c, code_c = torch.randn(B, 256, 17, 17, generator=g0), torch.randn(B, 256, generator=g0)
    ds.image_encoder = _ConstEncoder(regions_c, code_c)
    ds.text_encoder = rh.seeded_state_(M.RNN_ENCODER(ds.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM), 91).eval()
    for p in ds.text_encoder.parameters():
        p.requires_grad_(False)
    tr = T.condGANTrainer('', None, ds, device=torch.device("cpu"))
    tr.batch_size = B
    nets = [ds.text_encoder, ds.image_encoder, rh.seeded_state_(M.G_NET(nc), 92),
            [rh.seeded_state_(M.PAT_D_NET64(), 93)], [rh.seeded_state_(M.SHP_D_NET64(nc), 94)],
            rh.seeded_state_(M.OBJ_SS_D_NET(nc), 95), rh.seeded_state_(M.OBJ_LS_D_NET(nc), 96), 0]
    sds = {"G": _sd_of(nets[2]), "pat": [_sd_of(nets[3][0])], "shp": [_sd_of(nets[4][0])]}
    adam = lambda sd: torch.optim.Adam(tm.params_of(sd), lr=2e-4, betas=(0.5, 0.999))   # noqa: E731
    opts = {"G": adam(sds["G"]), "pat": [adam(sds["pat"][0])], "shp": [adam(sds["shp"][0])]}
    ema = [p.detach().clone() for p in tm.params_of(sds["G"])]
    for m in [nets[2], nets[5], nets[6]] + nets[3] + nets[4]:
        m.train()
    tr.build_models = lambda: nets
    tr.setup()
    assert tr.clabels_emb.shape == (nc, 50)

    np.random.seed(4)
    collated = default_collate([ds[0], ds[1]])                        # two images with boxes
    prepared = trainDataset.prepare_data(collated, None, nc)
    batch = trainDataset.batch_dict(prepared, tr.clabels_emb)
    noise = torch.randn(B, cfg.GAN.Z_DIM, generator=g0)
    eps = torch.randn(B, cfg.GAN.CONDITION_DIM, generator=g0)
    tr.netG.ca_net.fixed_eps = eps

    # the oracle's inputs: embeddings from its own encoder restatement (reference trainer.py:367-383)
    enc_sd = {k: v.detach() for k, v in ds.text_encoder.state_dict().items()}
    caps, lens = batch["captions"], batch["cap_lens"]
    words, sent = tm.rnn_encoder_forward(enc_sd, caps, lens, int(lens.max()))
    nw = words.size(2)
    glove = ds.glove_embed.weight.detach()[batch["glove_captions"].reshape(-1)].view(B, -1, 50)[:, :nw].transpose(1, 2)
    ob = dict(batch, words_embs=words, sent_emb=sent, glove_words_embs=glove, mask=(caps == 0)[:, :nw],
              noise=noise, ca_eps=eps)
    random.seed(5)
    want = tm.train_step(sds, opts, ema, ob, image_encoder=_ConstEncoder(regions_c, code_c), use_obj=False)
    random.seed(5)
    got = tr.train_step(batch, noise=noise)
    for k in ("errPatD0", "errShpD0", "errG", "kl"):
        assert abs(got[k].item() - want[k].item()) < 1e-4 * abs(want[k].item()) + 1e-6, (k, got[k].item(), want[k].item())
    assert rel_l2(got["fake_imgs"][0], want["fake_imgs"][0]) < 1e-5
    assert all(torch.isfinite(v).all() for v in got.values() if torch.is_tensor(v))


@pytest.mark.parametrize("device_imgs", [False, True])
def test_train_loop_checkpoints_and_resume(monkeypatch, tmp_path, device_imgs):
    """condGANTrainer.train() over a DataLoader of the committed data directory (reference loader
    tuples; with `device_imgs` the decoded 8-bit images, resized inside prepare_data), one epoch of two steps
    on the CPU shim: the checkpoint files have the reference's names,
    the generator is saved with the EMA weights swapped in, and cfg.TRAIN.NET_G resumes from them
    (reference trainer.py:155-193, 251-273)."""
    import os
    from conftest import ROOT
    import model as M
    import trainDataset
    import trainer as T
    from oracle import ref_harness as rh
    from miscc.config import cfg
    cpu_ops_shim.install(monkeypatch)
    monkeypatch.setattr(cfg.TREE, "BRANCH_NUM", 1)
    monkeypatch.setattr(cfg.TRAIN, "BATCH_SIZE", 2)
    monkeypatch.setattr(cfg.TRAIN, "NET_G", '')
    monkeypatch.setattr(cfg.TRAIN, "MAX_EPOCH", 1)
    monkeypatch.setattr(cfg.TRAIN, "FLAG", True)
    torch.set_num_threads(8)
    ds = trainDataset.TrainDataset(os.path.join(ROOT, "tests", "golden", "data_tiny"), "train", base_size=64,
                                   device_hmaps=True, device_imgs=device_imgs)
    g0 = torch.Generator().manual_seed(12)
    ds.image_encoder = _ConstEncoder(torch.randn(2, 256, 17, 17, generator=g0), torch.randn(2, 256, generator=g0))
    ds.text_encoder = rh.seeded_state_(M.RNN_ENCODER(ds.n_words, nhidden=cfg.TEXT.EMBEDDING_DIM), 91).eval()
    for p in ds.text_encoder.parameters():
        p.requires_grad_(False)
    torch.manual_seed(0)
    loader = trainDataset.build_loader(ds, 2, workers=0, shuffle=False)
    tr = T.condGANTrainer(str(tmp_path), loader, ds, device=torch.device("cpu"))
    tr.train()
    assert tr.gen_iterations == 2
    model_dir = tmp_path / "Model"
    names = sorted(os.listdir(str(model_dir)))
    assert names == ["netG_epoch_0.pth", "netG_epoch_1.pth", "netObjLSD.pth", "netObjSSD.pth",
                     "netPatD0.pth", "netShpD0.pth"]
    saved = torch.load(str(model_dir / "netG_epoch_1.pth"))
    assert list(saved.keys()) == list(tr.netG.state_dict().keys())
    # saved generator = EMA weights (flat order = parameter order), live generator = raw weights
    flat_saved = torch.cat([saved[k].reshape(-1) for k, _ in tr.netG.named_parameters()])
    assert torch.allclose(flat_saved, tr.avg_param_G) and not torch.allclose(flat_saved, tr.optimizerG.arena.flat)
    # resume
    monkeypatch.setattr(cfg.TRAIN, "NET_G", str(model_dir / "netG_epoch_1.pth"))
    tr2 = T.condGANTrainer(str(tmp_path), loader, ds, device=torch.device("cpu"))
    tr2.setup()
    assert tr2.start_epoch == 2
    assert torch.allclose(tr2.optimizerG.arena.flat, tr.avg_param_G)
    d0 = torch.load(str(model_dir / "netPatD0.pth"))
    assert all(torch.equal(d0[k], v) for k, v in tr2.netsPatD[0].state_dict().items())


def test_sampling_with_ema_weights(monkeypatch):
    """condGANTrainer.sample(): eval-mode generator with the EMA weights swapped in through the arena
    (epoch bumped both ways, training weights restored) against the oracle's g_net under bn_eval()
    on a state dict holding the EMA values."""
    import model as M
    import synth_batch
    import trainer as T
    from oracle import ref_harness as rh, torch_model as tm
    from miscc.config import cfg
    from miscc.utils import form_clabels_feat
    cpu_ops_shim.install(monkeypatch)
    monkeypatch.setattr(cfg.TREE, "BRANCH_NUM", 2)
    monkeypatch.setattr(cfg.TRAIN, "BATCH_SIZE", 2)
    monkeypatch.setattr(cfg.TRAIN, "NET_G", '')
    torch.set_num_threads(8)
    B = 2

    class DS(object):
        num_classes = 80
    tr = T.condGANTrainer('', None, DS(), device=torch.device("cpu"))
    tr.batch_size = B
    G = rh.seeded_state_(M.G_NET(80), 31)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for k, v in G.state_dict().items():
            if "running_mean" in k:
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)
            elif "running_var" in k:
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    nets = [None, None, G, [rh.seeded_state_(M.PAT_D_NET64(), 32), rh.seeded_state_(M.PAT_D_NET128(), 33)],
            [rh.seeded_state_(M.SHP_D_NET64(80), 34), rh.seeded_state_(M.SHP_D_NET128(80), 35)],
            rh.seeded_state_(M.OBJ_SS_D_NET(80), 36), rh.seeded_state_(M.OBJ_LS_D_NET(80), 37), 0]
    tr.build_models = lambda: nets
    tr.setup()
    raw = tr.optimizerG.arena.flat.clone()
    tr.avg_param_G.copy_(raw * 0.9 + 0.01)                      # an EMA buffer that differs from the live weights
    epoch0 = tr.optimizerG.arena.epoch[0]
    b = synth_batch.make_batch(B, seed=55, branch_num=2)
    tr.netG.ca_net.fixed_eps = b["ca_eps"]
    fake, att, bt_att = tr.sample(b, b["noise"], b["words_embs"], b["sent_emb"], b["glove_words_embs"], b["mask"])
    assert torch.equal(tr.optimizerG.arena.flat, raw) and tr.netG.training
    assert tr.optimizerG.arena.epoch[0] == epoch0 + 2
    # oracle: the same state dict with the EMA values in place of the parameters
    sd = {k: v.detach().clone() for k, v in tr.netG.state_dict().items()}
    off = 0
    for k, p in tr.netG.named_parameters():
        sd[k] = tr.avg_param_G[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    cl = form_clabels_feat(b["clabels_emb"], b["rois"][0], b["num_rois"])
    with tm.bn_eval(), torch.no_grad():
        want = tm.g_net(sd, b["noise"], b["sent_emb"], b["words_embs"], b["glove_words_embs"], cl, b["mask"],
                        b["hmaps"], b["rois"], b["fm_rois"], b["num_rois"], b["bt_masks"], b["fm_bt_masks"],
                        int(b["num_rois"].max()), b["ca_eps"], branch_num=2)
    assert len(fake) == 2
    for i in range(2):
        assert rel_l2(fake[i], want[0][i]) < 1e-5, i
    live, _, _ = tr.sample(b, b["noise"], b["words_embs"], b["sent_emb"], b["glove_words_embs"], b["mask"], use_ema=False)
    assert rel_l2(live[1], fake[1]) > 1e-3                      # the swap really changed the weights


def test_arena_refuses_a_weight_whose_gradient_arrives_in_two_pieces():
    """ops' direct weight-gradient sink reports a parameter once per USE, from inside that use's
    backward, and the armed arena hands a bucket to the all-reduce as soon as all its parameters reported.  A weight used
    twice per forward would ship a partial gradient.  The arena detects both forms -- two sink reports, or a sink report
    followed by a contribution through autograd -- and raises; the normal case (one sink report, then AccumulateGrad's
    post-hook, which fires even when autograd was handed None) passes."""
    import trainer as T

    def fresh():
        net = torch.nn.Sequential(torch.nn.Linear(4, 4, bias=False), torch.nn.Linear(4, 4, bias=False))
        arena = T.ParamArena(net)
        fired = []
        arena.arm(lambda s0, e0: fired.append((s0, e0)))
        return net, arena, fired

    # (1) one use through the sink: reports once, the hook that follows is the same use
    net, arena, fired = fresh()
    w1 = net[1].weight
    w1._og_grad_sink[1]()                                  # the sink's report of its single use (raw-pointer add: no version bump)
    arena._mark(1, False)                                  # AccumulateGrad's post-hook of the same pass (autograd got None)
    assert arena._sink_marked == {1}
    # (2) two sink reports of one parameter
    net, arena, fired = fresh()
    net[1].weight._og_grad_sink[1]()
    with pytest.raises(RuntimeError, match="more than once per forward"):
        net[1].weight._og_grad_sink[1]()
    # (3) a sink report, then a contribution through autograd (the version counter of the gradient view moves)
    net, arena, fired = fresh()
    net[1].weight._og_grad_sink[1]()
    with pytest.raises(RuntimeError, match="more than once per forward"):
        net(torch.randn(2, 4)).sum().backward()


def test_autograd_hooks_fire_for_a_parameter_whose_backward_function_returned_none():
    """ParamArena._mark / _autograd_piece rely on the installed torch running BOTH the tensor hook (with
    g = None) and AccumulateGrad's post-accumulate hook for a parameter whose backward function handed autograd no
    gradient -- the direct weight-gradient sink's case (ops._grad_sink).  Pinned here on the installed version: if a
    future torch returns early on an undefined gradient, `_hook_marked` stops seeing sink-only parameters (harmless: the
    sink itself marks them) and this test says so."""
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            return g @ w.t(), None

    w = torch.randn(4, 4, requires_grad=True)
    x = torch.randn(2, 4, requires_grad=True)
    w.grad = torch.zeros_like(w)
    seen = []
    w.register_post_accumulate_grad_hook(lambda p: seen.append("post"))
    w.register_hook(lambda g: seen.append(("tensor", g is None)))
    Fn.apply(x, w).sum().backward()
    assert seen == [("tensor", True), "post"], seen


def test_d_job_order_keeps_the_rng_drawing_jobs_in_the_reference_sequence():
    """a PARTIAL d_job_order list is completed with the unlisted jobs behind it; the check runs on the final
    order.  Every job but the patch discriminators draws from the python RNG (permute_seg), so they must stay in the
    reference sequence; unknown / repeated names raise."""
    import trainer as T
    names = ["errPatD0", "errPatD1", "errPatD2", "errShpD0", "errShpD1", "errShpD2", "errObjSSD", "errObjLSD"]
    assert T.resolve_d_job_order(names, ["errPatD2"]) == ["errPatD2", "errPatD0", "errPatD1"] + names[3:]
    assert T.resolve_d_job_order(names, ["errShpD0", "errPatD2"])[:2] == ["errShpD0", "errPatD2"]
    full = ["errPatD2", "errShpD0", "errShpD1", "errShpD2", "errObjSSD", "errObjLSD", "errPatD1", "errPatD0"]
    assert T.resolve_d_job_order(names, full) == full
    with pytest.raises(ValueError, match="out of the reference order"):
        T.resolve_d_job_order(names, ["errShpD1"])               # would run ShpD1 ahead of ShpD0
    with pytest.raises(ValueError, match="out of the reference order"):
        T.resolve_d_job_order(names, ["errObjLSD", "errObjSSD"])
    with pytest.raises(ValueError, match="do not exist"):
        T.resolve_d_job_order(names, ["errPatD7"])
    with pytest.raises(ValueError, match="do not exist"):
        T.resolve_d_job_order(names, ["errPatD0", "errPatD0"])
