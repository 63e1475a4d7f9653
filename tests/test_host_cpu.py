"""Host-side logic of the conv wrappers that needs no GPU: the algebra of the phased upBlock
convolution (ops._UpConv3x3Fn) with the three device kernels replaced by their torch CPU
definitions, and the filter-bank caches."""
import pytest
import torch
import torch.nn.functional as F


def _cpu_kernels(ops, monkeypatch):
    """conv forward / input gradient / weight gradient as torch defines them (same contracts as
    ops._conv_fwd / _conv_dgrad / _conv_wgrad)."""
    monkeypatch.setattr(ops, "_conv_fwd",
                        lambda x, w, bias, stride, pad, refl, upsample, act: F.conv2d(x, w, None, stride, pad))
    monkeypatch.setattr(ops, "_conv_dgrad",
                        lambda g, w, N, Cin, H, W, stride, pad, refl, upsample, cacheable=True:
                        torch.nn.grad.conv2d_input((N, Cin, H, W), w, g, stride=stride, padding=pad))
    monkeypatch.setattr(ops, "_conv_wgrad",
                        lambda x, g, Cout, k, stride, pad, refl, upsample:
                        torch.nn.grad.conv2d_weight(x, (Cout, x.shape[1], k, k), g, stride=stride, padding=pad))
    monkeypatch.setattr(ops, "_chk", lambda *a: None)


def test_phased_upblock_conv_is_the_lifted_conv(monkeypatch):
    """nearest x2 + 3x3 conv == transposed stride-2 4x4 conv with W4 = A w A^T (forward), and its
    gradients are the stride-2 forward / weight gradient folded back with A^T . A."""
    from objgan_hip import ops
    _cpu_kernels(ops, monkeypatch)
    g = torch.Generator().manual_seed(0)
    for (N, C, H, W, M) in [(2, 40, 8, 8, 96), (1, 33, 5, 7, 34), (2, 48, 4, 4, 64)]:
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(M, C, 3, 3, generator=g) / (C * 9) ** 0.5
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        yr = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, padding=1)
        gy = torch.randn(yr.shape, generator=g)
        yr.backward(gy)
        xd, wd = x.clone().requires_grad_(), w.clone().requires_grad_()
        assert ops._up_phased_ok(xd, wd, None, 1, 1, "zeros", True, None)
        yd = ops.conv2d(xd, wd, None, 1, 1, "zeros", True, None)
        yd.backward(gy)
        rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
        assert rel(yd.detach(), yr.detach()) < 2e-6
        assert rel(xd.grad, xr.grad) < 2e-6 and rel(wd.grad, wr.grad) < 2e-6


def test_phased_form_is_only_taken_where_it_is_defined():
    from objgan_hip import ops
    x = torch.zeros(2, 40, 8, 8)
    w = torch.zeros(64, 40, 3, 3)
    ok = lambda *a, **k: ops._up_phased_ok(*a, **k)          # noqa: E731
    assert ok(x, w, None, 1, 1, "zeros", True, None)
    assert not ok(x, w, None, 1, 1, "zeros", False, None)     # no upsample
    assert not ok(x, w, torch.zeros(64), 1, 1, "zeros", True, None)
    assert not ok(x, w, None, 1, 1, "zeros", True, "lrelu")
    assert not ok(x, w, None, 2, 1, "zeros", True, None)
    assert not ok(x, w, None, 1, 1, "reflect", True, None)
    assert not ok(x, w[:32], None, 1, 1, "zeros", True, None)  # thin outputs stay on the VALU kernels
    assert not ok(x[:, :32], w[:, :32], None, 1, 1, "zeros", True, None)
    prev = ops.get_conv_math()
    ops.set_conv_math("bf16")                                  # bf16 mode is defined on the original filters
    try:
        assert not ok(x, w, None, 1, 1, "zeros", True, None)
        ops.set_conv_math("bf16x3")                            # fp32 results: pre-summed taps are fine
        assert ok(x, w, None, 1, 1, "zeros", True, None)
    finally:
        ops.set_conv_math(prev)


def test_up_bank_cache_follows_weight_updates():
    from objgan_hip import ops
    A = ops._up_matrix("cpu")
    w = torch.randn(64, 40, 3, 3)
    b1, cached = ops._up_bank(w)
    assert cached and ops._up_bank(w)[0] is b1
    assert torch.allclose(b1, torch.einsum("pk,mckl,ql->cmpq", A, w, A))
    w.mul_(2.0)                                               # torch-side edit: _version
    b2, _ = ops._up_bank(w)
    assert b2 is b1 and torch.allclose(b2, torch.einsum("pk,mckl,ql->cmpq", A, w, A))
    # optimizer-owned weights: the arena epoch vouches for raw-pointer updates
    p = torch.nn.Parameter(torch.randn(64, 40, 3, 3))
    p._og_epoch = [0]
    b3, cached = ops._up_bank(p)
    assert cached
    with torch.no_grad():
        p.data.add_(1.0)                                      # what the fused Adam kernel does: no version bump
    p._og_epoch[0] += 1
    b4, _ = ops._up_bank(p)
    assert b4 is b3 and torch.allclose(b4, torch.einsum("pk,mckl,ql->cmpq", A, p.detach(), A))
    # trainable tensors nobody vouches for are recomposed every call
    q = torch.randn(64, 40, 3, 3, requires_grad=True)
    assert ops._up_bank(q)[1] is False


def test_inception_score_helpers_match_reference(tmp_path):
    """compute_inception_score / negative_log_posterior_probability against the reference's own
    functions (imported through the oracle harness where /root/reference exists) and a hand value."""
    import numpy as np
    from miscc import utils
    rng = np.random.RandomState(0)
    logits = rng.randn(37, 11)
    p = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    m, s = utils.compute_inception_score(p, 4)
    mc, sc = utils.negative_log_posterior_probability(p, 4)
    uniform = np.full((8, 5), 0.2)
    assert abs(utils.compute_inception_score(uniform, 2)[0] - 1.0) < 1e-12
    assert abs(utils.negative_log_posterior_probability(uniform, 2)[0] - np.log(5.0)) < 1e-12
    from oracle import ref_harness
    if ref_harness.available():
        ref = ref_harness.load_reference().utils
        assert np.allclose((m, s), ref.compute_inception_score(p, 4), rtol=1e-12, atol=0)
        assert np.allclose((mc, sc), ref.negative_log_posterior_probability(p, 4), rtol=1e-12, atol=0)
    # the trainer's epoch file (reference trainer.py:495-506)
    import types
    import torch
    import trainer
    fake = types.SimpleNamespace(batch_size=4, score_dir=str(tmp_path))
    out = trainer.condGANTrainer.write_scores(fake, [torch.from_numpy(p[:20]), torch.from_numpy(p[20:])], 3)
    txt = open(str(tmp_path / "scores_3.txt")).read().split("\n")
    assert txt[0] == "mean, std, mean_conf, std_conf " and len(txt[1].split(", ")) == 4
    assert np.allclose(out[:2], utils.compute_inception_score(p, 4))


def test_shape_generator_mirror_and_form_hmaps(monkeypatch):
    """Sampling path: SHP_G_NET has the reference's state-dict layout, and
    form_hmaps reproduces the reference output (tests/golden/shp_g_ref.pt, generated by the unmodified
    reference on CPU) when the bilinear kernel is replaced by its torch definition."""
    import os
    from conftest import ROOT
    import model as M
    import synth_batch
    from miscc import utils
    from miscc.config import cfg
    cfg.TREE.BRANCH_NUM = 3
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "shp_g_ref.pt"), weights_only=False)
    net = M.SHP_G_NET(gold["nbf"])
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == gold["state_keys"]
    monkeypatch.setattr(utils.ops, "bilinear_resize",
                        lambda x, oh, ow: F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True))
    z, fwd, bwd, fmaps, rois, num = synth_batch.make_shape_inputs(nbf=gold["nbf"])
    assert z.shape == (2, 3, 4 * gold["nbf"])
    hm, bt, fmbt = utils.form_hmaps(gold["fake_hmaps"].squeeze(2).clone(), num, rois, [64, 128, 256], gold["nbf"])

    def check(t, fp):
        t = t.double()
        assert tuple(t.shape) == tuple(fp["shape"])
        assert abs(float(t.sum()) - fp["sum"]) <= 1e-6 * max(1.0, abs(fp["sum"]))
        assert abs(float((t * t).sum()) - fp["sq"]) <= 1e-6 * max(1.0, abs(fp["sq"]))
        assert torch.allclose(t[..., ::8, ::8].float(), fp["sample"], atol=1e-6, rtol=0)
    for t, fp in zip(hm, gold["gen_hmaps"]):
        check(t, fp)
    for t, fp in zip(bt, gold["gen_bt_masks"]):
        check(t, fp)
    assert torch.allclose(fmbt, gold["gen_fm_bt_masks"], atol=1e-6, rtol=0)
    assert float(hm[0].max()) <= 1.0 + 1e-6 and float(hm[0].min()) >= 0.0


def test_load_params_invalidates_filter_bank_caches():
    """Swapping the EMA weights in (reference load_params) must change the cache keys of the packed
    filter banks: the values are written through the parameter, so its version counter moves."""
    from miscc.utils import load_params, copy_G_params
    from objgan_hip import ops
    net = torch.nn.Conv2d(40, 64, 3, bias=False)
    w = net.weight
    w._og_epoch = [0]                                     # as if owned by an optimizer arena
    stamp0 = (w._version, ops._epoch_of(w))               # what a cached bank is checked against
    bank0, _ = ops._up_bank(w)
    ref0 = bank0.clone()
    backup = copy_G_params(net)
    load_params(net, [torch.full_like(w, 0.25)])
    assert (w._version, ops._epoch_of(w)) != stamp0
    bank1, _ = ops._up_bank(w)
    assert not torch.allclose(bank1, ref0) and float(bank1.max()) == 1.0   # 4 taps of 0.25 summed
    load_params(net, backup)
    assert torch.allclose(ops._up_bank(w)[0], ref0)


def test_product_inception_wiring_matches_oracle_twin_on_the_cpu_shim(monkeypatch):
    """encoders.py (kernels only, no CPU path) with its `ops` pointed at the CPU definitions of the
    kernels: module wiring, BatchNorm folding and pooling geometry against oracle/torch_encoders.py."""
    import cpu_ops_shim
    import encoders
    from oracle import torch_encoders as te
    cpu_ops_shim.install(monkeypatch)
    enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 5)).eval()
    for p in enc.parameters():
        p.requires_grad_(False)
    x = torch.tanh(torch.randn(1, 3, 48, 48, generator=torch.Generator().manual_seed(2)))
    with torch.no_grad():
        r0, c0 = te.cpu_twin(enc)(x)
        r1, c1 = enc(x)
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    assert rel(r1, r0) < 1e-5 and rel(c1, c0) < 1e-5


def test_main_cli_mirrors_the_reference_arguments(tmp_path):
    """main.py: the reference's argparse surface (main.py:26-56) and its args -> cfg wiring (:58-88), then
    dataset / sharded loader / trainer construction on the committed six-image data directory."""
    import copy
    import os
    from conftest import ROOT
    import main as cli
    from miscc.config import cfg
    saved = copy.deepcopy(dict(cfg))
    try:
        data_dir = os.path.join(ROOT, "tests", "golden", "data_tiny")
        args = cli.parse_args(["--gpu", "0", "--FLAG", "--data_dir", data_dir, "--BATCH_SIZE", "2",
                               "--BRANCH_NUM", "3", "--OBJ_LAMBDA", "0.2", "--MAX_EPOCH", "1",
                               "--output_dir", str(tmp_path), "--manualSeed", "7"])
        cli.apply_args(args)
        assert cfg.TRAIN.FLAG and cfg.TRAIN.BATCH_SIZE == 2 and cfg.GPU_IDS == [0] and cfg.CUDA
        assert cfg.TRAIN.SMOOTH.OBJ_LAMBDA == 0.2 and cfg.TRAIN.MAX_EPOCH == 1
        assert cfg.TRAIN.NET_E == data_dir + "/pretrained/text_encoder100.pth"
        assert cfg.TEST.NET_SHP_G == data_dir + "/pretrained/shape_ckpt/shape_gen.pth"
        assert cli.seed_everything(args, rank=1) == 7
        dataset, loader, algo = cli.build_training(args, rank=0, world=1, device=torch.device("cpu"))
        assert len(dataset) == 4 and len(loader) == 2 and algo.n_words == dataset.n_words      # train split of the tiny set
        assert os.path.isdir(algo.model_dir) and algo.model_dir.startswith(str(tmp_path))
        ref_defaults = cli.parse_args([])
        assert ref_defaults.BATCH_SIZE == 24 and ref_defaults.gpu_ids == '-1' and not ref_defaults.FLAG
        assert not ref_defaults.device_imgs and not ref_defaults.device_hmaps and not dataset.device_imgs
        lean_args = cli.parse_args(["--data_dir", data_dir, "--device_imgs", "--device_hmaps", "--BATCH_SIZE", "2",
                                    "--output_dir", str(tmp_path)])
        lean_ds, lean_loader, _ = cli.build_training(lean_args, rank=0, world=1, device=torch.device("cpu"))
        assert lean_ds.device_imgs and lean_ds.device_hmaps
        assert lean_loader.collate_fn.__name__ == "collate_keep_images"
        cli.apply_args(ref_defaults)
        assert cfg.CUDA is False                      # '--gpu -1' switches CUDA off, like the reference
    finally:
        cfg.clear()
        for k, v in saved.items():
            cfg[k] = v


def test_bench_gpus_n_starts_n_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` outside a launcher must start N ranks (one per GPU, RCCL) -- the round-1
    bench silently ran one process: the re-exec goes through torch.distributed.run on 127.0.0.1
    and hands the original flags on."""
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_workloads_build_and_step_on_the_cpu_shim(monkeypatch):
    """bench.py --workload stage1 (BASELINE configs 1-2: stage-1 tree at 64x64, no object discriminators):
    the trainer bench.py builds for it steps once on the CPU definitions of the kernels and returns the
    stage-1 loss set; the default workload keeps the object discriminators."""
    import bench
    import cpu_ops_shim
    import synth_batch
    from miscc.config import cfg
    cpu_ops_shim.install(monkeypatch)
    saved = cfg.TREE.BRANCH_NUM
    try:
        assert bench.WORKLOADS["stage3_obj"][:2] == (3, True) and bench.WORKLOADS["stage3"][:2] == (3, False)
        tr = bench.build_trainer(torch.device("cpu"), 2, seed=5, with_is_monitor=False, workload="stage1")
        assert cfg.TREE.BRANCH_NUM == 1 and not tr.use_obj and len(tr.netsPatD) == 1
        out = tr.train_step(synth_batch.make_batch(2, seed=3, branch_num=1))
        assert len(out["fake_imgs"]) == 1 and tuple(out["fake_imgs"][0].shape) == (2, 3, 64, 64)
        assert "errObjSSD" not in out and torch.isfinite(out["errG"]).all()
    finally:
        cfg.TREE.BRANCH_NUM = saved


def test_gradient_buckets_leave_in_descending_order_whatever_the_completion_order():
    """ParamArena.arm / _mark / disarm: under data parallelism every rank must issue the generator's bucket
    all-reduces in the same sequence, even if a rank's batch prunes a branch of its graph and its parameters
    report in a different order (or not at all): buckets are handed over highest first, a bucket with a
    missing gradient holds the lower ones back until disarm()."""
    import trainer as T
    net = torch.nn.Sequential(*[torch.nn.Linear(8, 8) for _ in range(6)])
    arena = T.ParamArena(net)
    buckets = arena.make_buckets(k=4)
    assert len(buckets) == 4 and buckets[0][0] == 0 and buckets[-1][1] == arena.n
    assert all(a[1] == b[0] for a, b in zip(buckets, buckets[1:]))            # contiguous, whole parameters
    for order, skip in ((list(range(len(arena.params))), ()),                 # forward order: everything waits
                        (list(range(len(arena.params)))[::-1], ()),           # backward order: fires as it goes
                        ([5, 0, 11, 3, 2, 9, 1, 7, 10, 4, 8, 6], ()),
                        (list(range(len(arena.params)))[::-1], (arena.buckets[2][2][0],))):
        fired = []
        arena.arm(lambda s0, e0: fired.append((s0, e0)))
        for i in order:
            if i not in skip:
                arena._mark(i)
        during = list(fired)
        rest = arena.disarm()
        assert during + rest == [b[:2] for b in buckets[::-1]]                # the same sequence in every case
        if skip:
            assert during == [buckets[3][:2]] and rest == [b[:2] for b in buckets[2::-1]]
        elif order == list(range(len(arena.params)))[::-1]:
            assert rest == []
    arena._mark(0)                                                            # disarmed: hooks are inert


def test_bench_main_end_to_end_on_the_cpu_shim(monkeypatch, capsys):
    """bench.main() from argument parsing to the JSON line, with the kernels replaced by their CPU
    definitions and the device calls stubbed (stage-1 workload, batch 2, one timed step + the profiling
    pass): the line carries the contract's fields, the step count of the protocol is right, and the roofline
    object is derived from the library's per-kernel records (here: two synthetic launches, 100 TFLOP/s)."""
    import json
    import sys
    import bench
    import cpu_ops_shim
    import objgan_hip._lib as L
    import trainer as T
    from miscc.config import cfg
    cpu_ops_shim.install(monkeypatch)

    class FakeLib(object):
        enabled = []

        def objgan_prof_enable(self, on):
            self.enabled.append(on)
            return 1

        def objgan_prof_collect(self, ms, fl, cnt):      # two recorded launches of one kernel instance
            i = bench.cat_names("fp16x2").index("conv_igemm3_kernel<6, false, 4, 4, 1>")
            ms[i], fl[i], cnt[i] = 2.0, 2.0e11, 2
            return 1
    fake = FakeLib()
    monkeypatch.setattr(L, "load", lambda *a, **k: fake)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(bench, "_rank_device", lambda r: torch.device("cpu"))
    steps = []
    real_step = T.condGANTrainer.train_step
    monkeypatch.setattr(T.condGANTrainer, "train_step",
                        lambda self, *a, **k: (steps.append(1), real_step(self, *a, **k))[1])
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "stage1", "--batch", "2", "--steps", "1",
                                      "--warmup", "1", "--no-cpu-baseline", "--no-is-monitor"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    saved = cfg.TREE.BRANCH_NUM
    try:
        bench.main()
    finally:
        cfg.TREE.BRANCH_NUM = saved
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert len(steps) == 1 + 1 + 1 + 1 and fake.enabled == [1, 0]        # warm-up, timed, profiling pass, host-issue probe
    assert res["host_step"]["issue_ms"] > 0
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in res
    assert res["n_gpus"] == 1 and res["steps"] == 1 and res["unit"] == "images/sec" and res["value"] > 0
    assert res["metric"].endswith("at 64x64, batch 2 per GPU") and res["config"]["workload"].startswith("stage1_64x64")
    roof = res["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["kernel"] == "conv_igemm3_kernel<6, false, 4, 4, 1>"
    assert abs(roof["achieved"] - 100.0) < 1e-6 and abs(roof["frac"] - 100.0 / roof["peak"]) < 1e-3
    assert roof["traffic"] is None or roof["traffic"] > 1e8             # committed PMC record of that kernel
    assert res["conv_total"]["tflops"] == roof["achieved"] and res["kernel_breakdown"][0]["launches_per_step"] == 2.0
    assert "cpu_baseline" not in res and "side_configs" not in res and res["dtype"].startswith("fp32 (fp16x2")


def test_bench_respawn_passes_every_flag_through_and_numa_slices_are_disjoint():
    """`python bench.py --gpus 8 --math bf16 --batch 32 --d-streams 3` outside a launcher re-starts itself under
    torch.distributed.run: every flag of the original call reaches the ranks verbatim.  Rank pinning: the cores of a GPU's
    NUMA node are split evenly among the ranks whose GPUs hang off that node -- disjoint, NUMA-local slices."""
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--math", "bf16", "--batch", "32", "--d-streams", "3",
            "--workload", "stage3_obj", "--no-cpu-baseline"]
    cmd = bench.respawn_command(8, argv, 29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # two sockets of 64 cores + SMT siblings, GPUs 0-3 on node 0 and 4-7 on node 1
    nodes = {0: list(range(64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpus = [0, 0, 0, 0, 1, 1, 1, 1]
    slices = [bench.numa_slice(nodes, gpus, r) for r in range(8)]
    assert all(len(c) == 32 for c in slices)
    assert len(set().union(*map(set, slices))) == 256                       # disjoint and complete
    assert all(set(slices[r]) <= set(nodes[gpus[r]]) for r in range(8))     # NUMA-local
    assert bench.numa_slice({0: [0, 1]}, [0, 0, 0, 0], 1) is None           # fewer cores than ranks: no pinning


def test_bench_traffic_lookup_resolves_category_patterns():
    """`roofline.traffic` comes from the committed counter passes (profiles/pmc_traffic.json).  A timing category is either a
    kernel instance or -- the weight-gradient kernels, whose 16-byte and dword gather forms report into one category -- a
    pattern with `*` for one template argument: then the launch-weighted mean over the matching instances."""
    import json
    import bench
    kernels = {"conv_wgrad3_kernel<6, 4, false, 0, 8>": {"launches": 30, "hbm_bytes_per_launch": 100},
               "conv_wgrad3_kernel<6, 4, true, 0, 8>": {"launches": 10, "hbm_bytes_per_launch": 300},
               "conv_wgrad3_kernel<6, 4, false, 0, 4>": {"launches": 50, "hbm_bytes_per_launch": 7},
               "conv_igemm3_kernel<7, false, 5, 8, 1>": {"launches": 5, "hbm_bytes_per_launch": 42}}
    assert bench.pmc_lookup(kernels, "conv_igemm3_kernel<7, false, 5, 8, 1>") == 42
    assert bench.pmc_lookup(kernels, "conv_wgrad3_kernel<6, 4, *, 0, 8>") == (30 * 100 + 10 * 300) // 40
    assert bench.pmc_lookup(kernels, "conv_wgrad3_kernel<7, 4, *, 0, 8>") is None
    assert bench.pmc_lookup(kernels, "conv_thin_kernel") is None
    # the committed file answers for the categories the bench line names on the configuration it was collected on
    pmc = json.load(open(bench.PMC_TRAFFIC_JSON))
    assert pmc["conv_math"] == "fp16x2" and pmc["per_gpu_batch"] == 16
    # (round 6) the counters are keyed on the kernel sources they were collected on; bench.py withholds the figure otherwise
    assert len(pmc["csrc_sha16"]) == 16 and len(bench.csrc_hash()) == 16
    assert bench.pmc_lookup(pmc["kernels"], "conv_wgrad3_kernel<6, 4, *, 0, 8>") > 0
    assert bench.pmc_lookup(pmc["kernels"], "conv_igemm3_kernel<7, false, 5, 8, 1>") > 0


def test_graphed_callable_runs_eagerly_where_nothing_can_be_captured():
    """objgan_hip.graphs.GraphedCallable is an optimisation of the issue path: on a CPU tensor, with graphs switched off, or
    under a checked step of the fp16x2 guard (host syncs) it calls the wrapped chain directly -- same results, no capture;
    attribute access falls through to the wrapped callable."""
    from objgan_hip import graphs, ops

    class Chain(object):
        nef = 7

        def __call__(self, x):
            return (x * 2.0, x.sum(1))
    wrapped = graphs.GraphedCallable(Chain(), name="chain")
    x = torch.randn(3, 5, requires_grad=True)
    before = graphs.stats()
    a, b = wrapped(x)
    assert torch.equal(a, x * 2.0) and torch.equal(b, x.sum(1)) and wrapped.nef == 7
    prev = graphs.enabled()
    graphs.enable(False)
    try:
        assert torch.equal(wrapped(x)[0], x * 2.0)
    finally:
        graphs.enable(prev)
    ops.h2_guard_begin()
    try:
        assert torch.equal(wrapped(x)[0], x * 2.0)
    finally:
        assert ops.h2_guard_end() == []
    after = graphs.stats()
    assert after["captures"] == before["captures"] and after["replays"] == before["replays"]


def test_jpeg_index_cache_keys_on_file_identity():
    """ops.JpegIndexCache: an entry is used only for a file of the same byte count and row count (a re-written bigfile under
    the same key must not be decoded from a stale index)."""
    from objgan_hip import ops
    cache = ops.JpegIndexCache()
    idx = torch.zeros(3 * 48, dtype=torch.uint8)
    cache.put("a", 1000, idx)
    assert cache.get("a", 1000, 3, 48) is idx and cache.hits == 1
    assert cache.get("a", 1001, 3, 48) is None and cache.get("a", 1000, 4, 48) is None and cache.get("b", 1000, 3, 48) is None
    assert cache.misses == 3
