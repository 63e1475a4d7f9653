"""Data-parallel path on CPU: world_size-2 gloo processes exercise the gradient exchange that the
trainer runs over RCCL on the GPUs (flat gradient arenas, one all-reduce per network, averaging
folded into the optimizer step, cross-rank agreement on the conditional object-discriminator
update).  The arenas and the exchange logic are plain torch / torch.distributed and are the very
classes the GPU trainer uses; only the fused Adam kernel is replaced by its CPU oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(6, 5), nn.Tanh(), nn.Linear(5, 3))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd")]
    import trainer as T
    from oracle import torch_ref

    # adam on the CPU: the oracle formula instead of the HIP kernel (same call signature)
    def cpu_adam(p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, n=None):
        n = p.numel() if n is None else n
        pn, mn, vn = torch_ref.adam_step(p[:n], g[:n] * grad_scale, m[:n], v[:n], lr, b1, b2, eps, step)
        p[:n].copy_(pn); m[:n].copy_(mn); v[:n].copy_(vn)
    T.ops.adam_step_ = cpu_adam
    sys.path.insert(0, os.path.join(root, "tests"))
    import cpu_ops_shim
    cpu_ops_shim.adam_step_ = cpu_adam
    T.ops.adam_step_gated_ = cpu_ops_shim.adam_step_gated_

    net = _make_net(0)                                  # identical replicas
    arena = T.ParamArena(net)
    opt = T.ArenaAdam(arena, lr=1e-2)
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(8, 6, generator=g)
    y_all = torch.randn(8, 3, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)               # per-rank minibatch shard

    opt.zero_grad()
    loss = ((net(x_all[shard]) - y_all[shard]) ** 2).mean()
    loss.backward()
    arena.sync_grads()
    arena.grad[-1] = 1.0 if rank == 0 else 0.0          # only rank 0 "has boxes of this scale"
    h = dist.all_reduce(arena.grad, op=dist.ReduceOp.SUM, async_op=True)
    h.wait()
    grad_avg = arena.grad[:arena.n].clone() / world
    # the gate is the flag slot itself, read by the optimizer kernel: no host-side decision
    opt.step(grad_scale=1.0 / world, gated=True)
    active = opt.steps_taken == 1
    q.put((rank, active, grad_avg, arena.flat.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_single_process_average():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-process reference: mean of the two shard gradients == gradient of the mean shard loss
    net = _make_net(0)
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(8, 6, generator=g)
    y_all = torch.randn(8, 3, generator=g)
    grads = []
    for r in range(world):
        net.zero_grad()
        (((net(x_all[r * 4:r * 4 + 4]) - y_all[r * 4:r * 4 + 4]) ** 2).mean()).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in net.parameters()]))
    want = sum(grads) / world
    for rank, active, grad_avg, flat in res:
        assert active, "every rank must take the update when any rank is active"
        assert torch.allclose(grad_avg, want, atol=1e-6)
    assert torch.equal(res[0][3], res[1][3]), "replicas diverged"
    from oracle import torch_ref
    p0 = torch.cat([p.detach().reshape(-1) for p in _make_net(0).parameters()])
    z = torch.zeros_like(p0)
    p1, _, _ = torch_ref.adam_step(p0, want, z, z, 1e-2, 0.5, 0.999, 1e-8, 1)
    assert torch.allclose(res[0][3], p1, atol=1e-6)


def test_param_arena_keeps_views_and_survives_zero_grad_none():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd")]
    import trainer as T
    net = _make_net(1)
    before = torch.cat([p.detach().reshape(-1).clone() for p in net.parameters()])
    arena = T.ParamArena(net)
    assert torch.equal(arena.flat, before)
    net(torch.randn(2, 6)).sum().backward()
    g1 = arena.grad[:arena.n].clone()
    assert float(g1.abs().sum()) > 0                     # autograd accumulated INTO the arena
    net.zero_grad(set_to_none=True)                      # what the reference trainer would call
    net(torch.randn(2, 6)).sum().backward()              # grads now live outside the arena ...
    arena.sync_grads()                                   # ... and are pulled back in
    for p, gv in zip(arena.params, arena._views):
        assert p.grad.data_ptr() == gv.data_ptr()
    arena.flat.mul_(2.0)                                 # parameters are views of the arena
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), before * 2)


def _trainer_worker(rank, world, port, q):
    """One rank of the product trainer (stage-1 + stage-2 tree, both object discriminators) on the CPU
    shim of the kernels; rank 1's boxes are all small, so its large-scale object discriminator sees no
    box and contributes zeros -- it must still take the same update as rank 0.  (World size 4: ranks 2 and 3
    hold only LARGE boxes -- no gradient for the small-scale object discriminator from them.  A minibatch without any
    box at all is not a case: the reference's generator raises on it, model.py:553-572, and so does the product.)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd"), os.path.join(root, "tests")]
    import random
    import cpu_ops_shim
    import model as M
    import synth_batch
    import trainer as T
    from miscc.config import cfg
    from oracle import ref_harness as rh
    cpu_ops_shim.install_plain()
    torch.set_num_threads(4 if world <= 2 else (2 if world <= 4 else 1))
    cfg.TREE.BRANCH_NUM = 2
    cfg.TRAIN.BATCH_SIZE = 2
    cfg.TRAIN.NET_G = ''
    B = 2

    class Enc(object):
        def __init__(self):
            g0 = torch.Generator().manual_seed(5)
            self.r, self.c = torch.randn(B, 256, 17, 17, generator=g0), torch.randn(B, 256, generator=g0)

        def __call__(self, x):
            return self.r, self.c

    class DS(object):
        num_classes = 80
    ds = DS()
    ds.image_encoder = Enc()
    tr = T.condGANTrainer('', None, ds, device=torch.device("cpu"))
    tr.batch_size = B
    # rank-dependent initial weights: setup() must broadcast rank 0's
    nets = [None, ds.image_encoder, rh.seeded_state_(M.G_NET(80), 200 + rank),
            [rh.seeded_state_(c(), 210 + rank + i) for i, c in enumerate((M.PAT_D_NET64, M.PAT_D_NET128))],
            [rh.seeded_state_(c(80), 220 + rank + i) for i, c in enumerate((M.SHP_D_NET64, M.SHP_D_NET128))],
            rh.seeded_state_(M.OBJ_SS_D_NET(80), 230 + rank), rh.seeded_state_(M.OBJ_LS_D_NET(80), 240 + rank), 0]
    for m in [nets[2], nets[5], nets[6]] + nets[3] + nets[4]:
        m.train()
    orig_build = T.condGANTrainer.build_models

    def build(self):
        for net in [nets[2], nets[5], nets[6]] + nets[3] + nets[4]:
            for t in list(net.parameters()) + list(net.buffers()):
                dist.broadcast(t.data, src=0)
        return nets
    tr.build_models = lambda: build(tr)
    tr.setup()
    init_ls = tr.optimizerObjLSD.arena.flat.clone()
    b = synth_batch.make_batch(B, seed=300 + rank, branch_num=2)
    # box populations by rank % 4: 0 both scales, 1 only small boxes, 2 and 3 only large ones (world size 8: ranks 4-7 repeat)
    if rank % 4 == 1:                               # no box reaches the large-scale threshold on this rank
        for r in b["rois"]:
            r[:, :, 2:4] = r[:, :, 2:4].clamp(max=6.0 * r[:, :, 2:4].max() / 64.0)
        b["rois"][0][:, :, 2:4] = b["rois"][0][:, :, 2:4].clamp(max=6.0)
        b["rois"][1][:, :, 2:4] = b["rois"][1][:, :, 2:4].clamp(max=12.0)
        b["fm_rois"][:, :, 2:4] = b["fm_rois"][:, :, 2:4].clamp(max=3.0)
    if rank % 4 >= 2:                               # every box of this rank is large (>= 16 of the 32-px feature map)
        b["fm_rois"][:, :, 2:4] = b["fm_rois"][:, :, 2:4].clamp(min=17.0)
        b["rois"][0][:, :, 2:4] = b["rois"][0][:, :, 2:4].clamp(min=34.0)
        b["rois"][1][:, :, 2:4] = b["rois"][1][:, :, 2:4].clamp(min=68.0)
    tr.netG.ca_net.fixed_eps = b["ca_eps"]
    random.seed(7 + rank)
    out = tr.train_step(b, noise=b["noise"])
    if world > 2:
        # the flag slots behind the gradient arenas after the exchange = number of ranks that contributed a gradient
        q.put((rank, "errObjLSD" in out, "errObjSSD" in out,
               [o.arena.flat.numpy().copy() for o in [tr.optimizerG] + tr._d_optimizers()], tr.avg_param_G.numpy().copy(),
               float(tr.optimizerObjLSD.arena.grad[-1]), float(tr.optimizerObjSSD.arena.grad[-1]),
               tr.optimizerObjLSD.steps_taken, tr.optimizerObjSSD.steps_taken, init_ls.numpy().copy(), tuple(tr.g_buckets)))
        dist.barrier()
        dist.destroy_process_group()
        return
    # numpy copies: a tensor put on the queue travels through shared memory owned by this process
    assert tr.g_buckets[0] >= 3 and tr.g_buckets[1] <= 1, tr.g_buckets   # generator all-reduce overlapped its backward
    q.put((rank, "errObjLSD" in out, tr.optimizerG.arena.flat.numpy().copy(),
           tr.optimizersPatD[1].arena.flat.numpy().copy(), tr.optimizerObjLSD.arena.flat.numpy().copy(),
           init_ls.numpy().copy(), tr.avg_param_G.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_step_keeps_replicas_identical():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, ls0, g0, d0, objls0, init0, ema0), (r1, ls1, g1, d1, objls1, init1, ema1) = res
    assert ls0 and not ls1, "rank 0 has large boxes, rank 1 must have none for this test to bite"
    import numpy as np
    assert np.array_equal(init0, init1), "setup() did not broadcast rank 0's weights"
    assert np.array_equal(g0, g1) and np.array_equal(d0, d1) and np.array_equal(ema0, ema1), "replicas diverged"
    assert np.array_equal(objls0, objls1), "conditional object-discriminator update diverged across ranks"
    assert not np.array_equal(objls0, init0), "the large-scale object discriminator was not updated at all"


def test_four_rank_trainer_step_with_a_rank_without_large_boxes_and_a_rank_without_boxes():
    """World size 4: rank 0 has boxes of both scales, rank 1 only small boxes, ranks 2 and 3 only
    large ones.  Every rank must issue the same collectives in the same order (a mismatch hangs: the queue times out),
    all replicas must end bit-identical, both object discriminators must be updated on every rank, and the gated Adam
    must have divided by the number of CONTRIBUTING ranks: 3 for the large-scale discriminator, 2 for the small-scale.
    (No rank without ANY box: the reference's generator raises on such a minibatch and the product keeps that contract.)"""
    import numpy as np
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=600))
    finally:
        if len(res) < world:                       # a rank died: the others wait in a collective for ever
            for p in procs:
                p.terminate()
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    has_ls = [r[1] for r in res]
    has_ss = [r[2] for r in res]
    assert has_ls == [True, False, True, True] and has_ss == [True, True, False, False], (has_ls, has_ss)
    for r in res[1:]:
        for a0, a1 in zip(res[0][3], r[3]):
            assert np.array_equal(a0, a1), "replicas diverged (rank %d)" % r[0]
        assert np.array_equal(res[0][4], r[4])
    for r in res:
        assert r[5] == 3.0 and r[6] == 2.0, ("contributing-rank counts", r[5], r[6])
        assert r[7] == 1 and r[8] == 1, "a gated object-discriminator update was skipped"
        assert r[10][0] >= 3, r[10]           # generator buckets reduced inside the backward on every rank
    assert not np.array_equal(res[0][3][-1], res[0][9]), "the large-scale object discriminator was not updated"


@pytest.mark.timeout(1500)
def test_eight_rank_trainer_step_with_uneven_box_populations():
    """World size 8, the driver's SCALE configuration: ranks 0 and 4 hold boxes of both scales, ranks 1
    and 5 only small ones, ranks 2, 3, 6, 7 only large ones.  Every rank issues the same collectives in the same order
    (a mismatch hangs into the queue timeout), all eight replicas end bit-identical, and the gated Adam divided by the
    number of contributing ranks: 6 for the large-scale object discriminator, 4 for the small-scale one."""
    import numpy as np
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=1200))
    finally:
        if len(res) < world:
            for p in procs:
                p.terminate()
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, False, True, True] * 2 and [r[2] for r in res] == [True, True, False, False] * 2
    for r in res[1:]:
        for a0, a1 in zip(res[0][3], r[3]):
            assert np.array_equal(a0, a1), "replicas diverged (rank %d)" % r[0]
        assert np.array_equal(res[0][4], r[4])
    for r in res:
        assert r[5] == 6.0 and r[6] == 4.0, ("contributing-rank counts", r[5], r[6])
        assert r[7] == 1 and r[8] == 1, "a gated object-discriminator update was skipped"
        assert r[10][0] >= 3, r[10]


def _bench_protocol_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd")]
    import time
    import bench
    calls = {"step": 0, "begin": 0, "end": 0}
    acc = torch.zeros(1)

    def step():                                   # a "training step": one collective, like the gradient exchange
        calls["step"] += 1
        t = torch.ones(1)
        dist.all_reduce(t)
        acc.add_(t)
        if rank == 1:
            time.sleep(0.01)                      # the slow rank sets the time

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def hook(name):
        return (lambda: calls.__setitem__(name, calls[name] + 1)) if rank == 0 else None
    dt, prof_dt = bench.timed_passes(step, dist.barrier, max_over_ranks, 5, 2, 3, hook("begin"), hook("end"))
    dt0, none = bench.timed_passes(step, dist.barrier, max_over_ranks, 2, 0, 0)
    q.put((rank, calls, float(acc), dt, prof_dt, none))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bench_measurement_protocol_runs_every_pass_on_every_rank():
    """bench.timed_passes on two gloo ranks with a step that contains a collective: warm-up, the timed
    steps and the untimed profiling pass run on BOTH ranks (a pass on rank 0 alone leaves its collectives
    unmatched -- the test would hang into its timeout), only rank 0 gets the profiling hooks, and the reported
    time is the maximum over the ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_protocol_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, c0, acc0, dt0, prof0, n0), (r1, c1, acc1, dt1, prof1, n1) = res
    assert c0["step"] == c1["step"] == 2 + 5 + 3 + 2 and acc0 == acc1 == 2.0 * 12
    assert (c0["begin"], c0["end"]) == (1, 1) and (c1["begin"], c1["end"]) == (0, 0)
    assert dt0 == dt1 and dt0 >= 5 * 0.01 and prof0 is not None and prof1 is not None and n0 is None and n1 is None


def _bench_main_worker(rank, world, port, q):
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd"), os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(4)
    import bench
    import cpu_ops_shim
    import objgan_hip._lib as L
    cpu_ops_shim.install_plain()

    class FakeLib(object):
        def objgan_prof_enable(self, on):
            return 1

        def objgan_prof_collect(self, ms, fl, cnt):
            return 1
    L.load = lambda *a, **k: FakeLib()
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    bench._rank_device = lambda r: torch.device("cpu")
    real_init = dist.init_process_group
    dist.init_process_group = lambda backend=None, rank=None, world_size=None, device_id=None, **k: \
        real_init("gloo", rank=rank, world_size=world_size)
    sys.argv = ["bench.py", "--gpus", str(world), "--workload", "stage1", "--batch", "2", "--steps", "1",
                "--warmup", "1", "--no-is-monitor"]
    out, real_stdout = io.StringIO(), sys.stdout
    sys.stdout = out
    try:
        bench.main()
    finally:
        sys.stdout = real_stdout
    q.put((rank, [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]))


@pytest.mark.timeout(300)
def test_bench_main_on_two_ranks_prints_one_whole_job_line():
    """`bench.py --gpus 2` as the driver launches it (RANK / WORLD_SIZE in the environment), on two gloo
    ranks with the kernels' CPU definitions: both ranks walk through warm-up, timed steps, the profiling pass
    and the teardown (every collective matched -- otherwise this test hangs into its timeout), rank 0 alone
    prints the line, and the line counts the images of BOTH ranks."""
    import json
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_main_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert len(res[0]) == 1 and res[1] == []
    line = json.loads(res[0][0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    assert line["scaling"] == "weak" and "cpu_baseline" not in line
    assert abs(line["value"] - 4 * 1000.0 / line["ms_per_step"]) < 0.05 * line["value"]
    # the `comm` object of a data-parallel run: what was exchanged per step (stage 1: PatD64 + ShpD64 + G arenas, each
    # with its flag slot, the generator in buckets), measured in the profiling pass on every rank
    comm = line["comm"]
    assert comm["world_size"] == 2 and comm["backend"] == "gloo"
    nb = comm["arena_bytes"]
    assert set(nb) == {"PatD0", "ShpD0", "G"} and set(comm["standalone_allreduce_ms"]) == set(nb)
    assert comm["allreduce_bytes_per_step"] == sum(nb.values()) - 4        # the G buckets leave the flag slot out
    assert comm["collectives_per_step"] >= 3 and comm["exposed_wait_ms_per_step"] is None    # (no device events on CPU)


def _seed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    import random
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd")]
    import main as cli
    from miscc.config import cfg
    from torch.utils.data.distributed import DistributedSampler
    cfg.TRAIN.FLAG = True
    random.seed(1000 + 17 * rank)                        # every process would draw its own seed
    args = argparse.Namespace(manualSeed=None)
    seed = cli.seed_everything(args, rank, world, None)
    order = list(DistributedSampler(range(64), num_replicas=world, rank=rank, shuffle=True, seed=seed))
    q.put((rank, seed, order, random.random()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_without_manual_seed_agree_on_one_seed_and_read_disjoint_shards():
    """`main.py` without --manualSeed: the seed is drawn on rank 0 and broadcast, so the DistributedSampler of every
    rank shuffles the SAME permutation and the shards are disjoint and complete; the python / numpy generators still
    differ per rank (seed + rank: caption sampling, permute_seg)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, o0, r0), (_, s1, o1, r1) = res
    assert s0 == s1
    assert not (set(o0) & set(o1)) and sorted(o0 + o1) == list(range(64))
    assert r0 != r1


def _dying_rank_worker(rank, world, port):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "obj-gan_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OBJGAN_DIST_TIMEOUT_S="20")
    import bench
    bench.init_dist("gloo", rank, world)
    dist.barrier()
    if rank == 1:
        os._exit(3)                               # this rank dies between two steps
    t = torch.ones(1)
    dist.all_reduce(t)                            # ... and its peer must not wait for ever
    dist.barrier()
    sys.exit(0)


@pytest.mark.timeout(180)
def test_a_dying_rank_fails_its_peer_instead_of_hanging_it():
    """`bench.py --gpus N` must end with a non-zero exit code and no hang when a rank dies.  The process
    group is created with a finite timeout (bench.init_dist); the surviving rank's next collective raises, the rank exits
    non-zero (torch.distributed.run then terminates the job with that status)."""
    import time
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_dying_rank_worker, args=(r, world, port)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert all(not p.is_alive() for p in procs), "a rank is still waiting for its dead peer"
    assert procs[1].exitcode == 3 and procs[0].exitcode not in (0, None), [p.exitcode for p in procs]
    assert time.time() - t0 < 120
