#!/usr/bin/env python
"""Benchmark of the Obj-GAN image_generation G+D training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full iteration of the reference's training loop (reference
image_generation/trainer.py:357-472) on one synthetic COCO-shaped minibatch: three-stage G_NET
forward at 256x256 (fed by the frozen caption encoder and GloVe lookup), the three patch and three
shape discriminators, the two ROIAlign object discriminators, generator loss with DAMSM + KL, nine
Adam updates and the generator EMA (and the per-step Inception-score monitor on a side stream, as in
the reference).  Per-GPU batch 16; under
`--gpus N` (launched by torch.distributed.run, one rank per GPU, RCCL; without a launcher in the
environment bench.py starts the N ranks itself) every rank steps its own batches and gradients are
all-reduced: weak scaling, value = N * 16 * steps / time.  Every step gets a new minibatch.

One JSON line is printed by rank 0: the contract fields plus
  roofline      the dominant kernel instance (the MFMA implicit-GEMM conv with the largest share
                of the step): algorithmic flops of its launches / their hipEvent-measured duration
                (events recorded by the library on the launch stream, in a second untimed pass so
                that the headline carries no event overhead), against the fp32 MFMA peak;
                traffic = HBM bytes per launch from the rocprofv3 --pmc passes kept in profiles/
  cpu_baseline  the CPU oracle (oracle/torch_model.py, a port of the reference path) timed on this host's
                physical cores at the BASELINE batch: 1 warm-up + `--cpu-baseline-steps` (default 3) timed steps at
                batch 16 (~90 s each; child process with a hard time limit, batch-8 fallback).  The unmodified
                reference cannot run on the GPU box (no /root/reference there): `reference_vs_port` carries the
                reference-vs-port timing measured in the build container (tools/cpu_ref_vs_port.py)
  side_configs  (default N = 1 run only) two short side records measured after the headline, not the metric: the same
                step on the true fp32 MFMA, and BASELINE config 5 (bf16 mixed precision, batch 32)
  comm          (data-parallel runs) what the gradient exchange moved and what it cost: bytes all-reduced per step,
                collectives per step, time the compute stream stood still waiting for them, stand-alone all-reduce
                times of the arena sizes, RCCL version
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "obj-gan_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3                   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0                  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
# timing categories of the library = kernel instances, named as rocprofv3 prints them
MATH_IDS = {"fp32": 0, "bf16": 1, "bf16x3": 2, "fp16x2": 4}
# bf16x3: six bf16 MFMAs (K = 16) do the work of eight fp32 ones (K = 2) at 16x their rate; fp16x2: three fp16 MFMAs
BF16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
FP16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0


def cat_names(math):
    """rocprofv3's names of the kernel instances behind the library's timing categories (leading template
    arguments: tile height, LDS-free form / arithmetic), indexed by category id."""
    def names(mi, m, bfb=False, ng=1, rec=False):
        # (rec: the weight gradient that reads x from its fp16 record reports in the record categories' slots)
        wg4 = ["conv_wgrad_rec_kernel<%d, 4>" % tm if rec else "conv_wgrad3_kernel<%d, %d" % (tm, m) for tm in range(1, 8)]
        wg8 = ["conv_wgrad_rec_kernel<%d, 8>" % tm if rec else "conv_wgrad3_kernel<%d, %d, *, 0, 8>" % (tm, m) for tm in range(1, 8)]
        return (["conv_igemm3_kernel<%d, false, %d, 4, %d>" % (tm, mi, ng) for tm in range(1, 8)] +
                # (bf16 mode: the bf16-operand weight-gradient kernel reports in the slots of the LDS-staged one)
                [("conv_wgrad_bfb_kernel<%d," % tm) if bfb else ("conv_wgrad2_kernel<%d, %d" % (tm, min(m, 2))) for tm in range(1, 8)] +
                ["conv_thin_kernel", "conv_thin3x3_kernel", "conv_igemm_kernel", "conv_wgrad_kernel",
                 "conv_igemm3_kernel<1, true, %d, 4, 1>" % m] + wg4 +
                ["conv_igemm3_kernel<%d, false, %d, 8, %d>" % (tm, mi, ng) for tm in range(1, 8)] + wg8)
    if math == "fp16x2":       # categories 0..47: this mode's bf16x3 launches (small ones, LDS-staged weight gradients);
        base = names(2, 2)               # 48..95: the fp16x2 instances of the same kernel families; 96..143 / 144..191:
        pad = [""] * (48 - len(base))    # the forward / data-gradient kernel on pre-split records, one / two pixel groups
        return base + pad + names(4, 4) + pad + names(5, 5, rec=True) + pad + names(5, 5, ng=2, rec=True)
    m = MATH_IDS[math]
    # bf16 mode: the matrix kernels read the bf16 channel-blocked copy (template value 3)
    return names(3 if m == 1 else m, m, bfb=(m == 1))


PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def csrc_hash():
    """sha256 (16 hex digits) over the convolution kernel sources (obj-gan_amd/csrc/conv_igemm*, common.h): the key of the committed counter passes (tools/pmc_traffic.py)"""
    import hashlib
    root = os.path.join(ROOT, "obj-gan_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.startswith("conv_igemm") or f == "common.h":           # the sources of the measured (convolution) kernels
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_lookup(kernels, name):
    """HBM bytes per launch of a timing category from the committed counter passes: the category name is the kernel
    instance's, or (weight-gradient kernels: the 16-byte and the dword gather form report into one category) a pattern
    with `*` for one template argument -- then the launch-weighted mean over the matching instances."""
    ent = kernels.get(name)
    if ent is not None:
        return ent["hbm_bytes_per_launch"]
    if "*" not in name:
        return None
    import re
    pat = re.compile("^" + re.escape(name).replace("\\*", "[^,>]*") + ("" if name.rstrip().endswith(">") else ".*") + "$")
    hits = [v for k, v in kernels.items() if pat.match(k)]
    n = sum(v.get("launches", 0) for v in hits)
    if not hits or n <= 0:
        return None
    return int(sum(v["hbm_bytes_per_launch"] * v.get("launches", 0) for v in hits) / n)


WORKLOADS = {
    # BASELINE.json configs[3] (the configuration `metric` is quoted on, per GPU) -- the default
    "stage3_obj": (3, True, "stage3_256x256_full_GD_step: RNN_ENCODER+G_NET(3 stages)+PatD x3+ShpD x3+"
                            "ObjSSD+ObjLSD(ROIAlign)+DAMSM+KL+Adam x9+EMA"),
    # configs[2]: the three-stage tree without the two ROIAlign object discriminators
    "stage3": (3, False, "stage3_256x256_GD_step_without_object_discriminators: RNN_ENCODER+G_NET(3 stages)+"
                         "PatD x3+ShpD x3+DAMSM+KL+Adam x7+EMA"),
    # configs[1]: the stage-1 tree at 64x64
    "stage1": (1, False, "stage1_64x64_GD_step: RNN_ENCODER+G_NET(stage 1)+PatD64+ShpD64+DAMSM+KL+Adam x3+EMA"),
}


def build_trainer(device, batch_size, seed, with_is_monitor=True, workload="stage3_obj"):
    import encoders
    import trainer as T
    from miscc.config import cfg
    cfg.TREE.BRANCH_NUM = WORKLOADS[workload][0]
    cfg.TRAIN.BATCH_SIZE = batch_size
    cfg.TRAIN.NET_G = ''
    torch.manual_seed(seed)                     # same initial weights on every rank

    class SynthDataset(object):
        num_classes = 80
    ds = SynthDataset()
    import model as M
    # frozen text front-end of the step (reference trainer.py:91-100, 63-73): caption encoder + GloVe table
    ds.text_encoder = encoders.seeded_init_(M.RNN_ENCODER(1000, nhidden=cfg.TEXT.EMBEDDING_DIM), 2).to(device).eval()
    ds.glove_embed = torch.nn.Embedding(401, cfg.TEXT.GLOVE_EMBEDDING_DIM).to(device).eval()
    for p in list(ds.text_encoder.parameters()) + list(ds.glove_embed.parameters()):
        p.requires_grad_(False)
    trunk = encoders.seeded_init_(encoders.inception_v3(), 1)
    ds.image_encoder = encoders.CNN_ENCODER(256, trunk).to(device).eval()
    for p in ds.image_encoder.parameters():
        p.requires_grad_(False)
    if with_is_monitor:
        ds.inception_model = encoders.INCEPTION_V3(trunk).to(device).eval()
    tr = T.condGANTrainer('', None, ds, device=device)
    tr.batch_size = batch_size
    if not WORKLOADS[workload][1]:
        tr.use_obj = False
    tr.setup()
    return tr


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_baseline_worker(sample_batch, threads, timed_steps, seed=1234):
    """(child process) the oracle -- CPU port of the reference step (reference trainer.py:388-462 order) --
    on `threads` host threads: one warm-up step, then `timed_steps` timed ones."""
    import model as M
    import encoders
    import synth_batch
    from miscc.config import cfg
    from miscc.utils import weights_init
    from oracle import torch_model as tm
    from oracle import torch_encoders
    cfg.TREE.BRANCH_NUM = 3
    torch.set_num_threads(threads)
    torch.manual_seed(seed)

    def sd_of(m):
        m.apply(weights_init)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for k, v in sd.items():
            if v.dtype.is_floating_point and "running_" not in k:
                v.requires_grad_(True)
        return sd
    sds = {"G": sd_of(M.G_NET(80)),
           "pat": [sd_of(c()) for c in (M.PAT_D_NET64, M.PAT_D_NET128, M.PAT_D_NET256)],
           "shp": [sd_of(c(80)) for c in (M.SHP_D_NET64, M.SHP_D_NET128, M.SHP_D_NET256)],
           "objss": sd_of(M.OBJ_SS_D_NET(80)), "objls": sd_of(M.OBJ_LS_D_NET(80))}
    adam = lambda sd: torch.optim.Adam(tm.params_of(sd), lr=2e-4, betas=(0.5, 0.999))   # noqa: E731
    opts = {"G": adam(sds["G"]), "pat": [adam(s) for s in sds["pat"]], "shp": [adam(s) for s in sds["shp"]],
            "objss": adam(sds["objss"]), "objls": adam(sds["objls"])}
    ema = [p.detach().clone() for p in tm.params_of(sds["G"])]
    enc = torch_encoders.CpuImageEncoder(encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 1)).eval())
    batch = synth_batch.make_batch(sample_batch, seed=seed)

    def report(dt, note):
        print("CPU_BASELINE " + json.dumps(
            {"value": round(sample_batch / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
             "cpu": _cpu_model(), "host_cores": os.cpu_count() or 1,
             "sample": "full G+D step at batch %d (same networks, 256x256, fp32) on %d threads: %s"
                       % (sample_batch, threads, note)}), flush=True)
    t0 = time.time()
    tm.train_step(sds, opts, ema, batch, image_encoder=enc)
    warm = time.time() - t0
    report(warm, "first (un-warmed) step %.1f s" % warm)      # provisional: replaced by the timed steps below
    if timed_steps <= 0:
        return
    t0 = time.time()
    for _ in range(timed_steps):
        tm.train_step(sds, opts, ema, batch, image_encoder=enc)
    dt = (time.time() - t0) / max(1, timed_steps)
    report(dt, "1 warm-up (%.1f s) + %d timed step(s) of %.1f s each" % (warm, timed_steps, dt))


REF_VS_PORT_JSON = os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json")


def cpu_baseline(sample_batch=16, timed_steps=3, timeout_s=630):
    """The oracle timed on the host's physical cores at the BASELINE batch (1 warm-up + timed
    steps), in a child process with a hard time limit so that the default bench run always finishes; if
    the full-batch run does not fit the limit a batch-8 sample is reported instead.  One thread per
    PHYSICAL core: with all 256 hardware threads of the GPU box's two EPYC 9575F the same step did not
    finish in 420 s (fork-join cost of the many small operators), with 128 it takes about 90 s.  Default
    sample: one warm-up step + THREE timed steps at batch 16 (the first step runs ~5 % slower than a warmed one;
    should the timed steps not fit the limit, the un-warmed one is reported and says so)."""
    import subprocess
    threads = max(1, min(128, (os.cpu_count() or 2) // 2))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    note = ""
    for batch, limit in ((sample_batch, timeout_s), (8, timeout_s // 3)):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(batch), str(threads),
               str(timed_steps)]
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        try:
            out = proc.communicate(timeout=limit)[0]
        except subprocess.TimeoutExpired:
            proc.kill()
            out = proc.communicate()[0]
            note += "batch %d: stopped at the %d s limit; " % (batch, limit)
        last = None
        for line in out.decode(errors="replace").splitlines():
            if line.startswith("CPU_BASELINE "):
                last = json.loads(line[len("CPU_BASELINE "):])
        if last is not None:          # the timed result when it got that far, else the un-warmed first step
            if note:
                last["sample"] += " (" + note.strip() + ")"
            try:                      # reference-vs-port timing from the build container (the reference is not here)
                last["reference_vs_port"] = json.load(open(REF_VS_PORT_JSON))
            except (OSError, ValueError):
                last["reference_vs_port"] = None
            return last
        note += "batch %d: no step finished; " % batch
    return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "cpu": _cpu_model(),
            "sample": note}


def side_configs(steps=6, warmup=2, timeout_s=240):
    """Five short side records of the default N = 1 run, each a child `bench.py` on the same GPU after the headline is
    measured (they are NOT the metric): the same step on the true fp32 MFMA (`--math fp32`, the arithmetic of rounds
    1-2), on the bf16 MFMA with exactly split operands (`--math bf16x3`, round 3's), BASELINE config 5
    (`--math bf16 --batch 32`: bf16 matrix-core inputs, fp32 accumulation), and the default step with every minibatch
    crossing PCIe inside the timed region (`--host-batches`: the reference's host hand-over, prefetched)."""
    import subprocess
    out = {}
    for key, extra in (("fp32_mfma_b16", ["--math", "fp32"]), ("bf16x3_b16", ["--math", "bf16x3"]),
                       ("config5_bf16_b32", ["--math", "bf16", "--batch", "32"]),
                       ("pcie_inclusive_b16", ["--host-batches", "--no-kernel-timing"]),
                       ("jpeg_input_b16", ["--jpeg-input", "--no-kernel-timing"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-side-configs"] + extra
        try:
            txt = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s).stdout.decode()
            line = [ln for ln in txt.splitlines() if ln.startswith("{")][-1]
            r = json.loads(line)
            out[key] = {k: r.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype")}
            out[key]["per_gpu_batch"] = r["config"]["per_gpu_batch"]
            if "host_batches" in r["config"]:
                out[key]["host_batches"] = r["config"]["host_batches"]
            if "jpeg_input" in r["config"]:
                out[key]["jpeg_input"] = r["config"]["jpeg_input"]
            if "roofline" in r:
                out[key]["roofline"] = {k: r["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic")}
            if "conv_total" in r:
                out[key]["conv_total"] = r["conv_total"]
        except Exception as e:                                   # noqa: BLE001 (a side record never fails the headline)
            out[key] = {"error": repr(e)[:200]}
    return out


def respawn_command(n, argv, port):
    """the launcher command of `python bench.py --gpus N ...` outside a launcher: N ranks of THIS script with EVERY flag
    of the original call (--math / --batch / --d-streams / --workload ... travel to the ranks verbatim)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _respawn_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks, one per GPU, over RCCL.  A rank that dies
    takes the job down: torch.distributed.run terminates the other ranks and returns non-zero, which is this process's
    exit code (no line is printed); a rank that stops answering trips the process-group timeout (init_dist)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.call(respawn_command(n, sys.argv[1:], port), env=env)


def init_dist(backend, rank, world, device=None):
    """The process group of a data-parallel run, with a finite timeout (OBJGAN_DIST_TIMEOUT_S, default 600 s): a peer
    that died or hangs makes the next collective / barrier RAISE instead of waiting for ever -- the rank exits non-zero
    and the launcher reports the job as failed."""
    import datetime
    kw = {"timeout": datetime.timedelta(seconds=float(os.environ.get("OBJGAN_DIST_TIMEOUT_S", "600")))}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)


def parse_cpulist(text):
    """'0-15,128-143' -> [0..15, 128..143]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_slice(node_cpus, gpu_nodes, local_rank):
    """CPUs for the rank that drives GPU `local_rank`: the cores of the GPU's NUMA node (node_cpus: {node: [cpu ids]},
    gpu_nodes: [node of GPU 0, 1, ...]), split evenly among the ranks whose GPUs hang off the same node.  Eight ranks each
    issue ~4 900 launches per step from five streams; pinned to NUMA-local cores they neither migrate across sockets nor
    compete for the same cores."""
    node = gpu_nodes[local_rank]
    if node not in node_cpus or not node_cpus[node]:
        return None
    peers = [r for r, nd in enumerate(gpu_nodes) if nd == node]
    cpus = sorted(node_cpus[node])
    k, i = len(peers), peers.index(local_rank)
    per = len(cpus) // k
    if per < 1:
        return None
    return cpus[i * per:(i + 1) * per]


def pin_rank_to_numa(local_rank, world):
    """-> the CPU list this rank was pinned to (None: single rank, or the topology could not be read)"""
    if world <= 1 or os.environ.get("OBJGAN_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        gpu_nodes = []
        for i in range(torch.cuda.device_count()):
            bdf = torch.cuda.get_device_properties(i).pci_bus_id if hasattr(
                torch.cuda.get_device_properties(i), "pci_bus_id") else None
            node = -1
            if bdf is not None:
                dom = getattr(torch.cuda.get_device_properties(i), "pci_domain_id", 0)
                dev = getattr(torch.cuda.get_device_properties(i), "pci_device_id", 0)
                path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (dom, bdf, dev)
                if os.path.exists(path):
                    node = int(open(path).read().strip())
            gpu_nodes.append(node)
        allowed = set(os.sched_getaffinity(0))
        node_cpus = {}
        base = "/sys/devices/system/node"
        for d in sorted(os.listdir(base)) if os.path.isdir(base) else []:
            if d.startswith("node") and d[4:].isdigit():
                node_cpus[int(d[4:])] = [c for c in parse_cpulist(open(os.path.join(base, d, "cpulist")).read())
                                         if c in allowed]
        if any(n < 0 for n in gpu_nodes):           # no NUMA information: even slices of the allowed cores, in rank order
            node_cpus, gpu_nodes = {0: sorted(allowed)}, [0] * max(world, len(gpu_nodes))
        cpus = numa_slice(node_cpus, gpu_nodes, local_rank)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError, IndexError, RuntimeError):
        return None


def _write_shape_table(lib, path, steps):
    """Per-layer table of the profiling pass: (kind, tile, rows, K channels, taps, pixel grid) ->
    launches/step, ms/step, algorithmic TFLOP/s."""
    cap = 1 << 16
    ms = (ctypes.c_float * cap)()
    fl = (ctypes.c_double * cap)()
    meta = (ctypes.c_int * (10 * cap))()
    n = ctypes.c_int(0)
    lib.objgan_prof_dump(ms, fl, meta, cap, ctypes.byref(n))
    agg = {}
    for i in range(n.value):
        key = tuple(meta[10 * i + j] for j in range(10))
        e = agg.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += ms[i]
        e[2] += fl[i]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    kinds = {0: "gemm", 1: "wgrad", 2: "thin"}
    with open(path, "w") as f:
        f.write("# per-shape conv launches of the profiling pass (%d steps); stride < 0: strided output phases; "
                "wgrad stride x10: upsampled source, negative: reflect pad\n" % steps)
        f.write("%-6s %3s %5s %5s %3s %3s %5s %5s %4s %4s %8s %9s %8s\n" % (
            "kind", "TM", "rows", "C", "T", "N", "PH", "PW", "str", "spl", "n/step", "ms/step", "TFLOP/s"))
        for key, (cnt, tms, tfl) in rows:
            f.write("%-6s %3d %5d %5d %3d %3d %5d %5d %4d %4d %8.1f %9.3f %8.1f\n" % (
                (kinds.get(key[0], "?"),) + key[1:] + (cnt / steps, tms / steps, tfl / (tms * 1e-3) / 1e12 if tms > 0 else 0)))


def _rank_device(local_rank):
    torch.cuda.set_device(local_rank)
    return torch.device("cuda", local_rank)


def standalone_allreduce(tr, device, barrier, reps=3):
    """Every rank: stand-alone all-reduce of each network's gradient arena (the sizes the step exchanges), timed on
    the host between barriers -> {"standalone_allreduce_ms": {network: ms}, "arena_bytes": {network: bytes}}.
    Next to `exposed_wait_ms_per_step` this says how much of the exchange the step hides."""
    arenas = [("PatD%d" % i, o.arena) for i, o in enumerate(tr.optimizersPatD)] + \
             [("ShpD%d" % i, o.arena) for i, o in enumerate(tr.optimizersShpD)]
    if tr.use_obj:
        arenas += [("ObjSSD", tr.optimizerObjSSD.arena), ("ObjLSD", tr.optimizerObjLSD.arena)]
    arenas.append(("G", tr.optimizerG.arena))
    ms, nbytes = {}, {}
    for name, arena in arenas:
        buf = torch.zeros_like(arena.grad)
        dist.all_reduce(buf)                      # warm-up (connection set-up)
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(buf)
        barrier()
        ms[name] = round(1000.0 * (time.perf_counter() - t0) / reps, 3)
        nbytes[name] = buf.numel() * buf.element_size()
    return {"standalone_allreduce_ms": ms, "arena_bytes": nbytes}


def timed_passes(step, barrier, max_over_ranks, steps, warmup, prof_steps, prof_begin=None, prof_end=None):
    """The measurement protocol, on EVERY rank alike (a step holds collectives under data parallelism, so no
    pass may run on a subset of the ranks): `warmup` untimed steps; exactly `steps` steps bracketed by
    barrier + device synchronize on both sides, the MAX over ranks of that time; then a second, UNTIMED pass
    of `prof_steps` steps during which rank 0 brackets every conv launch with hipEvents on its stream and every
    rank books its collectives (`prof_begin` / `prof_end`) -- the headline carries no event overhead.
    -> (seconds of the timed pass, seconds per step of the profiling pass or None)"""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    prof_dt = None
    if prof_steps > 0:
        if prof_begin is not None:
            prof_begin()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            step()
        barrier()
        prof_dt = (time.perf_counter() - t1) / prof_steps
        if prof_end is not None:
            prof_end()
    return dt, prof_dt


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-baseline-worker":
        return _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 16; 32 with --math bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=3,
                    help="timed CPU steps after the warm-up step (default 3, SURVEY.md 8d; 0: report the un-warmed "
                         "first step)")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the two short side records of the default N = 1 run (true fp32 MFMA arithmetic; BASELINE "
                         "config 5 = bf16 mixed precision at batch 32)")
    ap.add_argument("--host-batches", action="store_true",
                    help="PCIe-inclusive side measurement: the minibatches stay in pinned HOST memory in the reference's "
                         "hand-over form (float32 images, 80-channel layout maps, box masks) and every step's batch is "
                         "uploaded inside the timed region (copy stream, one step ahead).  Never the headline value.")
    ap.add_argument("--jpeg-input", action="store_true",
                    help="real-input-shaped side measurement: every step's training images arrive as JPEG FILES (synthetic "
                         "480x640 photographs encoded once with Pillow, four batches in rotation): the file bytes cross PCIe, "
                         "are decoded on the device (csrc/jpeg.hip, entropy index after the first pass) and resized to the "
                         "branch sizes there, one step ahead on a side stream.  Never the headline value.")
    ap.add_argument("--no-is-monitor", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--force-ddp", action="store_true",
                    help="initialise the RCCL process group even at world size 1 (exercises the data-parallel "
                         "code path -- arena all-reduces, gated optimizer steps -- on a single-GPU box)")
    ap.add_argument("--shape-table", default=None, help="write the per-layer conv table of the profiling pass here")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="stage3_obj",
                    help="stage3_obj: the configuration BASELINE's metric is quoted on (default); stage3 / stage1: "
                         "BASELINE configs 3 and 2 (parity-test cases; their lines are side records, not the metric)")
    ap.add_argument("--d-streams", type=int, default=None,
                    help="HIP streams the eight discriminator updates are spread over (default: the trainer's)")
    ap.add_argument("--math", choices=("fp32", "bf16x3", "fp16x2", "bf16"), default="fp16x2",
                    help="fp32: fp32 operands on the fp32 MFMA; bf16x3: fp32 operands split exactly three ways on the "
                         "bf16 MFMA, six partial products; fp16x2 (default): fp32 operands as two fp16 pieces of x * 2^s on "
                         "the fp16 MFMA, three partial products -- fp32 accumulation and fp32 results in all three; bf16: "
                         "mixed precision of BASELINE config 5 (bf16 matrix-core inputs, fp32 accumulation / storage / "
                         "norms / optimizer)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_respawn_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    device = _rank_device(local_rank)
    pinned = pin_rank_to_numa(local_rank, world)
    use_dist = world > 1 or args.force_ddp
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        init_dist("nccl", rank, world, device)

    import synth_batch
    from objgan_hip import _lib, ops
    ops.set_conv_math(args.math)
    tr = build_trainer(device, args.batch, seed=1234, with_is_monitor=not args.no_is_monitor,
                       workload=args.workload)
    if args.d_streams is not None:
        tr.d_streams = args.d_streams
    if os.environ.get("OG_DIRECT_WGRAD") == "0":      # development A/B
        tr.direct_wgrad = False
    branch_num, _, workload_name = WORKLOADS[args.workload]
    side = 64 << (branch_num - 1)
    # a new minibatch every step, like training: four distinct per-rank batches in rotation (the trainer
    # carries nothing from step to step)
    nb = max(1, min(4, args.steps + args.warmup))
    batches = [synth_batch.make_batch(args.batch, seed=1234 + rank + 1000 * i, device=device, branch_num=branch_num)
               for i in range(nb)]
    it = [0]
    host_bytes = None
    if args.host_batches:
        # The reference's boundary hands HOST tensors over (prepare_data, trainDataset.py:79-127).  Four pinned host
        # batches in rotation, two device-side batch buffers: while step t computes on buffer t % 2, batch t + 1 crosses
        # PCIe into the other one on a copy stream (which first waits until step t - 1 has finished reading it).
        from miscc.utils import attach_host

        def tree(fn, a, b=None):
            if torch.is_tensor(a):
                return fn(a, b)
            if isinstance(a, (list, tuple)):
                return [tree(fn, x, None if b is None else b[i]) for i, x in enumerate(a)]
            return a
        host = [{k: tree(lambda t, _: t.pin_memory(), v) for k, v in
                 synth_batch.make_batch(args.batch, seed=1234 + rank + 1000 * i, branch_num=branch_num).items()}
                for i in range(nb)]
        host_bytes = sum(t.numel() * t.element_size() for v in host[0].values()
                         for t in (v if isinstance(v, (list, tuple)) else [v]) if torch.is_tensor(t))
        bufs = batches[:2] if nb >= 2 else batches * 2
        copy_stream = torch.cuda.Stream(device)
        ready, free = [None, None], [None, None]

        def refill(k, i):
            src = host[i % nb]
            with torch.cuda.stream(copy_stream):
                if free[k] is not None:
                    copy_stream.wait_event(free[k])

                def cp(dst, s_):
                    dst.copy_(s_, non_blocking=True)
                    if getattr(dst, "_og_host", None) is not None:
                        attach_host(dst, s_)
                    return dst
                for key, v in bufs[k].items():
                    tree(cp, v, src[key])
                ev = torch.cuda.Event()
                ev.record(copy_stream)
                ready[k] = ev
        refill(0, 0)

    jpeg_info = None
    if args.jpeg_input:
        # the loader side of the reference step (miscc/load.py:141-151: PIL decode + three PIL resizes per sample) on the
        # device: files -> jpeg_decode_batch -> resize_pil_bilinear_device, issued one step ahead
        import io
        import numpy as np
        from PIL import Image
        rng = np.random.RandomState(77 + rank)
        jfiles = []
        for i in range(nb * args.batch):
            a = rng.rand(480 // 4 + 2, 640 // 4 + 2, 3)
            im = Image.fromarray((a * 255).astype(np.uint8)).resize((640, 480), Image.BICUBIC)
            a = np.clip(np.asarray(im).astype(np.float32) + rng.randn(480, 640, 3) * 14.0, 0, 255).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(a).save(buf, "JPEG", quality=90, subsampling=2)
            jfiles.append(buf.getvalue())
        jcache = ops.JpegIndexCache()
        jstream = torch.cuda.Stream(device)
        sizes = [64 << b_ for b_ in range(branch_num)]
        jready = [None] * nb
        jpeg_info = {"files_per_step": args.batch, "bytes_per_step_and_gpu": int(sum(len(f) for f in jfiles) / nb),
                     "image_size": "480x640, 4:2:0, quality 90"}

        # the decode of the NEXT batch is issued on the step's own stream ahead of the step (4.2 ms of device time per batch of
        # 16): on a side stream (OBJGAN_JPEG_INLINE=0) it is an eleventh stream on a device-bound step and costs 2.5 ms more
        # (135.5 against 138.0 ms per step, 129.4 without JPEG input: profiles/r06_ab_variants.txt m)
        jinline = os.environ.get("OBJGAN_JPEG_INLINE", "1") == "1"

        def jload(i):
            k = i % nb
            with torch.cuda.stream(torch.cuda.current_stream() if jinline else jstream):
                fs = jfiles[k * args.batch:(k + 1) * args.batch]
                src, offs, hs, ws_ = ops.jpeg_decode_batch(fs, device, jcache, [(k, j) for j in range(args.batch)])
                imgs = ops.resize_pil_bilinear_device(src, offs, hs, ws_, sizes)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            jready[k] = (imgs, ev)
        for k_ in range(nb):                # "epoch 1": every file decoded once by one lane, its entropy index cached on the device
            jload(k_)
        torch.cuda.synchronize()
        jpeg_info["first_pass_index_misses"] = jcache.misses
        jload(0)

    def fresh(b):
        # The synthetic batches stay resident in HBM across steps, but nothing DERIVED from them may: objgan_hip.ops keeps the
        # |x| maxima / fp16 records of a tensor on the tensor OBJECT, and a training loop gets new tensors from its loader every
        # step.  Hand train_step new tensor objects over the same storage (detach(): no copy), so every step pays those passes.
        def f(v):
            if torch.is_tensor(v):
                d = v.detach()
                h = getattr(v, "_og_host", None)      # (the loader's own host copy of the box tables travels with the batch)
                if h is not None:
                    d._og_host = h
                return d
            if isinstance(v, (list, tuple)):
                return type(v)(f(x) for x in v)
            return v
        return dict((k_, f(v)) for k_, v in b.items())

    def step():
        if args.jpeg_input:
            k = it[0] % nb
            imgs, ev = jready[k]
            torch.cuda.current_stream().wait_event(ev)
            b_ = fresh(batches[k])
            b_["imgs"] = imgs
            for t_ in imgs:
                t_.record_stream(torch.cuda.current_stream())
            jload(it[0] + 1)
            out = tr.train_step(b_)
            it[0] += 1
            return out
        if args.host_batches:
            k = it[0] % 2
            torch.cuda.current_stream().wait_event(ready[k])
            refill(1 - k, it[0] + 1)
            out = tr.train_step(bufs[k])
            ev = torch.cuda.Event()
            ev.record()                   # (train_step has joined its side streams into this one when it returns)
            free[k] = ev
        else:
            out = tr.train_step(batches[it[0] % nb] if os.environ.get("OBJGAN_BENCH_KEEP_DERIVED") == "1" else fresh(batches[it[0] % nb]))
        it[0] += 1
        return out

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    lib = _lib.load()
    from objgan_hip import graphs
    graphs_on = graphs.enabled()
    timing = (rank == 0) and not args.no_kernel_timing
    prof_steps = 0 if args.no_kernel_timing else max(1, min(args.steps, 10))
    timed_streams = int(tr.d_streams)

    def prof_begin():
        # per-kernel durations are those of a kernel ALONE on the device: the profiling pass runs on one stream
        # (in the timed pass kernels of different streams overlap and stretch each other)
        tr.d_streams = 1
        graphs.enable(False)              # hipEvents around every convolution launch: the chains run eagerly here
        if timing:
            lib.objgan_prof_enable(1)

    def prof_end():
        if timing:
            lib.objgan_prof_enable(0)
        graphs.enable(graphs_on)
        tr.d_streams = timed_streams
    dt, prof_dt = timed_passes(step, barrier, max_over_ranks, args.steps, args.warmup, prof_steps,
                               prof_begin if timing else None, prof_end if timing else None)
    comm = None
    if use_dist and prof_steps > 0:
        # third, untimed pass on EVERY rank alike, in the stream configuration of the timed pass (the collectives are
        # issued from the side streams the discriminator updates run on, beside the small-map launches of the other
        # streams): events around every stream-side wait for a collective = time the compute stream stood still
        tr.enable_comm_stats(True)
        barrier()
        for _ in range(prof_steps):
            step()
        barrier()
        comm = tr.comm_summary(prof_steps)
        comm["measured_with_d_streams"] = timed_streams
        tr.enable_comm_stats(False)
        comm.update(standalone_allreduce(tr, device, barrier))
    # host-side cost of a step: with the queue empty, the time the host needs to ISSUE one step (it does not wait for
    # the device) and the time the device still needs afterwards.  host_issue close to ms_per_step = launch-bound.
    host = None
    if not args.no_kernel_timing:
        barrier()
        tr.phase_events = []
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        barrier()
        t2 = time.perf_counter()
        host = {"issue_ms": round(1000.0 * (t1 - t0), 2), "device_drain_ms": round(1000.0 * (t2 - t1), 2)}
        try:        # where the main stream was at the phase boundaries of this one step (the side streams' work sits inside)
            ev = tr.phase_events
            host["main_stream_phases_ms"] = {ev[i + 1][0]: round(ev[i][1].elapsed_time(ev[i + 1][1]), 2) for i in range(len(ev) - 1)}
        except Exception:                                       # noqa: BLE001 (CPU shim of the host tests)
            pass
        tr.phase_events = None
    if timing and args.shape_table:
        _write_shape_table(lib, args.shape_table, prof_steps)
    if use_dist:
        dist.barrier()

    if rank == 0:
        n_img = args.batch * world * args.steps
        res = {
            "metric": "G+D train-step images/sec at %dx%d, batch %d per GPU" % (side, side, args.batch),
            "value": round(n_img / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "fp32", "bf16x3": "fp32 (bf16x3: operands split exactly three ways on the bf16 MFMA)",
                      "fp16x2": "fp32 (fp16x2: operands as two fp16 pieces of x * 2^s on the fp16 MFMA)",
                      "bf16": "bf16-in/fp32-acc (mixed precision, config 5)"}[args.math],
            "data": "synthetic",
            "config": {"workload": workload_name + ("" if args.no_is_monitor else "+IS-monitor"),
                       "d_streams_timed_pass": timed_streams,
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": "dp%d" % world + (" (RCCL path forced)" if args.force_ddp and world == 1 else ""),
                       "fresh_batch_every_step": True,
                       "conv_math": {"fp32": "fp32 operands on the fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                                     "bf16x3": "fp32 operands split exactly into 3 bf16 pieces each, 6 partial products "
                                               "on v_mfma_f32_32x32x16_bf16, fp32 accumulation: fp32 results (error vs "
                                               "fp64 <= the fp32 MFMA's, profiles/r03_parity.txt)",
                                     "fp16x2": "fp32 operands as two fp16 pieces of x * 2^s (s from the tensor's maximum; residual "
                                               "<= 2^-23 |x|), 3 partial products on v_mfma_f32_32x32x16_f16, fp32 "
                                               "accumulation, scales undone exactly: fp32 results (error vs fp64 <= the "
                                               "fp32 MFMA's, profiles/r05_parity.txt); the forward / data-gradient "
                                               "launches that do enough arithmetic per operand element read the pixel "
                                               "operand as its PRE-SPLIT fp16 record (round 5, bit-identical results); "
                                               "launches below 1 GFLOP and the LDS-staged weight-gradient kernels run "
                                               "bf16x3 (six products on the bf16 MFMA)",
                                     "bf16": "operands rounded to bf16, fp32 accumulation"}[args.math]},
        }
        if jpeg_info is not None:
            jpeg_info["index_cache"] = {"hits": jcache.hits, "misses": jcache.misses}
            jpeg_info["what"] = ("real-input-shaped side measurement: the training images of every step arrive as JPEG files; "
                                 "bytes over PCIe, Huffman + IDCT + upsampling + colour + the three Pillow-exact resizes on "
                                 "the device, one step ahead on a side stream (first pass of a file: one lane; later passes: "
                                 "one lane per MCU row from the cached entropy index -- the first pass over the four batches happens before "
                                 "the warm-up, as epoch 1 would)")
            res["config"]["jpeg_input"] = jpeg_info
        if args.host_batches:
            res["config"]["host_batches"] = {
                "bytes_per_step_and_gpu": int(host_bytes),
                "what": "PCIe-INCLUSIVE side measurement: every step's minibatch (the reference's prepare_data hand-over: "
                        "float32 images at three scales, 80-channel layout maps, box masks, box tables, embeddings) is "
                        "uploaded from pinned host memory inside the timed region, one step ahead on a copy stream"}
        if timing:
            ms = (ctypes.c_double * 192)()
            fl = (ctypes.c_double * 192)()
            cnt = (ctypes.c_long * 192)()
            lib.objgan_prof_collect(ms, fl, cnt)
            CAT_NAMES = cat_names(args.math)
            cats = [(CAT_NAMES[i], ms[i], fl[i], cnt[i], i) for i in range(len(CAT_NAMES)) if cnt[i] > 0]
            cats.sort(key=lambda c: -c[1])
            if cats:
                name, tms, tfl, n, cat_index = cats[0]
                ach = tfl / (tms * 1e-3) / 1e12
                traffic, traffic_source = None, None          # HBM bytes / launch from the committed rocprofv3 --pmc passes
                if os.path.exists(PMC_TRAFFIC_JSON):
                    try:
                        pmc = json.load(open(PMC_TRAFFIC_JSON))
                        # the counters answer for the run AND the kernel sources they were collected on
                        here = csrc_hash()
                        same = (pmc.get("conv_math") == args.math and pmc.get("per_gpu_batch") == args.batch
                                and args.workload == "stage3_obj" and pmc.get("csrc_sha16") == here)
                        traffic = pmc_lookup(pmc.get("kernels", {}), name) if same else None
                        traffic_source = "profiles/pmc_traffic.json (%s; collected on csrc %s, this tree is %s%s)" % (
                            pmc.get("source", "committed rocprofv3 --pmc passes"), pmc.get("csrc_sha16"), here,
                            "" if pmc.get("csrc_sha16") == here else ": kernels changed since, traffic withheld")
                    except (ValueError, KeyError, OSError):
                        traffic = None
                peak = {"fp32": FP32_MFMA_PEAK_TFLOPS, "bf16": BF16_MFMA_PEAK_TFLOPS,
                        "bf16x3": round(BF16X3_PEAK_TFLOPS, 1), "fp16x2": round(FP16X2_PEAK_TFLOPS, 1)}[args.math]
                if args.math == "fp16x2" and cat_index < 48:      # the dominant instance is one of the mode's bf16x3 launches
                    peak = round(BF16X3_PEAK_TFLOPS, 1)
                res["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2),
                                   "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": traffic,
                                   # (a lookup in the committed counter passes, not a counter read of this run)
                                   "traffic_source": (traffic_source if traffic is not None else None),
                                   "algorithmic_gflop_per_launch": round(tfl / n / 1e9, 2),
                                   "launches": int(n), "avg_launch_ms": round(tms / n, 4),
                                   "share_of_step": round(tms / prof_steps / (1000.0 * prof_dt), 4),
                                   "measured_in": "second pass of %d steps ON ONE STREAM with hipEvents around every conv "
                                                  "launch (%.1f ms/step; the timed pass above runs without them and "
                                                  "on %d streams)" % (prof_steps, 1000.0 * prof_dt, timed_streams),
                                   "peak_note": ({"bf16x3": "2500 TFLOP/s dense bf16 MFMA / 6 products per fp32 MAC.  "
                                                            "Stand-alone probe of the instruction (tools/mfma_rate.hip, "
                                                            "profiles/r04_mfma_rate.txt): 2.46-2.48 PF on zero operands "
                                                            "(32 cycles per MFMA and SIMD at 2.38 GHz), 1.78-1.85 PF on "
                                                            "random / split-fp32 operands (same 32 cycles, clock held at "
                                                            "1.75-1.82 GHz by the power limit) = 297-308 TFLOP/s of "
                                                            "fp32-equivalent work for this arithmetic on real data",
                                                  "fp16x2": "2500 TFLOP/s dense fp16 MFMA (same rate as bf16) / 3 products "
                                                            "per fp32 MAC.  Stand-alone probe of the matrix pipe "
                                                            "(tools/mfma_rate.hip, profiles/r04_mfma_rate.txt): 32 cycles per "
                                                            "32x32x16 MFMA and SIMD in every arrangement, 1.78-1.85 PF on real "
                                                            "operands (clock held at 1.75-1.82 GHz by the power limit) = "
                                                            "593-617 TFLOP/s of fp32-equivalent work for this arithmetic; the "
                                                            "category also holds this tile class's launches below 1 GFLOP, "
                                                            "which run the bf16x3 instance"}
                                                 .get(args.math))}
                res["kernel_breakdown"] = [
                    {"kernel": c[0], "ms_per_step": round(c[1] / prof_steps, 3),
                     "tflops": round(c[2] / (c[1] * 1e-3) / 1e12, 2),
                     "launches_per_step": round(c[3] / prof_steps, 1)}
                    for c in cats[:10]]
                conv_ms = sum(c[1] for c in cats) / prof_steps
                conv_fl = sum(c[2] for c in cats) / prof_steps
                res["conv_total"] = {"ms_per_step": round(conv_ms, 2), "tflop_per_step": round(conv_fl / 1e12, 3),
                                     "tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                     "frac_of_peak": round(conv_fl / (conv_ms * 1e-3) / 1e12 / peak, 4)}
        if host is not None:
            if pinned:
                host["pinned_cpus_per_rank"] = len(pinned)
            gs = graphs.stats()
            host["hip_graphs"] = {"on": bool(gs["on"]), "captures": gs["captures"], "replays": gs["replays"],
                                  "fallbacks": gs["fallbacks"]}
            try:
                host["peak_device_memory_gb"] = round(torch.cuda.max_memory_allocated(device) / 2.0 ** 30, 2)
            except Exception:                                   # noqa: BLE001 (CPU shim of the host tests)
                pass
            res["host_step"] = host
        if comm is not None:
            comm["world_size"] = dist.get_world_size()
            comm["backend"] = dist.get_backend()
            try:
                comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                                   # noqa: BLE001 (CPU build / gloo)
                comm["rccl_version"] = None
            res["comm"] = comm
        if (world == 1 and not use_dist and not args.no_side_configs and not args.no_cpu_baseline
                and args.workload == "stage3_obj" and args.math == "fp16x2" and args.batch == 16):
            res["side_configs"] = side_configs()
        if world == 1 and not args.no_cpu_baseline and args.workload == "stage3_obj":
            res["cpu_baseline"] = cpu_baseline(timed_steps=args.cpu_baseline_steps,
                                               timeout_s=300 + 110 * args.cpu_baseline_steps)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
