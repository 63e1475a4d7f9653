#!/usr/bin/env python
"""Benchmark of the Obj-GAN image_generation G+D training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full iteration of the reference's training loop (reference
image_generation/trainer.py:357-472) on one synthetic COCO-shaped minibatch: three-stage G_NET
forward at 256x256 (fed by the frozen caption encoder and GloVe lookup), the three patch and three
shape discriminators, the two ROIAlign object discriminators, generator loss with DAMSM + KL, nine
Adam updates and the generator EMA (and the per-step Inception-score monitor on a side stream, as in
the reference).  Per-GPU batch 16; under
`--gpus N` (launched by torch.distributed.run, one rank per GPU, RCCL) every rank steps its own
batch and gradients are all-reduced: weak scaling, value = N * 16 * steps / time.

One JSON line is printed by rank 0: the contract fields plus
  roofline      the dominant kernel instance (the MFMA implicit-GEMM conv with the largest share
                of the step): algorithmic flops of its launches / their hipEvent-measured duration
                (events recorded by the library on the launch stream), against the fp32 MFMA peak;
                traffic = HBM bytes per launch from the rocprofv3 --pmc passes kept in profiles/
  cpu_baseline  the CPU oracle (oracle/torch_model.py, a port of the reference path) timed on this
                host's cores on a bounded sample (one step at batch 8, <= 16 threads,
                child process with a hard time limit)
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
CAT_NAMES = (["conv_igemm3_kernel<%d, false>" % tm for tm in range(1, 8)] +
             ["conv_wgrad2_kernel<%d>" % tm for tm in range(1, 8)] +
             ["conv_thin_kernel", "conv_thin3x3_kernel", "conv_igemm_kernel", "conv_wgrad_kernel",
              "conv_igemm3_kernel<1, true>", "conv_wgrad3_kernel<1>", "conv_wgrad3_kernel<2>"])
PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def build_trainer(device, batch_size, seed, with_is_monitor=True):
    import encoders
    import trainer as T
    from miscc.config import cfg
    cfg.TREE.BRANCH_NUM = 3
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
    tr.setup()
    return tr


def _cpu_baseline_worker(sample_batch, threads, seed=1234):
    """(child process) the oracle -- CPU port of the reference step -- on `threads` host threads."""
    import model as M
    import encoders
    import synth_batch
    from miscc.config import cfg
    from miscc.utils import weights_init
    from oracle import torch_model as tm
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
    enc = encoders.CNN_ENCODER(256, encoders.seeded_init_(encoders.inception_v3(), 1)).eval()
    batch = synth_batch.make_batch(sample_batch, seed=seed)
    t0 = time.time()
    tm.train_step(sds, opts, ema, batch, image_encoder=enc)
    dt = time.time() - t0
    print("CPU_BASELINE " + json.dumps(
        {"value": round(sample_batch / dt, 4), "unit": "images/sec", "cores": threads, "kind": "port",
         "sample": "1 full G+D step at batch %d (same networks, 256x256), %.1f s on %d threads of %d host cores"
                   % (sample_batch, dt, threads, os.cpu_count() or 1)}), flush=True)


def cpu_baseline(sample_batch=8, max_threads=16, timeout_s=240):
    """The oracle timed on the host cores, on a bounded sample (one step at a small batch), in a
    child process with a hard time limit so that the default bench run always finishes."""
    import subprocess
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(sample_batch), str(threads)]
    try:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             timeout=timeout_s).stdout.decode(errors="replace")
        for line in out.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        note = "worker produced no result"
    except subprocess.TimeoutExpired:
        note = "did not finish one batch-%d step within %d s on %d threads" % (sample_batch, timeout_s, threads)
    return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": note}


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-worker":
        return _cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-is-monitor", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--math", choices=("fp32", "bf16"), default="fp32",
                    help="fp32: exact fp32 MFMA (headline, parity path); bf16: mixed precision of BASELINE "
                         "config 5 (bf16 matrix-core inputs, fp32 accumulation / storage / norms / optimizer)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    import synth_batch
    from objgan_hip import _lib, ops
    ops.set_conv_math(args.math)
    tr = build_trainer(device, args.batch, seed=1234, with_is_monitor=not args.no_is_monitor)
    batch = synth_batch.make_batch(args.batch, seed=1234 + rank, device=device)   # per-rank data shard

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.train_step(batch)
    lib = _lib.load()
    timing = (rank == 0) and not args.no_kernel_timing
    barrier()
    if timing:
        lib.objgan_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if timing:
        lib.objgan_prof_enable(0)
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        n_img = args.batch * world * args.steps
        res = {
            "metric": "G+D train-step images/sec at 256x256, batch 16 per GPU",
            "value": round(n_img / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if args.math == "fp32" else "bf16-in/fp32-acc (mixed precision, config 5)",
            "data": "synthetic",
            "config": {"workload": "stage3_256x256_full_GD_step: RNN_ENCODER+G_NET(3 stages)+PatD x3+ShpD x3+"
                                   "ObjSSD+ObjLSD(ROIAlign)+DAMSM+KL+Adam x9+EMA"
                                   + ("" if args.no_is_monitor else "+IS-monitor"),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": "dp%d" % world},
        }
        if timing:
            ms = (ctypes.c_double * 32)()
            fl = (ctypes.c_double * 32)()
            cnt = (ctypes.c_long * 32)()
            lib.objgan_prof_collect(ms, fl, cnt)
            cats = [(CAT_NAMES[i], ms[i], fl[i], cnt[i]) for i in range(len(CAT_NAMES)) if cnt[i] > 0]
            cats.sort(key=lambda c: -c[1])
            if cats:
                name, tms, tfl, n = cats[0]
                ach = tfl / (tms * 1e-3) / 1e12
                traffic = None          # HBM bytes / launch from the committed rocprofv3 --pmc passes
                if os.path.exists(PMC_TRAFFIC_JSON):
                    try:
                        kern = json.load(open(PMC_TRAFFIC_JSON)).get("kernels", {})
                        # the profiler prints every template argument, the category names only the
                        # leading ones: "conv_igemm3_kernel<6, false>" is "...<6, false, false>" (fp32)
                        ent = kern.get(name) or kern.get(name[:-1] + ", false>" if name.endswith(">") else name)
                        traffic = ent["hbm_bytes_per_launch"] if ent else None
                    except (ValueError, KeyError, OSError):
                        traffic = None
                peak = FP32_MFMA_PEAK_TFLOPS if args.math == "fp32" else BF16_MFMA_PEAK_TFLOPS
                res["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2),
                                   "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "traffic": traffic if args.math == "fp32" else None,
                                   "algorithmic_gflop_per_launch": round(tfl / n / 1e9, 2),
                                   "launches": int(n), "avg_launch_ms": round(tms / n, 4),
                                   "share_of_step": round(tms / (1000.0 * dt), 4)}
                res["kernel_breakdown"] = [
                    {"kernel": c[0], "ms_per_step": round(c[1] / args.steps, 3),
                     "tflops": round(c[2] / (c[1] * 1e-3) / 1e12, 2), "launches_per_step": c[3] // args.steps}
                    for c in cats[:8]]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
