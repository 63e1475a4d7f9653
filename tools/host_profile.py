"""Host-side profile of the bench step (cProfile over a few steps on the GPU box): where the Python / launch time goes.
    python tools/host_profile.py [steps] > gpurun_out/host_profile.txt"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd")]
import torch                     # noqa: E402
import bench                     # noqa: E402
import synth_batch               # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
tr = bench.build_trainer(dev, 16, seed=1234)
batches = [synth_batch.make_batch(16, seed=1234 + 1000 * i, device=dev, branch_num=3) for i in range(4)]
for i in range(3):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.train_step(batches[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("plain: issue %.1f ms/step, drain %.1f ms" % (1000 * (t1 - t0) / steps, 1000 * (t2 - t1)))
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    tr.train_step(batches[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
