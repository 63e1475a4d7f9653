#!/usr/bin/env python
"""Where do the small torch-side launches of a training step come from?  Profiles ONE step with python
stacks and aggregates the aten ops that launch fills / copies / adds by the innermost frame inside this repo.
    python tools/op_sources.py > gpurun_out/op_sources.txt
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "obj-gan_amd")]
import torch                                               # noqa: E402
from torch.profiler import profile, ProfilerActivity       # noqa: E402

import bench                                               # noqa: E402
import synth_batch                                         # noqa: E402

dev = torch.device("cuda:0")
tr = bench.build_trainer(dev, 16, seed=1234)
batch = synth_batch.make_batch(16, seed=1234, device=dev)
for _ in range(2):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::mul", "aten::mul_",
         "aten::cat", "aten::zeros", "aten::index", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy",
         "aten::sum", "aten::div", "aten::sub", "aten::select", "aten::index_select", "aten::gather", "aten::where")
agg = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH or not ev.stack:
        continue
    frame = next((f for f in ev.stack if ROOT in f and "tools/op_sources" not in f), None)
    if frame is None:
        frame = "autograd engine / other: " + (ev.stack[0] if ev.stack else "?")
    agg[(ev.name, frame.replace(ROOT + "/", ""))] += 1
for (name, frame), n in agg.most_common(70):
    print("%5d  %-18s %s" % (n, name, frame))
