#!/usr/bin/env python
"""Which torch-side operators launch the small fills / copies / adds of a training step?  Profiles ONE step and
prints the aten operators by call count with their input shapes (the shapes identify the call sites).
    python tools/op_sources.py > gpurun_out/op_sources.txt
"""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.train_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True)
        if e.key.startswith("aten::") and e.key.split("::")[1] in (
            "copy_", "fill_", "zero_", "add_", "add", "mul", "mul_", "cat", "zeros", "clone", "contiguous", "_to_copy",
            "index", "index_put_", "sum", "div", "sub", "where", "masked_fill", "full_like", "zeros_like", "empty_like",
            "select", "slice", "expand", "reshape", "view", "transpose", "sigmoid", "exp", "binary_cross_entropy")]
rows.sort(key=lambda e: -e.count)
for e in rows[:90]:
    print("%5d  %-28s %s" % (e.count, e.key, str(e.input_shapes)[:150]))
