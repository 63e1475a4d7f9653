#!/usr/bin/env python3
"""Census of the cached filter banks after a few bench steps: per network (optimizer arena) the number of
banks, pack jobs, bank floats against parameter floats, and the largest banks with their keys."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "obj-gan_amd"))


def main():
    import bench
    import synth_batch
    from objgan_hip import ops
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 16, seed=1234)
    for i in range(3):
        tr.train_step(synth_batch.make_batch(16, seed=77 + i, device=dev))
    torch.cuda.synchronize()
    tot_b = tot_j = 0
    for gid, grp in ops._ARENA_BANKS.items():
        banks = grp["banks"]
        floats = sum(e.wt.numel() for e in banks)
        jobs = sum(len(e.jobs or ()) for e in banks)
        weights = {e.w.data_ptr(): e.w.numel() for e in banks}
        print("arena %x: %d banks, %d jobs, %.1f M bank floats for %.1f M weight floats in %d tensors"
              % (gid & 0xffffff, len(banks), jobs, floats / 1e6, sum(weights.values()) / 1e6, len(weights)))
        tot_b += floats
        tot_j += jobs
        ids = set(id(e) for e in banks)
        per_w = collections.defaultdict(list)
        for k, e in ops._PACK_CACHE.items():
            if id(e) in ids:
                per_w[(e.w.data_ptr(), tuple(e.w.shape))].append((e.wt.numel(), k[2:]))
        for (ptr, shape), lst in sorted(per_w.items(), key=lambda kv: -sum(x[0] for x in kv[1]))[:4]:
            print("   w %s: %d banks, %.2f M floats" % (shape, len(lst), sum(x[0] for x in lst) / 1e6))
            for n, k in sorted(lst, reverse=True)[:6]:
                print("        %.2f M  transpose=%s taps=%s layout=%s math=%s" % ((n / 1e6,) + tuple(k)))
    print("total: %.1f M bank floats (%.2f GB), %d jobs; cache entries %d"
          % (tot_b / 1e6, tot_b * 4 / 1e9, tot_j, len(ops._PACK_CACHE)))


if __name__ == "__main__":
    main()
