#!/bin/bash
# whole-step A/B of the number of HIP streams the discriminator updates / generator-loss terms run on
mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 1 2 3; do
  ( timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --d-streams $n ) > gpurun_out/r3o_bench_ds$n.log 2>&1
  grep "^{" gpurun_out/r3o_bench_ds$n.log | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('d_streams', r['config']['d_streams_timed_pass'], r['value'], r['ms_per_step'])"
done
