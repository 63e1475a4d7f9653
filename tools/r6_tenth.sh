#!/bin/bash
# round 6, tenth GPU call: stream counts of the discriminator phase with the generator's weight gradients on their own stream
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6j}
for ds in 5 4 6 3 5; do
  ( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing --d-streams $ds ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('d_streams=$ds', r['value'], r['ms_per_step'])"
done
for m in "0,1,2,0,1,2,3,4" "2,4,3,3,4,2,1,0" "0,0,1,1,2,2,3,4"; do
  ( OBJGAN_D_STREAM_MAP=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('map=$m', r['value'], r['ms_per_step'])"
done
