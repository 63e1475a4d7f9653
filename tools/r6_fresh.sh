#!/bin/bash
# round 6: bench.py hands train_step fresh tensor objects every step (no maxima / records of the synthetic batch carried across
# steps) -- A/B against the resident objects; --jpeg-input with the decode on the step's stream / on a side stream, same box
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6f}
for cfg in "OBJGAN_BENCH_KEEP_DERIVED=1" "OBJGAN_BENCH_KEEP_DERIVED=0" "OBJGAN_BENCH_KEEP_DERIVED=1" "OBJGAN_BENCH_KEEP_DERIVED=0" "JP=1,OBJGAN_JPEG_INLINE=0" "JP=1,OBJGAN_JPEG_INLINE=1" "JP=1,OBJGAN_JPEG_INLINE=0" "JP=1,OBJGAN_JPEG_INLINE=1"; do
  extra=""; case $cfg in JP=1*) extra="--jpeg-input";; esac
  ( env ${cfg//,/ } timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs $extra ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['value'], r['ms_per_step'], r['host_step'].get('issue_ms'), r['host_step'].get('launches_per_step'))"
done 2>&1 | tee gpurun_out/${TAG}_ab.txt
