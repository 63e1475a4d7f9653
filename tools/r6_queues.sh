#!/bin/bash
# round 6: HIP hardware-queue count (GPU_MAX_HW_QUEUES, runtime default 4) against the step's ~10 streams; with / without early DAMSM
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6q}
for cfg in ${CFGS:-"GPU_MAX_HW_QUEUES=4"}; do
  ( env ${cfg//,/ } timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['value'], r['ms_per_step'], r['host_step'].get('main_stream_phases_ms'))"
done 2>&1 | tee gpurun_out/${TAG}_ab.txt
