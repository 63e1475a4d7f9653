#!/bin/bash
# round 6, fourth GPU call: 16-byte-load record pass (bit-identical records), real passes hoisted beside the generator's forward
# pass (bit-identical step), A/B of the bench step with / without the hoist
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6d}
rm -f gpurun_out/parity_numbers.txt
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q --timeout=600 --durations=8 \
   -k "records or rccl_world_size_one or (full_training_step and 2-False) or graphed" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -12 gpurun_out/${TAG}_pytest.log | cut -c1-300
for g in 1 0 1 0; do
  ( OBJGAN_HOIST_REAL=$g timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench_h$g.log 2>&1
  tail -1 gpurun_out/${TAG}_bench_h$g.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('hoist=$g', r['value'], r['ms_per_step'], r.get('host_step'))"
done
