#!/bin/bash
# round 6, fifth GPU call: weight gradients of the generator's backward pass on a side stream -- bit-identity tests, A/B, phases
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6e}
( time timeout 1200 python -m pytest tests/test_modules_gpu.py -m gpu -q --timeout=600 --durations=5 \
   -k "rccl_world_size_one or (full_training_step and 2-False) or graphed_encoders" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -8 gpurun_out/${TAG}_pytest.log | cut -c1-300
for g in 1 0 1 0; do
  ( OBJGAN_ASYNC_WGRAD=$g timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench_a$g.log 2>&1
  tail -1 gpurun_out/${TAG}_bench_a$g.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('async_wgrad=$g', r['value'], r['ms_per_step'], r.get('host_step'))"
done
