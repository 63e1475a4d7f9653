#!/bin/bash
# round 6: bf16 mode with the operand copy made once per tensor (math 3): the mode's tests, A/B of config 5
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6c5}
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q --timeout=600 -k "bf16 and not 32" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
for g in 1 0 1 0; do
  ( OBJGAN_BF16_COPY_CACHE=$g timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing --math bf16 --batch 32 ) > gpurun_out/${TAG}_bench_c$g.log 2>&1
  tail -1 gpurun_out/${TAG}_bench_c$g.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('copy_cache=$g', r['value'], r['ms_per_step'], r['host_step']['peak_device_memory_gb'] if r.get('host_step') else '')"
done
