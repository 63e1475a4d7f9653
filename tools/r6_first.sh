#!/bin/bash
# round 6, first GPU call: the new parity cases (batch-16 reference golden, configs 2 / 3 at batch 16, bf16 mode at batch 32
# + its linear generator-gradient check), the hipGraph tests, and an A/B of the bench step with / without graphs
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6a}
rm -f gpurun_out/parity_numbers.txt
( time timeout 1500 python -m pytest tests/test_batch16_gpu.py tests/test_modules_gpu.py -m gpu -q --timeout=900 --durations=12 \
   -k "batch_16 or graphed or bf16_mode or without_object" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -25 gpurun_out/${TAG}_pytest.log | cut -c1-300
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null
for g in 1 0 1 0; do
  ( OBJGAN_GRAPHS=$g timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench_g$g.log 2>&1
  tail -1 gpurun_out/${TAG}_bench_g$g.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('graphs=$g', r['value'], r['ms_per_step'], r.get('host_step'))"
done
