#!/bin/bash
# round 6, sixth GPU call: weight gradient with BOTH operands pre-split (math 6) -- bit-identity test, micro-benchmark against
# the gather form (math 4) and the x-record form (math 5), then the step with every eligible weight gradient on it
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6f}
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout=600 -k "presplit_dy or on_records_equals" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
hipcc --offload-arch=gfx950 -O2 tools/conv_bench.cpp -Iinclude -L obj-gan_amd/objgan_hip -lobjgan_hip -Wl,-rpath,$R/obj-gan_amd/objgan_hip -o tools/conv_bench 2>&1 | tail -3
for m in 4 5 6; do echo "== math $m"; timeout 300 tools/conv_bench "" 5 $m 2>&1 | grep -v "hash" | cut -c1-175; done > gpurun_out/${TAG}_convbench.txt 2>&1
cat gpurun_out/${TAG}_convbench.txt | awk '/== math/ {print} /wgrad/ {print $1, $2, $3, $4, $5, "wgrad", $(NF-20), $(NF-19), $(NF-18), $(NF-17)}' | head -80
for cfg in "OBJGAN_REC_WGRAD=1" "OBJGAN_REC_WGRAD=all" "OBJGAN_REC_WGRAD=all OBJGAN_REC_WGRAD_DYP=0" "OBJGAN_REC_WGRAD=1 OBJGAN_ASYNC_WGRAD_D=1"; do
  ( env $cfg timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['value'], r['ms_per_step'], r['host_step'].get('main_stream_phases_ms'), r['roofline']['kernel'], r['roofline']['achieved'])"
done
