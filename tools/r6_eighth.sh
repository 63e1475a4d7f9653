#!/bin/bash
# round 6, eighth GPU call: record-form weight gradient with two column groups per wave (math 7): test, micro-benchmark, step A/B;
# ROIAlign backward (lane per few-tap element) test + timing
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6h}
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout=600 -k "presplit_dy or roi" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
hipcc --offload-arch=gfx950 -O2 tools/conv_bench.cpp -Iinclude -L obj-gan_amd/objgan_hip -lobjgan_hip -Wl,-rpath,$R/obj-gan_amd/objgan_hip -o tools/conv_bench 2>/dev/null
for m in 4 5 6 7; do echo "== math $m"; timeout 300 tools/conv_bench "" 5 $m 2>&1 | grep -v "hash" | cut -c1-175; done > gpurun_out/${TAG}_convbench.txt 2>&1
awk '/== math/ {print} / wgrad / {n=split($0,a,"|"); print substr(a[1],1,33) "|" a[3]}' gpurun_out/${TAG}_convbench.txt | grep -v "rgb_256\|outlogit\|patd_l1\|incep" | cut -c1-90
for cfg in "OBJGAN_REC_WGRAD=1" "OBJGAN_REC_WGRAD=1 OBJGAN_REC_WGRAD_MATH=7" "OBJGAN_REC_WGRAD=all OBJGAN_REC_WGRAD_MATH=7"; do
  ( env $cfg timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['value'], r['ms_per_step'], r['host_step'].get('main_stream_phases_ms'), r['roofline']['kernel'], r['roofline']['achieved'])"
done
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
grep -i "roi_" gpurun_out/${TAG}_prof1/prof_kernel_stats.csv | cut -c1-170
