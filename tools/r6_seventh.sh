#!/bin/bash
# round 6, seventh GPU call: ROIAlign backward (anchor ranges once per pixel), JPEG-input side line, single-stream kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6g}
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout=600 -k "roi" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-300
( timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing --jpeg-input ) > gpurun_out/${TAG}_bench_jpeg.log 2>&1
tail -1 gpurun_out/${TAG}_bench_jpeg.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('jpeg-input', r['value'], r['ms_per_step'], r['config'].get('jpeg_input'))"
( timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing ) > gpurun_out/${TAG}_bench_plain.log 2>&1
tail -1 gpurun_out/${TAG}_bench_plain.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('plain', r['value'], r['ms_per_step'])"
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
grep -i "roi_\|h2_records\|jpeg\|h2_pair" gpurun_out/${TAG}_prof1/prof_kernel_stats.csv | cut -c1-170
