#!/bin/bash
# round 6, ninth GPU call: dense ROIAlign backward, two-group record weight gradient (refill order fixed), JPEG rows over workgroups
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6i}
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_numbers.txt
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_jpeg_gpu.py -m gpu -q --timeout=600 -k "presplit_dy or roi or jpeg" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
grep -i jpeg gpurun_out/parity_numbers.txt
( timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing --jpeg-input ) > gpurun_out/${TAG}_bench_jpeg.log 2>&1
tail -1 gpurun_out/${TAG}_bench_jpeg.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('jpeg-input', r['value'], r['ms_per_step'])"
( timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing ) > gpurun_out/${TAG}_bench_plain.log 2>&1
tail -1 gpurun_out/${TAG}_bench_plain.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('plain', r['value'], r['ms_per_step'])"
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
grep -i "roi_" gpurun_out/${TAG}_prof1/prof_kernel_stats.csv | cut -c1-170
