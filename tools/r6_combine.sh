#!/bin/bash
# round 6: output maxima folded into the split-K combine -- kernel / encoder tests, stale-maxima check, launch census
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6k}
R=$GRAFT_REPO_ROOT
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q --timeout=600 -k "conv or encoder or stale_maxima or inception or graphed" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-300
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
grep -i "absmax_partials\|splitk_combine" gpurun_out/${TAG}_prof1/prof_kernel_stats.csv | cut -c1-140
for i in 1 2; do ( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs --no-kernel-timing ) > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('bench', r['value'], r['ms_per_step'])"; done
