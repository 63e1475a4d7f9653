#!/bin/bash
# round 5: record-based weight gradient -- kernel parity tests, conv micro-bench math 4 vs 5, step A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5g}
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q --timeout=300 -k "records or conv2d_forward_backward or as_accurate or per_row or wrong_maxima or adjoint or bit_reproducible" ) > gpurun_out/${TAG}_pytestk.log 2>&1; tail -12 gpurun_out/${TAG}_pytestk.log
( timeout 400 tools/conv_bench "" 5 4 ) > gpurun_out/${TAG}_convbench_m4.log 2>&1
( timeout 400 tools/conv_bench "" 5 5 ) > gpurun_out/${TAG}_convbench_m5.log 2>&1
paste -d'\n' <(grep -v hash gpurun_out/${TAG}_convbench_m4.log) <(grep -v hash gpurun_out/${TAG}_convbench_m5.log) | cut -c1-215
bash tools/ab_env.sh "OBJGAN_REC_WGRAD=0" "OBJGAN_REC_WGRAD=1" 2>&1 | tee gpurun_out/${TAG}_ab_wgrad.txt
