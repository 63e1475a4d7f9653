#!/bin/bash
# round 5, first GPU call: record-form parity tests, conv micro-bench math 4 vs 5, step A/B records on/off
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5a}
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout=300 -k "records or conv2d_forward_backward or bit_reproducible or never_consumes" ) > gpurun_out/${TAG}_pytestk.log 2>&1; tail -15 gpurun_out/${TAG}_pytestk.log
( timeout 400 tools/conv_bench "" 5 4 ) > gpurun_out/${TAG}_convbench_m4.log 2>&1
( timeout 400 tools/conv_bench "" 5 5 ) > gpurun_out/${TAG}_convbench_m5.log 2>&1
paste -d'\n' gpurun_out/${TAG}_convbench_m4.log gpurun_out/${TAG}_convbench_m5.log | cut -c1-200
bash tools/ab_env.sh "OBJGAN_H2_RECORDS=0" "OBJGAN_H2_RECORDS=1" 2>&1 | tee gpurun_out/${TAG}_ab_records.txt
