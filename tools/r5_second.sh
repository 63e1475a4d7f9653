#!/bin/bash
# round 5, second GPU call: full GPU suite with records as the default, conv micro-bench math 5, threshold sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5b}
( timeout 400 tools/conv_bench "" 5 5 ) > gpurun_out/${TAG}_convbench_m5.log 2>&1; cut -c1-175 gpurun_out/${TAG}_convbench_m5.log
bash tools/ab_env.sh "OBJGAN_H2_RECORDS=0" "OBJGAN_REC_MIN_I=2500" "OBJGAN_REC_MIN_I=1500;OBJGAN_REC_MIN_I_SHORT=600" "OBJGAN_REC_MIN_I=4000;OBJGAN_REC_MIN_I_SHORT=2000" "OBJGAN_REC_MIN_I=0;OBJGAN_REC_MIN_I_SHORT=0" "OBJGAN_REC_MIN_I=1e9;OBJGAN_REC_MIN_I_SHORT=1200" 2>&1 | tee gpurun_out/${TAG}_ab_records.txt
rm -f gpurun_out/parity_numbers.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout=420 --durations=15 ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -25 gpurun_out/${TAG}_pytest.log
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null
