#!/bin/bash
# HBM traffic of the bench step's MFMA convolution kernels + calibration of the counters, per
# MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, kernel trace
# only (no sys / hip / hsa tracing).  Output: gpurun_out/TAG_pmc_{fetch,write,cal_fetch,cal_write}/
#   tools/pmc_bench.sh TAG   then   python tools/pmc_traffic.py gpurun_out/TAG profiles/pmc_traffic.json
set -x
TAG=$1; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  s=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_cal_$s -o pmc -- $R/tools/pmc_calib > $R/gpurun_out/${TAG}_pmc_cal_$s.log 2>&1
  d=$R/gpurun_out/${TAG}_pmc_$s
  OBJGAN_H2_GUARD_EVERY=0 timeout 400 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "conv_igemm3_kernel|conv_wgrad" --output-format csv -d $d -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-is-monitor --d-streams 1 > $d.log 2>&1
  find $d -type f -name '*kernel_trace*' -size +3M -delete
done
cd $R; python tools/pmc_traffic.py gpurun_out/${TAG} gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.log 2>&1
cat gpurun_out/${TAG}_pmc_traffic.log
# keep the pull small: the per-dispatch tables stay on the box
find gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write -type f -size +2M -delete
