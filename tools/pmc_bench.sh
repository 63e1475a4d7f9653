#!/bin/bash
# HBM traffic of the bench step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel
# trace only (no sys/hip/hsa tracing), per MI355X_MICROARCH.md.  Output: gpurun_out/TAG_pmc_{fetch,write}/
set -x
TAG=$1; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/${TAG}_pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-is-monitor > $d.log 2>&1
  find $d -type f -name '*kernel_trace*' -size +3M -delete
done
cd $R; ls -la gpurun_out/${TAG}_pmc_*/ | head
