#!/bin/bash
# Memory-path PMC counters of the conv kernels on one conv_bench shape (one counter group per pass).
#   tools/pmc_conv2.sh TAG "shape filter"
TAG=$1; FILT="$2"; R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_REQ_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o pmc -- $R/tools/conv_bench "$FILT" 2 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
done
cd $R; python tools/pmc_parse.py $TAG
