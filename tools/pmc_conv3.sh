#!/bin/bash
# PMC counters of the conv kernels on one conv_bench shape, one counter group per pass.
#   tools/pmc_conv3.sh TAG "shape filter" MATH
TAG=$1; FILT="$2"; MATH=${3:-0}; R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_REQ_sum TCC_BUSY_sum" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o pmc -- $R/tools/conv_bench "$FILT" 2 $MATH > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
done
cd $R; python tools/pmc_parse.py $TAG > gpurun_out/${TAG}_pmc.txt; cat gpurun_out/${TAG}_pmc.txt
find gpurun_out -name "*.csv" -size +2M -delete
