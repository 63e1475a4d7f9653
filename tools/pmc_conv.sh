#!/bin/bash
# PMC counters of the conv kernels on one conv_bench shape (one counter group per pass).
#   tools/pmc_conv.sh TAG "shape filter"
set -x
TAG=$1; FILT="$2"; R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
if [ ! -f $R/gpurun_out/counters_list.txt ]; then rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1; fi
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -o pmc -- $R/tools/conv_bench "$FILT" 2 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
done
cd $R; ls -la gpurun_out/${TAG}_pmc*/ | head -40
