#!/bin/bash
# A/B of the conv kernels: tools/ab_conv.sh TAG "filter" -> v1 vs v2 (and TM caps)
TAG=$1; FILT="$2"; mkdir -p gpurun_out
for cfg in "OG_IGEMM_V1=1" "OG_IGEMM_TMMAX=8" "OG_IGEMM_TMMAX=4"; do
  echo "== $cfg"; env $cfg timeout 200 tools/conv_bench "$FILT" 5
done 2>&1 | tee gpurun_out/${TAG}_ab.log
