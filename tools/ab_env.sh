#!/bin/bash
# tools/ab_env.sh TAG FILTER "ENV1" "ENV2" ... : conv_bench A/B under different environment settings
mkdir -p gpurun_out
TAG=$1; FILT=$2; shift; shift
for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 100 tools/conv_bench "$FILT" 5
done > gpurun_out/${TAG}_ab.txt 2>&1
cat gpurun_out/${TAG}_ab.txt
