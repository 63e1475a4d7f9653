#!/bin/bash
# A/B of the extra-VALU-rows forms on the 194 / 388-channel layers (conv_bench only: ~1 GPU-minute)
mkdir -p gpurun_out
for cfg in "OG_NO_XROWS=1" "OG_NONE=1" "OG_XR_MINTILES=1"; do
  echo "== $cfg"; env $cfg timeout 100 tools/conv_bench "res" 5
done > gpurun_out/${1:-xr}_ab.txt 2>&1
cat gpurun_out/${1:-xr}_ab.txt
