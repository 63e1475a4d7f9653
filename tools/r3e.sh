#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for abl in 0 1 2 4 6 8 16 32 38 62 63; do
  for f in objd_l3 res2_128; do
    echo -n "ABL=$abl "; OG_ABLATE=$abl timeout 100 tools/conv_bench $f 5 2 | cut -c1-92
  done
done > gpurun_out/r3e_ablate.txt 2>&1
cat gpurun_out/r3e_ablate.txt
