#!/bin/bash
# main-loop ablation of the bf16x3 kernel (development build, OG_ABLATE bits): profiles/r03_ablation_bf16x3_mainloop.txt
mkdir -p gpurun_out; export TMPDIR=/tmp
for abl in 0 62 64 126 127; do
  for f in objd_l3 res2_128; do
    echo -n "NW4 ABL=$abl "; OG_NW8_MIN=0 OG_ABLATE=$abl timeout 100 tools/conv_bench $f 5 2 | cut -c1-92
  done
done > gpurun_out/r3e_ablate2.txt 2>&1
cat gpurun_out/r3e_ablate2.txt
