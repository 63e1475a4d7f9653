#!/bin/bash
# same-box sweep of the fp16x2 launch-plan knobs on tools/conv_bench (development build of the library: OBJGAN_DEV=1)
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 100 tools/conv_bench "" 8 4 | grep -E "res1_128|res2_128|res1_64|res1_32|up_256|objd_l1|objd_l2|objd_l3|d_l4|joint|roi_code|incep" | cut -c1-150; }
run X=0
run OG_H2_NW8_TM=3
run OG_H2_NW8_TM=2
run OG_NW8_MIN=256
run OG_NW8_MIN=1024
run OG_H2_PEN_PCT=200
run OG_H2_PEN_PCT=300
run OG_KGROUP_S1=2 OG_KGROUP_PH=2
run OG_KGROUP_S1=8 OG_KGROUP_PH=8
run OG_KGROUP_S2=4
run OG_SPLIT_TARGET=512
run OG_SPLIT_TARGET=2048
run X=0
