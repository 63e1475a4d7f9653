#!/bin/bash
# round 5, third GPU call (DEVELOPMENT build of the library: knobs from the environment): tile plans of the record form,
# who still packs banks / makes maximum passes per call
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5c}
run() { echo "== $1"; env $1 timeout 300 tools/conv_bench "" 5 5 2>&1 | grep -v hash | cut -c1-110; }
( run "OG_DUMMY=0"
  run "OG_REC_NG2_MAXTM=0"
  run "OG_REC_NW8_TM=99 OG_REC_TMMAX=4 OG_REC_NG2_MAXTM=4"
  run "OG_REC_NW8_TM=99 OG_REC_TMMAX=3"
  run "OG_REC_NG2_NW8=1 OG_REC_NW8_TM=1"
  run "OG_REC_NG2_MIN=256"
  run "OG_REC_NG2_MIN=1024" ) > gpurun_out/${TAG}_tileplans.txt 2>&1
cat gpurun_out/${TAG}_tileplans.txt
OBJGAN_PACK_LOG=1 OBJGAN_H2_LOG=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/${TAG}_packlog.out 2> gpurun_out/${TAG}_packlog.err
grep -E "PACKLOG|ABSMAX" gpurun_out/${TAG}_packlog.err | cut -c1-330 | head -60
