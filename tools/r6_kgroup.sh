#!/bin/bash
# round 6: K-group size of the stride-2 fp16x2 launches (the 5.4x over-fetch of the gather form on 96 -> 192 @ 256^2) on a
# development build (OBJGAN_DEV=1: the OG_* knobs read the environment); the shipped library is restored afterwards
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6kg}
R=$GRAFT_REPO_ROOT
cp obj-gan_amd/objgan_hip/libobjgan_hip.so /tmp/libobjgan_hip.so.ship; cp obj-gan_amd/objgan_hip/libobjgan_hip.so.flags /tmp/flags.ship
( OBJGAN_DEV=1 python -c "
import sys; sys.path[:0]=['obj-gan_amd']
from objgan_hip import build; build.build(force=True, verbose=False)" ) > gpurun_out/${TAG}_build.log 2>&1; tail -2 gpurun_out/${TAG}_build.log
hipcc --offload-arch=gfx950 -O2 tools/conv_bench.cpp -Iinclude -L obj-gan_amd/objgan_hip -lobjgan_hip -Wl,-rpath,$R/obj-gan_amd/objgan_hip -o tools/conv_bench 2>/dev/null
for m in 4 5; do for g in 0 1 2 4 8; do
  echo "== math $m OG_KGROUP_S2=$g"
  OG_KGROUP_S2=$g timeout 120 tools/conv_bench "" 6 $m 2>&1 | grep -E "objd_l1|objd_l2|objd_l3|d_l4 |d_l4s" | awk '{n=split($0,a,"|"); print substr(a[1],1,60) "|" a[2]}' | cut -c1-110
done; done 2>&1 | tee gpurun_out/${TAG}_kgroup.txt
cp /tmp/libobjgan_hip.so.ship obj-gan_amd/objgan_hip/libobjgan_hip.so; cp /tmp/flags.ship obj-gan_amd/objgan_hip/libobjgan_hip.so.flags
