#!/bin/bash
# round 5: issue order / stream map of the discriminator jobs (same box A/B)
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/ab_env.sh "OG_X=0" \
 "OBJGAN_D_ORDER=errPatD2,errShpD0,errShpD1,errShpD2,errObjSSD,errObjLSD,errPatD1,errPatD0" \
 "OBJGAN_D_ORDER=errPatD2,errShpD0,errShpD1,errShpD2,errObjSSD,errObjLSD,errPatD1,errPatD0;OBJGAN_D_STREAM_MAP=4,4,3,1,1,2,0,3" \
 "OBJGAN_D_ORDER=errShpD0,errShpD1,errShpD2,errObjSSD,errObjLSD,errPatD2,errPatD1,errPatD0" \
 "OBJGAN_D_ORDER=errPatD2,errPatD1,errPatD0" 2>&1 | tee gpurun_out/${1:-r5k}_ab_order.txt
