#!/bin/bash
for ab in 0 1 2 4 5 7 15; do echo "== OG_ABLATE=$ab"; OG_ABLATE=$ab tools/conv_bench "$1" 5 | cut -c1-110; done 2>&1 | tee gpurun_out/$2_ablate.log
