#!/bin/bash
for m in 0 1; do echo "== OG_V3=$m"; OG_V3=$m OG_WGRAD_PM=0 OG_NO_THIN=$2 tools/conv_bench "" 5 | cut -c1-170; done 2>&1 | tee gpurun_out/$1_abv3.log
