#!/bin/bash
for m in 0 1 2; do echo "== OG_WGRAD_PM=$m"; OG_WGRAD_PM=$m tools/conv_bench "" 5 | cut -c1-33,94-126,130-170; done 2>&1 | tee gpurun_out/$1_abw.log
