#!/bin/bash
# same-box A/B of bench.py variants: tools/ab_env.sh "ENV1=..;ENV2=.." ... ; prints ms per step for each, two rounds
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  echo -n "[$v]  "; env $(echo $v | tr ';' ' ') timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done
