#!/bin/bash
# same-box A/B at the step level (development build): side-stream count and the split-K target of the small-map launches
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo -n "[$*]  "; env $1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-side-configs $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
run X=0 ""
run X=0 "--d-streams 4"
run X=0 "--d-streams 5"
run OG_SPLIT_TARGET=0 ""
run OG_SPLIT_TARGET=256 ""
run X=0 "--d-streams 2"
run X=0 ""
