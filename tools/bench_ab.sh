#!/bin/bash
# tools/bench_ab.sh TAG "ENV1" "ENV2" ...: the full step under different environment settings (8 steps each)
mkdir -p gpurun_out
TAG=$1; shift
for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done > gpurun_out/${TAG}_bench_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_ab.txt
