#!/bin/bash
# First GPU call of the next round: verify and price everything that was written after the round-1 GPU
# budget was spent.  Roughly 8 GPU-minutes.  Logs under gpurun_out/TAG_*.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/next_round.sh r02a'
TAG=${1:-r02a}
mkdir -p gpurun_out; export TMPDIR=/tmp
# 1. opt-in tests: full-size properties, Winograd transforms, SHP_G_NET forward, env-selected variants
( time OG_TEST_EXPERIMENTAL=1 timeout 1500 python -m pytest tests -m gpu -q \
    -k "fullsize or full_size or winograd or shape_generator or experimental or adjoint or arena or distributions or statistics or without_object" ) \
    > gpurun_out/${TAG}_exp_pytest.log 2>&1
tail -15 gpurun_out/${TAG}_exp_pytest.log
# 2. conv micro-benchmark: default vs chunk-major K order vs 16-byte weight-gradient gathers
for cfg in "OG_NONE=1" "OG_KORDER=1" "OG_WGRAD_B128=1 OG_WGRAD3_MAXTM=7"; do
  echo "== $cfg"; env $cfg timeout 120 tools/conv_bench "" 5
done > gpurun_out/${TAG}_convbench_ab.txt 2>&1
# 3. the step: default vs each opt-in
for cfg in "OG_NONE=1" "OG_KORDER=1" "OBJGAN_WINOGRAD=1" "OBJGAN_WGRAD_INPLACE=1" "OG_WGRAD_B128=1 OG_WGRAD3_MAXTM=7"; do
  echo "== $cfg"; env $cfg timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print('   ',k) for k in d.get('kernel_breakdown',[])]"
done > gpurun_out/${TAG}_bench_ab.txt 2>&1
cat gpurun_out/${TAG}_bench_ab.txt | grep -A1 "^==" | head -30
