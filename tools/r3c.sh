#!/bin/bash
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; ( env "$@" timeout 200 tools/conv_bench "" 5 2 ) > gpurun_out/r3c_cb_$tag.log 2>&1; }
run A OG_NW8_MIN=0
run B OG_NW8_MIN=512
run C OG_NW8_MIN=512 OG_X3_WGRAD3_MAXTM=7
run D OG_NW8_MIN=512 OG_X3_WGRAD3_MAXTM=7 OG_IGEMM_TMMAX=6
run E OG_NW8_MIN=256 OG_X3_WGRAD3_MAXTM=7
for t in A B C D E; do echo == $t; cut -c1-118 gpurun_out/r3c_cb_$t.log | grep -v "^+"; done
( OG_X3_WGRAD3_MAXTM=7 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout=300 -k "conv2d or bf16x3 or never_consumes" ) > gpurun_out/r3c_pytest_conv.log 2>&1; tail -5 gpurun_out/r3c_pytest_conv.log
( OG_X3_WGRAD3_MAXTM=7 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --math bf16x3 --shape-table gpurun_out/r3c_shapes_C.txt ) > gpurun_out/r3c_bench_C.log 2>&1; grep "^{" gpurun_out/r3c_bench_C.log | cut -c1-250
( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --math bf16x3 --shape-table gpurun_out/r3c_shapes_B.txt ) > gpurun_out/r3c_bench_B.log 2>&1; grep "^{" gpurun_out/r3c_bench_B.log | cut -c1-250
