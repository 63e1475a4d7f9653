#!/bin/bash
# round-3 first GPU call: conv micro-bench fp32 vs bf16x3, conv parity tests, one bench line per mode
set -x
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 tools/conv_bench "" 5 0 ) > gpurun_out/r3a_convbench_fp32.log 2>&1
( timeout 200 tools/conv_bench "" 5 2 ) > gpurun_out/r3a_convbench_x3.log 2>&1
cat gpurun_out/r3a_convbench_x3.log
( time timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout=300 -k "conv2d or bf16x3 or never_consumes or frozen or lift" ) > gpurun_out/r3a_pytest_conv.log 2>&1; tail -15 gpurun_out/r3a_pytest_conv.log
( time timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --math bf16x3 --shape-table gpurun_out/r3a_shapes_x3.txt ) > gpurun_out/r3a_bench_x3.log 2>&1; tail -2 gpurun_out/r3a_bench_x3.log | cut -c1-1500
( time timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --math fp32 --shape-table gpurun_out/r3a_shapes_fp32.txt ) > gpurun_out/r3a_bench_fp32.log 2>&1; tail -2 gpurun_out/r3a_bench_fp32.log | cut -c1-600
