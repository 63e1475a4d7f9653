#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 200 tools/conv_bench "" 5 2 ) > gpurun_out/r3g_cb.log 2>&1; cut -c1-190 gpurun_out/r3g_cb.log
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout=300 ) > gpurun_out/r3g_pytest_k.log 2>&1; tail -7 gpurun_out/r3g_pytest_k.log
( timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --shape-table gpurun_out/r3g_shapes.txt ) > gpurun_out/r3g_bench.log 2>&1; grep "^{" gpurun_out/r3g_bench.log | cut -c1-250
