#!/bin/bash
# round 6, third GPU call: JPEG (entropy index), ROIAlign backward with LDS staging, fp16x2 guard; census of the per-call
# packs / maximum passes that are left; single-stream kernel stats of the current tree
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6c}
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_numbers.txt
( time timeout 1200 python -m pytest tests/test_jpeg_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout=600 --durations=8 \
   -k "jpeg or roi or guard or per_row" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -22 gpurun_out/${TAG}_pytest.log | cut -c1-300
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null; grep -i "jpeg\|guard" gpurun_out/${TAG}_parity.txt
( OBJGAN_PACK_LOG=1 OBJGAN_H2_LOG=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing ) > gpurun_out/${TAG}_census.log 2> gpurun_out/${TAG}_census.err
grep -c PACKLOG gpurun_out/${TAG}_census.err; grep PACKLOG gpurun_out/${TAG}_census.err | head -40 | cut -c1-260
grep ABSMAX gpurun_out/${TAG}_census.err | head -40 | cut -c1-260
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
head -60 gpurun_out/${TAG}_prof1/prof_kernel_stats.csv | cut -c1-160
