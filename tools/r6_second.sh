#!/bin/bash
# round 6, second GPU call: JPEG decode on the device, the bf16 mode at its own batch with the linear generator-gradient check
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6b}
rm -f gpurun_out/parity_numbers.txt
( time timeout 1500 python -m pytest tests/test_jpeg_gpu.py tests/test_modules_gpu.py -m gpu -q --timeout=900 --durations=12 \
   -k "jpeg or bf16_mode_full" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -25 gpurun_out/${TAG}_pytest.log | cut -c1-300
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null; grep -i "jpeg\|G gradient\|G Adam" gpurun_out/${TAG}_parity.txt
