#!/bin/bash
# round 5: full GPU suite at the current tree + the bf16-mode full-step numbers + short bench
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5h}
rm -f gpurun_out/parity_numbers.txt
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --durations=8 ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -25 gpurun_out/${TAG}_pytest.log | cut -c1-300
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null
grep -E "bf16 mode" gpurun_out/${TAG}_parity.txt | cut -c1-250
( timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench8.log 2>&1; tail -1 gpurun_out/${TAG}_bench8.log | cut -c1-400
