#!/bin/bash
# round 5, fourth GPU call: fp16 saturation probe, full GPU suite (accuracy-contract tests, census, verify mode), short bench
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5d}
hipcc --offload-arch=gfx950 -O2 tools/ovfl_probe.hip -o /tmp/ovfl_probe 2>/dev/null && /tmp/ovfl_probe | tee gpurun_out/${TAG}_ovfl_probe.txt
rm -f gpurun_out/parity_numbers.txt
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --durations=12 ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -40 gpurun_out/${TAG}_pytest.log
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null
grep -E "census|per-row|re-verified|identical discriminators" gpurun_out/${TAG}_parity.txt | cut -c1-250
( timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench8.log 2>&1; tail -1 gpurun_out/${TAG}_bench8.log | cut -c1-900
