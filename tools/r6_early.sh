#!/bin/bash
# round 6: DAMSM term issued beside the discriminator updates (trainer.early_damsm) -- bit-identity test + step A/B;
# tools/conv_bench of the four weight-gradient forms at the final sources
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6e2}
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q --timeout=600 -k "damsm_term_issued or graphed_encoders" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log | cut -c1-300
for cfg in "OBJGAN_EARLY_DAMSM=0" "OBJGAN_EARLY_DAMSM=1" "OBJGAN_EARLY_DAMSM=0" "OBJGAN_EARLY_DAMSM=1"; do
  ( env $cfg timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['value'], r['ms_per_step'], r['host_step'].get('main_stream_phases_ms'))"
done
hipcc --offload-arch=gfx950 -O2 tools/conv_bench.cpp -Iinclude -L obj-gan_amd/objgan_hip -lobjgan_hip -Wl,-rpath,$R/obj-gan_amd/objgan_hip -o tools/conv_bench 2>/dev/null
for m in 4 5 6 7; do echo "== math $m"; timeout 300 tools/conv_bench "" 5 $m 2>&1 | grep -v "hash" | cut -c1-175; done > gpurun_out/${TAG}_convbench.txt 2>&1
grep -c "wgrad" gpurun_out/${TAG}_convbench.txt
