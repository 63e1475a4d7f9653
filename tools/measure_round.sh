#!/bin/bash
# round 6, final measurement round at one tree, one box: GPU suite, counter passes (keyed on the kernel sources), driver-style
# bench line (reads the counter passes of this very call), single-stream kernel stats, whole-step PMC table, timeline, RCCL
# path at world size 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6z}
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_numbers.txt
( time timeout 2700 python -m pytest tests -m gpu -q --timeout=900 --durations=10 ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -16 gpurun_out/${TAG}_pytest.log | cut -c1-200
cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log | cut -c1-300
hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.cpp -o tools/pmc_calib 2>/dev/null
bash tools/pmc_bench.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1; tail -16 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-200
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
( time timeout 1700 python bench.py --steps 20 --warmup 5 --shape-table gpurun_out/${TAG}_conv_shapes.txt ) > gpurun_out/${TAG}_benchfull.log 2> gpurun_out/${TAG}_benchfull.err; tail -1 gpurun_out/${TAG}_benchfull.log | cut -c1-900
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
bash tools/pmc_step.sh ${TAG} > gpurun_out/${TAG}_pmcstep.log 2>&1
python tools/roofline_step.py gpurun_out/${TAG}_pmcstep.json gpurun_out/${TAG}_prof1/prof_kernel_stats.csv 4 gpurun_out/${TAG}_roofline_table.md > gpurun_out/${TAG}_roofline.log 2>&1; head -24 gpurun_out/${TAG}_roofline_table.md | cut -c1-200
bash tools/gpu_round.sh ${TAG} timeline > /dev/null 2>&1; head -4 gpurun_out/${TAG}_timeline.txt
( timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --force-ddp ) > gpurun_out/${TAG}_benchddp.log 2>&1; tail -1 gpurun_out/${TAG}_benchddp.log | cut -c1-300
for w in stage3 stage1; do ( timeout 300 python bench.py --steps 8 --warmup 2 --workload $w --no-cpu-baseline ) > gpurun_out/${TAG}_bench_$w.log 2>&1; tail -1 gpurun_out/${TAG}_bench_$w.log | cut -c1-300; done
