#!/bin/bash
# round 6, closing measurement at the final tree: counter passes (keyed on the convolution sources), the driver-style line, the
# profile passes
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6y}
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.cpp -o tools/pmc_calib 2>/dev/null
bash tools/pmc_bench.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1; tail -8 gpurun_out/${TAG}_pmc_traffic.log | cut -c1-200
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
( time timeout 1700 python bench.py --steps 20 --warmup 5 --shape-table gpurun_out/${TAG}_conv_shapes.txt ) > gpurun_out/${TAG}_benchfull.log 2> gpurun_out/${TAG}_benchfull.err; tail -1 gpurun_out/${TAG}_benchfull.log | cut -c1-700
bash tools/profile_round.sh ${TAG} 2>&1 | tail -6
