#!/bin/bash
# round 6: the driver's own sequence on the final commit -- GPU suite (-x -q), smoke, default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6v}
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-200
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log | cut -c1-300
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400; grep real gpurun_out/${TAG}_bench.err
