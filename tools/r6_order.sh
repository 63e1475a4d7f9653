#!/bin/bash
# round 6: issue order of the eight discriminator jobs, now free of the RNG sequence (permutations drawn up front)
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6o}
( timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q --timeout=600 -k "rccl_world_size_one or (full_training_step and 2-False) or discriminator_losses" ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log | cut -c1-200
for o in "" "errObjLSD,errObjSSD,errPatD2,errShpD2,errPatD1,errShpD1,errPatD0,errShpD0" "errObjSSD,errObjLSD,errShpD2,errPatD2,errShpD1,errPatD1,errShpD0,errPatD0" "" "errObjLSD,errObjSSD,errPatD2,errShpD2,errPatD1,errShpD1,errPatD0,errShpD0"; do
  ( OBJGAN_D_ORDER=$o timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-configs ) > gpurun_out/${TAG}_bench.log 2>&1
  tail -1 gpurun_out/${TAG}_bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('order=[$o]', r['value'], r['ms_per_step'], r['host_step']['main_stream_phases_ms'])"
done
