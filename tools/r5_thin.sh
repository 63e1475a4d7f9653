#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5p}
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -x -q --timeout=600 -k "conv2d_forward_backward or conv2d_cat or bit_reproducible or full_training_step_matches_oracle or discriminator_losses_and_grads or one_launch_bank_refresh or packed_filter_cache" ) > gpurun_out/${TAG}_pytestk.log 2>&1; tail -6 gpurun_out/${TAG}_pytestk.log | cut -c1-200
bash tools/ab_env.sh "OBJGAN_THIN4=0" "OBJGAN_THIN4=1" 2>&1 | tee gpurun_out/${TAG}_ab_thin4.txt
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${TAG}_prof1/prof_kernel_stats.csv")))
print("total ms/step", sum(float(r["TotalDurationNs"]) for r in rows)/4e6, "launches", sum(int(r["Calls"]) for r in rows)/4)
for r in rows:
    if "thin" in r["Name"]: print(r["Name"][:60], int(r["Calls"])//4, "%.3f ms/step"%(float(r["TotalDurationNs"])/4e6), "%.1f us"%(float(r["AverageNs"])/1e3))
PY
