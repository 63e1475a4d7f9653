#!/bin/bash
# round 6: the single-stream kernel statistics, whole-step PMC table and timeline of the final tree (no checked step of the
# fp16x2 guard inside the profiled steps: OBJGAN_H2_GUARD_EVERY=0 in the recipes)
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6p}
R=$GRAFT_REPO_ROOT
bash tools/gpu_round.sh ${TAG} prof1 > /dev/null 2>&1
bash tools/pmc_step.sh ${TAG} > gpurun_out/${TAG}_pmcstep.log 2>&1
python tools/roofline_step.py gpurun_out/${TAG}_pmcstep.json gpurun_out/${TAG}_prof1/prof_kernel_stats.csv 4 gpurun_out/${TAG}_roofline_table.md > gpurun_out/${TAG}_roofline.log 2>&1; head -30 gpurun_out/${TAG}_roofline_table.md | cut -c1-200
bash tools/gpu_round.sh ${TAG} timeline > /dev/null 2>&1; head -3 gpurun_out/${TAG}_timeline.txt
export TAG
python - <<'P'
import csv, os
rows = list(csv.DictReader(open("gpurun_out/%s_prof1/prof_kernel_stats.csv" % os.environ.get("TAG", "r6p"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 4e6
n = sum(int(r["Calls"]) for r in rows) / 4
nat = [r for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"] or "Cijk" in r["Name"]]
print("single stream: %.1f ms of kernels, %.0f launches per step; torch-native / runtime: %.1f ms, %.0f launches" % (
    tot, n, sum(float(r["TotalDurationNs"]) for r in nat) / 4e6, sum(int(r["Calls"]) for r in nat) / 4))
P
