#!/bin/bash
# round 6: single-stream kernel statistics of config 5 (bf16 mode, B = 32): what the per-call operand copies cost
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6b5}
R=$GRAFT_REPO_ROOT
cd /tmp && OBJGAN_H2_GUARD_EVERY=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-configs --no-kernel-timing --d-streams 1 --math bf16 --batch 32 > $R/gpurun_out/${TAG}_prof.log 2>&1
cd $R; find gpurun_out/${TAG}_prof -type f ! -name '*stats*' -size +1M -delete
head -40 gpurun_out/${TAG}_prof/prof_kernel_stats.csv | cut -c1-150
python - <<'P'
import csv
rows = list(csv.DictReader(open("gpurun_out/r6b5_prof/prof_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 4e6
cp = [r for r in rows if "nchw_to_nhwc_bf16" in r["Name"] or "f32_to_bf16" in r["Name"]]
print("single stream %.1f ms per step; operand copies %.1f ms, %d launches per step" % (tot, sum(float(r["TotalDurationNs"]) for r in cp) / 4e6, sum(int(r["Calls"]) for r in cp) / 4))
P
