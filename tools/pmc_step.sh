#!/bin/bash
# Whole-step PMC passes (every kernel of the bench step, not only the convolutions): HBM bytes and MFMA-pipe
# occupancy per kernel.  Separate rocprofv3 --pmc passes with the kernel trace only, as MI355X_MICROARCH.md
# prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; no sys / hip / hsa tracing next to --pmc).
#   tools/pmc_step.sh TAG   ->  gpurun_out/TAG_pmcstep.json ; then tools/roofline_step.py builds the table
set -x
TAG=$1; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p $R/gpurun_out; cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=/tmp/${TAG}_pmcstep$i
  OBJGAN_H2_GUARD_EVERY=0 timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --d-streams 1 > $R/gpurun_out/${TAG}_pmcstep$i.log 2>&1
done
cd $R; python tools/pmc_step.py /tmp/${TAG}_pmcstep gpurun_out/${TAG}_pmcstep.json
