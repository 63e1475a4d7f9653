#!/bin/bash
# Runs on the GPU box (via gpurun): [tests,] conv micro-bench, bench line, rocprofv3 kernel stats.
#   tools/gpu_round.sh TAG [test] [convbench] [bench] [benchfull] [benchddp] [prof] ...
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-run}; shift
R=$GRAFT_REPO_ROOT
for what in "$@"; do case $what in
test) rm -f gpurun_out/parity_numbers.txt
  ( time timeout 1500 python -m pytest tests -m gpu -x -v --timeout=420 --durations=40 ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log
  cp gpurun_out/parity_numbers.txt gpurun_out/${TAG}_parity.txt 2>/dev/null ;;
testk) ( time timeout 900 python -m pytest tests -m gpu -x -v --timeout=240 --durations=10 -k "$OG_K" ) > gpurun_out/${TAG}_pytestk.log 2>&1; tail -15 gpurun_out/${TAG}_pytestk.log ;;
convbench) ( timeout 300 tools/conv_bench "" 5 ) > gpurun_out/${TAG}_convbench.log 2>&1; cat gpurun_out/${TAG}_convbench.log ;;
bench) ( time timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --shape-table gpurun_out/${TAG}_shapes.txt ) > gpurun_out/${TAG}_bench.log 2>&1; tail -3 gpurun_out/${TAG}_bench.log ;;
benchfull) ( time timeout 1700 python bench.py --steps 20 --warmup 5 --shape-table gpurun_out/${TAG}_conv_shapes.txt ) > gpurun_out/${TAG}_benchfull.log 2>&1; tail -3 gpurun_out/${TAG}_benchfull.log ;;
benchddp) ( time timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --force-ddp ) > gpurun_out/${TAG}_benchddp.log 2>&1; tail -3 gpurun_out/${TAG}_benchddp.log ;;
benchbf16) ( time timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --math bf16 --batch 32 ) > gpurun_out/${TAG}_benchbf16.log 2>&1; tail -3 gpurun_out/${TAG}_benchbf16.log | cut -c1-600 ;;
benchcfg) for w in stage3 stage1; do ( time timeout 300 python bench.py --steps 8 --warmup 2 --workload $w ) > gpurun_out/${TAG}_bench_$w.log 2>&1; tail -3 gpurun_out/${TAG}_bench_$w.log | cut -c1-700; done ;;
bench8t) ( time timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --shape-table gpurun_out/${TAG}_conv_shapes.txt ) > gpurun_out/${TAG}_bench8t.log 2>&1; tail -3 gpurun_out/${TAG}_bench8t.log | cut -c1-400 ;;
bench8) ( time timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline ) > gpurun_out/${TAG}_bench8.log 2>&1; tail -3 gpurun_out/${TAG}_bench8.log | cut -c1-400 ;;
smoke) ( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
pmcstep) bash tools/pmc_step.sh ${TAG} > gpurun_out/${TAG}_pmcstep.log 2>&1; tail -3 gpurun_out/${TAG}_pmcstep.log ;;
pmc) bash tools/pmc_bench.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1; tail -20 gpurun_out/${TAG}_pmc_traffic.log ;;
opsrc) ( timeout 600 python tools/op_sources.py ) > gpurun_out/${TAG}_opsrc.txt 2> gpurun_out/${TAG}_opsrc.err; head -40 gpurun_out/${TAG}_opsrc.txt ;;
benchenv) ( time env $OG_ENV timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline ) > gpurun_out/${TAG}_benchenv.log 2>&1; tail -3 gpurun_out/${TAG}_benchenv.log | cut -c1-400 ;;
profenv) cd /tmp && env $OG_ENV timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_profenv -o prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_profenv.log 2>&1
  cd $R; find gpurun_out/${TAG}_profenv -type f ! -name '*stats*' -size +1M -delete ;;
prof1) cd /tmp && OBJGAN_H2_GUARD_EVERY=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof1 -o prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --d-streams 1 > $R/gpurun_out/${TAG}_prof1.log 2>&1
  cd $R; find gpurun_out/${TAG}_prof1 -type f ! -name '*stats*' -size +1M -delete ;;
prof) cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_prof.log 2>&1
  cd $R; find gpurun_out/${TAG}_prof -type f | head; find gpurun_out/${TAG}_prof -type f ! -name '*stats*' -size +1M -delete ;;
timeline) cd /tmp && OBJGAN_H2_GUARD_EVERY=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_tl -o tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_tl.log 2>&1
  cd $R; python tools/timeline.py gpurun_out/${TAG}_tl 4 100 | tee gpurun_out/${TAG}_timeline.txt; python tools/timeline.py gpurun_out/${TAG}_tl 4 30 | tee -a gpurun_out/${TAG}_timeline.txt
  find gpurun_out/${TAG}_tl -type f -size +40M -delete ;;
esac; done
