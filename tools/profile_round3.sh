#!/bin/bash
# Round profile of the DEFAULT bench.py run (planar_hand T=50 N=1e4 + the pendulum sub-report), run on
# the GPU box through gpurun.  Under gpurun_out/prof_$TAG:
#   stats/           rocprofv3 --kernel-trace --stats
#   pmc_<COUNTER>/   one separate --pmc pass per counter (HBM bytes; VALU instruction / busy counters)
TAG=${1:-r01c}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
# One workload per profiled process: with ~50k dispatches in one process (planar_hand + the pendulum
# sub-report) the FETCH_SIZE pass of rocprofv3 segfaulted inside its dispatch interception; the two
# halves run fine on their own and land in the same output directory.
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary > $OUT/pmc_$C.log 2>&1
  echo "pass $C (planar_hand) rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --workload pendulum > $OUT/pmc_${C}_pendulum.log 2>&1
  echo "pass $C (pendulum) rc=$?"
done
cd $R
python tools/pmc_summary3.py $OUT $TAG
# the raw traces are tens of MB (every launch is a row): keep only the summaries
rm -rf $OUT/stats $OUT/pmc_*/
