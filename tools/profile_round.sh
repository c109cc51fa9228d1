#!/bin/bash
# Round profile of bench.py, run on the GPU box through gpurun.  Under gpurun_out/prof_$TAG:
#   stats/           rocprofv3 --kernel-trace --stats of the whole DEFAULT run (every sub-report)
#   pmc_<COUNTER>/   one separate --pmc pass per counter and per workload (HBM bytes; VALU counters)
# One workload per profiled process (a --pmc pass over ~50k dispatches in one process segfaulted inside
# rocprofv3's dispatch interception in an earlier round).  Afterwards, in the build container:
#   tools/adopt_profile.sh $TAG     copies the summaries into profiles/ and stamps them with the commit
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
echo "stats rc=$?"
# per-workload stats passes: the average launch duration of each benchmarked kernel at ITS size
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_main -- python3 $R/bench.py $ARGS --no-secondary > $OUT/stats_main.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_first -- python3 $R/bench.py $ARGS --no-secondary --mode first_order > $OUT/stats_first.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pgs -- python3 $R/bench.py $ARGS --no-secondary --contact-solver pgs > $OUT/stats_pgs.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pendulum -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary --workload pendulum > $OUT/stats_pendulum.log 2>&1
echo "per-workload stats rc=$?"
if [ "$2" = "stats-only" ]; then cd $R; python tools/pmc_summary.py $OUT $TAG > /dev/null; rm -rf $OUT/stats $OUT/stats_*/ ; exit 0; fi
# matrix-core evidence for the bounded descent (ctrlbox_mfma_kernel: 2 waves, f64 MFMA) -- the loop of the main workload runs it
for C in SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary > $OUT/pmc_$C.log 2>&1
  echo "pass $C (planar_hand loop: descent kernel) rc=$?"
done
# LDS side of the uniform-geometry kernel (table lookups)
for C in SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary > $OUT/pmc_$C.log 2>&1
  echo "pass $C (planar_hand zero-order-B) rc=$?"
done
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary > $OUT/pmc_$C.log 2>&1
  echo "pass $C (planar_hand zero-order-B, exact step QP) rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary --mode first_order > $OUT/pmc_${C}_first.log 2>&1
  echo "pass $C (planar_hand first-order) rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py $ARGS --no-secondary --contact-solver pgs > $OUT/pmc_${C}_pgs.log 2>&1
  echo "pass $C (planar_hand zero-order-B, sweeps) rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary --workload pendulum > $OUT/pmc_${C}_pendulum.log 2>&1
  echo "pass $C (pendulum) rc=$?"
done
cd $R
python tools/pmc_summary.py $OUT $TAG
# the raw traces are tens of MB (every launch is a row): keep only the summaries
rm -rf $OUT/stats $OUT/stats_*/ $OUT/pmc_*/
