#!/bin/bash
# Kernel-trace pass only (no PMC) of the default bench.py run: refreshes kernel_stats.csv after a change that
# does not alter the counters.  Output under gpurun_out/prof_$TAG like tools/profile_round.sh.
TAG=${1:-stats}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "stats rc=$?"
cd $R
python tools/pmc_summary.py $OUT $TAG
rm -rf $OUT/stats
