#!/bin/bash
# Round profile (run on the GPU box through gpurun).  Produces, under gpurun_out/prof_$TAG:
#   stats/   rocprofv3 --kernel-trace --stats of `bench.py --steps 100`
#   fetch/   separate pass: --pmc FETCH_SIZE      write/  separate pass: --pmc WRITE_SIZE
# for the headline N (default 10000) and for N=1000000 (the HBM-streaming regime).
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for N in 10000 1000000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_N$N -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --N $N > $OUT/stats_N$N.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_N$N -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --N $N > $OUT/fetch_N$N.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_N$N -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --N $N > $OUT/write_N$N.log 2>&1
done
cd $R
python tools/pmc_summary.py $OUT $TAG
