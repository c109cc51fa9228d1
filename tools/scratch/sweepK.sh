#!/bin/bash
# times the zero-order-B exact kernel for builds with different first-attempt sweep counts
cd $GRAFT_REPO_ROOT
for K in 16 24 32 40; do
  L=irs_mpc_amd/csrc/libirs_hip_K$K.so; [ $K = 32 ] && L=irs_mpc_amd/csrc/libirs_hip.so
  for N in 10000 100000; do
    IRS_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-secondary --steps 500 --warmup 50 --N $N 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K=$K N=$N  %.1f us  %.3e  loop %.0f it/s' % (d['ms_per_step']*1e3, d['value'], d['ilqr_iters_per_s']))"
  done
done
