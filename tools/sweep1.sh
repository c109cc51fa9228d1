#!/bin/bash
# planner sweep (run on the GPU box); output -> gpurun_out/sweep1.log
out=gpurun_out/sweep1.log; : > $out
for N in 10000; do
  for S in 1 16384; do IRS_SINGLE_MAX=$S python tools/tune_smooth.py pendulum zero 30 $N >> $out 2>&1; done
  for W in 240 480 960; do IRS_SINGLE_MAX=1 IRS_MAX_WG=$W python tools/tune_smooth.py pendulum zero 30 $N >> $out 2>&1; done
done
for N in 100000 1000000; do
  for W in 512 1024 2048 4096 8192; do IRS_MAX_WG=$W python tools/tune_smooth.py pendulum zero 30 $N >> $out 2>&1; done
done
for W in 512 2048 8192; do IRS_MAX_WG=$W python tools/tune_smooth.py quadrotor first 50 10000 >> $out 2>&1; IRS_MAX_WG=$W python tools/tune_smooth.py quadrotor zero 50 10000 >> $out 2>&1; done
python tools/tune_smooth.py pendulum zero 30 10000 rng >> $out 2>&1
python tools/tune_smooth.py pendulum zero 30 1000000 rng >> $out 2>&1
python tools/tune_smooth.py quadrotor first 50 100000 >> $out 2>&1
grep -v amdgpu.ids $out
