#!/bin/bash
out=gpurun_out/sweep2.log; : > $out
python tools/tune_smooth.py pendulum zero 30 10000 >> $out 2>&1
python tools/tune_smooth.py pendulum zero 30 1000 >> $out 2>&1
for W in 256 384 512 1024; do IRS_MAX_WG=$W IRS_SPT=1 python tools/tune_smooth.py pendulum zero 30 100000 >> $out 2>&1; done
for S in 8 16 32 64; do IRS_SPT=$S python tools/tune_smooth.py pendulum zero 30 100000 >> $out 2>&1; done
for W in 512 768 1024 1536; do IRS_MAX_WG=$W python tools/tune_smooth.py pendulum zero 30 1000000 >> $out 2>&1; done
for W in 128 256 512; do IRS_MAX_WG_HEAVY=$W python tools/tune_smooth.py quadrotor first 50 10000 >> $out 2>&1; IRS_MAX_WG_HEAVY=$W python tools/tune_smooth.py quadrotor zero 50 10000 >> $out 2>&1; done
for W in 256 512; do IRS_MAX_WG_HEAVY=$W python tools/tune_smooth.py quadrotor first 50 100000 >> $out 2>&1; done
python tools/tune_smooth.py pendulum zero 30 10000 rng >> $out 2>&1
python tools/tune_smooth.py pendulum zero 30 1000000 rng >> $out 2>&1
python tools/tune_smooth.py pendulum first 30 1000000 >> $out 2>&1
grep -v amdgpu.ids $out
