"""Backward (irs_tvlqr_riccati) vs forward (irs_closed_loop_rollout) launch times."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import PendulumDynamics, QuadrotorDynamics, device as dev  # noqa: E402
model, T = sys.argv[1], int(sys.argv[2])
sysd = PendulumDynamics(0.05) if model == "pendulum" else QuadrotorDynamics(0.05)
dm = sysd.dm(); n, m = dm.n, dm.m
u_trj = dev.to_dev(np.full((T, m), 0.1 if model == "pendulum" else 2.0))
Q, Qd, R = dev.to_dev(np.eye(n)), dev.to_dev(10 * np.eye(n)), dev.to_dev(np.eye(m))
xd = dev.to_dev(np.zeros((T + 1, n))); x0 = dev.to_dev(np.zeros(n))
x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
At, Bt, ct = dm.exact_linearize(x_trj, u_trj)
K, k, info = dev.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd)
def tm(fn, reps=300):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
tb = tm(lambda: dev.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd))
tf = tm(lambda: dm.closed_loop_rollout(K, k, x0, Q, R, xd))
to = tm(lambda: dm.rollout_cost(x0, u_trj, Q, R, xd))
print("%s T=%d: riccati %.1f us (%.2f/step) | closed-loop rollout %.1f us (%.2f/step) | open-loop rollout %.1f us" % (model, T, tb, tb / T, tf, tf / T, to))
