"""Times the fused Riccati + closed-loop rollout launch (descent_kernel) for a model."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import PendulumDynamics, QuadrotorDynamics, device as dev  # noqa: E402

model, T = sys.argv[1], int(sys.argv[2])
sysd = PendulumDynamics(0.05) if model == "pendulum" else QuadrotorDynamics(0.05)
dm = sysd.dm()
n, m = dm.n, dm.m
u0 = 0.1 if model == "pendulum" else 2.0
u_trj = dev.to_dev(np.full((T, m), u0))
Q, Qd, R = dev.to_dev(np.eye(n)), dev.to_dev(10 * np.eye(n)), dev.to_dev(np.eye(m))
xd = dev.to_dev(np.zeros((T + 1, n)))
x0 = dev.to_dev(np.zeros(n))
x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
At, Bt, ct = dm.exact_linearize(x_trj, u_trj)
plan = dev.DescentPlan(dm, At, Bt, ct, Q, Qd, R, xd, x0)
st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    plan.run(st)
torch.cuda.synchronize()
reps = 500
t0 = time.perf_counter()
for _ in range(reps):
    plan.run(st)
torch.cuda.synchronize()
w = (time.perf_counter() - t0) / reps
print("%s T=%d descent (Riccati+rollout+cost): %.2f us per launch, %.3f us/step, info=%d"
      % (model, T, w * 1e6, w * 1e6 / T, int(plan.out["info"].item())))
