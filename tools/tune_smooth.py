"""Times one smoothing launch configuration (median of HIP-event-bracketed launches and
back-to-back wall time).  Planner knobs come from the environment (IRS_SPT,
IRS_SINGLE_MAX, IRS_MAX_WG), so sweep them with one process per setting:

    IRS_MAX_WG=1024 python tools/tune_smooth.py pendulum zero 30 100000
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import PendulumDynamics, QuadrotorDynamics, device as dev, _lib  # noqa: E402

model, mode_s, T, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng_mode = len(sys.argv) > 5 and "rng" in sys.argv[5:]
fuse = "nofuse" not in sys.argv[5:]
from irs_mpc_amd import BicycleDynamics, BoxPivotingDynamics, PlanarHandDynamics, ThreeCartDynamics  # noqa: E402
sysd = {"pendulum": PendulumDynamics(0.05), "quadrotor": QuadrotorDynamics(0.05), "bicycle": BicycleDynamics(0.1),
        "three_cart": ThreeCartDynamics(0.05), "planar_hand": PlanarHandDynamics(0.1),
        "box_pivoting": BoxPivotingDynamics(0.1)}[model]
mode = {"zero": _lib.SMOOTH_ZERO_ORDER_AB, "first": _lib.SMOOTH_FIRST_ORDER, "zeroB": _lib.SMOOTH_ZERO_ORDER_B}[mode_s]
dm = sysd.dm()
n, m = dm.n, dm.m
u0 = 0.1 if model == "pendulum" else 2.0
std = 1.0 if model == "pendulum" else 0.1
u_trj = dev.to_dev(np.full((T, m), u0))
x0 = np.zeros(n)
if model == "planar_hand":      # examples/planar_hand/run_planar_hand.py:31-44
    x0 = sysd.get_x_from_q_dict({"sphere": [0.0, 0.35, 0.0], "arm_left": [-np.pi / 4] * 2, "arm_right": [np.pi / 4] * 2})
    u_trj = dev.to_dev(np.tile(x0[sysd.get_u_indices_into_x()], (T, 1)))
if model == "box_pivoting":     # examples/box_pivoting/run_box_pivoting.py:20-43 (hand sweeping to the right)
    x0 = sysd.get_x_from_q_dict({"box": [0.0, 0.5, 0.0], "hand": [-0.5, 0.5]})
    u_trj = dev.to_dev(np.stack([np.array([-0.5 + (t + 1.0) / T, 0.5]) for t in range(T)]))
Q, R = dev.to_dev(np.eye(n)), dev.to_dev(np.eye(m))
xd = dev.to_dev(np.zeros((T + 1, n)))
x_trj, _ = dm.rollout_cost(dev.to_dev(x0), u_trj, Q, R, xd)
if rng_mode:
    plan = dev.SmoothPlan(dm, mode, x_trj, u_trj, rng=dict(N=N, std_x=[std] * n, std_u=[std] * m, seed=1, iter=1), fuse=fuse)
else:
    g = torch.Generator(device="cuda").manual_seed(0)
    dx = std * torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
    du = std * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
    plan = dev.SmoothPlan(dm, mode, x_trj, u_trj, dx=dx, du=du, fuse=fuse)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    plan.run(st)
torch.cuda.synchronize()
reps = 100
ss = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
ee = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
for i in range(reps):
    ss[i].record()
    plan.run(st)
    ee[i].record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in zip(ss, ee))
t0 = time.perf_counter()
for _ in range(reps):
    plan.run(st)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
byt = 4 * ((n + m) if mode != _lib.SMOOTH_ZERO_ORDER_B else m) * N * T
print("%s %s T=%d N=%d %s env[SPT=%s SINGLE=%s MAXWG=%s]: event med %.2f us min %.2f | wall %.2f us | %.0f GB/s (wall) %.3g samples/s"
      % (model, mode_s, T, N, ("rng" if rng_mode else "supplied") + ("" if fuse else " NOFUSE"), os.environ.get("IRS_SPT"), os.environ.get("IRS_SINGLE_MAX"),
         os.environ.get("IRS_MAX_WG"), ts[len(ts) // 2] * 1e3, ts[0] * 1e3, wall * 1e6, byt / wall / 1e9, N * T / wall))
