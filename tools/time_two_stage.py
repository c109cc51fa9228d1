"""Times the multi-GPU step's device work on one GPU: accumulate launch + solve launch (without the
all-reduce), with and without reuse of the nominal steps (irs_smooth_finalize_ws)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import PlanarHandDynamics, device as dev, _lib  # noqa: E402

T, N = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sysd = PlanarHandDynamics(0.1)
dm = sysd.dm()
x0 = sysd.get_x_from_q_dict({"sphere": [0.0, 0.35, 0.0], "arm_left": [-np.pi / 4] * 2, "arm_right": [np.pi / 4] * 2})
u_trj = dev.to_dev(np.tile(x0[sysd.get_u_indices_into_x()], (T, 1)))
Q, R = dev.to_dev(np.eye(7)), dev.to_dev(np.eye(4))
x_trj, _ = dm.rollout_cost(dev.to_dev(x0), u_trj, Q, R, dev.to_dev(np.zeros((T + 1, 7))))
du = 0.3 * torch.randn((T, N, 4), device="cuda", dtype=torch.float32)
MODE = _lib.SMOOTH_ZERO_ORDER_B
plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=None, du=du, fuse=False, n_total=N)
fused = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=None, du=du, fuse=True)
st = torch.cuda.current_stream().cuda_stream
for label, ws in (("finalize recomputes the nominal steps", None), ("finalize reuses them (finalize_ws)", plan.ws)):
    out = None
    for _ in range(20):
        plan.run(st)
        out = dm.smooth_finalize(MODE, N, x_trj, u_trj, plan.sums, out=out, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        plan.run(st)
        out = dm.smooth_finalize(MODE, N, x_trj, u_trj, plan.sums, out=out, workspace=ws)
    torch.cuda.synchronize()
    print("two-stage, %-40s %.1f us/step" % (label, (time.perf_counter() - t0) / 500 * 1e6))
for _ in range(20):
    fused.run(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    fused.run(st)
torch.cuda.synchronize()
print("fused single launch %37s %.1f us/step" % ("", (time.perf_counter() - t0) / 500 * 1e6))
