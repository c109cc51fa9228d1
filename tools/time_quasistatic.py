"""Times the quasistatic (du-cost, trust-region) descent of the planar hand at BASELINE's horizon:
active-set solver (and ADMM with --admm: seconds per descent), u_bounds_abs vs u_bounds_rel.
    python tools/time_quasistatic.py [T] [N] [--admm]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import PlanarHandDynamics, device as dev, _lib  # noqa: E402

args = [v for v in sys.argv[1:] if not v.startswith("--")]
T = int(args[0]) if len(args) > 0 else 50
N = int(args[1]) if len(args) > 1 else 10000
sysd = PlanarHandDynamics(0.1)
dm = sysd.dm()
parts = lambda obj, arm_l, arm_r: sysd.get_x_from_q_dict({"sphere": obj, "arm_left": arm_l, "arm_right": arm_r})
uidx = sysd.get_u_indices_into_x()
x0 = parts([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
u_trj = dev.to_dev(np.tile(x0[uidx], (T, 1)))
q = parts([1e-3, 1e-3, 10.0], [1e-3, 1e-3], [1e-3, 1e-3])
Q, Qd, R = dev.to_dev(np.diag(q)), dev.to_dev(np.diag(100 * q)), dev.to_dev(5.0 * np.eye(4))
xd = dev.to_dev(np.tile(x0 + parts([0.3, -0.1, 0.5], [0, 0], [0, 0]), (T + 1, 1)))
x0d = dev.to_dev(x0)
x_trj, _ = dm.rollout_cost(x0d, u_trj, Q, R, xd)
g = torch.Generator(device="cuda").manual_seed(0)
du = 0.3 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
o = dm.smooth(_lib.SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
At, Bt, ct = o["At"], o["Bt"], o["ct"]
idx = torch.as_tensor(uidx, device="cuda")
nom = x_trj[:-1].index_select(1, idx).contiguous()
cases = {"abs": dict(u_lo=(nom - 0.05).contiguous(), u_hi=(nom + 0.05).contiguous()),
         "rel": dict(du_lo=torch.full((T, 4), -0.03, dtype=torch.float64, device="cuda"),
                     du_hi=torch.full((T, 4), 0.03, dtype=torch.float64, device="cuda")),
         "none": dict()}
for name, b in cases.items():
    for solver, label in ((2, "active-set"),) + (((1, "ADMM"),) if "--admm" in sys.argv else ()):
        kw = dict(solver=solver, rho=100.0, relax=1.6, max_iter=20000 if solver == 1 else 2000, eps=1e-9)
        out = dm.quasistatic_box_descent(At, Bt, ct, Q, Qd, R, xd, x0d, **b, **kw)
        torch.cuda.synchronize()
        reps = 5 if solver == 1 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            dm.quasistatic_box_descent(At, Bt, ct, Q, Qd, R, xd, x0d, **b, **kw, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("T=%d %-4s %-10s %9.3f ms/descent  cost %.6f  info %s"
              % (T, name, label, dt * 1e3, float(out["cost"].item()), out["info"].cpu().numpy().tolist()), flush=True)
