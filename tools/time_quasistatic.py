"""Times the quasistatic (du-cost, one control box) descent at BASELINE's horizons: the two active-set
solvers (2 = lanes + LDS, 3 = matrix-core tiles), u_bounds_abs vs u_bounds_rel vs none, cold start.
    python tools/time_quasistatic.py [planar_hand|box_pivoting] [T] [N] [--admm]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import device as dev, _lib  # noqa: E402
import bench  # noqa: E402

args = [v for v in sys.argv[1:] if not v.startswith("--")]
name = args[0] if args else "planar_hand"
T = int(args[1]) if len(args) > 1 else None
N = int(args[2]) if len(args) > 2 else 10000
w = bench.Workload(name, T)
T = w.T
dm = w.system.dm()
m = dm.m
Q, Qd, R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
xd, x0d, u_trj = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
x_trj, _ = dm.rollout_cost(x0d, u_trj, Q, R, xd)
g = torch.Generator(device="cuda").manual_seed(0)
du = w.std_u * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
o = dm.smooth(_lib.SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
At, Bt, ct = o["At"], o["Bt"], o["ct"]
idx = torch.as_tensor(w.idx, device="cuda")
nom = x_trj[:-1].index_select(1, idx).contiguous()
wa, wr = (0.05, 0.03) if name == "planar_hand" else (0.05, 0.015)
cases = {"abs": dict(u_lo=(nom - wa).contiguous(), u_hi=(nom + wa).contiguous()),
         "rel": dict(du_lo=torch.full((T, m), -wr, dtype=torch.float64, device="cuda"),
                     du_hi=torch.full((T, m), wr, dtype=torch.float64, device="cuda")),
         "none": dict()}
ref = {}
for cname, b in cases.items():
    for solver, label in ((3, "mfma tiles"), (2, "lanes"),) + (((1, "ADMM"),) if "--admm" in sys.argv else ()):
        if not dm.quasistatic_descent_supported(T, solver):
            print("T=%d %-4s %-10s unsupported" % (T, cname, label))
            continue
        kw = dict(solver=solver, rho=100.0, relax=1.6, max_iter=20000 if solver == 1 else 2000, eps=1e-9)
        out = dm.quasistatic_box_descent(At, Bt, ct, Q, Qd, R, xd, x0d, **b, **kw)
        torch.cuda.synchronize()
        reps = 5 if solver == 1 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            dm.quasistatic_box_descent(At, Bt, ct, Q, Qd, R, xd, x0d, **b, **kw, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        u = out["u_new"].cpu().numpy()
        d = np.abs(u - ref.setdefault(cname, u)).max()
        if "--stamps" in sys.argv and solver == 3:
            import ctypes
            lib = _lib.load()
            if hasattr(lib, "irs_cbm_print_stamps"):
                lib.irs_cbm_print_stamps.restype = None
                lib.irs_cbm_print_stamps()
        print("%s T=%d %-4s %-10s %9.3f ms/descent  cost %.6f  info %s  |u - first solver's| %.1e"
              % (name, T, cname, label, dt * 1e3, float(out["cost"].item()), out["info"].cpu().numpy().tolist(), d), flush=True)
