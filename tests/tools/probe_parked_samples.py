"""Diagnostic: the sample pass of the exact contact models (parked samples, smooth.hip DEFER) against the per-sample
lanes of irs_contact_samples_f32 at several N -- run on the GPU box:  gpurun -- python tests/tools/probe_parked_samples.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import irs_mpc_amd as amd
from irs_mpc_amd import device as dev
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
from oracle import irs_oracle as orc
for system in ("box_pivoting", "planar_hand"):
    if system == "planar_hand":
        sys_d, sys_o = amd.PlanarHandDynamics(0.1), orc.PlanarHandOracle(0.1)
        x = orc.PlanarHandOracle.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    else:
        sys_d, sys_o = amd.BoxPivotingDynamics(0.1), orc.BoxPivotOracle(0.1)
        x = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
    idx = sys_o.indices_u_into_x; u = x[idx].copy(); n, m = sys_o.dim_x, sys_o.dim_u
    free = np.setdiff1d(np.arange(n), idx)
    for N in (50, 64, 130, 700, 5000, 20000):
        du = (0.05 * np.random.default_rng(5).normal(size=(N, m))).astype(np.float32)
        Xn, Bs, mask = sys_d.dm().contact_samples_f32(dev.to_dev(x), dev.to_dev(u), dev.to_dev(du, dev.F32))
        Xn, Bs = Xn.cpu().numpy().astype(float), Bs.cpu().numpy().astype(float)
        o1 = sys_d.dm().smooth(SMOOTH_FIRST_ORDER, dev.to_dev(np.stack([x, x])), dev.to_dev(u[None]), None, dev.to_dev(du[None], dev.F32))
        e1 = np.abs(o1["Bt"].cpu().numpy()[0][free] - Bs.mean(0)[free]).max()
        o2 = sys_d.dm().smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(np.stack([x, x])), dev.to_dev(u[None]), None, dev.to_dev(du[None], dev.F32))
        # zero-order B from the per-sample steps
        f0 = sys_o.dynamics(x.astype(np.float32).astype(float), u.astype(np.float32).astype(float))
        Z = np.hstack([du.astype(float), np.ones((N, 1))])
        # estimator: least squares of (f - xbar) on du with mean handling as the kernel: B = (sum z z')^-1 sum z (f - f0)'
        G = du.astype(float).T @ du.astype(float); H = du.astype(float).T @ (Xn - f0[None])
        B2 = np.linalg.solve(G, H).T
        e2 = np.abs(o2["Bt"].cpu().numpy()[0][free] - B2[free]).max()
        print("%s N=%5d first-order |smooth - mean(lanes)| %.2e   zero-order-B |smooth - lstsq(lanes)| %.2e" % (system, N, e1, e2), flush=True)
