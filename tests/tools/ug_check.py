#!/usr/bin/env python3
"""Uniform-geometry sample pass (csrc/smooth_ug.hip) against the general kernel (IRS_UG=0) and the oracle, and
timed: python tests/tools/ug_check.py [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import irs_mpc_amd as amd  # noqa: E402
from irs_mpc_amd import device as dev  # noqa: E402
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B  # noqa: E402
from oracle import irs_oracle as orc  # noqa: E402

HAND = orc.PlanarHandOracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000


def run(mode, xd, ud, du, ug):
    os.environ["IRS_UG"] = "1" if ug else "0"
    dm = amd.PlanarHandDynamics(0.1).dm()
    o = dm.smooth(mode, xd, ud, None, du)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}


def timeit(mode, xd, ud, du, ug, reps=200):
    os.environ["IRS_UG"] = "1" if ug else "0"
    dm = amd.PlanarHandDynamics(0.1).dm()
    plan = dev.SmoothPlan(dm, mode, xd, ud, dx=None, du=du, fuse=True)
    for _ in range(20):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    sys_o = HAND(0.1)
    x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    idx = sys_o.indices_u_into_x
    # (1) small problem against the oracle and the general kernel
    T, Ns = 6, 3000
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    rng = np.random.default_rng(3)
    for std in (0.3, 0.05):
        du = (std * rng.normal(size=(T, Ns, 4))).astype(np.float32)
        xd, ud, dud = dev.to_dev(x_trj), dev.to_dev(u_trj), dev.to_dev(du, dev.F32)
        Ao, Bo, co = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
        for ug in (False, True):
            o = run(SMOOTH_ZERO_ORDER_B, xd, ud, dud, ug)
            print("zero-order-B std %.2f ug=%d  info %s  |B-Bo| %.2e  |c-co| %.2e" % (
                std, ug, o["info"].tolist(), np.abs(o["Bt"] - Bo).max(), np.abs(o["ct"] - co).max()))
        A1, B1, c1 = orc.first_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
        for ug in (False, True):
            o = run(SMOOTH_FIRST_ORDER, xd, ud, dud, ug)
            print("first-order   std %.2f ug=%d  info %s  |B-Bo| %.2e  |c-co| %.2e" % (
                std, ug, o["info"].tolist(), np.abs(o["Bt"] - B1).max(), np.abs(o["ct"] - c1).max()))
    # (2) the benchmark's size: both kernels against each other, and timed
    T = 50
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    g = torch.Generator(device="cuda").manual_seed(1234)
    du = 0.3 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    for mode, name in ((SMOOTH_ZERO_ORDER_B, "zero-order-B"), (SMOOTH_FIRST_ORDER, "first-order")):
        a, b = run(mode, xd, ud, du, False), run(mode, xd, ud, du, True)
        b2 = run(mode, xd, ud, du, True)
        print("%s T=50 N=%d: |B_ug - B_old| %.2e  |c_ug - c_old| %.2e  info %d/%d  rerun bit-equal %s" % (
            name, N, np.abs(a["Bt"] - b["Bt"]).max(), np.abs(a["ct"] - b["ct"]).max(), int(np.abs(a["info"]).sum()),
            int(np.abs(b["info"]).sum()), bool((b["Bt"] == b2["Bt"]).all() and (b["ct"] == b2["ct"]).all())))
        t_old, t_ug = timeit(mode, xd, ud, du, False), timeit(mode, xd, ud, du, True)
        print("   time per launch: general %.1f us, uniform-geometry %.1f us  (%.2fx)  -> %.3e rollouts*timesteps/s" % (
            t_old, t_ug, t_old / t_ug, N * T / (t_ug * 1e-6)))


if __name__ == "__main__":
    main()
