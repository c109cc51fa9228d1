"""Prints the actual device-vs-oracle errors of the contact sample pass (the quantities the parity tests and
smoke() bound), so that their tolerances can be set from measurements:   python tests/tools/contact_tolerance_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import irs_mpc_amd as amd  # noqa: E402
from irs_mpc_amd import device as dev  # noqa: E402
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B  # noqa: E402
from oracle import irs_oracle as orc  # noqa: E402

HAND = orc.PlanarHandOracle
IDX = np.array([1, 4, 2, 5])


def err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max())


def hand_case(solver, T, N, std, seed, settle=4, wiggle=True):
    sys_d = amd.PlanarHandDynamics(0.1, contact_solver=solver)
    sys_o = HAND(0.1, pgs_iters=0 if solver == "exact" else 50)
    x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    for _ in range(settle):
        x0 = sys_o.dynamics(x0, np.array([-np.pi / 4, -np.pi / 4, np.pi / 4, np.pi / 4]))
    u_trj = np.tile(x0[IDX], (T, 1))
    if wiggle:
        u_trj = u_trj + 0.02 * np.sin(np.arange(T))[:, None] * np.array([1, -1, -1, 1])
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (std * np.random.default_rng(seed).normal(size=(T, N, 4))).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    o0 = dm.smooth(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    _, B0, c0 = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    o1 = dm.smooth(SMOOTH_FIRST_ORDER, xd, ud, None, dev.to_dev(du, dev.F32))
    _, B1, c1 = orc.first_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    print("hand %-5s T=%d N=%d std=%.2f: zero-order-B  dB %.2e dc %.2e | first-order dB %.2e dc %.2e" % (
        solver, T, N, std, err(o0["Bt"].cpu().numpy(), B0), err(o0["ct"].cpu().numpy(), c0),
        err(o1["Bt"].cpu().numpy(), B1), err(o1["ct"].cpu().numpy(), c1)), flush=True)


def box_case(solver, T, N, std, seed):
    BOX = orc.BoxPivotOracle
    sys_o = BOX(0.1, pgs_iters=0 if solver == "exact" else 50)
    sys_d = amd.BoxPivotingDynamics(0.1, contact_solver=solver)
    x0 = BOX.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
    u_trj = np.tile(x0[sys_o.indices_u_into_x], (T, 1)) + np.linspace(0, 1, T)[:, None] * np.array([0.1, 0.0])
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (std * np.random.default_rng(seed).normal(size=(T, N, 2))).astype(np.float32)
    dm = sys_d.dm()
    o0 = dm.smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    _, B0, c0 = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    o1 = dm.smooth(SMOOTH_FIRST_ORDER, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    _, B1, c1 = orc.first_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    print("box  %-5s T=%d N=%d std=%.2f: zero-order-B  dB %.2e dc %.2e | first-order dB %.2e dc %.2e" % (
        solver, T, N, std, err(o0["Bt"].cpu().numpy(), B0), err(o0["ct"].cpu().numpy(), c0),
        err(o1["Bt"].cpu().numpy(), B1), err(o1["ct"].cpu().numpy(), c1)), flush=True)


if __name__ == "__main__":
    for solver in ("exact", "pgs"):
        hand_case(solver, 6, 2000, 0.1, 12)
        hand_case(solver, 6, 3000, 0.1, 13)
        hand_case(solver, 6, 2000, 0.3, 31, settle=0, wiggle=False)
        hand_case(solver, 8, 256, 0.1, 0, settle=0, wiggle=False)          # smoke()'s case
        box_case(solver, 4, 1500, 0.05, 77)
        box_case(solver, 6, 1500, 0.02, 5)
