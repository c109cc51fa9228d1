#!/usr/bin/env python3
"""How much the inexactness of the default contact solver (50 over-relaxed sweeps + active-set polish) moves
what the optimiser sees: B-hat of the planar hand from the SAME samples through the default functor and
through the exact one (contact_solver="exact"), both estimators, on the benchmark's nominal trajectory.

    python tools/compare_contact_solvers.py [--N 10000]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd  # noqa: E402
from irs_mpc_amd import device as dev  # noqa: E402
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10000)
    ap.add_argument("--T", type=int, default=50)
    a = ap.parse_args()
    sd, se = amd.PlanarHandDynamics(0.1), amd.PlanarHandDynamics(0.1, contact_solver="exact")
    parts = lambda o, l, r: sd.get_x_from_q_dict({"sphere": o, "arm_left": l, "arm_right": r})
    idx = sd.get_u_indices_into_x()
    x0 = parts([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    u = np.tile(x0[idx], (a.T, 1))
    ud = dev.to_dev(u)
    xd, _ = se.dm().rollout_cost(dev.to_dev(x0), ud, dev.to_dev(np.eye(7)), dev.to_dev(np.eye(4)),
                                 dev.to_dev(np.zeros((a.T + 1, 7))))
    unact = [i for i in range(7) if i not in set(idx.tolist())]
    g = torch.Generator(device="cuda").manual_seed(0)
    z = torch.randn((a.T, a.N, 4), generator=g, device="cuda", dtype=torch.float32)
    for std in (0.3, 0.1, 0.03):
        du = (std * z).contiguous()
        for name, mode in (("zero_order_B", SMOOTH_ZERO_ORDER_B), ("first_order", SMOOTH_FIRST_ORDER)):
            Bd = sd.dm().smooth(mode, xd, ud, None, du)["Bt"].cpu().numpy()[:, unact]
            Be = se.dm().smooth(mode, xd, ud, None, du)["Bt"].cpu().numpy()[:, unact]
            print("std %.2f %-13s max |B_default - B_exact| %.2e   (max |B_exact| %.3f, Monte-Carlo std of an entry ~ %.1e)"
                  % (std, name, np.abs(Bd - Be).max(), np.abs(Be).max(), np.abs(Be).max() / np.sqrt(a.N)))


if __name__ == "__main__":
    main()
