#!/usr/bin/env python3
"""Launch time of the planar-hand sample pass over N, uniform-geometry kernel vs the general one."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import irs_mpc_amd as amd  # noqa: E402
from irs_mpc_amd import device as dev  # noqa: E402
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B  # noqa: E402
from oracle import irs_oracle as orc  # noqa: E402

HAND = orc.PlanarHandOracle


def timeit(mode, xd, ud, du, ug, reps=100):
    os.environ["IRS_UG"] = "1" if ug else "0"
    dm = amd.PlanarHandDynamics(0.1).dm()
    plan = dev.SmoothPlan(dm, mode, xd, ud, dx=None, du=du, fuse=True)
    for _ in range(10):
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
    T = 50
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    modes = ((SMOOTH_ZERO_ORDER_B, "zeroB"),) if "--zero" in sys.argv else ((SMOOTH_ZERO_ORDER_B, "zeroB"), (SMOOTH_FIRST_ORDER, "first"))
    both = "--both" in sys.argv
    diags = [int(v[7:]) for v in sys.argv if v.startswith("--diag=")] or [0]
    for N in [int(v) for v in sys.argv[1:] if v.isdigit()] or [64, 1000, 10000, 100000]:
        g = torch.Generator(device="cuda").manual_seed(1234)
        du = 0.3 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
        for mode, name, dg in [(m_, n_ + (" diag %d" % d_ if d_ else ""), d_) for m_, n_ in modes for d_ in diags]:
            os.environ["IRS_DIAG"] = str(dg)
            t_ug = timeit(mode, xd, ud, du, True)
            s = "N=%6d %s: ug %.1f us (%.3e /s)" % (N, name, t_ug, N * T / (t_ug * 1e-6))
            if both:
                s += "  general %.1f us" % timeit(mode, xd, ud, du, False)
            if "--verify" in sys.argv:
                outs = []
                for ug in (True, False):
                    os.environ["IRS_UG"] = "1" if ug else "0"
                    o = amd.PlanarHandDynamics(0.1).dm().smooth(mode, xd, ud, None, du)
                    outs.append((o["Bt"].cpu().numpy(), o["ct"].cpu().numpy(), int(o["info"].abs().sum().item())))
                s += "  |dB| %.1e |dc| %.1e info %d" % (np.abs(outs[0][0] - outs[1][0]).max(), np.abs(outs[0][1] - outs[1][1]).max(), outs[0][2])
            print(s, flush=True)


if __name__ == "__main__":
    main()
