#!/usr/bin/env python3
"""Wall time per iteration of IrsLqrQuasistatic.iterate (the host twin's loop, bookkeeping included) on the
planar-hand problem of examples/run_quasistatic.py, after a warm-up solve; optional cProfile listing.

    python tools/time_host_loop.py [--profile] [--gradient-mode first_order] [--N 10000]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd  # noqa: E402
from examples.run_quasistatic import problem  # noqa: E402


def make(a):
    h = 0.1
    q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, x_trj_d = problem(a.T, h)
    p = amd.IrsLqrQuasistaticParameters()
    p.Q_dict, p.Qd_dict, p.R_dict = Q_dict, Qd_dict, R_dict
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, x_trj_d, u_traj_0, a.T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.5 * h, np.ones(4) * 0.5 * h])
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)
    p.std_u_initial = np.ones(4) * 0.3
    p.num_samples = a.N
    p.gradient_mode = a.gradient_mode
    p.publish_every_iteration = False
    p.device_rng_seed = 0
    s = amd.IrsLqrQuasistatic(q_dynamics, p)
    s.verbose = False
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--N", type=int, default=10000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--gradient-mode", default="zero_order_B")
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    make(a).iterate(2)                      # library load, first launches
    s = make(a)
    pr = cProfile.Profile() if a.profile else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    s.iterate(a.iters)
    if pr:
        pr.disable()
    el = time.perf_counter() - t0
    print("%d descents in %.3f s = %.2f ms per iteration (%.1f iterations/s); best cost %.4f"
          % (a.iters + 1, el, 1e3 * el / (a.iters + 1), (a.iters + 1) / el, s.cost_best))
    if pr:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
