#!/usr/bin/env python3
"""Runs one of the reference's analytic examples on the GPU:

    python examples/run.py pendulum zero_order            # examples/pendulum/pendulum_zero_order.py
    python examples/run.py quadrotor first_order --device-rng
    python examples/run.py bicycle exact --csv bicycle_easy_exact.csv
    python examples/run.py pendulum cem

method = zero_order | first_order | exact | cem.  The sampling closure is the scripts'
(host NumPy RNG, `--seed`) unless --device-rng draws on the GPU.  Prints the cost history
(the reference saves it with np.savetxt, e.g. examples/quadrotor/quadrotor_cem.py:60-61).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd  # noqa: E402
from examples.problems import PROBLEMS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("system", choices=sorted(PROBLEMS))
    ap.add_argument("method", choices=["zero_order", "first_order", "exact", "cem"])
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--N", type=int, default=None, help="samples per timestep (overrides the script's num_samples)")
    ap.add_argument("--device-rng", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--csv", default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()

    sysd, params, sm, cem, iters = PROBLEMS[a.system](*([a.T] if a.T else []))
    iters = a.iters if a.iters is not None else iters
    N = a.N or sm["N"]
    np.random.seed(a.seed)
    if a.method == "cem":
        cp = amd.CemParameters()
        for k in ("Q", "Qd", "R", "x0", "xd_trj", "u_trj_initial"):
            setattr(cp, k, getattr(params, k))
        cp.initial_std, cp.batch_size, cp.n_elite = cem["initial_std"], cem["batch_size"], cem["n_elite"]
        solver = amd.CrossEntropyMethod(sysd, cp)
    elif a.method == "exact":
        solver = amd.IrsLqrExact(sysd, params)
    else:
        sx, su, pw = np.asarray(sm["std_x"], float), np.asarray(sm["std_u"], float), sm["power"]
        if a.device_rng:
            sampling = amd.GaussianSmoothing(sx, su, N, power=pw, seed=a.seed)
        else:
            def sampling(xbar, ubar, it):       # the scripts' closure, e.g. pendulum_zero_order.py:38-43
                dx = np.random.normal(0.0, sx / (it ** pw), size=(N, len(sx)))
                du = np.random.normal(0.0, su / (it ** pw), size=(N, len(su)))
                return dx, du
        cls = amd.IrsLqrZeroOrder if a.method == "zero_order" else amd.IrsLqrFirstOrder
        solver = cls(sysd, params, sampling)
    solver.verbose = not a.quiet
    t0 = time.time()
    solver.iterate(iters)
    print("Final cost: " + str(solver.cost))
    print("Elapsed time: " + str(time.time() - t0))
    print("cost history:", " ".join("%.6f" % c for c in solver.cost_lst))
    if a.csv:
        np.savetxt(a.csv, np.array(solver.cost_lst))


if __name__ == "__main__":
    main()
