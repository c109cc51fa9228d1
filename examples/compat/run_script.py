#!/usr/bin/env python3
"""Runs an UNMODIFIED example script of the reference on the GPU through the import shim of this directory
(README.md).  Usage: run_script.py SCRIPT.py [--csv OUT.csv] [--seed S]"""
import argparse
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def run(script, seed=None, workdir=None):
    """Executes `script` with the shim first on sys.path; returns its module globals.  `workdir`: run there, with
    the `examples/<system>/analysis` directories some scripts np.savetxt into (relative to the reference's root,
    e.g. examples/bicycle/bicycle_cem_easy.py:49) created first."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    script = os.path.abspath(script)
    if workdir is not None:
        for s in ("pendulum", "quadrotor", "bicycle", "three_cart"):
            os.makedirs(os.path.join(workdir, "examples", s, "analysis"), exist_ok=True)
        os.chdir(workdir)
    for p in (ROOT, HERE):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in [m for m in sys.modules if m == "irs_lqr" or m.startswith("irs_lqr.") or m.endswith("_dynamics")]:
        del sys.modules[name]           # a real `irs_lqr` imported earlier must not shadow the shim
    if seed is not None:
        import numpy as np
        np.random.seed(seed)
    return runpy.run_path(script, run_name="__main__")


def cost_history(g):
    """The cost list of whatever solver object the script left in its globals (`solver` in every analytic example)."""
    for v in g.values():
        lst = getattr(v, "cost_lst", None)
        if lst is None:
            lst = getattr(v, "cost_all_list", None)
        if lst is not None and hasattr(v, "iterate"):
            return list(lst)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("script")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--workdir", default=None, help="run the script in this directory (its own np.savetxt targets land there)")
    a = ap.parse_args()
    if a.csv:
        a.csv = os.path.abspath(a.csv)
    g = run(a.script, a.seed, a.workdir)
    costs = cost_history(g)
    if a.csv and costs is not None:
        import numpy as np
        os.makedirs(os.path.dirname(os.path.abspath(a.csv)), exist_ok=True)
        np.savetxt(a.csv, np.array(costs))
    if costs is not None:
        print("cost history:", " ".join("%.6f" % c for c in costs))


if __name__ == "__main__":
    main()
