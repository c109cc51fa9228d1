import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import irs_mpc_amd as amd
from examples.problems import pendulum, quadrotor
for name, prob, T, N, k in (("pendulum", pendulum, 30, 10000, 99), ("quadrotor", quadrotor, 50, 10000, 3)):
    for bounded in (False, True):
        best = 1e9
        for ep in range(4):
            sysd, p, smp, _, _ = prob(T)
            if not bounded:
                p.xbound = p.ubound = None
            s = amd.GaussianSmoothing(np.asarray(smp["std_x"], float), np.asarray(smp["std_u"], float), N, power=0.5, seed=11)
            cls = amd.IrsLqrFirstOrder if name == "quadrotor" else amd.IrsLqrZeroOrder
            sol = cls(sysd, p, s)
            sol.verbose = False if hasattr(sol, "verbose") else None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sol.iterate(k)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(name, "bounded" if bounded else "unbounded", "%.1f us / iteration, %.0f it/s" % (1e6 * best / (k + 1), (k + 1) / best))
