"""Times one bounded descent (bicycle_exact.py problem) on the device."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import irs_mpc_amd as amd
T = 100
p = amd.IrsLqrParameters()
p.Q, p.Qd, p.R = np.diag([5, 5, 3, 0.1, 0.1]), np.diag([50., 50, 30, 1, 1]), np.diag([1, 0.1])
p.x0 = np.zeros(5); p.xd_trj = np.tile(np.array([3.0, 1.0, np.pi / 2, 0, 0]), (T + 1, 1))
p.u_trj_initial = np.tile(np.array([0.1, 0.0]), (T, 1))
p.xbound = [-np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4]), np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4])]
sol = amd.IrsLqrExact(amd.BicycleDynamics(0.1), p); sol.verbose = False
t0 = time.perf_counter(); sol.iterate(10); torch.cuda.synchronize(); el = time.perf_counter() - t0
print("bicycle exact, bounded, T=100: 11 descents in %.3f s (%.1f ms each); ADMM max iters last descent %d" % (el, el / 11 * 1e3, int(sol._last["box_info"][1].item())))
print("costs", np.round(sol.cost_lst, 3))
print("gold ", np.round(np.loadtxt(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/bicycle_easy_exact.csv")), 3))
