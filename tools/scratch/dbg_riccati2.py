import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irs_mpc_amd import device as dev
from oracle import irs_oracle as orc
T = 50
s = orc.QuadrotorOracle(0.05)
Q = np.diag([10., 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0]); Qd = 10.0 * np.diag([10., 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1]); R = np.eye(4)
xd = np.zeros((T + 1, 12))
for i in range(T + 1): xd[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
u0 = np.tile(np.array([2.0, 2.0, 2.0, 2.0]), (T, 1))
x = orc.rollout(s, np.zeros(12), u0)
At, Bt, ct = orc.exact_TV(s, x, u0)
K, k, info = dev.tvlqr_riccati(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd)], alpha_R=0.5)
Ko, ko = orc.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd, alpha_R=0.5)
K, k = K.cpu().numpy(), k.cpu().numpy()
print("info", int(info.item()))
for t in (49, 48, 45, 40, 30, 20, 10, 0):
    print("t=%d |dK| %.3e (|K| %.3e) |dk| %.3e (|k| %.3e)" % (t, np.abs(K[t] - Ko[t]).max(), np.abs(Ko[t]).max(), np.abs(k[t] - ko[t]).max(), np.abs(ko[t]).max()))
