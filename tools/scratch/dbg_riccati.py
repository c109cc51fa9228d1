import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irs_mpc_amd import device as dev
from oracle import irs_oracle as orc
rng = np.random.default_rng(0)
for n, m, T in ((5, 2, 4), (12, 4, 4)):
    At = np.eye(n) + 0.1 * rng.normal(size=(T, n, n)); Bt = rng.normal(size=(T, n, m)); ct = 0.1 * rng.normal(size=(T, n))
    Q, Qd, R = np.eye(n) * 2.0, np.eye(n) * 5.0, np.eye(m) * 0.7
    xd = rng.normal(size=(T + 1, n))
    K, k, info = dev.tvlqr_riccati(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd)], alpha_R=0.5)
    Ko, ko = orc.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd, alpha_R=0.5)
    K, k = K.cpu().numpy(), k.cpu().numpy()
    print("n=%d m=%d info=%d" % (n, m, int(info.item())))
    for t in range(T - 1, -1, -1):
        print("  t=%d  |dK| %.3e  |dk| %.3e" % (t, np.abs(K[t] - Ko[t]).max(), np.abs(k[t] - ko[t]).max()))
    if n == 5:
        print(np.round(K[T - 1], 4)); print(np.round(Ko[T - 1], 4))
