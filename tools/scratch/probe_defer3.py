import sys, numpy as np, torch
sys.path.insert(0, '.')
import irs_mpc_amd as amd
from irs_mpc_amd import device as dev
from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER
from oracle import irs_oracle as orc
sys_d, sys_o = amd.BoxPivotingDynamics(0.1), orc.BoxPivotOracle(0.1)
x = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
idx = sys_o.indices_u_into_x; u = x[idx].copy(); n, m = sys_o.dim_x, sys_o.dim_u
free = np.setdiff1d(np.arange(n), idx)
duall = (0.05 * np.random.default_rng(5).normal(size=(4096, m))).astype(np.float32)
du = duall[64:128].copy()
Xn, Bs, mask = sys_d.dm().contact_samples_f32(dev.to_dev(x), dev.to_dev(u), dev.to_dev(du, dev.F32))
Bs = Bs.cpu().numpy().astype(float); mask = mask.cpu().numpy()
for i in range(64):
    o1 = sys_d.dm().smooth(SMOOTH_FIRST_ORDER, dev.to_dev(np.stack([x, x])), dev.to_dev(u[None]), None, dev.to_dev(du[None, i:i+1], dev.F32))
    B1 = o1["Bt"].cpu().numpy()[0]
    e = np.abs(B1[free] - Bs[i][free]).max()
    print("i=%2d du=(%.9g,%.9g) mask=%03x err=%.2e %s" % (i, du[i,0], du[i,1], mask[i], e, "BAD" if e > 1e-4 else ""), flush=True)
    if e > 1e-4:
        print("   smooth", B1[free].ravel()); print("   lanes ", Bs[i][free].ravel())
