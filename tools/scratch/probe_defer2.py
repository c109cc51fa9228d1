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
for lo, hi in ((0, 64), (64, 128), (0, 66), (0, 128), (0, 192), (0, 256), (0, 257), (0, 512), (100, 164), (128, 192)):
    du = duall[lo:hi].copy(); N = hi - lo
    Xn, Bs, mask = sys_d.dm().contact_samples_f32(dev.to_dev(x), dev.to_dev(u), dev.to_dev(du, dev.F32))
    Bs = Bs.cpu().numpy().astype(float)
    o1 = sys_d.dm().smooth(SMOOTH_FIRST_ORDER, dev.to_dev(np.stack([x, x])), dev.to_dev(u[None]), None, dev.to_dev(du[None], dev.F32))
    e1 = np.abs(o1["Bt"].cpu().numpy()[0][free] - Bs.mean(0)[free]).max()
    print("samples [%d,%d) first-order |smooth - mean(lanes)| %.2e" % (lo, hi, e1), flush=True)
