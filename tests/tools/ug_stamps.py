#!/usr/bin/env python3
"""Phase stamps of the uniform-geometry kernel (a -DIRS_UG_STAMPS build of csrc/smooth_ug.hip, tools/ug_variants.sh):
per workgroup, s_memtime at start / geometry done / table done / sample loop done / statistics ready / end, and the
nominal wave's finish.  python tests/tools/ug_stamps.py [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import irs_mpc_amd as amd  # noqa: E402
from irs_mpc_amd import _lib, device as dev  # noqa: E402
from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B  # noqa: E402
from oracle import irs_oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
HAND = orc.PlanarHandOracle
sys_o = HAND(0.1)
x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
T = 50
u_trj = np.tile(x0[sys_o.indices_u_into_x], (T, 1))
x_trj = orc.rollout(sys_o, x0, u_trj)
xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
g = torch.Generator(device="cuda").manual_seed(1234)
du = 0.3 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
dm = amd.PlanarHandDynamics(0.1).dm()
plan = dev.SmoothPlan(dm, SMOOTH_ZERO_ORDER_B, xd, ud, dx=None, du=du, fuse=True)
for _ in range(20):
    plan.run()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nwg = 250
buf = (ctypes.c_ulonglong * (nwg * 12))()
rc = lib.irs_debug_ug_stamps(buf, nwg)
st = np.array(buf, dtype=np.uint64).reshape(nwg, 12).astype(np.int64)
names = ["start", "geometry done", "table done", "wave 0 loop done", "statistics ready", "end", "", "", "nominal wave done",
         "wave 1 loop done"]
print("rc", rc, " s_memtime ticks since the workgroup's own start (the counters of the 8 XCDs are not aligned); median / min / max")
for k in (1, 2, 3, 4, 5, 8, 9):
    ok = st[:, k] > 0
    v = (st[:, k] - st[:, 0])[ok]
    if len(v):
        print("%-20s median %7d  min %7d  max %7d  (n=%d)" % (names[k], np.median(v), v.min(), v.max(), len(v)))
