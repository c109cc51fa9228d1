"""Per-iteration cost of the bounded descent inside the bench's 20-iteration planar-hand (or box-pivoting) loop.

    gpurun -- python tools/descent_loop_profile.py [planar_hand|box_pivoting] [--stamps] [--dump=K ...]

Replays bench.py's episode (fresh device-drawn samples, bounds re-centred, first tail's active set handed on) with a
synchronisation after every descent: ms per descent, info, cost; with the diagnostic library of tools/stamp_descent.sh
(IRS_HIP_LIB=gpurun_out/libirs_hip_stamps.so, --stamps) also the solver wave's phase totals of every descent."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from irs_mpc_amd import _lib, device as dev  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "planar_hand"
    stamps = "--stamps" in sys.argv
    dump = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--dump=")]      # iterations whose inputs to save
    w = bench.Workload(name)
    N = 10000 if name == "planar_hand" else 6250
    dm = w.system.dm()
    m, T = dm.m, w.T
    Q, Qd, R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
    xd, x0, u_trj = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
    x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
    idx_t = torch.as_tensor(w.idx, device=x_trj.device)
    kind, wd = w.bounds

    def bound_rows(x):
        if kind == "abs":
            nom = x[:-1].index_select(1, idx_t)
            return dict(u_lo=(nom - wd).contiguous(), u_hi=(nom + wd).contiguous())
        return dict(du_lo=torch.full((T, m), -wd, dtype=dev.F64, device=x.device),
                    du_hi=torch.full((T, m), wd, dtype=dev.F64, device=x.device))

    lib = _lib.load()
    if stamps:
        lib.irs_cbm_print_stamps.restype = None
    rngd = dict(N=N, std_x=None, std_u=[w.std_u] * m, seed=4321, iter=1)
    xs = [x_trj.clone(), torch.empty_like(x_trj)]
    us = [u_trj.clone(), torch.empty_like(u_trj)]
    plan = dev.SmoothPlan(dm, w.mode, xs[0], us[0], rng=rngd, fuse=True)
    act = torch.zeros((T, m), dtype=dev.F64, device=x_trj.device)
    louts = [None, None]
    for ep in range(2):
        xs[0].copy_(x_trj)
        us[0].copy_(u_trj)
        act.zero_()
        tot = 0.0
        for it in range(1, 21):
            a_, b_ = (it - 1) % 2, it % 2
            plan.set_iter(it, None, [w.std_schedule(it)] * m)
            plan.set_trajectory(xs[a_], us[a_])
            plan.run()
            br = bound_rows(xs[a_])
            torch.cuda.synchronize()
            if ep == 1 and it in dump:
                import numpy as np
                np.savez(os.path.join("gpurun_out", "descent_inputs_%s_it%d.npz" % (name, it)),
                         At=plan.out["At"].cpu().numpy(), Bt=plan.out["Bt"].cpu().numpy(), ct=plan.out["ct"].cpu().numpy(),
                         Q=w.Q, Qd=w.Qd, R=w.R, xd=w.xd, x0=w.x0, act=act.cpu().numpy(), kind=kind,
                         **{k_: v_.cpu().numpy() for k_, v_ in br.items()})
            t0 = time.perf_counter()
            louts[b_] = dm.quasistatic_box_descent(plan.out["At"], plan.out["Bt"], plan.out["ct"], Q, Qd, R, xd, x0,
                                                   solver=0, max_iter=2000, eps=1e-9, out=louts[b_], act=act, **br)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
            tot += ms
            xs[b_], us[b_] = louts[b_]["x_new"], louts[b_]["u_new"]
            if ep == 1:
                u = us[b_]
                lo = br.get("u_lo"); hi = br.get("u_hi")
                if lo is not None:
                    nact = int(((u <= lo + 1e-9) | (u >= hi - 1e-9)).sum().item())
                else:
                    nact = -1
                print("it %2d  %.3f ms  cost %.4f  info %s  controls at a bound (realised) %d / %d" % (
                    it, ms, float(louts[b_]["cost"].item()), louts[b_]["info"].cpu().numpy().tolist(), nact, T * m),
                    flush=True)
                if stamps:
                    sys.stderr.flush()
                    lib.irs_cbm_print_stamps()
        if ep == 1:
            print("sum of descents: %.2f ms over 20 iterations" % tot)


if __name__ == "__main__":
    main()
