"""BASELINE configs[2]: quadrotor first-order smoothing, T=50, N=10000, + on-device Riccati.
Times the two launches of one iLQR iteration (and the zero-order variant of the sample pass)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irs_mpc_amd import QuadrotorDynamics, device as dev, _lib  # noqa: E402
T, N = 50, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dm = QuadrotorDynamics(0.05).dm(); n, m = dm.n, dm.m
Q = dev.to_dev(np.diag([10., 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0])); Qd = dev.to_dev(10.0 * np.diag([10., 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1])); R = dev.to_dev(np.eye(4))
xd = np.zeros((T + 1, 12))
for i in range(T + 1): xd[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
xd = dev.to_dev(xd); x0 = dev.to_dev(np.zeros(12)); u_trj = dev.to_dev(np.full((T, 4), 2.0))
x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
g = torch.Generator(device="cuda").manual_seed(0)
dx = 0.1 * torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
du = 0.1 * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
res = {}
for name, mode in (("first_order", _lib.SMOOTH_FIRST_ORDER), ("zero_order", _lib.SMOOTH_ZERO_ORDER_AB)):
    plan = dev.SmoothPlan(dm, mode, x_trj, u_trj, dx=dx, du=du)
    plan.run(st)
    des = dev.DescentPlan(dm, plan.out["At"], plan.out["Bt"], plan.out["ct"], Q, Qd, R, xd, x0)
    def tm(fn, reps=300):
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end: fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    ts = tm(lambda: plan.run(st)); td = tm(lambda: des.run(st)); ti = tm(lambda: (plan.run(st), des.run(st)))
    res[name] = dict(smooth_us=ts * 1e6, descent_us=td * 1e6, iteration_us=ti * 1e6, sample_steps_per_s=N * T / ts,
                     smooth_GBps=4 * (n + m) * N * T / ts / 1e9, ilqr_iters_per_s=1 / ti)
print(json.dumps({"workload": "quadrotor T=50 N=%d (BASELINE configs[2])" % N, **res}))
