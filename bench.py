#!/usr/bin/env python3
"""Headline benchmark: rollouts x timesteps / s of the randomised-smoothing pass
(get_TV_matrices: sample pass + reduction + solve -> A_t,B_t,c_t) and iLQR
iterations / s (that + Riccati + closed-loop rollout + cost), BASELINE.json config 1:
pendulum zero-order, T=30, N=10000 samples per timestep PER GPU, samples resident in
HBM (f32).  One process per GPU; N>1 is launched by torch.distributed.run.

    python bench.py --gpus 1 --steps 200 --warmup 20

A "step" = one smoothing pass over the (T x N) sample grid.  With --gpus N every rank
holds its own N samples per timestep (weak scaling) and the (T,P) f64 statistics are
all-reduced (RCCL) inside every step.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(T, N, seconds=12.0):
    """The oracle (NumPy restatement of irs_lqr_zero_order.py:38-63, same structure
    as the reference: Python loop over t, vectorised dynamics_batch, SVD lstsq) timed
    on ONE host core on the same workload shape.  Reported, never the target."""
    from oracle import irs_oracle as orc
    s = orc.PendulumOracle(0.05)
    u = np.tile(np.array([0.1]), (T, 1))
    x = orc.rollout(s, np.zeros(2), u)
    rng = np.random.default_rng(0)
    dx = rng.normal(size=(T, N, 2)).astype(np.float32).astype(np.float64)
    du = rng.normal(size=(T, N, 1)).astype(np.float32).astype(np.float64)
    orc.zero_order_TV(s, x, u, dx, du)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.zero_order_TV(s, x, u, dx, du)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds or reps >= 2000:
            break
    return {"value": T * N * reps / el, "unit": "rollouts*timesteps/s", "cores": 1, "kind": "port",
            "sample": "%d passes of the same T=%d N=%d workload (oracle.zero_order_TV, supplied samples, "
                      "1 thread, %.1f s)" % (reps, T, N, el),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--N", type=int, default=10000, help="samples per timestep per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also time N=1e3,1e5,1e6 (extra keys)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from irs_mpc_amd import PendulumDynamics, device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_AB as MODE
    from irs_mpc_amd.distributed import all_reduce_sums

    T = args.T
    system = PendulumDynamics(0.05)
    dm = system.dm()
    n, m = dm.n, dm.m
    Q, Qd, R = dev.to_dev(np.diag([1., 1.])), dev.to_dev(np.diag([20., 20.])), dev.to_dev(np.diag([1.]))
    xd = dev.to_dev(np.tile(np.array([np.pi, 0.]), (T + 1, 1)))
    x0 = dev.to_dev(np.zeros(2))
    u_trj = dev.to_dev(np.tile(np.array([0.1]), (T, 1)))
    x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)

    def make_samples(N):
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        dx = torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
        du = torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
        return dx, du

    def smooth_step(dx, du, sums, n_total):
        dm.smooth_accumulate(MODE, x_trj, u_trj, dx, du, sums=sums)
        all_reduce_sums(sums)
        return dm.smooth_finalize(MODE, n_total, x_trj, u_trj, sums)

    def ilqr_step(dx, du, sums, n_total):
        At, Bt, ct, info = smooth_step(dx, du, sums, n_total)
        K, k, _ = dev.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd, alpha_R=0.5)
        return dm.closed_loop_rollout(K, k, x0, Q, R, xd)

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def kernel_time_ms(dx, du, sums, reps):
        """Average duration of the sample pass measured with HIP events on the stream
        the kernels are launched on (torch's current stream is the one passed to the C ABI)."""
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for i in range(reps):
            starts[i].record()
            dm.smooth_accumulate(MODE, x_trj, u_trj, dx, du, sums=sums)
            ends[i].record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
        return float(np.mean(ts)), float(ts[len(ts) // 2])

    def run(N, steps, warmup):
        dx, du = make_samples(N)
        sums = torch.empty((T, dm.sums_len(MODE)), dtype=torch.float64, device="cuda")
        n_total = N * world
        el = timed(lambda: smooth_step(dx, du, sums, n_total), steps, warmup)
        el_it = timed(lambda: ilqr_step(dx, du, sums, n_total), steps, warmup)
        k_mean, k_med = kernel_time_ms(dx, du, sums, min(steps, 200))
        return el, el_it, k_mean, k_med

    N = args.N
    el, el_it, k_mean, k_med = run(N, args.steps, args.warmup)
    bytes_per_sample_step = 4 * (n + m)            # dx,du read once, f32
    alg_bytes = bytes_per_sample_step * N * T      # per launch (per GPU)
    achieved = alg_bytes / (k_mean * 1e-3) / 1e9

    out = {
        "metric": "rollouts*timesteps/s (randomized-smoothing pass) + iLQR-iters/s",
        "value": world * N * T * args.steps / el,
        "unit": "rollouts*timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "pendulum zero-order smoothing (BASELINE configs[1])", "T": T,
                   "N_per_gpu": N, "N_total": N * world, "samples": "supplied, resident in HBM (f32)",
                   "parallelism": "samples sharded over %d GPU(s), 1 all-reduce of (T,P) f64 per step" % world},
        "ilqr_iters_per_s": args.steps / el_it,
        "ms_per_ilqr_iter": 1e3 * el_it / args.steps,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "smooth_accum_kernel (+ reduce_partials)", "alg_bytes_per_launch": alg_bytes,
                     "avg_launch_ms": k_mean, "median_launch_ms": k_med},
    }
    if args.sweep and world == 1:
        sweep = {}
        for Ns in (1000, 100000, 1000000):
            e, ei, km, kmed = run(Ns, max(20, args.steps // 4), 5)
            st = max(20, args.steps // 4)
            sweep[str(Ns)] = {"value": Ns * T * st / e, "ilqr_iters_per_s": st / ei,
                              "kernel_GBps": bytes_per_sample_step * Ns * T / (km * 1e-3) / 1e9}
        out["sweep_N"] = sweep
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(T, N)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
