#!/usr/bin/env python3
"""Headline benchmark: rollouts x timesteps / s of the randomised-smoothing pass
(get_TV_matrices: sample pass + reduction + solve -> A_t,B_t,c_t) and iLQR
iterations / s (that + Riccati + closed-loop rollout + cost), BASELINE.json configs[1]:
pendulum zero-order, T=30, N=10000 samples per timestep PER GPU, samples resident in
HBM (f32).  One process per GPU; N>1 is launched by torch.distributed.run.

    python bench.py --gpus 1 --steps 20000 --warmup 2000

A "step" = one smoothing pass over the (T x N) sample grid = ONE kernel launch on one
GPU.  With --gpus N every rank holds its own N samples per timestep (weak scaling) and
the (T,P) f64 statistics are all-reduced (RCCL) inside every step, followed by the
solve launch.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--N", type=int, default=10000, help="samples per timestep per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="run all ranks on GPU 0 with gloo (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--sweep", action="store_true", help="also time N=1e3,1e5,1e6 (extra keys)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if args.rehearse_one_gpu:       # testing aid: every rank on GPU 0, gloo collectives
        local, args.backend = 0, "gloo"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

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
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, steps, warmup):
        """W untimed steps, then EXACTLY `steps` steps between barrier+synchronize pairs
        (host clock, max over ranks).  Also returns the same region measured by a HIP
        event pair recorded on the launch stream (device clock)."""
        t_end = time.perf_counter() + 0.25        # bring the clocks up before the W warmup steps
        while time.perf_counter() < t_end:
            fn()
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev_ms

    def run(N, steps, warmup):
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        dx = torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
        du = torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
        n_total = N * world
        if world == 1:
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=True)
            tv = plan.out

            def smooth_step():
                plan.run(stream)
        else:
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=False, n_total=n_total)
            tv = {}

            def smooth_step():
                plan.run(stream)
                all_reduce_sums(plan.sums)
                tv["At"], tv["Bt"], tv["ct"], tv["info"] = dm.smooth_finalize(MODE, n_total, x_trj, u_trj, plan.sums)

        smooth_step()
        descent = dev.DescentPlan(dm, tv["At"], tv["Bt"], tv["ct"], Q, Qd, R, xd, x0)

        def ilqr_step():
            smooth_step()
            if world > 1:       # finalize allocated fresh outputs
                c = descent.call
                c.At, c.Bt, c.ct = tv["At"].data_ptr(), tv["Bt"].data_ptr(), tv["ct"].data_ptr()
            descent.run(stream)

        el, ev_ms = timed(smooth_step, steps, warmup)
        el_it, _ = timed(ilqr_step, steps, warmup)
        # Kernel time of the sample pass: HIP events recorded on the stream the kernel is
        # launched on (torch's current stream is the one handed to the C ABI), bracketing
        # `steps` launches of the timed region when the step is a single launch, otherwise
        # a dedicated loop of sample-pass launches.  Per-launch average = elapsed / launches
        # (back-to-back launches: includes the ~1 us dispatch gap, no per-event overhead).
        if world == 1:
            k_ms = ev_ms / steps
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                plan.run(stream)
            e1.record()
            torch.cuda.synchronize()
            k_ms = e0.elapsed_time(e1) / steps
        return el, el_it, k_ms

    N = args.N
    el, el_it, k_mean = run(N, args.steps, args.warmup)
    bytes_per_sample_step = 4 * (n + m)            # dx,du read once, f32
    alg_bytes = bytes_per_sample_step * N * T      # per launch (per GPU)
    achieved = alg_bytes / (k_mean * 1e-3) / 1e9

    # HBM traffic per launch from the PMC passes of tools/profile_round.sh (separate
    # rocprofv3 --pmc runs of this same command; FETCH_SIZE doubled as the gfx950 note in
    # MI355X_MICROARCH.md prescribes, and it then matches the known byte count).
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path)).get("pendulum_zero_T%d_N%d" % (T, N))
        if pmc and "FETCH_SIZE_raw_avg" in pmc:
            traffic = (2.0 * pmc["FETCH_SIZE_raw_avg"] + pmc.get("WRITE_SIZE_raw_avg", 0.0)) * 1024.0
            traffic_src = "profiles/pmc_latest.json"

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
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "smooth_kernel<PendulumModel, ZERO_ORDER_AB> (sample pass + reduction + solve, "
                               "one launch)",
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k_mean,
                     "timing": "HIP event pair on the launch stream around the timed launches / launches"},
    }
    if args.sweep and world == 1:
        sweep = {}
        for Ns in (1000, 100000, 1000000):
            st = max(20, args.steps // (4 if Ns <= 100000 else 16))
            e, ei, km = run(Ns, st, 5)
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
