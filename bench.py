#!/usr/bin/env python3
"""Headline benchmark: rollouts x timesteps / s of the randomised-smoothing pass
(get_TV_matrices: sample pass + reduction + solve -> A_t,B_t,c_t) and iLQR iterations / s
(that + TV-LQR backward pass + closed-loop rollout + cost).

Default workload = the configuration BASELINE.json's metric is quoted on: planar_hand
(quasi-dynamic contact, irs_lqr_quasistatic), T=50, N=10000 u-perturbations per timestep PER
GPU, zero-order-B smoothing, samples resident in HBM (f32).  `--workload pendulum` runs
BASELINE configs[1] (pendulum zero-order AB, T=30, N=10000), the HBM-bound case; the default run
reports it too under "pendulum" so both kernels are tracked from one JSON line.

    python bench.py --gpus 1 --steps 2000 --warmup 200

A "step" = one smoothing pass over the (T x N) sample grid = ONE kernel launch on one GPU.  With
--gpus N every rank holds its own N samples per timestep (weak scaling) and the (T,P) f64
statistics are all-reduced (RCCL) inside every step, followed by the solve launch -- the three
launches replayed as one HIP graph.  One process per GPU; N>1 is launched by
torch.distributed.run.  Prints ONE JSON line on rank 0 (stdout carries nothing else).

iLQR iterations / s of the planar hand are measured on the optimisation run_planar_hand.py performs
("ilqr_loop": 20 iterations per episode, every one re-linearised around the previous result with fresh
device-drawn samples, trust region re-centred, first tail warm-started from the previous descent);
"ilqr_first_iter_per_s" is the first iteration alone, repeated from a cold start.  The default run adds
sub-reports: "first_order" (the reference's planar-hand gradient mode), "exact_contact_solver" (the step
QP solved exactly), "pendulum" (configs[1]) and "cpu_baseline" (the oracle on one core and on a pool of
16 worker processes, timed before the GPU is touched).  --mode first_order / --contact-solver exact make
those the timed workload; --force-unfused times the multi-GPU step with a 1-rank RCCL group on one GPU.
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

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_F32_PEAK_TFLOPS = 157.3  # peak FP32 (vector), same guide


class Workload:
    """Everything that differs between the benchmarked configurations."""

    def __init__(self, name, T=None, mode=None, host_only=False, contact_solver="pgs"):
        self.name = name
        self.contact_solver = contact_solver
        if host_only:           # a CPU-baseline worker process: only what the oracle needs, no GPU library
            self._host_only(T, mode)
            return
        from irs_mpc_amd import PendulumDynamics, PlanarHandDynamics
        from irs_mpc_amd import _lib
        if name == "pendulum":
            self.T = T or 30
            self.system = PendulumDynamics(0.05)
            self.mode, self.mode_name = _lib.SMOOTH_ZERO_ORDER_AB, "ZERO_ORDER_AB"
            self.x0 = np.zeros(2)
            self.u_trj = np.tile(np.array([0.1]), (self.T, 1))
            self.Q, self.Qd, self.R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
            self.xd = np.tile(np.array([np.pi, 0.]), (self.T + 1, 1))
            self.std_x, self.std_u = 1.0, 1.0
            self.label = "pendulum zero-order smoothing (BASELINE configs[1])"
            self.kernel = "smooth_kernel<PendulumModel, ZERO_ORDER_AB>"
            self.flops_per_sample = None
        elif name == "planar_hand":
            self.T = T or 50
            self.system = PlanarHandDynamics(0.1, contact_solver=contact_solver)
            self.mode, self.mode_name = _lib.SMOOTH_ZERO_ORDER_B, "ZERO_ORDER_B"
            # examples/planar_hand/run_planar_hand.py:31-44 (initial grasp), :113-131 (costs, goal)
            sd = self.system
            parts = lambda obj, arm_l, arm_r: sd.get_x_from_q_dict({"sphere": obj, "arm_left": arm_l, "arm_right": arm_r})
            self.idx = sd.get_u_indices_into_x()
            self.x0 = parts([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
            self.u_trj = np.tile(self.x0[self.idx], (self.T, 1))
            q = parts([1e-3, 1e-3, 10.0], [1e-3, 1e-3], [1e-3, 1e-3])
            self.Q, self.Qd, self.R = np.diag(q), np.diag(100 * q), 5.0 * np.eye(4)
            self.xd = np.tile(self.x0 + parts([0.3, -0.1, 0.5], [0, 0], [0, 0]), (self.T + 1, 1))
            self.std_x, self.std_u = 0.0, 0.3          # run_planar_hand.py:146
            self.label = "planar_hand quasi-dynamic contact, zero-order-B smoothing (the metric's config)"
            self.kernel = "smooth_kernel<PlanarHandModel, ZERO_ORDER_B>"
            # per one-step evaluation of the contact QP (csrc/contact_models.hpp):
            # QP assembly ~1.5 kFLOP + 152 FLOP per projected sweep + the active-set polish (masked LDL' 196,
            # one solve 120, W dl 128, tests ~26) + the Gram update
            self.flops_per_sample = 1500 + 152 * int(self.system.pgs_iters) + 470 + 76
            if mode == "first_order":
                # gradient_mode "first_order" (examples/planar_hand/planar_hand_setup.py:28): every sample's
                # step is differentiated through its active constraints inside the sample pass -- masked
                # LDL' of the 8x8 dual Hessian (196 FLOP), 4 right-hand sides (480), J'Y (448), B (56), sum (28)
                self.mode, self.mode_name = _lib.SMOOTH_FIRST_ORDER, "FIRST_ORDER"
                self.label = "planar_hand quasi-dynamic contact, first-order smoothing (per-sample active-set derivative)"
                self.kernel = "smooth_kernel<PlanarHandModel, FIRST_ORDER>"
                self.flops_per_sample = 1500 + 152 * int(self.system.pgs_iters) + 470 + 1208
            if contact_solver == "exact":
                # the dual active-set solve takes a data-dependent number of steps: no fixed flop count
                self.label += " [step QP solved exactly: dual active-set method]"
                self.kernel = self.kernel.replace("PlanarHandModel", "PlanarHandExactModel")
                self.flops_per_sample = None
        else:
            raise ValueError(name)

    def _host_only(self, T, mode):
        from oracle import irs_oracle as orc
        if self.name == "pendulum":
            self.T = T or 30
            self.x0, self.u_trj = np.zeros(2), np.tile(np.array([0.1]), (self.T, 1))
            self.std_u, self.mode_name = 1.0, "ZERO_ORDER_AB"
        else:
            self.T = T or 50
            self.x0 = orc.PlanarHandOracle.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
            self.u_trj = np.tile(self.x0[orc.PlanarHandOracle.PERM[3:]], (self.T, 1))
            self.std_u, self.mode_name = 0.3, "FIRST_ORDER" if mode == "first_order" else "ZERO_ORDER_B"

    def bytes_per_sample(self, n, m):
        """the f32 perturbations are read once."""
        return 4 * (m if self.name == "planar_hand" else n + m)

    def oracle(self):
        from oracle import irs_oracle as orc
        return orc, (orc.PendulumOracle(0.05) if self.name == "pendulum" else orc.PlanarHandOracle(0.1))


def _cpu_problem(w, N):
    """The bounded CPU sample of workload `w`: (one full pass as a callable over a range of timesteps,
    samples per timestep, name of the oracle function)."""
    orc, s = w.oracle()
    T = w.T
    rng = np.random.default_rng(0)
    x = orc.rollout(s, w.x0, w.u_trj)
    if w.name == "pendulum":
        Ns = N
        dx = rng.normal(size=(T, Ns, 2)).astype(np.float32).astype(np.float64)
        du = rng.normal(size=(T, Ns, 1)).astype(np.float32).astype(np.float64)

        def part(t0, t1):
            orc.zero_order_TV(s, x[t0:t1 + 1], w.u_trj[t0:t1], dx[t0:t1], du[t0:t1])
        what = "oracle.zero_order_TV"
    else:
        Ns = min(N, 2000)       # the vectorised PGS loop costs ~ms per 1000 samples per timestep
        du = (w.std_u * rng.normal(size=(T, Ns, 4))).astype(np.float32).astype(np.float64)
        if w.mode_name == "FIRST_ORDER":
            def part(t0, t1):
                orc.first_order_B_decoupled(s, x[t0:t1 + 1], w.u_trj[t0:t1], du[t0:t1])
            what = "oracle.first_order_B_decoupled"
        else:
            def part(t0, t1):
                orc.zero_order_B_decoupled(s, x[t0:t1 + 1], w.u_trj[t0:t1], du[t0:t1])
            what = "oracle.zero_order_B_decoupled"
    return part, Ns, what


def _pool_worker(args):
    """One worker of the pooled CPU baseline: its share of the timesteps, `reps` times, one thread."""
    name, T, mode, N, t0, t1, reps = args
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        import threadpoolctl
        threadpoolctl.threadpool_limits(1)
    except Exception:       # noqa: BLE001
        pass
    part, _, _ = _cpu_problem(Workload(name, T, mode, host_only=True), N)
    t_start = time.perf_counter()
    for _ in range(reps):
        part(t0, t1)
    return time.perf_counter() - t_start


def cpu_baseline(w, N, seconds=12.0, pool_cores=16):
    """The oracle (NumPy restatement with the reference's structure: Python loop over t, vectorised
    dynamics_batch, SVD lstsq) timed on the GPU box's host on a bounded sample of the same workload:
    on ONE core, and -- the honest analogue of the reference's 18-30 ZMQ worker processes, which split
    the timesteps among themselves (irs_lqr_quasistatic.py:245-263) -- on a pool of `pool_cores`
    single-threaded processes (the one-GPU box's CPU share).  Reported, never the target."""
    import multiprocessing as mp
    part, Ns, what = _cpu_problem(w, N)
    T = w.T

    def once():
        part(0, T)
    try:
        import threadpoolctl
        limit = threadpoolctl.threadpool_limits(1)      # "cores": 1 means one BLAS thread too
    except Exception:       # noqa: BLE001
        limit = None
    once()
    reps, t0 = 0, time.perf_counter()
    while True:
        once()
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds or reps >= 2000:
            break
    if limit is not None:
        limit.restore_original_limits()
    out = {"value": T * Ns * reps / el, "unit": "rollouts*timesteps/s", "cores": 1, "kind": "port",
           "sample": "%d passes of T=%d N=%d of the same workload (%s, supplied samples, 1 thread, %.1f s)"
                     % (reps, T, Ns, what, el),
           "host_cpus": os.cpu_count()}
    # pooled: the timesteps dealt out to `cores` processes, each repeating its share `reps_p` times
    cores = max(1, min(pool_cores, os.cpu_count() or 1, T))
    try:
        reps_p = max(1, int(reps * 6.0 / max(el, 1e-9) * cores))          # ~6-8 s of work per worker
        bounds = [round(i * T / cores) for i in range(cores + 1)]
        jobs = [(w.name, w.T, "first_order" if w.mode_name == "FIRST_ORDER" else None, N, bounds[i], bounds[i + 1], reps_p)
                for i in range(cores) if bounds[i + 1] > bounds[i]]
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(len(jobs)) as pool:
            busy = pool.map_async(_pool_worker, jobs).get(timeout=180)
        wall = time.perf_counter() - t0
        out["pool"] = {"value": T * Ns * reps_p / max(busy), "unit": "rollouts*timesteps/s", "cores": len(jobs),
                       "sample": "%d passes of T=%d N=%d, timesteps dealt out to %d single-threaded processes; "
                                 "slowest worker %.1f s (%.1f s with process start-up)"
                                 % (reps_p, T, Ns, len(jobs), max(busy), wall)}
    except Exception as e:      # noqa: BLE001 -- the pooled figure is an extra; never fail the bench on it
        out["pool"] = {"error": repr(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 2000 (planar_hand) / 20000 (pendulum)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="planar_hand", choices=["planar_hand", "pendulum"])
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--N", type=int, default=10000, help="samples per timestep per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the pendulum (configs[1]) sub-report")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="run all ranks on GPU 0 with gloo (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--sweep", action="store_true", help="also time N=1e3,1e5,1e6 (extra keys)")
    ap.add_argument("--mode", default=None, choices=["first_order"],
                    help="planar_hand: gradient_mode first_order instead of zero_order_B as the timed workload")
    ap.add_argument("--contact-solver", default="pgs", choices=["pgs", "exact"],
                    help="planar_hand: 50 over-relaxed projected sweeps (default) or the exact dual active-set solve")
    ap.add_argument("--no-graph", action="store_true",
                    help="several ranks: issue the step's launches one by one instead of replaying a HIP graph")
    ap.add_argument("--force-unfused", action="store_true",
                    help="one GPU: time the multi-GPU step (accumulate + all-reduce + solve) with a 1-rank RCCL group")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20000 if args.workload == "pendulum" else 2000
    if args.warmup is None:
        args.warmup = max(1, args.steps // 10)

    # stdout carries exactly ONE JSON line: native libraries print there too (RCCL's version banner at
    # communicator creation), so file descriptor 1 is pointed at stderr and the line goes to a private copy
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    # the CPU baseline runs FIRST, before this process initialises the GPU: its pooled leg starts worker
    # processes (fresh interpreters), which a process that already holds the GPU must not do on this pool
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_unfused:
        cpu_base = cpu_baseline(Workload(args.workload, args.T, args.mode, host_only=True), args.N)
    if args.rehearse_one_gpu:       # testing aid: every rank on GPU 0, gloo collectives
        local, args.backend = 0, "gloo"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    elif args.force_unfused:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    unfused = world > 1 or args.force_unfused

    from irs_mpc_amd import device as dev
    from irs_mpc_amd.distributed import all_reduce_sums, capture_step

    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, steps, warmup, prewarm=50):
        """W untimed steps, then EXACTLY `steps` steps between barrier+synchronize pairs
        (host clock, max over ranks).  Also returns the same region measured by a HIP
        event pair recorded on the launch stream (device clock)."""
        # bring the clocks up before the W warmup steps.  Time-based on one GPU; a FIXED count with
        # several ranks (fn contains a collective: every rank must issue the same number of them)
        if world == 1:
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                fn()
        else:
            for _ in range(prewarm):
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

    step_info = {}

    def run(w, N, steps, warmup):
        """Times the smoothing step and the full iLQR iteration of workload `w`."""
        dm = w.system.dm()
        n, m, T, MODE = dm.n, dm.m, w.T, w.mode
        Q, Qd, R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
        xd, x0, u_trj = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
        x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        dx = None
        if w.name == "pendulum":
            dx = w.std_x * torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
        du = w.std_u * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
        n_total = N * world
        if not unfused:
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=True)
            tv = plan.out

            def smooth_step():
                plan.run(stream)
        else:
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=False, n_total=n_total)
            tv = {}

            def launches():
                # sample pass -> (T,P) f64 statistics; ONE all-reduce; the solve (every rank, redundantly)
                plan.run()
                if args.force_unfused:
                    dist.all_reduce(plan.sums)
                else:
                    all_reduce_sums(plan.sums)
                tv["out"] = dm.smooth_finalize(MODE, n_total, x_trj, u_trj, plan.sums, out=tv.get("out"),
                                               workspace=plan.ws)
                tv["At"], tv["Bt"], tv["ct"], tv["info"] = tv["out"]

            launches()
            smooth_step, step_info["graph"] = launches, False
            if not args.no_graph and args.backend == "nccl":
                # the three launches as one HIP graph: one host call per step
                def agree(flag):
                    """min over the ranks: every rank takes the same branch below (a rank that fell back alone
                    would issue a different number of collectives and hang the others)"""
                    if world == 1:
                        return bool(flag)
                    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return bool(t.item())

                torch.cuda.synchronize()
                ref = [tv[k].clone() for k in ("At", "Bt", "ct")]
                replay = None
                try:
                    replay = capture_step(launches)       # warm-up collectives run on every rank; the capture runs none
                except Exception as e:      # noqa: BLE001 -- an RCCL build that cannot be captured
                    step_info["graph_error"] = repr(e)[:200]
                    torch.cuda.synchronize()
                if agree(replay is not None):
                    for k in ("At", "Bt", "ct"):
                        tv[k].zero_()
                    replay()
                    torch.cuda.synchronize()
                    # the replayed step must reproduce the eagerly issued one bit for bit (fixed-order sums)
                    same = all(torch.equal(a, tv[k]) for a, k in zip(ref, ("At", "Bt", "ct")))
                    if agree(same):
                        smooth_step, step_info["graph"] = replay, True
                    else:
                        step_info.setdefault("graph_error", "replayed step differs from the eager one on some rank; "
                                                            "eager step timed")
                elif "graph_error" not in step_info:
                    step_info["graph_error"] = "capture failed on another rank; eager step timed"

        smooth_step()
        if w.name == "planar_hand":
            # IrsLqrQuasistatic.local_descent (irs_lqr_quasistatic.py:286-345) as run_planar_hand.py
            # sets it up: du cost, trust region u_bounds_abs = +-0.5 h around the nominal actuated
            # positions (:138-139), T re-solved tail QPs, contact dynamics in the loop; one launch
            nom = x_trj[:-1].index_select(1, torch.as_tensor(w.idx, device=x_trj.device))
            u_lo, u_hi = (nom - 0.5 * w.system.h).contiguous(), (nom + 0.5 * w.system.h).contiguous()
            qs_out = {}

            def descent_run():
                qs_out["o"] = dm.quasistatic_box_descent(tv["At"], tv["Bt"], tv["ct"], Q, Qd, R, xd, x0,
                                                         u_lo=u_lo, u_hi=u_hi, solver=0, max_iter=2000,
                                                         eps=1e-9, out=qs_out.get("o"))
        else:
            descent = dev.DescentPlan(dm, tv["At"], tv["Bt"], tv["ct"], Q, Qd, R, xd, x0)

            def descent_run():
                descent.run(stream)

        def ilqr_step():
            smooth_step()
            descent_run()

        el, ev_ms = timed(smooth_step, steps, warmup)
        it_steps = max(1, steps // (4 if w.name == "pendulum" else 20))
        el_it, _ = timed(ilqr_step, it_steps, max(1, it_steps // 10))
        el_it *= steps / it_steps
        loop = None
        if w.name == "planar_hand":
            # The optimisation run_planar_hand.py performs: num_iters = 20 iterations
            # (planar_hand_setup.py:36), every one linearised around the previous one's result with FRESH
            # samples of std 0.3 / iter^0.8 (run_planar_hand.py:142-146; drawn on the device), the trust
            # region re-centred on the new nominal, and the first tail's active set handed on from the
            # previous descent (irs_quasistatic_box_descent_ws) -- cold at the start of every episode.
            # Nothing returns to the host inside an episode.
            K_LOOP = 20
            idx_t = torch.as_tensor(w.idx, device=x_trj.device)
            rngd = dict(N=N, std_x=None, std_u=[w.std_u] * m, seed=4321, iter=1)
            xs = [x_trj.clone(), torch.empty_like(x_trj)]
            us = [u_trj.clone(), torch.empty_like(u_trj)]
            if not unfused:
                lplan = dev.SmoothPlan(dm, MODE, xs[0], us[0], rng=rngd, fuse=True)
            else:
                lplan = dev.SmoothPlan(dm, MODE, xs[0], us[0], rng=rngd, fuse=False, n_total=n_total,
                                       sample_offset=rank * N)
            act = torch.zeros((T, m), dtype=dev.F64, device=x_trj.device)
            louts = [None, None]
            ltv = {}
            lcost = []

            def episode():
                xs[0].copy_(x_trj)
                us[0].copy_(u_trj)
                act.zero_()
                del lcost[:]
                for it in range(1, K_LOOP + 1):
                    a_, b_ = (it - 1) % 2, it % 2
                    lplan.set_iter(it, None, [w.std_u / it ** 0.8] * m)
                    lplan.set_trajectory(xs[a_], us[a_])
                    lplan.run()
                    if not unfused:
                        At_, Bt_, ct_ = lplan.out["At"], lplan.out["Bt"], lplan.out["ct"]
                    else:
                        if args.force_unfused:
                            dist.all_reduce(lplan.sums)
                        else:
                            all_reduce_sums(lplan.sums)
                        ltv["out"] = dm.smooth_finalize(MODE, n_total, xs[a_], us[a_], lplan.sums,
                                                        out=ltv.get("out"), workspace=lplan.ws)
                        At_, Bt_, ct_ = ltv["out"][:3]
                    nom_ = xs[a_][:-1].index_select(1, idx_t)
                    louts[b_] = dm.quasistatic_box_descent(At_, Bt_, ct_, Q, Qd, R, xd, x0,
                                                           u_lo=nom_ - 0.5 * w.system.h, u_hi=nom_ + 0.5 * w.system.h,
                                                           solver=0, max_iter=2000, eps=1e-9, out=louts[b_], act=act)
                    xs[b_], us[b_] = louts[b_]["x_new"], louts[b_]["u_new"]
                    lcost.append(louts[b_]["cost"].clone())

            n_ep = max(2, it_steps // K_LOOP)
            el_loop, _ = timed(episode, n_ep, 1, prewarm=1)
            costs = [float(c.item()) for c in lcost]
            qi = louts[K_LOOP % 2]["info"].cpu().numpy()
            # reported, not asserted: a run with more GPUs draws other samples, and a line with a flag is worth
            # more than no line (the first iteration alone, above, is deterministic and IS asserted)
            loop = {"iters_per_s": n_ep * K_LOOP / el_loop, "ms_per_iter": 1e3 * el_loop / (n_ep * K_LOOP),
                    "descends": bool(min(costs) < 0.9 * costs[0]),
                    "qp_converged_last_iteration": bool(qi[0] == 0 and qi[2] == 0),
                    "iterations_per_episode": K_LOOP, "episodes": n_ep,
                    "cost_first_iteration": costs[0], "cost_best": min(costs),
                    "what": "the 20-iteration optimisation of run_planar_hand.py from its initial trajectory: every "
                            "iteration re-linearises around the previous result with fresh device-drawn samples "
                            "(std 0.3/iter^0.8), re-centres the trust region and warm-starts the first tail's active "
                            "set from the previous descent; cold start at the head of each episode"}
        if w.name == "planar_hand":
            qi = qs_out["o"]["info"].cpu().numpy()
            assert qi[0] == 0 and qi[2] == 0, "bounded TV-LQR did not converge: %s" % qi
        # Kernel time of the sample pass: HIP events recorded on the stream the kernel is
        # launched on (torch's current stream is the one handed to the C ABI), bracketing
        # `steps` launches of the timed region when the step is a single launch, otherwise
        # a dedicated loop of sample-pass launches.  Per-launch average = elapsed / launches
        # (back-to-back launches: includes the ~1 us dispatch gap, no per-event overhead).
        if not unfused:
            k_ms = ev_ms / steps
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                plan.run(stream)
            e1.record()
            torch.cuda.synchronize()
            k_ms = e0.elapsed_time(e1) / steps
        info = int(tv["info"].abs().sum().item())
        assert info == 0, "smoothing solve reported a non-SPD Gram matrix"
        return el, el_it, k_ms, (n, m), loop

    def roofline(w, N, k_ms, nm):
        alg_bytes = w.bytes_per_sample(*nm) * N * w.T      # per launch (per GPU)
        hbm = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC passes of tools/profile_round.sh (separate rocprofv3
        # --pmc runs of this same command; FETCH_SIZE doubled as the gfx950 note in
        # MI355X_MICROARCH.md prescribes).
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            tag = {"ZERO_ORDER_AB": "zero", "ZERO_ORDER_B": "zeroB", "FIRST_ORDER": "first"}[w.mode_name]
            key = "%s_%s_T%d_N%d" % (w.name + ("_exact" if getattr(w, "contact_solver", "pgs") == "exact" else ""),
                                     tag, w.T, N)
            pmc = json.load(open(pmc_path)).get(key)
            if pmc and "FETCH_SIZE_raw_avg" in pmc:
                traffic = (2.0 * pmc["FETCH_SIZE_raw_avg"] + pmc.get("WRITE_SIZE_raw_avg", 0.0)) * 1024.0
                traffic_src = "profiles/pmc_latest.json"
        r = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
             "traffic": traffic, "traffic_source": traffic_src,
             "kernel": w.kernel + " (sample pass + reduction + solve, one launch)",
             "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k_ms,
             "timing": "HIP event pair on the launch stream around the timed launches / launches"}
        if w.flops_per_sample:
            # the contact step is arithmetic on registers (PGS sweeps): f32 VALU is what bounds it
            tf = w.flops_per_sample * N * w.T / (k_ms * 1e-3) / 1e12
            r.update({"bound": "valu_f32", "achieved": tf, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                      "frac": tf / VALU_F32_PEAK_TFLOPS, "alg_flops_per_launch": w.flops_per_sample * N * w.T,
                      "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS},
                      "note": "no GEMM shape in a per-sample projected Gauss-Seidel solve: the bound is the f32 "
                              "vector rate, not HBM or MFMA; the HBM view of the same launch is under 'hbm'"})
        return r

    w = Workload(args.workload, args.T, args.mode, contact_solver=args.contact_solver)
    N, T = args.N, w.T
    el, el_it, k_mean, nm, loop = run(w, N, args.steps, args.warmup)
    out = {
        "metric": "rollouts*timesteps/s (randomized-smoothing pass) + iLQR-iters/s",
        "value": world * N * T * args.steps / el,
        "unit": "rollouts*timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": w.label, "T": T, "N_per_gpu": N, "N_total": N * world, "mode": w.mode_name,
                   "samples": "supplied, resident in HBM (f32)",
                   "parallelism": "samples sharded over %d GPU(s), 1 all-reduce of (T,P) f64 per step" % world},
        "ilqr_iters_per_s": loop["iters_per_s"] if loop else args.steps / el_it,
        "ms_per_ilqr_iter": loop["ms_per_iter"] if loop else 1e3 * el_it / args.steps,
        "ilqr_loop": loop,
        "ilqr_first_iter_per_s": args.steps / el_it,
        "ilqr_iters_note": ("ilqr_iters_per_s = the real 20-iteration loop (ilqr_loop); ilqr_first_iter_per_s = the "
                            "first iteration alone, repeated from a cold start (no warm start, supplied samples)"
                            if loop else "one iteration repeated: its cost does not depend on the trajectory"),
        "ilqr_iter": ("smoothing launch + IrsLqrQuasistatic.local_descent (du cost, u_bounds_abs = +-0.5h trust "
                      "region, T re-solved tail QPs by the active-set solver, contact dynamics in the loop)"
                      if w.name == "planar_hand" else
                      "smoothing launch + Riccati + closed-loop rollout + cost (bounds inactive)"),
        "roofline": roofline(w, N, k_mean, nm),
    }
    if unfused:
        out["config"]["step"] = ("accumulate launch + all-reduce + solve launch, replayed as one HIP graph"
                                 if step_info.get("graph") else
                                 "accumulate launch + all-reduce + solve launch, issued one by one")
        if "graph_error" in step_info:
            out["config"]["graph_error"] = step_info["graph_error"]
    if args.sweep and world == 1:
        sweep = {}
        for Ns in (1000, 100000) + ((1000000,) if w.name == "pendulum" else ()):
            st = max(20, args.steps // (4 if Ns <= 100000 else 16))
            e, ei, km, _, _ = run(w, Ns, st, 5)
            sweep[str(Ns)] = {"value": Ns * T * st / e, "ilqr_iters_per_s": st / ei,
                              "kernel_GBps": w.bytes_per_sample(*nm) * Ns * T / (km * 1e-3) / 1e9}
        out["sweep_N"] = sweep
    if world == 1 and not unfused and not args.no_secondary and w.name == "planar_hand" and args.mode is None:
        # the reference's planar_hand set-up runs gradient_mode "first_order" (planar_hand_setup.py:28)
        w1 = Workload("planar_hand", args.T, "first_order")
        st1 = max(20, args.steps // 2)
        e1, ei1, km1, nm1, loop1 = run(w1, N, st1, max(1, st1 // 10))
        out["first_order"] = {"config": {"workload": w1.label, "T": w1.T, "N_per_gpu": N, "mode": w1.mode_name},
                              "value": N * w1.T * st1 / e1, "unit": "rollouts*timesteps/s", "steps": st1,
                              "ms_per_step": 1e3 * e1 / st1, "ilqr_iters_per_s": loop1["iters_per_s"],
                              "ilqr_loop": loop1, "ilqr_first_iter_per_s": st1 / ei1,
                              "roofline": roofline(w1, N, km1, nm1)}
    if world == 1 and not unfused and not args.no_secondary and w.name == "planar_hand" and args.mode is None \
            and args.contact_solver == "pgs":
        # the same workload with the step QP solved exactly (IRS_MODEL_PLANAR_HAND_EXACT)
        wx = Workload("planar_hand", args.T, None, contact_solver="exact")
        stx = max(20, args.steps // 4)
        ex_, eix, kmx, nmx, loopx = run(wx, N, stx, max(1, stx // 10))
        out["exact_contact_solver"] = {"config": {"workload": wx.label, "T": wx.T, "N_per_gpu": N, "mode": wx.mode_name},
                                       "value": N * wx.T * stx / ex_, "unit": "rollouts*timesteps/s", "steps": stx,
                                       "ms_per_step": 1e3 * ex_ / stx, "avg_launch_ms": kmx,
                                       "ilqr_iters_per_s": loopx["iters_per_s"], "ilqr_loop": loopx,
                                       "ilqr_first_iter_per_s": stx / eix}
    if world == 1 and not unfused and not args.no_secondary and w.name != "pendulum":
        w2 = Workload("pendulum")
        st2 = 10000
        e2, ei2, km2, nm2, _ = run(w2, N, st2, 1000)
        out["pendulum"] = {"config": {"workload": w2.label, "T": w2.T, "N_per_gpu": N, "mode": w2.mode_name},
                           "value": N * w2.T * st2 / e2, "unit": "rollouts*timesteps/s", "steps": st2,
                           "ms_per_step": 1e3 * e2 / st2, "ilqr_iters_per_s": st2 / ei2,
                           "roofline": roofline(w2, N, km2, nm2)}
    if rank == 0:
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1 or args.force_unfused:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
