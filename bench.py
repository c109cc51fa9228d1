#!/usr/bin/env python3
"""Headline benchmark: rollouts x timesteps / s of the randomised-smoothing pass
(get_TV_matrices: sample pass + reduction + solve -> A_t,B_t,c_t) and iLQR iterations / s
(that + the TV-LQR descent: T re-solved tail QPs + true-dynamics rollout + cost).

Default workload = the configuration BASELINE.json's metric is quoted on: planar_hand
(quasi-dynamic contact, irs_lqr_quasistatic), T=50, N=10000 u-perturbations per timestep PER
GPU, zero-order-B smoothing, every sample's step QP solved EXACTLY (what the reference's simulator
does), samples resident in HBM (f32).

    python bench.py --gpus 1 --steps 2000 --warmup 200

A "step" = one smoothing pass over the (T x N) sample grid = ONE kernel launch on one GPU.  With
--gpus N every rank holds its own N samples per timestep (weak scaling) and the (T,P) f64
statistics are all-reduced (RCCL) inside every step, followed by the solve launch -- the three
launches replayed as one HIP graph.  One process per GPU; N>1 is launched by
torch.distributed.run.  Prints ONE JSON line on rank 0 (stdout carries nothing else).

iLQR iterations / s of the contact workloads are measured on the optimisation the reference's script
performs ("ilqr_loop": 20 iterations per episode, every one re-linearised around the previous result with
fresh device-drawn samples, bounds re-centred, first tail warm-started from the previous descent);
"ilqr_first_iter_per_s" is the first iteration alone, repeated from a cold start.

Sub-reports of the default single-GPU run (all in the same JSON line):
  "sweep_N"             the metric's workload at N = 1e3 and 1e5 (north_star's N points)
  "first_order"         the reference's planar-hand gradient mode (planar_hand_setup.py:28)
  "pgs_contact_solver"  the opt-in approximate step-QP solver (50 projected sweeps + polish)
  "general_kernel"      the metric's workload through the general contact kernel (IRS_UG=0), for the record
  "pendulum"            BASELINE configs[1]: zero-order AB, T=30, N=1e4 (the HBM-bound case)
  "quadrotor"           BASELINE configs[2]: first-order, T=50, N=1e4 + on-device Riccati descent
  "box_pivoting"        BASELINE configs[4] at its per-GPU size (T=80, N=5e4/8): iRS-LQR (zero-order-B,
                        rate-limited descent) against CEM at the same simulator-step budget per iteration
                        (batch_size = N): iterations / s and cost after k iterations for both
  "cpu_baseline"        the oracle on the host: one core (bounded sample), and a pool of
                        min(usable CPUs, T) single-threaded processes at the GPU leg's N; median of >= 5
--mode first_order / --contact-solver pgs / --workload X make those the timed workload; --force-unfused
times the multi-GPU step with a 1-rank RCCL group on one GPU.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_F32_PEAK_TFLOPS = 157.3  # peak FP32 (vector), same guide
# 256 CUs x 4 SIMDs; max clock.  A wave64 VALU instruction EXECUTES in 2 cycles on the SIMD-32 (same guide, :54,473):
# that is the chip's capacity; one wave ALONE on a SIMD sustains one instruction per 4 cycles (:489), the ceiling of a
# kernel that holds one wave per SIMD (the general contact kernel; the uniform-geometry kernel holds two).
SIMDS, CLOCK_GHZ, ISSUE_CYCLES, ISSUE_CYCLES_ONE_WAVE = 1024, 2.4, 2, 4

# Algorithmic flops of ONE u-perturbed contact step of the planar hand inside the sample pass
# (csrc/contact_models.hpp).  In the u-only modes the state is not perturbed, so the
# contact geometry (J, phi, W = J D^-1 J', 1/W_ii) is the SAME for every sample of a timestep: it is
# loop-invariant (the compiler hoists it; the PMC instruction count confirms it) and NOT counted.
#   r = phi - J D^-1 b: only the 4 actuated entries of b move      8 x 4 x 2 =   64
#   one projected sweep over the 8 rows (residual form)                           152
#   masked LDL' of the 8 x 8 dual Hessian + one solve on it            196 + 120 = 316
#   slacks g = r + W lam                                                          128
#   primal recovery q+ = q + D^-1 (J' lam - b)                          112 + 14 = 126
#   zero-order-B statistics (Gram 10 + cross 28 FMAs)                              76
#   first-order extras: 4 right-hand sides through the factor, J'Y, B, sum       1208
_F = dict(r=64, sweep=152, ldl_solve=316, slack=128, primal=126, gram=76, first=1208, polish=470)
# ... and of the uniform-geometry kernel (csrc/smooth_ug.hip), what EVERY sample does:
#   r = r0 + C du                                                        8 x 4 x 2 =   64
#   6 projected sweeps, W in registers (per row: fma, max, sub, 8 fma)     6 x 8 x 20 =  960
#   4 primal-dual active-set iterations (table row: 8 mul + 64 fma)         4 x 136 =  544
#   statistics: Gram 10 fma + du lam' 32 fma + sum du 4                               88
# (the parked ~10 % of the samples' dual active-set steps are NOT counted: a lower bound)
UG_SWEEPS, UG_PDAS = 6, 4
_FUG = dict(r=64, sweep=160, pdas=136, stats=88, first=8)


def contact_flops_per_sample(solver, pgs_iters, first_order):
    """(flops, formula).  "exact": the dual active-set loop runs a data-dependent number of steps after its
    warm start (0-3 typical, ~500 flops each) and up to 3 repair rounds: only the work EVERY sample does is
    counted -- 32 warm-up sweeps, one factorisation + solve, the slacks -- so the figure is a LOWER bound of
    the algorithmic flops and roofline.frac a lower bound with it."""
    tail = _F["primal"] + (_F["first"] if first_order else _F["gram"])
    if solver == "ug":
        f = _FUG["r"] + UG_SWEEPS * _FUG["sweep"] + UG_PDAS * _FUG["pdas"] + (_FUG["first"] if first_order else _FUG["stats"])
        return f, "64 (r) + %d*160 (sweeps) + %d*136 (active-set iterations by table row) + %s; parked samples' " \
                  "dual active-set steps NOT counted (lower bound)" % (UG_SWEEPS, UG_PDAS,
                                                                        "8 (active-set key)" if first_order else "88 (statistics)")
    if solver == "exact":
        f = _F["r"] + 32 * _F["sweep"] + _F["ldl_solve"] + _F["slack"] + tail
        return f, "64 (r) + 32*152 (warm-up sweeps) + 316 (masked LDL' + solve) + 128 (slacks) + 126 (primal) + %s; " \
                  "data-dependent active-set steps NOT counted (lower bound)" % ("1208 (derivative)" if first_order else "76 (Gram)")
    f = _F["r"] + pgs_iters * _F["sweep"] + _F["polish"] + tail
    return f, "64 (r) + %d*152 (sweeps) + 470 (polish) + 126 (primal) + %s" % (
        pgs_iters, "1208 (derivative)" if first_order else "76 (Gram)")


class Workload:
    """Everything that differs between the benchmarked configurations."""

    def __init__(self, name, T=None, mode=None, host_only=False, contact_solver="exact"):
        self.name = name
        self.contact_solver = contact_solver
        self.bounds = None                  # ("abs" | "rel", half width) of the quasistatic descent
        self.std_x = 0.0
        self.flops_per_sample, self.flops_formula = None, None
        if host_only:           # a CPU-baseline worker process: only what the oracle needs, no GPU library
            self._host_only(T, mode)
            return
        import irs_mpc_amd as amd
        from irs_mpc_amd import _lib
        if name == "pendulum":
            # examples/pendulum/pendulum_zero_order.py:11-43 (BASELINE configs[1])
            self.T = T or 30
            self.system = amd.PendulumDynamics(0.05)
            self.mode, self.mode_name = _lib.SMOOTH_ZERO_ORDER_AB, "ZERO_ORDER_AB"
            self.x0 = np.zeros(2)
            self.u_trj = np.tile(np.array([0.1]), (self.T, 1))
            self.Q, self.Qd, self.R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
            self.xd = np.tile(np.array([np.pi, 0.]), (self.T + 1, 1))
            self.std_x, self.std_u = 1.0, 1.0
            self.label = "pendulum zero-order smoothing (BASELINE configs[1])"
            self.kernel = "smooth_kernel<PendulumModel, ZERO_ORDER_AB>"
        elif name == "quadrotor":
            # examples/quadrotor/quadrotor_first_order.py:12-52 with T = 50 (BASELINE configs[2])
            self.T = T or 50
            self.system = amd.QuadrotorDynamics(0.05)
            self.mode, self.mode_name = _lib.SMOOTH_FIRST_ORDER, "FIRST_ORDER"
            from examples.problems import quadrotor as quadrotor_problem
            _, p, _, _, _ = quadrotor_problem(self.T)
            self.x0, self.u_trj, self.Q, self.Qd, self.R, self.xd = p.x0, p.u_trj_initial, p.Q, p.Qd, p.R, p.xd_trj
            self.std_x, self.std_u = 0.1, 0.1
            self.label = "quadrotor first-order smoothing + on-device Riccati descent (BASELINE configs[2])"
            self.kernel = "smooth_kernel<QuadrotorModel, FIRST_ORDER>"
        elif name == "planar_hand":
            self.T = T or 50
            self.system = amd.PlanarHandDynamics(0.1, contact_solver=contact_solver)
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
            self.std_u = 0.3                            # run_planar_hand.py:146
            self.std_schedule = lambda it: 0.3 / it ** 0.8          # :142-146
            self.bounds = ("abs", 0.5 * sd.h)           # :138-139: u_bounds_abs = +-0.5 h
            first = mode == "first_order"
            if first:
                # gradient_mode "first_order" (examples/planar_hand/planar_hand_setup.py:28): every sample's
                # step is differentiated through its active constraints inside the sample pass
                self.mode, self.mode_name = _lib.SMOOTH_FIRST_ORDER, "FIRST_ORDER"
            model = "PlanarHandExactModel" if contact_solver == "exact" else "PlanarHandModel"
            # the exact 8-row model in a u-only mode runs the uniform-geometry kernel unless IRS_UG=0 forces the general one
            self.uniform_geometry = contact_solver == "exact" and os.environ.get("IRS_UG", "1") != "0"
            self.label = "planar_hand quasi-dynamic contact, %s smoothing, step QP %s" % (
                "first-order (per-sample active-set derivative)" if first else "zero-order-B",
                "solved exactly (dual active-set method)" if contact_solver == "exact"
                else "by %d projected sweeps + polish (opt-in)" % int(sd.pgs_iters))
            if not first and contact_solver == "exact":
                self.label += " -- the metric's config"
            self.kernel = "%s<%s, %s>" % ("smooth_ug_kernel" if self.uniform_geometry else "smooth_kernel", model, self.mode_name)
            self.flops_per_sample, self.flops_formula = contact_flops_per_sample(
                "ug" if self.uniform_geometry else contact_solver, int(sd.pgs_iters), first)
        elif name == "box_pivoting":
            # examples/box_pivoting/run_box_pivoting.py:20-131 (BASELINE configs[4]: T = 80)
            self.T = T or 80
            from examples.run_quasistatic import box_problem
            sd, x0, u0, Qd_, Qdd_, Rd_, xd = box_problem(self.T)
            if contact_solver != "exact":
                sd = amd.BoxPivotingDynamics(0.1, contact_solver=contact_solver)
            self.system = sd
            self.mode, self.mode_name = _lib.SMOOTH_ZERO_ORDER_B, "ZERO_ORDER_B"
            self.idx = sd.get_u_indices_into_x()
            self.x0, self.u_trj, self.xd = x0, u0, xd
            self.Q, self.Qd, self.R = sd.get_Q_from_Q_dict(Qd_), sd.get_Q_from_Q_dict(Qdd_), sd.get_R_from_R_dict(Rd_)
            self.std_schedule = lambda it: 0.1 ** (0.5 * it)        # :122-126
            self.std_u = self.std_schedule(1)
            self.bounds = ("rel", 0.15 * sd.h)          # :119-120: u_bounds_rel = +-0.15 h
            self.label = "box_pivoting quasi-dynamic contact (12 contact rows), zero-order-B smoothing, step QP solved exactly"
            self.kernel = "smooth_kernel<BoxPivotExactModel, ZERO_ORDER_B>"
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
        """the f32 perturbations are read once (u only in the u-only modes)."""
        return 4 * (m if self.name in ("planar_hand", "box_pivoting") else n + m)

    def oracle(self):
        from oracle import irs_oracle as orc
        if self.name == "pendulum":
            return orc, orc.PendulumOracle(0.05)
        return orc, orc.PlanarHandOracle(0.1, pgs_iters=0 if self.contact_solver == "exact" else 50)


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed; never the target)
# ------------------------------------------------------------------------------------------------
def _cpu_problem(w, N):
    """(one pass over a range of timesteps as a callable, name of the oracle function) at N samples."""
    orc, s = w.oracle()
    T = w.T
    rng = np.random.default_rng(0)
    x = orc.rollout(s, w.x0, w.u_trj)
    if w.name == "pendulum":
        dx = rng.normal(size=(T, N, 2)).astype(np.float32).astype(np.float64)
        du = rng.normal(size=(T, N, 1)).astype(np.float32).astype(np.float64)

        def part(t0, t1):
            orc.zero_order_TV(s, x[t0:t1 + 1], w.u_trj[t0:t1], dx[t0:t1], du[t0:t1])
        return part, "oracle.zero_order_TV"
    du = (w.std_u * rng.normal(size=(T, N, 4))).astype(np.float32).astype(np.float64)
    if w.mode_name == "FIRST_ORDER":
        def part(t0, t1):
            orc.first_order_B_decoupled(s, x[t0:t1 + 1], w.u_trj[t0:t1], du[t0:t1])
        return part, "oracle.first_order_B_decoupled"

    def part(t0, t1):
        orc.zero_order_B_decoupled(s, x[t0:t1 + 1], w.u_trj[t0:t1], du[t0:t1])
    return part, "oracle.zero_order_B_decoupled"


def _one_thread():
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        import threadpoolctl
        return threadpoolctl.threadpool_limits(1)
    except Exception:       # noqa: BLE001
        return None


def _pool_worker(args):
    """One worker of the pooled CPU baseline: its share of the timesteps, `reps` times, one thread.
    Returns the per-repetition times."""
    name, T, mode, solver, N, t0, t1, reps = args
    _one_thread()
    part, _ = _cpu_problem(Workload(name, T, mode, host_only=True, contact_solver=solver), N)
    part(t0, t1)                                    # warm-up (imports, first-touch)
    out = []
    for _ in range(reps):
        ts = time.perf_counter()
        part(t0, t1)
        out.append(time.perf_counter() - ts)
    return out


def usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:       # noqa: BLE001
        return os.cpu_count() or 1


def cpu_baseline(w, N, budget_s=15.0, reps=5):
    """The oracle (NumPy restatement with the reference's structure: Python loop over t, vectorised
    dynamics_batch, lstsq) timed on the GPU box's host, same workload, supplied samples:
      * ONE core (one BLAS thread): median of `reps` passes at N_1 = the largest of {N, N/2, N/5, N/10, ...}
        whose pass fits budget_s / reps (the rate of this loop does not depend on N once N >~ 1e3:
        it is per-sample work; the sample is stated);
      * "pool": the honest analogue of the reference's 18-30 ZMQ worker processes, which split the
        timesteps among themselves (irs_lqr_quasistatic.py:245-263) -- min(usable CPUs, T) single-threaded
        processes, each with its share of the timesteps at the GPU leg's N; a repetition costs what its
        slowest worker takes; median of `reps`.
    Reported, never the target."""
    import multiprocessing as mp
    T = w.T
    limit = _one_thread()
    # size the single-core sample: time a small probe, scale
    probe_N = min(N, 200 if w.name != "pendulum" else 2000)
    part, what = _cpu_problem(w, probe_N)
    part(0, T)
    ts = time.perf_counter()
    part(0, T)
    per_sample = (time.perf_counter() - ts) / (probe_N * T)
    N1 = N
    for div in (1, 2, 5, 10, 20, 50, 100):
        N1 = max(probe_N, N // div)
        if per_sample * N1 * T * reps <= budget_s:
            break
    part, what = _cpu_problem(w, N1)
    part(0, T)
    times = []
    for _ in range(reps):
        ts = time.perf_counter()
        part(0, T)
        times.append(time.perf_counter() - ts)
    med = statistics.median(times)
    if limit is not None:
        limit.restore_original_limits()
    out = {"value": T * N1 / med, "unit": "rollouts*timesteps/s", "cores": 1, "kind": "port",
           "sample": "median of %d passes of T=%d N=%d of the same workload (%s, supplied samples, 1 thread, %.1f s "
                     "per pass)%s" % (reps, T, N1, what, med,
                                      "" if N1 == N else "; the GPU leg runs N=%d: per-sample work, the rate carries over" % N),
           "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus()}
    cores = max(1, min(usable_cpus(), T))
    try:
        bounds = [round(i * T / cores) for i in range(cores + 1)]
        span = max(bounds[i + 1] - bounds[i] for i in range(cores))
        Np = N
        for div in (1, 2, 5, 10, 20, 50, 100):          # keep a repetition of the slowest worker under ~3 s
            Np = max(probe_N, N // div)
            if per_sample * Np * span * 1.5 <= 3.0:
                break
        jobs = [(w.name, w.T, "first_order" if w.mode_name == "FIRST_ORDER" else None, w.contact_solver, Np,
                 bounds[i], bounds[i + 1], reps) for i in range(cores) if bounds[i + 1] > bounds[i]]
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(len(jobs)) as pool:
            per_worker = pool.map_async(_pool_worker, jobs).get(timeout=240)
        wall = time.perf_counter() - t0
        rep_times = [max(wk[r] for wk in per_worker) for r in range(reps)]
        medp = statistics.median(rep_times)
        out["pool"] = {"value": T * Np / medp, "unit": "rollouts*timesteps/s", "cores": len(jobs),
                       "sample": "median of %d passes of T=%d N=%d, timesteps dealt out to %d single-threaded "
                                 "processes (min(usable CPUs, T)); a pass = its slowest worker, %.2f s (%.1f s in all "
                                 "with process start-up)" % (reps, T, Np, len(jobs), medp, wall)}
    except Exception as e:      # noqa: BLE001 -- the pooled figure is an extra; never fail the bench on it
        out["pool"] = {"error": repr(e)[:200]}
    return out


def pmc_entry(w, N):
    """The PMC passes of tools/profile_round.sh for this kernel (separate rocprofv3 --pmc runs of this same
    command), as adopted into profiles/pmc_latest.json together with the commit they were taken at."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None, None
    allp = json.load(open(path))
    tag = {"ZERO_ORDER_AB": "zero", "ZERO_ORDER_B": "zeroB", "FIRST_ORDER": "first"}[w.mode_name]
    key = "%s_%s_T%d_N%d" % (w.name + ("_exact" if w.contact_solver == "exact" and w.name in ("planar_hand", "box_pivoting")
                                       else ""), tag, w.T, N)
    return allp.get(key), allp.get("_meta")


def roofline(w, N, k_ms, nm):
    alg_bytes = w.bytes_per_sample(*nm) * N * w.T      # per launch (per GPU)
    hbm = alg_bytes / (k_ms * 1e-3) / 1e9
    pmc, meta = pmc_entry(w, N)
    traffic, issue = None, None
    src = None
    if pmc:
        src = {"file": "profiles/pmc_latest.json", "profile": meta}
        if "FETCH_SIZE_raw_avg" in pmc:
            # HBM bytes per launch: FETCH_SIZE (KB) doubled as the gfx950 note of MI355X_MICROARCH.md prescribes
            traffic = (2.0 * pmc["FETCH_SIZE_raw_avg"] + pmc.get("WRITE_SIZE_raw_avg", 0.0)) * 1024.0
        if "SQ_INSTS_VALU_raw_avg" in pmc and pmc.get("kernel_avg_ns"):
            # share of the chip's VALU issue slots the launch used, all from the profile (its own kernel time)
            base = pmc["SQ_INSTS_VALU_raw_avg"] / SIMDS / CLOCK_GHZ / pmc["kernel_avg_ns"]
            issue = {"frac": base * ISSUE_CYCLES,
                     "formula": "SQ_INSTS_VALU x %d cycles (a wave64 VALU instruction on the SIMD-32) / %d SIMDs / %.1f GHz / "
                                "kernel_avg_ns (both from the profile): the share of the CHIP's VALU capacity"
                                % (ISSUE_CYCLES, SIMDS, CLOCK_GHZ),
                     "frac_of_one_wave_per_simd_ceiling": base * ISSUE_CYCLES_ONE_WAVE,
                     "one_wave_note": "x %d cycles: what ONE wave per SIMD can issue at most -- the ceiling of the general "
                                      "contact kernel (388 registers); the uniform-geometry kernel holds two waves per SIMD"
                                      % ISSUE_CYCLES_ONE_WAVE,
                     "SQ_INSTS_VALU": pmc["SQ_INSTS_VALU_raw_avg"], "kernel_avg_ns": pmc["kernel_avg_ns"],
                     "valu_insts_per_sample": pmc["SQ_INSTS_VALU_raw_avg"] * 64.0 / (N * w.T)}
    r = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": src,
         "kernel": w.kernel + " (sample pass + reduction + solve, one launch)",
         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k_ms,
         "formula": "achieved = %d B x N x T / avg_launch_ms" % w.bytes_per_sample(*nm),
         "timing": "HIP event pair on the launch stream around the timed launches / launches"}
    if w.flops_per_sample:
        # the contact step is arithmetic on registers: the f32 vector rate is what bounds it
        tf = w.flops_per_sample * N * w.T / (k_ms * 1e-3) / 1e12
        r.update({"bound": "valu_f32", "achieved": tf, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": tf / VALU_F32_PEAK_TFLOPS, "alg_flops_per_launch": w.flops_per_sample * N * w.T,
                  "flops_per_sample": w.flops_per_sample,
                  "formula": "achieved = flops_per_sample x N x T / avg_launch_ms;  flops_per_sample = "
                             + w.flops_formula + ";  the contact geometry of a timestep (J, phi, W) is loop-invariant in "
                             "the u-only modes and not counted",
                  "valu_issue_slots": issue,
                  "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                          "formula": "%d B x N x T / avg_launch_ms (SURVEY 8(d))" % w.bytes_per_sample(*nm)},
                  "note": "no GEMM shape in a per-sample dual solve: the bound is the f32 vector rate, not HBM or "
                          "MFMA; the HBM view of the same launch is under 'hbm', the measured share of VALU issue "
                          "slots under 'valu_issue_slots'"})
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 2000 (contact workloads) / 20000 (pendulum)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="planar_hand", choices=["planar_hand", "pendulum", "quadrotor", "box_pivoting"])
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--N", type=int, default=10000, help="samples per timestep per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip every sub-report (timed workload only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="run all ranks on GPU 0 with gloo (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--sweep", action="store_true", help="also time N=1e6 for the pendulum (extra key)")
    ap.add_argument("--mode", default=None, choices=["first_order"],
                    help="planar_hand: gradient_mode first_order instead of zero_order_B as the timed workload")
    ap.add_argument("--contact-solver", default="exact", choices=["pgs", "exact"],
                    help="contact workloads: the exact dual active-set solve of every step QP (default; the "
                         "reference's semantics) or 50 over-relaxed projected sweeps + polish")
    ap.add_argument("--no-graph", action="store_true",
                    help="several ranks: issue the step's launches one by one instead of replaying a HIP graph")
    ap.add_argument("--collective", default="auto", choices=["auto", "torch", "direct", "peer"],
                    help="several ranks: the all-reduce issued by the HIP library on an RCCL communicator it owns, "
                         "the whole step captured into a HIP graph inside the library (csrc/collective.hip; 'direct'), "
                         "or through torch.distributed ('torch').  'auto' (default) = direct whenever the backend is "
                         "nccl and every rank can bind RCCL and join the communicator, else torch.  'peer' (opt-in, "
                         "not yet validated on several physical GPUs): no collective library at all -- every rank "
                         "reads the other ranks' statistics out of IPC-mapped device memory in one small launch")
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_unfused \
            and args.workload in ("planar_hand", "pendulum"):
        cpu_base = cpu_baseline(Workload(args.workload, args.T, args.mode, host_only=True,
                                         contact_solver=args.contact_solver), args.N)
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

    step_info = {}
    if unfused and args.collective in ("auto", "direct"):
        # the library-owned communicator is the default data path of a multi-GPU step; every rank must take the
        # same branch (a rank that fell back alone would issue different collectives and hang the others), so
        # the ranks agree FIRST on a rank-local pre-check (can RCCL be bound here, is the backend RCCL at all)
        # and only then enter the collective create; a failure there is agreed on the same way
        from irs_mpc_amd import _lib as _irs_lib
        from irs_mpc_amd.distributed import DirectComm

        def _agree(flag):
            if world == 1:
                return bool(flag)
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        why = None
        if not _agree(args.backend == "nccl" and _irs_lib.load().irs_comm_available() == 1):
            why = "RCCL not bound on every rank (backend %s)" % args.backend
        else:
            comm = None
            try:
                comm = DirectComm()
            except Exception as e:      # noqa: BLE001
                why = "irs_comm_create failed: " + repr(e)[:160]
            if _agree(comm is not None):
                step_info["comm"] = comm
            else:
                why = why or "irs_comm_create failed on another rank"
        if "comm" in step_info:
            args.collective = "direct"
        elif args.collective == "direct":
            raise SystemExit("--collective direct: " + why)
        else:
            args.collective, step_info["collective_fallback"] = "torch", why

    stream = torch.cuda.current_stream().cuda_stream
    peers = []                  # --collective peer: the exchanges made (one per timed workload), destroyed at the end

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
        # the first event pair of a process is slow (lazy initialisation: ~45 us for the first record, and the launches
        # behind it issue at half the usual rate; tools/short_run_timing.py): a 20-step region -- the driver's -- would
        # carry that as +3 us per step.  One throw-away pair, used the way the timed region uses its own, before it.
        w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0.record()
        fn()
        w1.record()
        while not w1.query():
            pass
        torch.cuda.synchronize()
        w0.elapsed_time(w1)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()                    # (the device is idle here: recorded before the host clock starts, it still
        t0 = time.perf_counter()        # brackets exactly the K steps on the device, and its ~6 us of host time is not theirs)
        for _ in range(steps):
            fn()
        ev1.record()
        while not ev1.query():          # spin until the last step has retired: a blocking synchronize() alone wakes the
            pass                        # host ~100 us late, which a 20-step region (the driver's default) would carry
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev_ms

    def bound_rows(w, x_trj, idx_t):
        """Absolute bound rows of the quasistatic descent around the nominal trajectory x_trj
        (irs_lqr_quasistatic.py:303-325)."""
        kind, wd = w.bounds
        if kind == "abs":
            nom = x_trj[:-1].index_select(1, idx_t)
            return dict(u_lo=(nom - wd).contiguous(), u_hi=(nom + wd).contiguous())
        m = idx_t.numel()
        return dict(du_lo=torch.full((w.T, m), -wd, dtype=dev.F64, device=x_trj.device),
                    du_hi=torch.full((w.T, m), wd, dtype=dev.F64, device=x_trj.device))

    def run(w, N, steps, warmup, loop_iters=20):
        """Times the smoothing step and the full iLQR iteration of workload `w`."""
        dm = w.system.dm()
        n, m, T, MODE = dm.n, dm.m, w.T, w.mode
        Q, Qd, R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
        xd, x0, u_trj = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
        x_trj, _ = dm.rollout_cost(x0, u_trj, Q, R, xd)
        g = torch.Generator(device="cuda").manual_seed(1234 + rank)
        dx = None
        if w.std_x > 0:
            dx = w.std_x * torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
        du = w.std_u * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
        n_total = N * world
        contact = w.bounds is not None
        if not unfused:
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=True)
            tv = plan.out

            def smooth_step():
                plan.run(stream)
        elif args.collective in ("direct", "peer"):
            # the library issues accumulate -> all-reduce (RCCL, or the IPC peer exchange) -> solve itself and
            # replays them as one HIP graph
            from irs_mpc_amd.distributed import CollectiveStep, PeerExchange
            plan = dev.SmoothPlan(dm, MODE, x_trj, u_trj, dx=dx, du=du, fuse=True, n_total=n_total)
            tv = plan.out
            if args.collective == "peer":
                step_info["comm"] = PeerExchange(plan.sums.numel())
                peers.append(step_info["comm"])
            cstep = CollectiveStep(plan, step_info["comm"])
            cstep.run()
            torch.cuda.synchronize()
            ref = [tv[k].clone() for k in ("At", "Bt", "ct")]
            cstep.capture()
            cstep.run()
            torch.cuda.synchronize()
            assert all(torch.equal(a_, tv[k]) for a_, k in zip(ref, ("At", "Bt", "ct"))), "replayed step differs"
            step_info["graph"] = True
            step_info["collective"] = ("direct (library-owned RCCL communicator, graph captured in the library)"
                                       if args.collective == "direct" else
                                       "peer (IPC-mapped exchange regions, one launch, no collective library; graph "
                                       "captured in the library)")
            smooth_step = cstep.run
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
        if contact:
            # IrsLqrQuasistatic.local_descent (irs_lqr_quasistatic.py:286-345) as the reference's script sets
            # it up: du cost, ONE control box (trust region / rate limit), T re-solved tail QPs, contact
            # dynamics in the loop; one launch
            idx_t = torch.as_tensor(w.idx, device=x_trj.device)
            brows = bound_rows(w, x_trj, idx_t)
            qs_out = {}

            def descent_run():
                qs_out["o"] = dm.quasistatic_box_descent(tv["At"], tv["Bt"], tv["ct"], Q, Qd, R, xd, x0,
                                                         solver=0, max_iter=2000, eps=1e-9, out=qs_out.get("o"),
                                                         **brows)
        else:
            descent = dev.DescentPlan(dm, tv["At"], tv["Bt"], tv["ct"], Q, Qd, R, xd, x0)

            def descent_run():
                descent.run(stream)

        def ilqr_step():
            smooth_step()
            descent_run()

        el, ev_ms = timed(smooth_step, steps, warmup)
        it_steps = max(1, steps // (20 if contact else 4))
        el_it, _ = timed(ilqr_step, it_steps, max(1, it_steps // 10))
        el_it *= steps / it_steps
        loop = None
        if contact:
            # The optimisation the reference's script performs: `loop_iters` iterations (planar_hand_setup.py:36
            # num_iters = 20), every one linearised around the previous one's result with FRESH samples of the
            # script's std schedule (drawn on the device), the bounds re-centred on the new nominal, and the
            # first tail's active set handed on from the previous descent (irs_quasistatic_box_descent_ws) --
            # cold at the start of every episode.  Nothing returns to the host inside an episode.
            K_LOOP = loop_iters
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
                    lplan.set_iter(it, None, [w.std_schedule(it)] * m)
                    lplan.set_trajectory(xs[a_], us[a_])
                    lplan.run()
                    if not unfused:
                        At_, Bt_, ct_ = lplan.out["At"], lplan.out["Bt"], lplan.out["ct"]
                    else:
                        if args.collective == "peer":
                            step_info["comm"].all_reduce_sums(lplan.sums)
                        elif args.force_unfused:
                            dist.all_reduce(lplan.sums)
                        else:
                            all_reduce_sums(lplan.sums)
                        ltv["out"] = dm.smooth_finalize(MODE, n_total, xs[a_], us[a_], lplan.sums,
                                                        out=ltv.get("out"), workspace=lplan.ws)
                        At_, Bt_, ct_ = ltv["out"][:3]
                    louts[b_] = dm.quasistatic_box_descent(At_, Bt_, ct_, Q, Qd, R, xd, x0, solver=0, max_iter=2000,
                                                           eps=1e-9, out=louts[b_], act=act,
                                                           **bound_rows(w, xs[a_], idx_t))
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
                    "cost_first_iteration": costs[0], "cost_best": min(costs), "cost_last": costs[-1],
                    "what": "the %d-iteration optimisation of the reference's script from its initial trajectory: every "
                            "iteration re-linearises around the previous result with fresh device-drawn samples "
                            "(the script's std schedule), re-centres the bounds and warm-starts the first tail's active "
                            "set from the previous descent; cold start at the head of each episode" % K_LOOP}
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
        assert info == 0, "smoothing solve reported a non-SPD Gram matrix or a non-finite statistic"
        return el, el_it, k_ms, (n, m), loop

    def public_api(w_, N_, k_iters, episodes=3):
        """iLQR iterations / s through the PUBLIC classes a user of the reference switches to: `solver.iterate(k)` of
        IrsLqrZeroOrder / IrsLqrFirstOrder (one irs_iterate library call: csrc/iterate.hip) or IrsLqrQuasistatic (all
        descents enqueued without a host synchronisation, one read-back), quiet, samples drawn on the device
        (GaussianSmoothing / params.device_rng_seed: the extensions that keep the perturbations off the host; the
        reference's host closures cost 20 ms of NumPy draws per iteration at this size).  Timed: iterate() alone,
        k + 1 descents per call, a fresh solver per episode (its construction -- initial rollout -- is set-up)."""
        import irs_mpc_amd as amd
        m_ = w_.system.dim_u

        def make(script_bounds=True):
            if w_.bounds is None:
                p = amd.IrsLqrParameters()
                p.Q, p.Qd, p.R, p.x0, p.xd_trj, p.u_trj_initial = w_.Q, w_.Qd, w_.R, w_.x0, w_.xd, w_.u_trj
                if script_bounds:
                    # the boxes the reference's scripts pass (pendulum_zero_order.py:22-29: +-1e4; quadrotor_first_order.py:
                    # 29-38: +-1e5 and the attitude limits): finite, so every iteration also tests whether any tail's
                    # unconstrained plan leaves them (csrc/iterate.hip) before it may skip the bounded QPs
                    from examples.problems import PROBLEMS
                    pp = PROBLEMS[w_.name](w_.T)[1]
                    p.xbound, p.ubound = pp.xbound, pp.ubound
                smp = amd.GaussianSmoothing(np.full(w_.system.dim_x, w_.std_x), np.full(m_, w_.std_u), N_, power=0.5, seed=11)
                cls = amd.IrsLqrFirstOrder if w_.mode_name == "FIRST_ORDER" else amd.IrsLqrZeroOrder
                sol = cls(w_.system, p, smp)
            else:
                sd = w_.system
                p = amd.IrsLqrQuasistaticParameters()
                names = sd.models_unactuated + sd.models_actuated
                p.Q_dict = {k_: np.diag(w_.Q)[sd.position_indices[k_]] for k_ in names}
                p.Qd_dict = {k_: np.diag(w_.Qd)[sd.position_indices[k_]] for k_ in names}
                p.R_dict, i0 = {}, 0
                for k_ in sd.models_actuated:
                    nk = len(sd.position_indices[k_])
                    p.R_dict[k_] = np.diag(w_.R)[i0:i0 + nk]
                    i0 += nk
                p.x0, p.x_trj_d, p.u_trj_0, p.T = w_.x0, w_.xd, w_.u_trj, w_.T
                kind, wd = w_.bounds
                box = np.array([-np.ones(m_) * wd, np.ones(m_) * wd])
                if kind == "abs":
                    p.u_bounds_abs = box
                else:
                    p.u_bounds_rel = box
                p.sampling = lambda u_initial, it: np.full(m_, w_.std_schedule(it))
                p.std_u_initial = np.full(m_, w_.std_u)
                p.num_samples = N_
                p.gradient_mode = "first_order" if w_.mode_name == "FIRST_ORDER" else "zero_order_B"
                p.publish_every_iteration = False
                p.device_rng_seed = 11
                sol = amd.IrsLqrQuasistatic(sd, p)
            sol.verbose = False
            return sol

        make().iterate(min(2, k_iters))            # warm-up (allocations, lazy initialisation)
        torch.cuda.synchronize()
        best = None
        for _ in range(episodes):
            sol = make()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sol.iterate(k_iters)
            dt = time.perf_counter() - t0          # iterate() returns host arrays: the read-back is inside
            best = dt if best is None or dt < best else best
        hist = sol.cost_lst if hasattr(sol, "cost_lst") else sol.cost_all_list
        extra = {}
        if w_.bounds is None:
            make(False).iterate(min(2, k_iters))
            b2 = None
            for _ in range(episodes):
                sol2 = make(False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sol2.iterate(k_iters)
                dt = time.perf_counter() - t0
                b2 = dt if b2 is None or dt < b2 else b2
            extra = {"bounds": "the script's xbound / ubound (finite: the plan test runs every iteration)",
                     "iters_per_s_without_bounds": (k_iters + 1) / b2}
        return {"iters_per_s": (k_iters + 1) / best, "ms_per_iter": 1e3 * best / (k_iters + 1), **extra,
                "call": "%s.iterate(%d): %d descents per call, best of %d calls" % (type(sol).__name__, k_iters, k_iters + 1, episodes),
                "cost_first": float(hist[0]), "cost_last": float(hist[-1]),
                "path": ("one irs_iterate library call (descents enqueued back to back, one read-back)"
                         if w_.bounds is None else "host loop without device synchronisation, one read-back")}

    def sub_report(w_, N_, steps_, warm_=None, sweep_N=(), **kw):
        e, ei, km, nm_, lp = run(w_, N_, steps_, warm_ if warm_ is not None else max(1, steps_ // 10), **kw)
        r = {"config": {"workload": w_.label, "T": w_.T, "N_per_gpu": N_, "mode": w_.mode_name},
             "value": N_ * w_.T * steps_ / e, "unit": "rollouts*timesteps/s", "steps": steps_,
             "ms_per_step": 1e3 * e / steps_, "avg_launch_ms": km,
             "ilqr_iters_per_s": lp["iters_per_s"] if lp else steps_ / ei,
             "ilqr_first_iter_per_s": steps_ / ei, "roofline": roofline(w_, N_, km, nm_)}
        if lp:
            r["ilqr_loop"] = lp
        if world == 1 and not unfused:
            try:
                # (quadrotor: the script's own 3 iterations, examples/quadrotor/quadrotor_first_order.py:57 -- without a line
                # search the optimisation breaks down a few iterations later, at this horizon, in the reference too)
                api = public_api(w_, N_, (kw.get("loop_iters", 20) - 1) if w_.bounds is not None
                                 else (3 if w_.name == "quadrotor" else 99))
                r["ilqr_iters_per_s_public_api"], r["public_api"] = api["iters_per_s"], api
            except Exception as e:      # noqa: BLE001 -- an optimisation that breaks down is reported, not fatal
                r["ilqr_iters_per_s_public_api"], r["public_api"] = None, {"error": repr(e)[:200]}
        if sweep_N:                                   # north_star's other N points for this workload
            r["sweep_N"] = {}
            for Ns in sweep_N:
                st = max(20, steps_ // (4 if Ns <= 10000 else 16))
                e2, ei2, km2, nm2, _ = run(w_, Ns, st, max(2, st // 10), **kw)
                r["sweep_N"][str(Ns)] = {"value": Ns * w_.T * st / e2, "ms_per_step": 1e3 * e2 / st, "avg_launch_ms": km2,
                                         "ilqr_iters_per_s": st / ei2,
                                         "hbm_GBps": w_.bytes_per_sample(*nm2) * Ns * w_.T / (km2 * 1e-3) / 1e9}
        return r

    def run_cem(w, B, iters, n_ep=2):
        """CrossEntropyMethodQuasistatic.local_descent (cem_quasistatic.py:168-211) at batch_size = B: B contact
        rollouts of T steps + their quasistatic costs, the elites, the refit -- `iters` iterations per episode
        from the script's initial trajectory; candidates drawn on the device (synthetic data)."""
        dm = w.system.dm()
        T, m = w.T, dm.m
        Q, Qd, R = dev.to_dev(w.Q), dev.to_dev(w.Qd), dev.to_dev(w.R)
        xd, x0, u0 = dev.to_dev(w.xd), dev.to_dev(w.x0), dev.to_dev(w.u_trj)
        n_elite = max(2, B // 20)                                   # run_box_pivoting_cem.py:118-119: 5 of 100
        gen = torch.Generator(device="cuda").manual_seed(99)
        best = []

        def episode():
            u, std = u0.clone(), torch.full((T, m), 0.2, dtype=dev.F64, device="cuda")      # :120 initial_std
            del best[:]
            for _ in range(iters):
                cand = u[None] + std[None] * torch.randn((B, T, m), generator=gen, device="cuda", dtype=dev.F64)
                costs = dm.cem_rollout_costs_quasistatic(cand, x0, Q, Qd, R, xd)
                _, u, std = dm.cem_refit(cand, costs, n_elite)
                best.append(dm.cem_rollout_costs_quasistatic(u[None].contiguous(), x0, Q, Qd, R, xd))

        el, _ = timed(episode, n_ep, 1, prewarm=1)
        costs = [float(c.item()) for c in best]
        return {"iters_per_s": n_ep * iters / el, "ms_per_iter": 1e3 * el / (n_ep * iters), "batch_size": B,
                "n_elite": n_elite, "initial_std": 0.2, "iterations_per_episode": iters,
                "cost_after_k_iterations": costs[-1], "cost_best": min(costs),
                "sim_steps_per_iteration": B * T,
                "what": "CEM (cem_quasistatic.py:168-211) at batch_size = N: the same number of simulator steps per "
                        "iteration as the iRS-LQR sample pass; the mean's cost after each refit"}

    w = Workload(args.workload, args.T, args.mode, contact_solver=args.contact_solver)
    N, T = args.N, w.T
    contact = w.bounds is not None
    el, el_it, k_mean, nm, loop = run(w, N, args.steps, args.warmup)
    out = {
        "metric": "rollouts*timesteps/s (randomized-smoothing pass) + iLQR-iters/s",
        "value": world * N * T * args.steps / el,
        "unit": "rollouts*timesteps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * el / args.steps,
        "timed_region_s": el,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "parity": (("planar hand PINNED by the reference's own result files: the first entry of examples/planar_hand/"
                    "analysis/planar_hand_spin_{exact,zero_order_B,first_order}.csv (the script's initial 30-step "
                    "simulator rollout) is reproduced to 2e-10 by the oracle and by the device functor (geometry, "
                    "stiffness, mass; friction identified only as >= 0.5: the grasp sticks); contact step + derivative "
                    "pinned by the simulator data of box_pushing; later entries of those files depend on Gurobi's "
                    "multipliers and the simulator's truncated least squares and are not reproducible (DESIGN.md 3)")
                   if w.name == "planar_hand" else
                   "partial: contact step and derivative pinned on box_pushing (simulator data), optimiser by the "
                   "reference's result files; the box-pivoting geometry is NOT pinned (its result files' first entry is "
                   "1.1 % off for every mass / friction: DESIGN.md 3)") if contact
        else "pinned by reference-run fixtures and result files",
        "config": {"workload": w.label, "T": T, "N_per_gpu": N, "N_total": N * world, "mode": w.mode_name,
                   "contact_solver": w.contact_solver if contact else None,
                   "samples": "supplied, resident in HBM (f32)",
                   "parallelism": "samples sharded over %d GPU(s), 1 all-reduce of (T,P) f64 per step" % world},
        "ilqr_iters_per_s": loop["iters_per_s"] if loop else args.steps / el_it,
        "ms_per_ilqr_iter": loop["ms_per_iter"] if loop else 1e3 * el_it / args.steps,
        "ilqr_loop": loop,
        "ilqr_first_iter_per_s": args.steps / el_it,
        "ilqr_iters_note": ("ilqr_iters_per_s = the real 20-iteration loop (ilqr_loop); ilqr_first_iter_per_s = the "
                            "first iteration alone, repeated from a cold start (no warm start, supplied samples)"
                            if loop else "one iteration repeated: its cost does not depend on the trajectory"),
        "ilqr_iter": ("smoothing launch + IrsLqrQuasistatic.local_descent (du cost, one control box, T re-solved tail "
                      "QPs by the active-set solver, contact dynamics in the loop)"
                      if contact else "smoothing launch + Riccati + closed-loop rollout + cost (bounds inactive)"),
        "roofline": roofline(w, N, k_mean, nm),
    }
    if world == 1 and not unfused:
        try:
            api = public_api(w, N, 19 if contact else (3 if w.name == "quadrotor" else 99))
            out["ilqr_iters_per_s_public_api"], out["public_api"] = api["iters_per_s"], api
        except Exception as e:      # noqa: BLE001
            out["ilqr_iters_per_s_public_api"], out["public_api"] = None, {"error": repr(e)[:200]}
    if unfused:
        out["config"]["step"] = ("accumulate launch + all-reduce + solve launch, replayed as one HIP graph"
                                 if step_info.get("graph") else
                                 "accumulate launch + all-reduce + solve launch, issued one by one")
        if "graph_error" in step_info:
            out["config"]["graph_error"] = step_info["graph_error"]
        out["config"]["collective"] = step_info.get("collective", "torch.distributed all_reduce (%s)" % args.backend)
        if "collective_fallback" in step_info:
            out["config"]["collective_fallback"] = step_info["collective_fallback"]
    secondary = world == 1 and not unfused and not args.no_secondary
    if secondary:
        # north_star's N points for the timed workload
        sweep = {}
        for Ns in (1000, 100000) + ((1000000,) if (args.sweep and w.name == "pendulum") else ()):
            st = max(20, args.steps // (4 if Ns <= 10000 else 16))
            e, ei, km, _, lp = run(w, Ns, st, max(2, st // 10), loop_iters=20)
            sweep[str(Ns)] = {"value": Ns * T * st / e, "ms_per_step": 1e3 * e / st, "avg_launch_ms": km,
                              "ilqr_iters_per_s": lp["iters_per_s"] if lp else st / ei,
                              "hbm_GBps": w.bytes_per_sample(*nm) * Ns * T / (km * 1e-3) / 1e9,
                              "valu_TFLOPs": (w.flops_per_sample * Ns * T / (km * 1e-3) / 1e12) if w.flops_per_sample else None}
        out["sweep_N"] = sweep
    if secondary and w.name == "planar_hand" and args.mode is None and args.contact_solver == "exact":
        st = max(20, args.steps // 2)
        # the reference's planar_hand set-up runs gradient_mode "first_order" (planar_hand_setup.py:28)
        out["first_order"] = sub_report(Workload("planar_hand", args.T, "first_order"), N, st)
        # the opt-in approximate step-QP solver
        out["pgs_contact_solver"] = sub_report(Workload("planar_hand", args.T, None, contact_solver="pgs"), N,
                                               max(20, args.steps // 4))
        if w.uniform_geometry:
            # the same workload through the GENERAL contact kernel (csrc/smooth.hip: per-lane geometry and LDL', parked
            # samples; rounds 1-2's kernel), for the record
            os.environ["IRS_UG"] = "0"
            try:
                out["general_kernel"] = sub_report(Workload("planar_hand", args.T, None), N, max(20, args.steps // 4))
                out["general_kernel"]["note"] = ("IRS_UG=0: smooth_kernel<PlanarHandExactModel, ZERO_ORDER_B>, one wave per "
                                                 "SIMD; the default since round 3 is smooth_ug_kernel (uniform geometry)")
            finally:
                del os.environ["IRS_UG"]
    if secondary and w.name != "pendulum":
        out["pendulum"] = sub_report(Workload("pendulum"), 10000, max(200, 5 * args.steps), 1000, sweep_N=(1000, 100000))
    if secondary and w.name != "quadrotor":
        out["quadrotor"] = sub_report(Workload("quadrotor"), 10000, max(100, args.steps), sweep_N=(1000, 100000))
    if secondary and w.name != "box_pivoting":
        # BASELINE configs[4]: T = 80, N = 5e4 over 8 GPUs -> 6250 samples per timestep per GPU
        wb = Workload("box_pivoting")
        Nb = 50000 // 8
        K = 10
        rb = sub_report(wb, Nb, max(20, args.steps // 4), loop_iters=K)
        rb["config"]["N_total_at_8_gpus"] = 50000
        rb["cem_same_budget"] = run_cem(wb, Nb, K)
        lp = rb.get("ilqr_loop") or {}
        rb["comparison"] = {"budget_sim_steps_per_iteration": Nb * wb.T, "iterations": K,
                            "irs_lqr": {"iters_per_s": lp.get("iters_per_s"), "cost_after_k_iterations": lp.get("cost_last"),
                                        "cost_best": lp.get("cost_best")},
                            "cem": {"iters_per_s": rb["cem_same_budget"]["iters_per_s"],
                                    "cost_after_k_iterations": rb["cem_same_budget"]["cost_after_k_iterations"],
                                    "cost_best": rb["cem_same_budget"]["cost_best"]},
                            "ref": "examples/box_pivoting/run_box_pivoting.py:96-135 vs run_box_pivoting_cem.py:100-135"}
        out["box_pivoting"] = rb
    if rank == 0:
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if os.environ.get("IRS_PRINT_STAMPS"):      # diagnostic library only (tools/stamp_descent.sh): phases of the LAST descent
        from irs_mpc_amd import _lib
        lib = _lib.load()
        if hasattr(lib, "irs_cbm_print_stamps"):
            lib.irs_cbm_print_stamps.restype = None
            lib.irs_cbm_print_stamps()
    for px in peers:
        step_info_t = px.status()
        assert step_info_t[1] == 0, "peer exchange: %d of %d launches timed out waiting for a rank" % (step_info_t[1], step_info_t[0])
        px.destroy()
    if world > 1 or args.force_unfused:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
