"""Host mirror of the reference's quasistatic optimiser, driving the HIP library.

    IrsLqrQuasistaticParameters      irs_lqr/irs_lqr_quasistatic.py:12-41
    IrsLqrQuasistatic                irs_lqr/irs_lqr_quasistatic.py:44-390

Same parameter object, attributes (`x_trj, u_trj, cost, x_trj_best, u_trj_best, cost_best,
current_iter, x_trj_list, u_trj_list, cost_all_list, cost_Qu_list, ...`) and methods.  What differs
underneath:

  * `get_TV_matrices` / `get_TV_matrices_batch` are the same device launch (sample pass over all
    T x N perturbed one-step evaluations + reduction + solve); the ZMQ PUSH/PULL sockets, the
    "Press Enter when the workers are ready" prompt and the worker processes
    (irs_lqr_quasistatic.py:118-131, zmq_parallel_cmp/) are gone.  With torch.distributed
    initialised the samples are sharded over the ranks and the sums all-reduced once.
  * `local_descent`'s T re-solved, bounded QPs (solve_tvlqr + Gurobi, :326-345) run in one launch
    (csrc/boxqp.hip, ADMM around a shared Riccati factorisation of the [x; u_prev] problem).
  * `q_dynamics` is a device-backed functor (e.g. PlanarHandDynamics), not the external
    quasistatic_simulator.  All four gradient modes run: "zero_order_B", "zero_order_AB" (the damped
    joint fit of calc_AB_zero_order), "first_order" (the mean over the u-perturbed samples of the
    step's active-set derivative -- the simulator's Dq_nextDqa_cmd -- computed per sample inside the
    sample pass) and "exact" (that derivative at the nominal point).  The sample-pass kernels return
    the decoupled pair (decouple_AB = True: every example of the reference); with decouple_AB = False
    the full pair is assembled from the pass's statistics and the f64 Jacobian lanes (slower paths).

`params.sampling(std_u_initial, iter)` returns the std of the u-perturbations like the reference;
the draws are made on the host by `np.random.normal(0, std_u, (num_samples, dim_u))` once per
time step in time order -- exactly the reference's calls (quasistatic_dynamics.py:258), so
identical seeds give identical samples -- unless `params.device_rng_seed` is set, in which case the
perturbations are drawn on the device (Philox) and never touch the host.
"""
import numpy as np
import torch

from . import device as dev
from . import distributed as dist_util
from ._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
from .quasistatic_base import QuasistaticOptimizerBase, quasistatic_eval_cost  # noqa: F401
from .tv_lqr import get_solver


class IrsLqrQuasistaticParameters:
    """irs_lqr/irs_lqr_quasistatic.py:12-41."""

    def __init__(self):
        # Necessary arguments defining optimal control problem.
        self.Q_dict = None
        self.Qd_dict = None
        self.R_dict = None
        self.x0 = None
        self.x_trj_d = None
        self.u_trj_0 = None
        self.T = None

        # Optional arguments defining bounds.
        self.x_bounds_abs = None
        self.u_bounds_abs = None
        self.x_bounds_rel = None
        self.u_bounds_rel = None

        # Necessary arguments related to sampling.
        self.sampling = None
        self.std_u_initial = None
        self.num_samples = 100

        # Arguments related to various options.
        self.decouple_AB = True
        self.use_workers = True
        # Supports "first_order", "exact", "zero_order_B", "zero_order_AB"
        self.gradient_mode = "zero_order_B"
        self.solver_name = "gurobi"
        self.task_stride = 1
        self.publish_every_iteration = True

        # ---- extensions (absent in the reference) ----
        self.device_rng_seed = None     # int: draw the perturbations on the device
        self.qp_solver = 0              # 0 auto, 1 ADMM, 2 active set (lanes), 3 active set (matrix-core tiles)
        self.qp_rho = 100.0             # ADMM penalty / iteration limit / tolerance of the bounded QPs
        self.qp_max_iter = 20000
        self.qp_eps = 1e-9


class IrsLqrQuasistatic(QuasistaticOptimizerBase):
    def __init__(self, q_dynamics, params):
        for name in ("x_bounds_abs", "u_bounds_abs", "x_bounds_rel", "u_bounds_rel", "decouple_AB", "use_workers",
                     "gradient_mode", "task_stride", "std_u_initial", "sampling", "num_samples"):
            setattr(self, name, getattr(params, name))
        if self.gradient_mode not in ("zero_order_B", "zero_order_AB", "first_order", "exact"):
            raise RuntimeError(f"AB mode {self.gradient_mode} is not supported.")   # quasistatic_dynamics.py:238
        if self.x_bounds_rel is not None:
            raise NotImplementedError("x_bounds_rel ('should be rarely used', irs_lqr_quasistatic.py:315) "
                                      "is not implemented on the device")
        dm = q_dynamics.dm()
        one_box = self.x_bounds_abs is None and (self.u_bounds_abs is None or self.u_bounds_rel is None)
        self._solver = int(getattr(params, "qp_solver", 0))
        if self._solver == 0:
            # one control box: the exact active-set method -- on matrix-core tiles (any horizon: beyond the
            # LDS-resident size its records move to HBM) where the model fits the tile, else on lanes
            self._solver = 1
            if one_box:
                self._solver = 3 if dm.quasistatic_descent_supported(params.T, 3) else (
                    2 if dm.quasistatic_descent_supported(params.T, 2) else 1)
        if not dm.quasistatic_descent_supported(params.T, self._solver):
            raise NotImplementedError("horizon T=%d does not fit the LDS-resident QP factorisation of solver %d"
                                      % (params.T, self._solver))
        self._setup(q_dynamics, params, params.x_trj_d)
        self._act = None
        self._idx = torch.as_tensor(np.asarray(self.indices_u_into_x), device=self._x0.device)
        # kept for interface parity; the bounded QPs are solved on the device
        self.solver = get_solver(params.solver_name)

    # ---- linearisation: irs_lqr_quasistatic.py:196-273 ------------------------
    def _get_TV_matrices_dev(self, x_trj, u_trj):
        std_u = np.broadcast_to(np.asarray(self.sampling(self.std_u_initial, self.current_iter), float),
                                (self.dim_u,))
        if self.gradient_mode == "zero_order_AB":
            return self._zero_order_AB_dev(x_trj, u_trj, std_u)
        if self.gradient_mode == "exact":
            return self._exact_dev(x_trj, u_trj)
        if self.gradient_mode == "first_order" and not self.decouple_AB:
            return self._first_order_full_dev(x_trj, u_trj, std_u)
        # "zero_order_B": least-squares fit of B; "first_order": mean of the per-sample derivative
        MODE = SMOOTH_FIRST_ORDER if self.gradient_mode == "first_order" else SMOOTH_ZERO_ORDER_B
        rank, world = dist_util.rank_world()
        N = self.num_samples
        seed = getattr(self.params, "device_rng_seed", None)
        lo, hi = dist_util.shard_range(N, rank, world)
        if seed is None:
            # the reference's draws, in its order (quasistatic_dynamics.py:258 inside calc_AB_batch)
            du = np.stack([np.random.normal(0, std_u, size=[N, self.dim_u]) for _ in range(self.T)])
            du = dev.to_dev(np.ascontiguousarray(du[:, lo:hi], np.float32), dev.F32)
        if world == 1:
            if seed is None:
                o = self._dm.smooth(MODE, x_trj, u_trj, None, du)
            else:
                o = self._dm.smooth_rng(MODE, x_trj, u_trj, N, None, std_u, int(seed),
                                        self.current_iter)
            self._smooth_info = o["info"]
            if MODE == SMOOTH_ZERO_ORDER_B and not self.decouple_AB:
                return self._zero_order_B_full_dev(x_trj, u_trj, o["sums"])
            return o["At"], o["Bt"], o["ct"]
        if seed is None:
            sums = self._dm.smooth_accumulate(MODE, x_trj, u_trj, None, du)
        else:
            sums = self._dm.smooth_accumulate_rng(MODE, x_trj, u_trj, hi - lo, None, std_u,
                                                  int(seed), self.current_iter, sample_offset=lo)
        dist_util.all_reduce_sums(sums)
        # the accumulate launch left the f64 nominal contact steps in its workspace: reuse them
        ws = self._dm._workspace(MODE, self.T, hi - lo, x_trj.device)
        At, Bt, ct, info = self._dm.smooth_finalize(MODE, N, x_trj, u_trj, sums, workspace=ws)
        self._smooth_info = info
        if MODE == SMOOTH_ZERO_ORDER_B and not self.decouple_AB:
            return self._zero_order_B_full_dev(x_trj, u_trj, sums)
        return At, Bt, ct

    # ---- decouple_AB = False: no example of the reference uses it; served from what the device already
    #      offers (the statistics of the sample pass, the f64 active-set Jacobian), a few small launches more
    def _zero_order_B_full_dev(self, x_trj, u_trj, sums):
        """calc_B_zero_order without decouple_AB_matrices (quasistatic_dynamics.py:242-266), read off the
        (all-reduced) statistics of the sample pass: `QuasistaticDeviceDynamics.zero_order_B_from_sums_dev`."""
        At, Bt, ct, info = self.q_dynamics.zero_order_B_from_sums_dev(x_trj[:-1], u_trj, sums)
        self._smooth_info = self._smooth_info | info
        return At, Bt, ct

    def _first_order_full_dev(self, x_trj, u_trj, std_u):
        """calc_AB_first_order without decouple_AB_matrices (quasistatic_dynamics.py:193-208): the mean over
        the u-perturbed samples of the FULL [Dq_nextDq | Dq_nextDqa_cmd] -- `calc_AB_batch_dev` over the T
        nominal points (f64 Jacobian lanes, one per sample; ranks own shards and all-reduce the sums)."""
        At, Bt, ct, self._smooth_info = self.q_dynamics.calc_AB_batch_dev(
            x_trj[:-1], u_trj, self.num_samples, std_u, "first_order",
            seed=getattr(self.params, "device_rng_seed", None), it=self.current_iter)
        return At, Bt, ct

    def _exact_dev(self, x_trj, u_trj):
        """gradient_mode "exact" (calc_AB_exact, quasistatic_dynamics.py:189-191): the step's active-set
        derivative at the nominal points, one f64 lane per time step; then decouple_AB_matrices if asked
        and c_t = f - A x - B u (irs_lqr_quasistatic.py:218-225)."""
        At, Bt, ct = self._dm.exact_linearize(x_trj, u_trj)
        self._smooth_info = torch.zeros(self.T, dtype=torch.int32, device=x_trj.device)
        if self.decouple_AB:
            f = ct + torch.einsum("tij,tj->ti", At, x_trj[:-1]) + torch.einsum("tij,tj->ti", Bt, u_trj)
            Bt[:, self._idx, :] = torch.eye(self.dim_u, dtype=At.dtype, device=At.device)
            At[:] = torch.eye(self.dim_x, dtype=At.dtype, device=At.device)
            At[:, :, self._idx] = 0.0
            ct = (f - torch.einsum("tij,tj->ti", At, x_trj[:-1]) - torch.einsum("tij,tj->ti", Bt, u_trj)).contiguous()
        return At, Bt, ct

    def _zero_order_AB_dev(self, x_trj, u_trj, std_u):
        """gradient_mode "zero_order_AB" (quasistatic_dynamics.py:268-300): x AND u are perturbed
        (dx ~ N(0, 1e-3), drawn first, as there), `damp`-weighted identity rows regularise the least squares
        (`calc_AB_batch_dev`: accumulate the statistics, all-reduce, damp^2 on the Gram diagonal, solve); then
        decouple_AB_matrices (:275-284) and c_t = f - A x - B u with the decoupled pair, f recovered from the
        undecoupled solve."""
        n, m = self.dim_x, self.dim_u
        At, Bt, ct, self._smooth_info = self.q_dynamics.calc_AB_batch_dev(
            x_trj[:-1], u_trj, self.num_samples, std_u, "zero_order_AB",
            seed=getattr(self.params, "device_rng_seed", None), it=self.current_iter)
        if not self.decouple_AB:
            return At, Bt, ct
        # f = c + A x + B u with the fitted pair; then overwrite the structure and rebuild c
        f = ct + torch.einsum("tij,tj->ti", At, x_trj[:-1]) + torch.einsum("tij,tj->ti", Bt, u_trj)
        eye_m = torch.eye(m, dtype=At.dtype, device=At.device)
        Bt[:, self._idx, :] = eye_m
        At[:] = torch.eye(n, dtype=At.dtype, device=At.device)
        At[:, :, self._idx] = 0.0
        ct = f - torch.einsum("tij,tj->ti", At, x_trj[:-1]) - torch.einsum("tij,tj->ti", Bt, u_trj)
        return At, Bt, ct.contiguous()

    def get_TV_matrices(self, x_trj, u_trj):
        T = u_trj.shape[0]
        assert self.T == T
        At, Bt, ct = self._get_TV_matrices_dev(dev.to_dev(np.asarray(x_trj, float)),
                                               dev.to_dev(np.asarray(u_trj, float)))
        if bool((self._smooth_info != 0).any().item()):
            raise ValueError("randomized-smoothing least squares is rank deficient")
        return At.cpu().numpy(), Bt.cpu().numpy(), ct.cpu().numpy()

    # the worker pool is the GPU: same launch
    get_TV_matrices_batch = get_TV_matrices

    def decouple_AB_matrices(self, At, Bt):
        """irs_lqr_quasistatic.py:275-284 (the device solve already returns this form)."""
        Bt[:, self.indices_u_into_x, :] = np.eye(self.dim_u)
        At[:] = np.eye(At.shape[1])
        At[:, :, self.indices_u_into_x] = 0.0
        return At, Bt

    # ---- irs_lqr_quasistatic.py:286-345 ---------------------------------------
    def _bounds_dev(self, x_trj):
        """Absolute per-time bound rows from the reference's trust-region offsets (:303-325)."""
        def rows(center, b):
            if b is None:
                return None, None
            lo = center + torch.as_tensor(np.asarray(b[0], float), device=center.device)
            hi = center + torch.as_tensor(np.asarray(b[1], float), device=center.device)
            return lo.contiguous(), hi.contiguous()

        x_lo, x_hi = rows(x_trj, self.x_bounds_abs)
        u_lo, u_hi = rows(x_trj[:-1].index_select(1, self._idx), self.u_bounds_abs)
        du_lo, du_hi = rows(torch.zeros((self.T, self.dim_u), dtype=dev.F64, device=x_trj.device),
                            self.u_bounds_rel)
        return x_lo, x_hi, u_lo, u_hi, du_lo, du_hi

    def _local_descent_dev(self, x_trj, u_trj):
        At, Bt, ct = self._get_TV_matrices_dev(x_trj, u_trj)
        p = self.params
        if self._solver in (2, 3) and getattr(self, "_act", None) is None:
            # the active set of the first tail, handed from one iteration's descent to the next: consecutive
            # iterations bind nearly the same bounds (the QP's solution does not depend on the start)
            self._act = torch.zeros((self.T, self.dim_u), dtype=dev.F64, device=x_trj.device)
        o = self._dm.quasistatic_box_descent(At, Bt, ct, self._Q, self._Qd, self._R, self._xd,
                                             x_trj[0].contiguous(), *self._bounds_dev(x_trj),
                                             solver=self._solver, rho=getattr(p, "qp_rho", 100.0),
                                             max_iter=getattr(p, "qp_max_iter", 20000),
                                             eps=getattr(p, "qp_eps", 1e-9),
                                             act=self._act if self._solver in (2, 3) else None)
        self._last = dict(At=At, Bt=Bt, ct=ct, info=o["info"])
        return o["x_new"], o["u_new"], o["cost"]

    def local_descent(self, x_trj, u_trj):
        x_new, u_new, _ = self._local_descent_dev(dev.to_dev(np.asarray(x_trj, float)),
                                                  dev.to_dev(np.asarray(u_trj, float)))
        self._check_last()
        return x_new.cpu().numpy(), u_new.cpu().numpy()

    def _check_last(self):
        info = self._last["info"].cpu().numpy()
        if bool((self._smooth_info != 0).any().item()):
            raise ValueError("randomized-smoothing least squares is rank deficient")
        if info[0] != 0 or info[2] != 0:
            # like solve_tvlqr's `raise ValueError` when the solver fails (tv_lqr.py:139-140)
            raise ValueError("TV_LQR failed. Optimization problem is not solved.")

    # ---- outer loop: the trajectory stays on the device ------------------------------------------------
    def iterate(self, max_iterations):
        """irs_lqr_quasistatic.py:347-390.  A quiet run (verbose = False, no trajectory publishing) enqueues ALL its
        descents -- sample pass, bound rows, the T re-solved tail QPs with the contact dynamics in the loop -- without
        ever waiting for the device: every descent linearises around the device-resident result of the previous one,
        and trajectories, costs and solver flags are read back ONCE at the end, where the reference's bookkeeping
        (history lists, five cost terms, best-so-far, the ValueError of a failed solve) is replayed in order.  A
        verbose run prints a cost per iteration and therefore synchronises per iteration (the base-class loop)."""
        if self.verbose or self.publish_every_iteration:
            return super().iterate(max_iterations)
        state, it0, recs = self._start(), self.current_iter, []
        while True:
            x_new_d, u_new_d, _ = self._local_descent_dev(*state)
            recs.append((x_new_d, u_new_d, self._last["info"], self._smooth_info))
            if self.current_iter > max_iterations:
                break
            state = (x_new_d, u_new_d)
            self.current_iter += 1
        xs = torch.stack([r[0] for r in recs]).cpu().numpy()            # the one read-back
        us = torch.stack([r[1] for r in recs]).cpu().numpy()
        infos = torch.stack([r[2] for r in recs]).cpu().numpy()
        sbad = torch.stack([(r[3] != 0).any() for r in recs]).cpu().numpy()
        self.current_iter = it0
        for i in range(len(recs)):
            if sbad[i]:
                raise ValueError("randomized-smoothing least squares is rank deficient")
            if infos[i][0] != 0 or infos[i][2] != 0:
                raise ValueError("TV_LQR failed. Optimization problem is not solved.")    # tv_lqr.py:139-140
            cost_new = self._log(xs[i], us[i])
            if self.current_iter > max_iterations:
                break
            self.cost, self.x_trj, self.u_trj = cost_new, xs[i], us[i]
            self.current_iter += 1
        return self.x_trj, self.u_trj, self.cost

    def _start(self):
        return dev.to_dev(np.asarray(self.x_trj, float)), dev.to_dev(np.asarray(self.u_trj, float))

    def _descend(self, state):
        x_new_d, u_new_d, _ = self._local_descent_dev(*state)
        self._check_last()
        return x_new_d.cpu().numpy(), u_new_d.cpu().numpy(), (x_new_d, u_new_d)
