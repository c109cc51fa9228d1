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
    quasistatic_simulator; gradient modes that need the simulator's analytic derivatives
    ("first_order", "exact", and "zero_order_B" without decouple_AB) raise NotImplementedError.

`params.sampling(std_u_initial, iter)` returns the std of the u-perturbations like the reference;
the draws are made on the host by `np.random.normal(0, std_u, (num_samples, dim_u))` once per
time step in time order -- exactly the reference's calls (quasistatic_dynamics.py:258), so
identical seeds give identical samples -- unless `params.device_rng_seed` is set, in which case the
perturbations are drawn on the device (Philox) and never touch the host.
"""
import time

import numpy as np
import torch

from . import device as dev
from . import distributed as dist_util
from ._lib import SMOOTH_ZERO_ORDER_B
from .tv_lqr import get_solver


def quasistatic_eval_cost(q_dynamics, x_trj, u_trj, x_trj_d, Q_dict, Qd_dict, R):
    """irs_lqr_quasistatic.py:153-194 (= cem_quasistatic.py:124-165): the five cost terms
    (unactuated / actuated, running / final, input-rate), vectorised over time -- O(T n)
    bookkeeping on trajectories already on the host."""
    qd = q_dynamics
    idx = qd.get_u_indices_into_x()
    e = np.asarray(x_trj, float) - np.asarray(x_trj_d, float)

    def q_cost(models, rows, Q_dict_):
        c = 0.
        for model in models:
            ei = rows[..., qd.position_indices[model]]
            c += float((ei * np.asarray(Q_dict_[model], float) * ei).sum())
        return c

    cost_Qu_final = q_cost(qd.models_unactuated, e[-1], Qd_dict)
    cost_Qa_final = q_cost(qd.models_actuated, e[-1], Qd_dict)
    cost_Qu = q_cost(qd.models_unactuated, e[:-1], Q_dict)
    cost_Qa = q_cost(qd.models_actuated, e[:-1], Q_dict)
    du = np.diff(np.vstack([np.asarray(x_trj)[0, idx][None], np.asarray(u_trj)]), axis=0)
    cost_R = float(np.einsum("ti,ij,tj->", du, R, du))
    return cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final, cost_R


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
        self.qp_solver = 0              # 0 auto, 1 ADMM, 2 active set (include/irs_hip.h)
        self.qp_rho = 100.0             # ADMM penalty / iteration limit / tolerance of the bounded QPs
        self.qp_max_iter = 20000
        self.qp_eps = 1e-9


class IrsLqrQuasistatic:
    def __init__(self, q_dynamics, params):
        self.q_dynamics = q_dynamics
        self.dim_x = q_dynamics.dim_x
        self.dim_u = q_dynamics.dim_u

        self.params = params

        self.T = params.T
        self.x0 = params.x0
        self.Q_dict = params.Q_dict
        self.Q = self.q_dynamics.get_Q_from_Q_dict(self.Q_dict)
        self.Qd_dict = params.Qd_dict
        self.Qd = self.q_dynamics.get_Q_from_Q_dict(self.Qd_dict)
        self.R_dict = params.R_dict
        self.R = self.q_dynamics.get_R_from_R_dict(self.R_dict)
        self.x_trj_d = params.x_trj_d
        self.u_trj_0 = params.u_trj_0
        self.x_bounds_abs = params.x_bounds_abs
        self.u_bounds_abs = params.u_bounds_abs
        self.x_bounds_rel = params.x_bounds_rel
        self.u_bounds_rel = params.u_bounds_rel
        self.indices_u_into_x = q_dynamics.get_u_indices_into_x()

        self.decouple_AB = params.decouple_AB
        self.use_workers = params.use_workers
        self.gradient_mode = params.gradient_mode
        self.task_stride = params.task_stride
        self.publish_every_iteration = params.publish_every_iteration

        if self.gradient_mode != "zero_order_B" or not self.decouple_AB:
            raise NotImplementedError(
                "gradient_mode=%r with decouple_AB=%r needs the quasistatic simulator's analytic "
                "derivatives (q_sim.get_Dq_nextDq); the device functors provide zero_order_B with "
                "decouple_AB=True" % (self.gradient_mode, self.decouple_AB))
        if self.x_bounds_rel is not None:
            raise NotImplementedError("x_bounds_rel ('should be rarely used', irs_lqr_quasistatic.py:315) "
                                      "is not implemented on the device")

        # device-resident problem data (f64)
        self._dm = q_dynamics.dm()
        one_box = self.x_bounds_abs is None and (self.u_bounds_abs is None or self.u_bounds_rel is None)
        self._solver = int(getattr(params, "qp_solver", 0))
        if self._solver == 0:
            self._solver = 2 if one_box and self._dm.quasistatic_descent_supported(self.T, 2) else 1
        if not self._dm.quasistatic_descent_supported(self.T, self._solver):
            raise NotImplementedError("horizon T=%d does not fit the LDS-resident QP factorisation" % self.T)
        self._Q, self._Qd, self._R = (dev.to_dev(np.asarray(a, float)) for a in (self.Q, self.Qd, self.R))
        self._x0 = dev.to_dev(np.asarray(self.x0, float))
        self._xd = dev.to_dev(np.asarray(self.x_trj_d, float))
        self._idx = torch.as_tensor(np.asarray(self.indices_u_into_x), device=self._x0.device)

        self.x_trj = self.rollout(self.x0, self.u_trj_0)
        self.u_trj = self.u_trj_0  # T x m

        (cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final,
         cost_R) = self.eval_cost(self.x_trj, self.u_trj)
        self.cost = cost_Qu + cost_Qu_final + cost_Qa + cost_Qa_final + cost_R

        self.x_trj_best = None
        self.u_trj_best = None
        self.cost_best = np.inf

        # sampling standard deviation.
        self.std_u_initial = params.std_u_initial
        self.sampling = params.sampling
        self.num_samples = params.num_samples

        # logging
        self.x_trj_list = [self.x_trj]
        self.u_trj_list = [self.u_trj]

        self.cost_all_list = [self.cost]
        self.cost_Qu_list = [cost_Qu]
        self.cost_Qu_final_list = [cost_Qu_final]
        self.cost_Qa_list = [cost_Qa]
        self.cost_Qa_final_list = [cost_Qa_final]
        self.cost_R_list = [cost_R]

        self.current_iter = 1
        self.start_time = time.time()
        self.verbose = True

        # solver: kept for interface parity; the bounded QPs are solved on the device
        self.solver = get_solver(params.solver_name)

    # ---- irs_lqr_quasistatic.py:133-140 ---------------------------------------
    def rollout(self, x0, u_trj):
        T = u_trj.shape[0]
        assert T == self.T
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    # ---- irs_lqr_quasistatic.py:142-194 ---------------------------------------
    @staticmethod
    def calc_Q_cost(models_list, x_dict, xd_dict, Q_dict):
        cost = 0.
        for model in models_list:
            dx_i = x_dict[model] - xd_dict[model]
            cost += (dx_i * Q_dict[model] * dx_i).sum()
        return cost

    def eval_cost(self, x_trj, u_trj):
        T = u_trj.shape[0]
        assert T == self.T and x_trj.shape[0] == T + 1
        return quasistatic_eval_cost(self.q_dynamics, x_trj, u_trj, self.x_trj_d, self.Q_dict, self.Qd_dict, self.R)

    # ---- linearisation: irs_lqr_quasistatic.py:196-273 ------------------------
    def _get_TV_matrices_dev(self, x_trj, u_trj):
        std_u = np.broadcast_to(np.asarray(self.sampling(self.std_u_initial, self.current_iter), float),
                                (self.dim_u,))
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
                o = self._dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
            else:
                o = self._dm.smooth_rng(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, N, None, std_u, int(seed),
                                        self.current_iter)
            self._smooth_info = o["info"]
            return o["At"], o["Bt"], o["ct"]
        if seed is None:
            sums = self._dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
        else:
            sums = self._dm.smooth_accumulate_rng(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, hi - lo, None, std_u,
                                                  int(seed), self.current_iter, sample_offset=lo)
        dist_util.all_reduce_sums(sums)
        At, Bt, ct, info = self._dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, x_trj, u_trj, sums)
        self._smooth_info = info
        return At, Bt, ct

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
        o = self._dm.quasistatic_box_descent(At, Bt, ct, self._Q, self._Qd, self._R, self._xd,
                                             x_trj[0].contiguous(), *self._bounds_dev(x_trj),
                                             solver=self._solver, rho=getattr(p, "qp_rho", 100.0),
                                             max_iter=getattr(p, "qp_max_iter", 20000),
                                             eps=getattr(p, "qp_eps", 1e-9))
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

    # ---- irs_lqr_quasistatic.py:347-390 ---------------------------------------
    def iterate(self, max_iterations):
        x_dev = dev.to_dev(np.asarray(self.x_trj, float))
        u_dev = dev.to_dev(np.asarray(self.u_trj, float))
        while True:
            if self.verbose:
                print('Iter {:02d},'.format(self.current_iter),
                      'cost: {:0.4f}.'.format(self.cost),
                      'time: {:0.2f}.'.format(time.time() - self.start_time))

            x_new_d, u_new_d, _ = self._local_descent_dev(x_dev, u_dev)
            self._check_last()
            x_trj_new, u_trj_new = x_new_d.cpu().numpy(), u_new_d.cpu().numpy()
            (cost_Qu, cost_Qu_final, cost_Qa, cost_Qa_final,
             cost_R) = self.eval_cost(x_trj_new, u_trj_new)
            cost = cost_Qu + cost_Qu_final + cost_Qa + cost_Qa_final + cost_R
            self.x_trj_list.append(x_trj_new)
            self.u_trj_list.append(u_trj_new)
            self.cost_Qu_list.append(cost_Qu)
            self.cost_Qu_final_list.append(cost_Qu_final)
            self.cost_Qa_list.append(cost_Qa)
            self.cost_Qa_final_list.append(cost_Qa_final)
            self.cost_R_list.append(cost_R)
            self.cost_all_list.append(cost)

            if self.publish_every_iteration:
                self.q_dynamics.publish_trajectory(x_trj_new)

            if self.cost_best > cost:
                self.x_trj_best = x_trj_new
                self.u_trj_best = u_trj_new
                self.cost_best = cost

            if self.current_iter > max_iterations:
                break

            # Go over to next iteration.
            self.cost = cost
            self.x_trj = x_trj_new
            self.u_trj = u_trj_new
            x_dev, u_dev = x_new_d, u_new_d
            self.current_iter += 1

        return self.x_trj, self.u_trj, self.cost
