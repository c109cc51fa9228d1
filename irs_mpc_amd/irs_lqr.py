"""Host mirror of the reference optimizers, driving the HIP library.

    IrsLqrParameters, IrsLqr            irs_lqr/irs_lqr.py:7-31, :34-218
    IrsLqrZeroOrder                     irs_lqr/irs_lqr_zero_order.py:5-63
    IrsLqrFirstOrder                    irs_lqr/irs_lqr_first_order.py:6-54
    IrsLqrExact                         irs_lqr/irs_lqr_exact.py:6-31

Same constructor signatures, attributes (`x_trj, u_trj, cost, iter, T, x_trj_lst,
u_trj_lst, cost_lst`), methods and error behaviour, so the example scripts run with
only their import root changed.  What differs underneath: one iteration is a handful
of kernel launches (sample pass -> solve -> Riccati -> closed-loop rollout) with the
trajectories resident in HBM; NumPy arrays appear only at the public boundary.

`sampling` may be
  * a Python callable `dx, du = sampling(x_t, u_t, iter)` as in the reference: the
    host draws (identical seeds => identical samples), the device evaluates;
  * a `GaussianSmoothing`: samples are drawn on the device (Philox).
With torch.distributed initialised (world_size > 1) the N samples of every timestep
are sharded over the ranks and the per-timestep sums are all-reduced once per
iteration (replaces zmq_parallel_cmp/array_io.py + the worker processes).
"""
import time

import ctypes

import numpy as np
import torch

from . import device as dev
from . import distributed as dist_util
from ._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_AB
from .sampling import GaussianSmoothing
from .tv_lqr import get_solver


class IrsLqrParameters:
    """irs_lqr/irs_lqr.py:7-31."""

    def __init__(self):
        self.Q = None
        self.Qd = None
        self.R = None
        self.x0 = None
        self.xd_trj = None
        self.u_trj_initial = None
        self.xbound = None
        self.ubound = None
        self.solver_name = "osqp"


class IrsLqr:
    def __init__(self, system, params):
        self.system = system
        self.params = params
        self.check_valid_system(self.system)
        self.check_valid_params(self.params, self.system)

        self.Q = params.Q
        self.Qd = params.Qd
        self.R = params.R
        self.x0 = params.x0
        self.xd_trj = params.xd_trj
        self.u_trj = params.u_trj_initial
        self.xbound = params.xbound
        self.ubound = params.ubound
        self.solver = get_solver(params.solver_name)

        self.T = self.u_trj.shape[0]
        self.dim_x = self.system.dim_x
        self.dim_u = self.system.dim_u

        # device-resident problem data (f64)
        self._dm = system.dm()
        self._Q = dev.to_dev(np.asarray(self.Q, float))
        self._Qd = dev.to_dev(np.asarray(self.Qd, float))
        self._R = dev.to_dev(np.asarray(self.R, float))
        self._x0 = dev.to_dev(np.asarray(self.x0, float))
        self._xd = dev.to_dev(np.asarray(self.xd_trj, float))

        self.x_trj = self.rollout(self.x0, self.u_trj)
        self.cost = self.evaluate_cost(self.x_trj, self.u_trj)

        self.x_trj_lst = [self.x_trj]
        self.u_trj_lst = [self.u_trj]
        self.cost_lst = [self.cost]

        self.start_time = time.time()
        self.iter = 1
        self.verbose = True

    # ---- validation: irs_lqr/irs_lqr.py:73-103 ----------------------------
    def check_valid_system(self, system):
        if system.dim_x == 0:
            raise RuntimeError("System has zero states. Did you forget to set dim_x?")
        elif system.dim_u == 0:
            raise RuntimeError("System has zero inputs. Did you forget to set dim_u?")
        try:
            system.dynamics(np.zeros(system.dim_x), np.zeros(system.dim_u))
        except Exception:
            raise RuntimeError("Could not evaluate dynamics. Have you implemented it?")

    def check_valid_params(self, params, system):
        if params.Q.shape != (system.dim_x, system.dim_x):
            raise RuntimeError("Q matrix must be diagonal with dim_x x dim_x.")
        if params.Qd.shape != (system.dim_x, system.dim_x):
            raise RuntimeError("Qd matrix must be diagonal with dim_x x dim_x.")
        if params.R.shape != (system.dim_u, system.dim_u):
            raise RuntimeError("R matrix must be diagonal with dim_u x dim_u.")

    # ---- irs_lqr/irs_lqr.py:105-137 ----------------------------------------
    def rollout(self, x0, u_trj):
        x_trj, _ = self._dm.rollout_cost(dev.to_dev(np.asarray(x0, float)), dev.to_dev(np.asarray(u_trj, float)),
                                         self._Q, self._R, self._xd)
        return x_trj.cpu().numpy()

    def evaluate_cost(self, x_trj, u_trj):
        cost = dev.evaluate_cost(dev.to_dev(np.asarray(x_trj, float)), dev.to_dev(np.asarray(u_trj, float)),
                                 self._Q, self._R, self._xd)
        return float(cost.item())

    # ---- linearisation -----------------------------------------------------
    def get_TV_matrices(self, x_trj, u_trj):
        At, Bt, ct = self._get_TV_matrices_dev(dev.to_dev(np.asarray(x_trj, float)),
                                               dev.to_dev(np.asarray(u_trj, float)))
        self._check_smooth_info()
        return At.cpu().numpy(), Bt.cpu().numpy(), ct.cpu().numpy()

    def _get_TV_matrices_dev(self, x_trj, u_trj):
        raise NotImplementedError("This class is virtual.")

    # ---- irs_lqr/irs_lqr.py:148-186 ----------------------------------------
    def local_descent(self, x_trj, u_trj):
        x_new, u_new, _ = self._local_descent_dev(dev.to_dev(np.asarray(x_trj, float)),
                                                  dev.to_dev(np.asarray(u_trj, float)))
        return x_new.cpu().numpy(), u_new.cpu().numpy()

    def _local_descent_dev(self, x_trj, u_trj):
        At, Bt, ct = self._get_TV_matrices_dev(x_trj, u_trj)
        # T MPC re-solves of the tail QP == one Riccati pass + closed-loop rollout while no box bound is
        # active in ANY tail's plan; one launch.
        o = self._dm.tvlqr_descent(At, Bt, ct, self._Q, self._Qd, self._R, self._xd, x_trj[0].contiguous(),
                                   alpha_R=0.5)
        self._last = dict(At=At, Bt=Bt, ct=ct, K=o["K"], k=o["k"], info=o["info"])
        self._box_used = False
        box = self._box_bounds()
        if box is None or self._tail_plans_within_bounds(At, Bt, ct, o["K"], o["k"], o["x_new"]):
            return o["x_new"], o["u_new"], o["cost"]
        if not self._dm.box_descent_supported(self.T):
            raise NotImplementedError(
                "a box bound is active and horizon T=%d does not fit the LDS-resident factorisation of the "
                "bounded TV-LQR kernel (tv_lqr.py:112-123)" % self.T)
        # genuine bounds (tv_lqr.py:112-123): T warm-started tail QPs, one launch
        ob = self._dm.tvlqr_box_descent(At, Bt, ct, self._Q, self._Qd, self._R, self._xd,
                                        x_trj[0].contiguous(), *box, alpha_R=0.5,
                                        rho=getattr(self.params, "qp_rho", 10.0),
                                        max_iter=getattr(self.params, "qp_max_iter", 5000),
                                        eps=getattr(self.params, "qp_eps", 1e-8))
        cost = dev.evaluate_cost(ob["x_new"], ob["u_new"], self._Q, self._R, self._xd)
        self._last = dict(At=At, Bt=Bt, ct=ct, K=None, k=None, info=ob["info"][:1], box_info=ob["info"])
        self._box_used = True
        return ob["x_new"], ob["u_new"], cost

    def _box_bounds(self):
        """(xlo, xhi, ulo, uhi) device vectors (+-inf where a component is unbounded), or None when
        `xbound` and `ubound` are absent or infinite throughout.  Every FINITE entry is a genuine bound --
        also the 1e4 / 1e5 the reference's scripts use to say "none": those simply never become active."""
        if getattr(self, "_box_cache", None) is not None:
            return self._box_cache if self._box_cache != () else None

        def vec(b, dim):
            if b is None:
                return np.full(dim, -np.inf), np.full(dim, np.inf)
            return (np.broadcast_to(np.asarray(b[0], float), (dim,)).copy(),
                    np.broadcast_to(np.asarray(b[1], float), (dim,)).copy())

        xlo, xhi = vec(self.xbound, self.dim_x)
        ulo, uhi = vec(self.ubound, self.dim_u)
        self._box_host = (xlo, xhi, ulo, uhi)
        if not any(np.isfinite(v).any() for v in self._box_host):
            self._box_cache = ()
            return None
        self._box_cache = tuple(dev.to_dev(a) for a in self._box_host)
        return self._box_cache

    def _tail_plans_within_bounds(self, At, Bt, ct, K, k, x_new):
        """True iff the unconstrained solution of EVERY tail QP (start t, realised state x_t) respects the
        box: then it is the solution of the bounded QP too (tv_lqr.py:112-123) and the Riccati descent above is
        exact.  By Bellman the unconstrained plan of tail t is the policy (K_s, k_s), s >= t, rolled out on
        the LINEAR model from x_t; all T plans advance together, one batched step per time index."""
        xlo, xhi, ulo, uhi = self._box_host
        A, B, c = At.cpu().numpy(), Bt.cpu().numpy(), ct.cpu().numpy()
        Kh, kh, xs = K.cpu().numpy(), k.cpu().numpy(), x_new.cpu().numpy()
        T = self.T
        X = np.zeros((T, self.dim_x))
        for s in range(T):
            X[s] = xs[s]                                    # tail s starts from the realised state (x_0 of a
            Xa = X[:s + 1]                                  # tail is data, not a decision variable)
            U = Xa @ Kh[s].T + kh[s]
            if (U < ulo).any() or (U > uhi).any():
                return False
            Xa = Xa @ A[s].T + U @ B[s].T + c[s]
            if (Xa < xlo).any() or (Xa > xhi).any():
                return False
            X[:s + 1] = Xa
        return True

    def _check_smooth_info(self):
        info = getattr(self, "_smooth_info", None)
        if info is not None and bool((info != 0).any().item()):
            t = int(torch.nonzero(info)[0].item())
            raise ValueError("randomized-smoothing least squares is rank deficient at t=%d "
                             "(Gram matrix not positive definite; need more samples or a non-zero std)" % t)

    def _check_box_solved(self):
        """Like OSQP hitting its iteration limit (tv_lqr.py:139-140): an unconverged bounded tail QP is a
        failure."""
        if getattr(self, "_box_used", False) and int(self._last["box_info"][2].item()) != 0:
            raise ValueError("TV_LQR failed. Optimization problem is not solved.")

    # ---- irs_lqr/irs_lqr.py:188-218 ----------------------------------------
    def _fused_spec(self):
        """(mode, N, sampling) if the whole loop can run inside the library (irs_iterate): the linearisation needs no
        host closure per time step -- exact, or a GaussianSmoothing drawn on the device -- on one GPU; else None."""
        return None

    def _iterate_fused(self, max_iterations, spec, timing=None):
        """IrsLqr.iterate through ONE C-ABI call (include/irs_hip.h, irs_iterate): every descent of the loop is
        enqueued back to back -- linearise, Riccati + closed-loop rollout + cost, and with finite bounds the plan test
        and the bounded descent behind its device-side flag; the histories come back in one read at the end.  Same
        bookkeeping as the loop below: k + 1 descents, the last one logged but not adopted, errors raised where the
        reference raises them (the lists hold everything up to the failing descent)."""
        from . import _lib
        mode, N, smp = spec
        n_desc = max_iterations - self.iter + 2
        if n_desc <= 0:
            return self.x_trj, self.u_trj, self.cost
        n, m, T, dm = self.dim_x, self.dim_u, self.T, self._dm
        device = self._Q.device
        c = _lib.IterateCall()
        c.model, c.n_params = dm.model_id, dm._np
        for i, v in enumerate(dm.params):
            c.params[i] = v
        c.mode, c.T, c.N, c.n_descents = mode, T, int(N), n_desc
        if smp is not None:
            sx = np.stack([smp.stds(self.iter + i)[0] for i in range(n_desc)]).astype(np.float64)
            su = np.stack([smp.stds(self.iter + i)[1] for i in range(n_desc)]).astype(np.float64)
            sx, su = np.ascontiguousarray(sx), np.ascontiguousarray(su)
            c.std_x, c.std_u = sx.ctypes.data, su.ctypes.data
            c.seed, c.iter0 = int(smp.seed), int(self.iter)
        c.Q, c.Qd, c.R, c.xd_trj = (t_.data_ptr() for t_ in (self._Q, self._Qd, self._R, self._xd))
        c.alpha_R = 0.5
        box = self._box_bounds()
        if box is not None:
            c.xlo, c.xhi, c.ulo, c.uhi = (b.data_ptr() for b in box)
            c.qp_rho = float(getattr(self.params, "qp_rho", 10.0))
            c.qp_max_iter = int(getattr(self.params, "qp_max_iter", 5000))
            c.qp_eps = float(getattr(self.params, "qp_eps", 1e-8))
        x0d, u0d = dev.to_dev(np.asarray(self.x_trj, float)), dev.to_dev(np.asarray(self.u_trj, float))
        c.x_trj0, c.u_trj0 = x0d.data_ptr(), u0d.data_ptr()
        xh = torch.empty((n_desc, T + 1, n), dtype=dev.F64, device=device)
        uh = torch.empty((n_desc, T, m), dtype=dev.F64, device=device)
        ch = torch.empty((n_desc,), dtype=dev.F64, device=device)
        ih = torch.zeros((n_desc, 8), dtype=torch.int32, device=device)
        c.x_hist, c.u_hist, c.cost_hist, c.info_hist = xh.data_ptr(), uh.data_ptr(), ch.data_ptr(), ih.data_ptr()
        need = self._dm.lib.irs_iterate_scratch_bytes(dm.model_id, mode, T, int(N))
        scratch = getattr(self, "_fused_scratch", None)
        if scratch is None or scratch.numel() < need:
            scratch = self._fused_scratch = torch.empty((need,), dtype=torch.uint8, device=device)
        c.scratch, c.scratch_bytes = scratch.data_ptr(), scratch.numel()
        tm = _lib.Timing() if timing is not None else None
        _lib.check(self._dm.lib.irs_iterate(ctypes.byref(c), ctypes.byref(tm) if tm is not None else None, dev._stream()),
                   "irs_iterate")
        if timing is not None:
            timing.update({k: getattr(tm, k) for k, _ in _lib.Timing._fields_})
        xs, us, cs, infos = xh.cpu().numpy(), uh.cpu().numpy(), ch.cpu().numpy(), ih.cpu().numpy()     # the ONE read-back
        self._last = dict(At=None, Bt=None, ct=None, K=None, k=None, info=ih[-1, :1])
        for i in range(n_desc):
            row = infos[i]
            if row[0] != 0:
                raise ValueError("TV_LQR failed. Optimization problem is not solved.")
            if row[1] != 0:
                raise ValueError("randomized-smoothing least squares is rank deficient (Gram matrix not positive "
                                 "definite; need more samples or a non-zero std)")
            if row[6] != 0:
                raise NotImplementedError(
                    "a box bound is active and horizon T=%d does not fit the LDS-resident factorisation of the "
                    "bounded TV-LQR kernel (tv_lqr.py:112-123)" % self.T)
            if row[2] != 0 and row[5] != 0:
                raise ValueError("TV_LQR failed. Optimization problem is not solved.")
            self.x_trj_lst.append(xs[i])
            self.u_trj_lst.append(us[i])
            self.cost_lst.append(float(cs[i]))
            if self.iter > max_iterations:
                break
            self.cost, self.x_trj, self.u_trj = float(cs[i]), xs[i], us[i]
            self.iter += 1
        return self.x_trj, self.u_trj, self.cost

    def iterate(self, max_iterations, timing=None):
        """irs_lqr/irs_lqr.py:188-218: max_iterations+1 descents, the last one logged but
        not adopted.  Quiet runs whose linearisation needs no host closure go through ONE library call
        (`_iterate_fused`); otherwise, per iteration: 2 kernel launches (3 with host-drawn samples' upload)
        and one read-back of (x_new, u_new, cost) for the history lists.  `timing` (a dict, fused path only):
        filled with the library's per-phase device times (irs_timing)."""
        spec = self._fused_spec()
        if spec is not None and not self.verbose:
            return self._iterate_fused(max_iterations, spec, timing)
        x_dev = dev.to_dev(np.asarray(self.x_trj, float))
        u_dev = dev.to_dev(np.asarray(self.u_trj, float))
        while True:
            x_new_d, u_new_d, cost_d = self._local_descent_dev(x_dev, u_dev)
            x_trj_new = x_new_d.cpu().numpy()
            u_trj_new = u_new_d.cpu().numpy()
            cost_new = float(cost_d.item())
            if int(self._last["info"].item()) != 0:
                raise ValueError("TV_LQR failed. Optimization problem is not solved.")
            self._check_smooth_info()
            self._check_box_solved()

            if self.verbose:
                print("Iteration: {:02d} ".format(self.iter) + " || " +
                      "Current Cost: {0:05f} ".format(cost_new) + " || " +
                      "Elapsed time: {0:05f} ".format(time.time() - self.start_time))

            self.x_trj_lst.append(x_trj_new)
            self.u_trj_lst.append(u_trj_new)
            self.cost_lst.append(cost_new)

            if self.iter > max_iterations:
                break

            self.cost = cost_new
            self.x_trj = x_trj_new
            self.u_trj = u_trj_new
            x_dev, u_dev = x_new_d, u_new_d
            self.iter += 1

        return self.x_trj, self.u_trj, self.cost


class _IrsLqrSampled(IrsLqr):
    MODE = None

    def __init__(self, system, params, sampling):
        super().__init__(system, params)
        self.sampling = sampling

    def _draw_host(self, x_trj, u_trj):
        """The reference's per-timestep closure calls (irs_lqr_zero_order.py:50),
        in the same order, stacked to (T,N,n) / (T,N,m)."""
        xh = x_trj.cpu().numpy()
        uh = u_trj.cpu().numpy()
        dxs, dus = [], []
        for t in range(self.T):
            dx, du = self.sampling(xh[t], uh[t], self.iter)
            dxs.append(np.asarray(dx, np.float32))
            dus.append(np.asarray(du, np.float32))
        return np.stack(dxs), np.stack(dus)

    def _fused_spec(self):
        on_device = isinstance(self.sampling, GaussianSmoothing) and getattr(self.sampling, "on_device", True)
        if not on_device or dist_util.rank_world()[1] != 1:
            return None
        return self.MODE, self.sampling.num_samples, self.sampling

    def _get_TV_matrices_dev(self, x_trj, u_trj):
        rank, world = dist_util.rank_world()
        on_device = isinstance(self.sampling, GaussianSmoothing) and getattr(self.sampling, "on_device", True)
        if world == 1:
            # single GPU: sample pass + reduction + solve in ONE launch
            if on_device:
                sx, su = self.sampling.stds(self.iter)
                o = self._dm.smooth_rng(self.MODE, x_trj, u_trj, self.sampling.num_samples, sx, su,
                                        self.sampling.seed, self.iter)
            else:
                dx, du = self._draw_host(x_trj, u_trj)
                o = self._dm.smooth(self.MODE, x_trj, u_trj, dev.to_dev(dx, dev.F32), dev.to_dev(du, dev.F32))
            self._smooth_info = o["info"]
            return o["At"], o["Bt"], o["ct"]
        if on_device:
            N = self.sampling.num_samples
            lo, hi = dist_util.shard_range(N, rank, world)
            sx, su = self.sampling.stds(self.iter)
            sums = self._dm.smooth_accumulate_rng(self.MODE, x_trj, u_trj, hi - lo, sx, su,
                                                  self.sampling.seed, self.iter, sample_offset=lo)
        else:
            dx, du = self._draw_host(x_trj, u_trj)
            N = du.shape[1]
            lo, hi = dist_util.shard_range(N, rank, world)
            dxd = dev.to_dev(np.ascontiguousarray(dx[:, lo:hi]), dev.F32)
            dud = dev.to_dev(np.ascontiguousarray(du[:, lo:hi]), dev.F32)
            sums = self._dm.smooth_accumulate(self.MODE, x_trj, u_trj, dxd, dud)
        dist_util.all_reduce_sums(sums)
        At, Bt, ct, info = self._dm.smooth_finalize(self.MODE, N, x_trj, u_trj, sums)
        self._smooth_info = info
        return At, Bt, ct


class IrsLqrZeroOrder(_IrsLqrSampled):
    """irs_lqr/irs_lqr_zero_order.py:5-63."""
    MODE = SMOOTH_ZERO_ORDER_AB

    def compute_least_squares(self, dxdu, deltaf):
        """irs_lqr_zero_order.py:27-36, for callers that use it stand-alone: ABhat = lstsq(dxdu, deltaf)[0]',
        split into (Ahat (n,n), Bhat (n,m)) -- normal equations + Jacobi-scaled Cholesky on the device
        (irs_least_squares), the solve `get_TV_matrices` ends with."""
        Z = dev.to_dev(np.asarray(dxdu, float))
        dF = dev.to_dev(np.asarray(deltaf, float))
        n, m = self.dim_x, self.dim_u
        assert Z.shape[1] == n + m and dF.shape == (Z.shape[0], n)
        A = torch.empty((n, n), dtype=dev.F64, device=Z.device)
        B = torch.empty((n, m), dtype=dev.F64, device=Z.device)
        info = torch.empty((1,), dtype=torch.int32, device=Z.device)
        from ._lib import check, load
        check(load().irs_least_squares(n, m, Z.shape[0], Z.data_ptr(), dF.data_ptr(), A.data_ptr(), B.data_ptr(),
                                       info.data_ptr(), dev._stream()), "irs_least_squares")
        if int(info.item()) != 0:
            raise ValueError("randomized-smoothing least squares is rank deficient (or has non-finite data)")
        return A.cpu().numpy(), B.cpu().numpy()


class IrsLqrFirstOrder(_IrsLqrSampled):
    """irs_lqr/irs_lqr_first_order.py:6-54."""
    MODE = SMOOTH_FIRST_ORDER


class IrsLqrExact(IrsLqr):
    """irs_lqr/irs_lqr_exact.py:6-31."""

    def _fused_spec(self):
        from ._lib import ITERATE_EXACT
        return ITERATE_EXACT, 0, None

    def _get_TV_matrices_dev(self, x_trj, u_trj):
        return self._dm.exact_linearize(x_trj, u_trj)
