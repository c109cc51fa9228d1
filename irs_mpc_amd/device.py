"""Thin tensor-level wrappers over the C ABI.  PyTorch is used for device memory,
the current HIP stream and (elsewhere) torch.distributed -- nothing else.

Every function takes/returns torch tensors that live on the GPU; nothing here
synchronises with the host.
"""
import ctypes

import torch

from . import _lib
from ._lib import (SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_AB, SMOOTH_ZERO_ORDER_B,
                   check, dbl_array)

F64 = torch.float64
F32 = torch.float32


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("irs_mpc_amd needs an AMD GPU (no CPU fallback exists).")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t, dtype):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def to_dev(a, dtype=F64):
    """numpy / tensor -> contiguous device tensor of `dtype`."""
    dev = require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device=dev, dtype=dtype).contiguous()
    return torch.as_tensor(a).to(device=dev, dtype=dtype).contiguous()


class DeviceModel:
    """A registered device functor (irs_model_id) with its constants bound."""

    def __init__(self, model_id, params):
        self.lib = _lib.load()
        self.model_id = int(model_id)
        self.params = [float(p) for p in params]
        n, m, npar = (_lib.c_int(), _lib.c_int(), _lib.c_int())
        check(self.lib.irs_model_info(self.model_id, n, m, npar), "irs_model_info")
        self.n, self.m = n.value, m.value
        if npar.value != len(self.params):
            raise ValueError("model %d expects %d params, got %d" % (model_id, npar.value, len(self.params)))
        self._p = dbl_array(self.params)
        self._np = len(self.params)
        self._ws = {}

    # ---- DynamicalSystem plugin surface -----------------------------------
    def dynamics_batch(self, X, U):
        B = X.shape[0]
        Xn = torch.empty((B, self.n), dtype=F64, device=X.device)
        check(self.lib.irs_dynamics_batch(self.model_id, self._p, self._np, _ptr(X, F64), _ptr(U, F64),
                                          B, _ptr(Xn, F64), _stream()), "irs_dynamics_batch")
        return Xn

    def jacobian_xu_batch(self, X, U):
        B = X.shape[0]
        J = torch.empty((B, self.n, self.n + self.m), dtype=F64, device=X.device)
        check(self.lib.irs_jacobian_xu_batch(self.model_id, self._p, self._np, _ptr(X, F64), _ptr(U, F64),
                                             B, _ptr(J, F64), _stream()), "irs_jacobian_xu_batch")
        return J

    def contact_samples_f32(self, x, u, du):
        """Per-sample f32 lanes of the FIRST_ORDER sample pass of a contact model (irs_contact_samples_f32):
        x (n), u (m) f64, du (B,m) f32 -> Xn (B,n) f32, Bs (B,n,m) f32, active_mask (B) i32."""
        B = du.shape[0]
        Xn = torch.empty((B, self.n), dtype=F32, device=du.device)
        Bs = torch.empty((B, self.n, self.m), dtype=F32, device=du.device)
        mask = torch.empty((B,), dtype=torch.int32, device=du.device)
        check(self.lib.irs_contact_samples_f32(self.model_id, self._p, self._np, _ptr(x, F64), _ptr(u, F64),
                                               _ptr(du, F32), B, _ptr(Xn, F32), _ptr(Bs, F32), mask.data_ptr(),
                                               _stream()), "irs_contact_samples_f32")
        return Xn, Bs, mask

    def rollout_cost(self, x0, u_trj, Q, R, xd_trj):
        T = u_trj.shape[0]
        x_trj = torch.empty((T + 1, self.n), dtype=F64, device=u_trj.device)
        cost = torch.empty((1,), dtype=F64, device=u_trj.device)
        check(self.lib.irs_rollout_cost(self.model_id, self._p, self._np, T, _ptr(x0, F64), _ptr(u_trj, F64),
                                        _ptr(Q, F64), _ptr(R, F64), _ptr(xd_trj, F64), _ptr(x_trj, F64),
                                        _ptr(cost, F64), _stream()), "irs_rollout_cost")
        return x_trj, cost

    # ---- smoothing --------------------------------------------------------
    def sums_len(self, mode):
        return self.lib.irs_sums_len(self.model_id, mode)

    def _workspace(self, mode, T, N, device):
        need = self.lib.irs_smooth_workspace_bytes(self.model_id, mode, T, N)
        key = (mode, device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty((max(need, 8192),), dtype=torch.uint8, device=device)
            check(self.lib.irs_workspace_init(ws.data_ptr(), ws.numel(), _stream()), "irs_workspace_init")
            self._ws[key] = ws
        return ws

    def _tv_outputs(self, T, device, out):
        if out is not None:
            return out
        return dict(sums=None,
                    At=torch.empty((T, self.n, self.n), dtype=F64, device=device),
                    Bt=torch.empty((T, self.n, self.m), dtype=F64, device=device),
                    ct=torch.empty((T, self.n), dtype=F64, device=device),
                    info=torch.empty((T,), dtype=torch.int32, device=device))

    def smooth(self, mode, x_trj, u_trj, dx, du, out=None):
        """Whole get_TV_matrices in one launch (single GPU), samples supplied.
        Returns dict(sums, At, Bt, ct, info); pass `out` (a previous result) to reuse buffers."""
        T, N = du.shape[0], du.shape[1]
        o = self._tv_outputs(T, du.device, out)
        if o["sums"] is None:
            o["sums"] = torch.empty((T, self.sums_len(mode)), dtype=F64, device=du.device)
        ws = self._workspace(mode, T, N, du.device)
        check(self.lib.irs_smooth(self.model_id, self._p, self._np, mode, T, N, _ptr(x_trj, F64),
                                  _ptr(u_trj, F64), _ptr(dx, F32), _ptr(du, F32), _ptr(o["sums"], F64),
                                  _ptr(o["At"], F64), _ptr(o["Bt"], F64), _ptr(o["ct"], F64),
                                  o["info"].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "irs_smooth")
        return o

    def smooth_rng(self, mode, x_trj, u_trj, N, std_x, std_u, seed, it, out=None):
        T = u_trj.shape[0]
        o = self._tv_outputs(T, u_trj.device, out)
        if o["sums"] is None:
            o["sums"] = torch.empty((T, self.sums_len(mode)), dtype=F64, device=u_trj.device)
        ws = self._workspace(mode, T, N, u_trj.device)
        sx = dbl_array(std_x) if std_x is not None else None
        check(self.lib.irs_smooth_rng(self.model_id, self._p, self._np, mode, T, N, _ptr(x_trj, F64),
                                      _ptr(u_trj, F64), sx, dbl_array(std_u), int(seed), int(it),
                                      _ptr(o["sums"], F64), _ptr(o["At"], F64), _ptr(o["Bt"], F64),
                                      _ptr(o["ct"], F64), o["info"].data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream()), "irs_smooth_rng")
        return o

    def tvlqr_descent(self, At, Bt, ct, Q, Qd, R, xd_trj, x0, alpha_R=0.5, out=None):
        """Riccati backward pass + closed-loop rollout + cost in one launch."""
        T = At.shape[0]
        dev = At.device
        o = out
        if o is None:
            o = dict(K=torch.empty((T, self.m, self.n), dtype=F64, device=dev),
                     k=torch.empty((T, self.m), dtype=F64, device=dev),
                     x_new=torch.empty((T + 1, self.n), dtype=F64, device=dev),
                     u_new=torch.empty((T, self.m), dtype=F64, device=dev),
                     cost=torch.empty((1,), dtype=F64, device=dev),
                     info=torch.empty((1,), dtype=torch.int32, device=dev))
        check(self.lib.irs_tvlqr_descent(self.model_id, self._p, self._np, T, _ptr(At, F64), _ptr(Bt, F64),
                                         _ptr(ct, F64), _ptr(Q, F64), _ptr(Qd, F64), _ptr(R, F64),
                                         float(alpha_R), _ptr(xd_trj, F64), _ptr(x0, F64), _ptr(o["K"], F64),
                                         _ptr(o["k"], F64), _ptr(o["x_new"], F64), _ptr(o["u_new"], F64),
                                         _ptr(o["cost"], F64), o["info"].data_ptr(), _stream()),
              "irs_tvlqr_descent")
        return o

    # ---- box-constrained descent ----------------------------------------------
    BOX_LDS_LIMIT = 160 * 1024 - 512

    def box_descent_supported(self, T):
        return 0 < self.lib.irs_tvlqr_box_lds_bytes(self.model_id, int(T)) <= self.BOX_LDS_LIMIT

    def tvlqr_box_descent(self, At, Bt, ct, Q, Qd, R, xd_trj, x0, xlo, xhi, ulo, uhi, alpha_R=0.5,
                          rho=10.0, relax=1.6, max_iter=5000, eps=1e-8):
        """local_descent with active abs bounds (T warm-started tail QPs by ADMM around one
        Riccati factorisation).  Returns dict(x_new, u_new, info[3])."""
        T = At.shape[0]
        dev = At.device
        o = dict(x_new=torch.empty((T + 1, self.n), dtype=F64, device=dev),
                 u_new=torch.empty((T, self.m), dtype=F64, device=dev),
                 info=torch.empty((3,), dtype=torch.int32, device=dev))
        check(self.lib.irs_tvlqr_box_descent(self.model_id, self._p, self._np, T, _ptr(At, F64), _ptr(Bt, F64),
                                             _ptr(ct, F64), _ptr(Q, F64), _ptr(Qd, F64), _ptr(R, F64),
                                             float(alpha_R), _ptr(xd_trj, F64), _ptr(x0, F64), _ptr(xlo, F64),
                                             _ptr(xhi, F64), _ptr(ulo, F64), _ptr(uhi, F64), float(rho),
                                             float(relax), int(max_iter), float(eps), _ptr(o["x_new"], F64),
                                             _ptr(o["u_new"], F64), o["info"].data_ptr(), _stream()),
              "irs_tvlqr_box_descent")
        return o

    SOLVER_AUTO, SOLVER_ADMM, SOLVER_ACTIVE_SET, SOLVER_ACTIVE_SET_MFMA = 0, 1, 2, 3

    def quasistatic_descent_supported(self, T, solver=1):
        """Whether `solver` can run horizon T (3: always, for models that fit the matrix-core tile -- beyond
        the LDS-resident size its records go to a workspace in HBM; 1, 2: while their data fit LDS)."""
        lds = self.lib.irs_quasistatic_box_lds_bytes(self.model_id, int(T), int(solver))
        if int(solver) == 3:
            return lds > 0
        return 0 < lds <= self.BOX_LDS_LIMIT

    def _descent_workspace(self, T, solver, device):
        need = self.lib.irs_quasistatic_descent_workspace_bytes(self.model_id, int(T), int(solver))
        if need == 0:
            return None
        ws = self._ws.get(("descent", device))
        if ws is None or ws.numel() < need:
            ws = torch.empty((need,), dtype=torch.uint8, device=device)
            self._ws[("descent", device)] = ws
        return ws

    def quasistatic_box_descent(self, At, Bt, ct, Q, Qd, R, xd_trj, x0, x_lo=None, x_hi=None, u_lo=None,
                                u_hi=None, du_lo=None, du_hi=None, solver=0, rho=10.0, relax=1.6,
                                max_iter=5000, eps=1e-8, out=None, act=None):
        """IrsLqrQuasistatic.local_descent after get_TV_matrices (irs_lqr_quasistatic.py:286-345) +
        eval_cost.  Bounds are absolute per-time rows ((T+1,n) / (T,m)) or None.  solver: 0 auto,
        1 ADMM, 2 active set (one control box, no x bounds; lanes, LDS-resident), 3 the same on matrix-core
        tiles (any horizon).  `act` (T,m) f64 in {-1,0,+1}, in/out: the active set the first tail starts
        from / converged to (hand it from one iteration's descent to the next; zeros = cold start).
        Returns dict(x_new, u_new, cost, info[3])."""
        T = At.shape[0]
        dev = At.device
        o = out
        if o is None:
            o = dict(x_new=torch.empty((T + 1, self.n), dtype=F64, device=dev),
                     u_new=torch.empty((T, self.m), dtype=F64, device=dev),
                     cost=torch.empty((1,), dtype=F64, device=dev),
                     info=torch.empty((3,), dtype=torch.int32, device=dev))
        for b, shape in ((x_lo, (T + 1, self.n)), (x_hi, (T + 1, self.n)), (u_lo, (T, self.m)),
                         (u_hi, (T, self.m)), (du_lo, (T, self.m)), (du_hi, (T, self.m))):
            assert b is None or tuple(b.shape) == shape, (tuple(b.shape), shape)
        assert act is None or tuple(act.shape) == (T, self.m)
        ws = self._descent_workspace(T, solver, dev) if int(solver) in (0, 3) else None
        check(self.lib.irs_quasistatic_box_descent_wsx(
            self.model_id, self._p, self._np, T, _ptr(At, F64), _ptr(Bt, F64), _ptr(ct, F64), _ptr(Q, F64),
            _ptr(Qd, F64), _ptr(R, F64), _ptr(xd_trj, F64), _ptr(x0, F64), _ptr(x_lo, F64), _ptr(x_hi, F64),
            _ptr(u_lo, F64), _ptr(u_hi, F64), _ptr(du_lo, F64), _ptr(du_hi, F64), int(solver), float(rho),
            float(relax), int(max_iter), float(eps), _ptr(o["x_new"], F64), _ptr(o["u_new"], F64), _ptr(o["cost"], F64),
            o["info"].data_ptr(), _ptr(act, F64), ws.data_ptr() if ws is not None else None,
            ws.numel() if ws is not None else 0, _stream()), "irs_quasistatic_box_descent_wsx")
        return o

    # ---- CEM baseline -------------------------------------------------------
    def cem_rollout_costs(self, u_cand, x0, Q, R, xd_trj):
        """costs (B) of the B candidate sequences u_cand (B,T,m): rollout + evaluate_cost each."""
        B, T = u_cand.shape[0], u_cand.shape[1]
        costs = torch.empty((B,), dtype=F64, device=u_cand.device)
        check(self.lib.irs_cem_rollout_costs(self.model_id, self._p, self._np, T, B, _ptr(u_cand, F64),
                                             _ptr(x0, F64), _ptr(Q, F64), _ptr(R, F64), _ptr(xd_trj, F64),
                                             _ptr(costs, F64), _stream()), "irs_cem_rollout_costs")
        return costs

    def cem_rollout_costs_quasistatic(self, u_cand, x0, Q, Qd, R, xd_trj):
        """costs (B) with the quasistatic eval_cost (du input cost, terminal Qd)."""
        B, T = u_cand.shape[0], u_cand.shape[1]
        costs = torch.empty((B,), dtype=F64, device=u_cand.device)
        check(self.lib.irs_cem_rollout_costs_quasistatic(self.model_id, self._p, self._np, T, B, _ptr(u_cand, F64),
                                                         _ptr(x0, F64), _ptr(Q, F64), _ptr(Qd, F64), _ptr(R, F64),
                                                         _ptr(xd_trj, F64), _ptr(costs, F64), _stream()),
              "irs_cem_rollout_costs_quasistatic")
        return costs

    def cem_refit(self, u_cand, costs, n_elite):
        """Elite selection + mean/std refit: returns elite_idx (n_elite), u_new (T,m), std_new (T,m)."""
        B, T, m = u_cand.shape
        dev = u_cand.device
        idx = torch.empty((n_elite,), dtype=torch.int32, device=dev)
        u_new = torch.empty((T, m), dtype=F64, device=dev)
        std_new = torch.empty((T, m), dtype=F64, device=dev)
        check(self.lib.irs_cem_refit(T, m, B, int(n_elite), _ptr(u_cand, F64), _ptr(costs, F64), idx.data_ptr(),
                                     _ptr(u_new, F64), _ptr(std_new, F64), _stream()), "irs_cem_refit")
        return idx, u_new, std_new

    def smooth_accumulate(self, mode, x_trj, u_trj, dx, du, sums=None):
        """Sample pass on supplied samples: dx (T,N,n) f32 (None for ZERO_ORDER_B), du (T,N,m) f32."""
        T, N = du.shape[0], du.shape[1]
        if sums is None:
            sums = torch.empty((T, self.sums_len(mode)), dtype=F64, device=du.device)
        ws = self._workspace(mode, T, N, du.device)
        check(self.lib.irs_smooth_accumulate(self.model_id, self._p, self._np, mode, T, N,
                                             _ptr(x_trj, F64), _ptr(u_trj, F64), _ptr(dx, F32), _ptr(du, F32),
                                             _ptr(sums, F64), ws.data_ptr(), ws.numel(), _stream()),
              "irs_smooth_accumulate")
        return sums

    def smooth_accumulate_rng(self, mode, x_trj, u_trj, N, std_x, std_u, seed, it, sample_offset=0, sums=None):
        T = u_trj.shape[0]
        if sums is None:
            sums = torch.empty((T, self.sums_len(mode)), dtype=F64, device=u_trj.device)
        ws = self._workspace(mode, T, N, u_trj.device)
        sx = dbl_array(std_x) if std_x is not None else None
        check(self.lib.irs_smooth_accumulate_rng(self.model_id, self._p, self._np, mode, T, N,
                                                 _ptr(x_trj, F64), _ptr(u_trj, F64), sx, dbl_array(std_u),
                                                 int(seed), int(it), int(sample_offset), _ptr(sums, F64),
                                                 ws.data_ptr(), ws.numel(), _stream()),
              "irs_smooth_accumulate_rng")
        return sums

    def rng_samples(self, T, N, std_x, std_u, seed, it, sample_offset=0):
        dev = require_gpu()
        dx = torch.empty((T, N, self.n), dtype=F32, device=dev)
        du = torch.empty((T, N, self.m), dtype=F32, device=dev)
        check(self.lib.irs_rng_samples(self.n, self.m, T, N, dbl_array(std_x), dbl_array(std_u), int(seed),
                                       int(it), int(sample_offset), _ptr(dx, F32), _ptr(du, F32), _stream()),
              "irs_rng_samples")
        return dx, du

    def smooth_finalize(self, mode, N_total, x_trj, u_trj, sums, out=None, workspace=None):
        """Solve step on (all-reduced) sums -> (At, Bt, ct, info); `out` = a previous result to reuse;
        `workspace` = the workspace tensor of the accumulate call that produced this rank's sums
        (contact models: its f64 nominal steps are reused)."""
        T = u_trj.shape[0]
        dev = u_trj.device
        if out is not None:
            At, Bt, ct, info = out
        else:
            At = torch.empty((T, self.n, self.n), dtype=F64, device=dev)
            Bt = torch.empty((T, self.n, self.m), dtype=F64, device=dev)
            ct = torch.empty((T, self.n), dtype=F64, device=dev)
            info = torch.empty((T,), dtype=torch.int32, device=dev)
        check(self.lib.irs_smooth_finalize_ws(self.model_id, self._p, self._np, mode, T, int(N_total),
                                              _ptr(x_trj, F64), _ptr(u_trj, F64), _ptr(sums, F64), _ptr(At, F64),
                                              _ptr(Bt, F64), _ptr(ct, F64), info.data_ptr(),
                                              workspace.data_ptr() if workspace is not None else None,
                                              workspace.numel() if workspace is not None else 0, _stream()),
              "irs_smooth_finalize_ws")
        return At, Bt, ct, info

    def exact_linearize(self, x_trj, u_trj):
        T = u_trj.shape[0]
        dev = u_trj.device
        At = torch.empty((T, self.n, self.n), dtype=F64, device=dev)
        Bt = torch.empty((T, self.n, self.m), dtype=F64, device=dev)
        ct = torch.empty((T, self.n), dtype=F64, device=dev)
        check(self.lib.irs_exact_linearize(self.model_id, self._p, self._np, T, _ptr(x_trj, F64),
                                           _ptr(u_trj, F64), _ptr(At, F64), _ptr(Bt, F64), _ptr(ct, F64),
                                           _stream()), "irs_exact_linearize")
        return At, Bt, ct

    def closed_loop_rollout(self, K, k, x0, Q, R, xd_trj):
        T = K.shape[0]
        dev = K.device
        x_new = torch.empty((T + 1, self.n), dtype=F64, device=dev)
        u_new = torch.empty((T, self.m), dtype=F64, device=dev)
        cost = torch.empty((1,), dtype=F64, device=dev)
        check(self.lib.irs_closed_loop_rollout(self.model_id, self._p, self._np, T, _ptr(K, F64), _ptr(k, F64),
                                               _ptr(x0, F64), _ptr(Q, F64), _ptr(R, F64), _ptr(xd_trj, F64),
                                               _ptr(x_new, F64), _ptr(u_new, F64), _ptr(cost, F64), _stream()),
              "irs_closed_loop_rollout")
        return x_new, u_new, cost


class SmoothPlan:
    """A pre-marshalled get_TV_matrices call (irs_smooth_call): `run()` is ONE FFI call
    that enqueues ONE kernel (fused path) and touches no Python-side allocation.

    Samples are either supplied (`dx`, `du` device f32 tensors) or drawn on the device
    (`rng=dict(N=..., std_x=..., std_u=..., seed=..., iter=...)`).  With `fuse=False`
    only `sums` is produced (multi-GPU path: all-reduce it, then `dm.smooth_finalize`)."""

    def __init__(self, dm, mode, x_trj, u_trj, dx=None, du=None, rng=None, fuse=True, n_total=None,
                 sample_offset=0):
        self.dm, self.mode = dm, mode
        T = u_trj.shape[0]
        device = u_trj.device
        c = _lib.SmoothCall()
        c.model, c.n_params = dm.model_id, dm._np
        for i, v in enumerate(dm.params):
            c.params[i] = v
        c.mode, c.T = mode, T
        if rng is not None:
            N = int(rng["N"])
            c.use_rng = 1
            c.seed, c.sample_offset = int(rng["seed"]), int(sample_offset)
            self.set_iter(rng.get("iter", 1), rng.get("std_x"), rng["std_u"], call=c)
        else:
            N = du.shape[1]
            c.use_rng = 0
            c.dx, c.du = _ptr(dx, F32), _ptr(du, F32)
        c.N = N
        self.N = N
        self.sums = torch.empty((T, dm.sums_len(mode)), dtype=F64, device=device)
        c.sums = self.sums.data_ptr()
        self.out = None
        if fuse:
            self.out = dm._tv_outputs(T, device, None)
            self.out["sums"] = self.sums
            c.At, c.Bt, c.ct = (self.out[k].data_ptr() for k in ("At", "Bt", "ct"))
            c.info = self.out["info"].data_ptr()
        c.n_total = int(n_total if n_total is not None else N)
        self.ws = dm._workspace(mode, T, N, device)
        c.workspace, c.workspace_bytes = self.ws.data_ptr(), self.ws.numel()
        self.call = c
        self._keep = (dx, du)
        self.set_trajectory(x_trj, u_trj)
        self._fn = dm.lib.irs_smooth_run
        self._ref = ctypes.byref(c)

    def set_trajectory(self, x_trj, u_trj):
        self._xu = (x_trj, u_trj)
        self.call.x_trj, self.call.u_trj = _ptr(x_trj, F64), _ptr(u_trj, F64)

    def set_samples(self, dx, du):
        self._keep = (dx, du)
        self.call.dx, self.call.du = _ptr(dx, F32), _ptr(du, F32)

    def set_iter(self, it, std_x, std_u, call=None):
        c = call if call is not None else self.call
        c.iter = int(it)
        if std_x is not None:
            for i, v in enumerate(std_x):
                c.std_x[i] = float(v)
        for i, v in enumerate(std_u):
            c.std_u[i] = float(v)

    def run(self, stream=None):
        rc = self._fn(self._ref, _stream() if stream is None else stream)
        if rc != 0:
            check(rc, "irs_smooth_run")
        return self.out if self.out is not None else self.sums


class DescentPlan:
    """A pre-marshalled irs_descent_call: Riccati + closed-loop rollout + cost, one launch."""

    def __init__(self, dm, At, Bt, ct, Q, Qd, R, xd_trj, x0, alpha_R=0.5, x_new=None, u_new=None):
        T = At.shape[0]
        device = At.device
        self.out = dict(K=torch.empty((T, dm.m, dm.n), dtype=F64, device=device),
                        k=torch.empty((T, dm.m), dtype=F64, device=device),
                        x_new=x_new if x_new is not None else torch.empty((T + 1, dm.n), dtype=F64, device=device),
                        u_new=u_new if u_new is not None else torch.empty((T, dm.m), dtype=F64, device=device),
                        cost=torch.empty((1,), dtype=F64, device=device),
                        info=torch.empty((1,), dtype=torch.int32, device=device))
        c = _lib.DescentCall()
        c.model, c.n_params = dm.model_id, dm._np
        for i, v in enumerate(dm.params):
            c.params[i] = v
        c.T, c.alpha_R = T, float(alpha_R)
        self._keep = (At, Bt, ct, Q, Qd, R, xd_trj, x0)
        c.At, c.Bt, c.ct = _ptr(At, F64), _ptr(Bt, F64), _ptr(ct, F64)
        c.Q, c.Qd, c.R = _ptr(Q, F64), _ptr(Qd, F64), _ptr(R, F64)
        c.xd_trj, c.x0 = _ptr(xd_trj, F64), _ptr(x0, F64)
        for name in ("K", "k", "x_new", "u_new", "cost"):
            setattr(c, name, self.out[name].data_ptr())
        c.info = self.out["info"].data_ptr()
        self.call = c
        self._fn = dm.lib.irs_descent_run
        self._ref = ctypes.byref(c)

    def run(self, stream=None):
        rc = self._fn(self._ref, _stream() if stream is None else stream)
        if rc != 0:
            check(rc, "irs_descent_run")
        return self.out


def evaluate_cost(x_trj, u_trj, Q, R, xd_trj):
    lib = _lib.load()
    T, n, m = u_trj.shape[0], x_trj.shape[1], u_trj.shape[1]
    cost = torch.empty((1,), dtype=F64, device=x_trj.device)
    check(lib.irs_evaluate_cost(n, m, T, _ptr(x_trj, F64), _ptr(u_trj, F64), _ptr(Q, F64), _ptr(R, F64),
                                _ptr(xd_trj, F64), _ptr(cost, F64), _stream()), "irs_evaluate_cost")
    return cost


def tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd_trj, alpha_R=0.5):
    """Backward pass; returns K (T,m,n), k (T,m), info (1) device tensors."""
    lib = _lib.load()
    T, n, m = At.shape[0], At.shape[1], Bt.shape[2]
    dev = At.device
    K = torch.empty((T, m, n), dtype=F64, device=dev)
    k = torch.empty((T, m), dtype=F64, device=dev)
    info = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib.irs_tvlqr_riccati(n, m, T, _ptr(At, F64), _ptr(Bt, F64), _ptr(ct, F64), _ptr(Q, F64),
                                _ptr(Qd, F64), _ptr(R, F64), float(alpha_R), _ptr(xd_trj, F64), _ptr(K, F64),
                                _ptr(k, F64), info.data_ptr(), _stream()), "irs_tvlqr_riccati")
    return K, k, info


def tvlqr_linear_rollout(At, Bt, ct, K, k, x0):
    lib = _lib.load()
    T, n, m = At.shape[0], At.shape[1], Bt.shape[2]
    dev = At.device
    xs = torch.empty((T + 1, n), dtype=F64, device=dev)
    us = torch.empty((T, m), dtype=F64, device=dev)
    check(lib.irs_tvlqr_linear_rollout(n, m, T, _ptr(At, F64), _ptr(Bt, F64), _ptr(ct, F64), _ptr(K, F64),
                                       _ptr(k, F64), _ptr(x0, F64), _ptr(xs, F64), _ptr(us, F64), _stream()),
          "irs_tvlqr_linear_rollout")
    return xs, us
