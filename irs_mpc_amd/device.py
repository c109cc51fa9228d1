"""Thin tensor-level wrappers over the C ABI.  PyTorch is used for device memory,
the current HIP stream and (elsewhere) torch.distributed -- nothing else.

Every function takes/returns torch tensors that live on the GPU; nothing here
synchronises with the host.
"""
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
            ws = torch.empty((max(need, 256),), dtype=torch.uint8, device=device)
            self._ws[key] = ws
        return ws

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

    def smooth_finalize(self, mode, N_total, x_trj, u_trj, sums):
        T = u_trj.shape[0]
        dev = u_trj.device
        At = torch.empty((T, self.n, self.n), dtype=F64, device=dev)
        Bt = torch.empty((T, self.n, self.m), dtype=F64, device=dev)
        ct = torch.empty((T, self.n), dtype=F64, device=dev)
        info = torch.empty((T,), dtype=torch.int32, device=dev)
        check(self.lib.irs_smooth_finalize(self.model_id, self._p, self._np, mode, T, int(N_total),
                                           _ptr(x_trj, F64), _ptr(u_trj, F64), _ptr(sums, F64), _ptr(At, F64),
                                           _ptr(Bt, F64), _ptr(ct, F64), info.data_ptr(), _stream()),
              "irs_smooth_finalize")
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
