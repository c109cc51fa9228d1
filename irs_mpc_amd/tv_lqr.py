"""Host mirror of irs_lqr/tv_lqr.py: get_solver (:11-27) and solve_tvlqr (:30-145).

The reference builds a Drake MathematicalProgram and calls OSQP/Gurobi.  Here ONE such QP is solved on the
GPU, whatever its options:

  * no bound given, or none active: the backward Riccati pass (irs_tvlqr_riccati) + the linear-model rollout
    of the resulting affine policy (irs_tvlqr_linear_rollout) -- exact;
  * an active `x_bound_abs` / `u_bound_abs` / `u_bound_rel`, and / or `indices_u_into_x` (the
    position-controlled form: cost on du_t = u_t - u_{t-1}, du_0 = u_0 - x0[idx], tv_lqr.py:96-108):
    irs_tvlqr_box_solve -- ADMM around one Riccati factorisation (csrc/boxqp.hip), converged to `eps`
    (OSQP's default is 1e-3; this runs to 1e-8).

The reference's callers are the MPC loops of `local_descent`; those run as whole-descent kernels
(irs_tvlqr_descent / irs_tvlqr_box_descent / irs_quasistatic_box_descent_wsx) and do not go through here.
The bounded kernel is compiled per (dim_x, dim_u) of the registered device models: (2,1), (12,4), (5,2), (6,2),
(7,4); position-controlled: (7,4) with indices [1,4,2,5], (5,2) with [0,2], (2,1) with [0]; other sizes raise.  `x_bound_rel` ("should be rarely used", irs_lqr_quasistatic.py:315) is not supported in the
position-controlled form; in the plain form the reference bounds free variables with it (dxt / dut are only
tied to x, u inside the `indices_u_into_x` branch, tv_lqr.py:93-104), i.e. it has no effect, and neither has it here.
"""
import numpy as np

from . import device as dev
from ._lib import check, dbl_array, load

_SOLVERS = ("osqp", "snopt", "clp", "gurobi")


class RiccatiSolver:
    """Stand-in for the Drake solver handle get_solver returns (tv_lqr.py:11-27)."""

    def __init__(self, name):
        self.name = name


def get_solver(solver_name: str):
    if solver_name in _SOLVERS:
        return RiccatiSolver(solver_name)
    raise ValueError("Do not recognize solver.")


def _model_for(n, m, indices_u_into_x):
    """A registered device model of this size (its dynamics are not used: only the compiled dimensions and, for
    the position-controlled form, indices_u_into_x)."""
    from . import systems as sy
    if indices_u_into_x is None:
        table = {(2, 1): lambda: sy.PendulumDynamics(0.05), (12, 4): lambda: sy.QuadrotorDynamics(0.05),
                 (5, 2): lambda: sy.BicycleDynamics(0.1), (6, 2): lambda: sy.ThreeCartDynamics(0.05),
                 (7, 4): lambda: sy.PlanarHandDynamics(0.1)}
    else:
        table = {(7, 4): lambda: sy.PlanarHandDynamics(0.1), (5, 2): lambda: sy.BoxPivotingDynamics(0.1),
                 (2, 1): lambda: sy.BoxOnBoxDynamics()}
    make = table.get((n, m))
    if make is None:
        raise NotImplementedError("solve_tvlqr with bounds: no compiled kernel for dim_x=%d, dim_u=%d" % (n, m))
    system = make()
    if indices_u_into_x is not None and list(system.get_u_indices_into_x()) != list(np.asarray(indices_u_into_x)):
        raise NotImplementedError("solve_tvlqr: indices_u_into_x %s has no compiled kernel (this size: %s)"
                                  % (list(indices_u_into_x), list(system.get_u_indices_into_x())))
    return system.dm()


def _rows(b, T_rows, width, dv):
    """(lo, hi) per-time device rows from the reference's (2, rows, width) bound array."""
    if b is None:
        return None, None
    b = np.asarray(b, float)
    lo = np.broadcast_to(b[0], (b[0].shape[0] if b[0].ndim == 2 else T_rows, width))[:T_rows]
    hi = np.broadcast_to(b[1], (b[1].shape[0] if b[1].ndim == 2 else T_rows, width))[:T_rows]
    return dev.to_dev(np.array(lo, dtype=float, copy=True)), dev.to_dev(np.array(hi, dtype=float, copy=True))


def solve_tvlqr(At, Bt, ct, Q, Qd, R, x0, x_trj_d, solver=None, indices_u_into_x=None,
                x_bound_abs=None, u_bound_abs=None, x_bound_rel=None, u_bound_rel=None,
                xinit=None, uinit=None, rho=10.0, max_iter=20000, eps=1e-8):
    """Same signature and return value (xt_star (T+1,n), ut_star (T,m)) as tv_lqr.py:30; raises
    ValueError("TV_LQR failed...") like :139-140 when the solve does not converge.  `rho`, `max_iter`, `eps`
    (extensions) tune the bounded solve."""
    At = np.asarray(At, float)
    T, n, m = At.shape[0], At.shape[1], np.asarray(Bt).shape[2]
    At_d, Bt_d = dev.to_dev(At), dev.to_dev(np.asarray(Bt, float))
    ct_d = dev.to_dev(np.asarray(ct, float).reshape(T, -1))
    Q_d, Qd_d, R_d = dev.to_dev(np.asarray(Q, float)), dev.to_dev(np.asarray(Qd, float)), dev.to_dev(np.asarray(R, float))
    xd_d = dev.to_dev(np.asarray(x_trj_d, float))
    x0_d = dev.to_dev(np.asarray(x0, float))
    position = indices_u_into_x is not None
    if position and x_bound_rel is not None:
        raise NotImplementedError("x_bound_rel ('should be rarely used', irs_lqr_quasistatic.py:315) is not supported")
    if not position:
        # Drake: AddQuadraticCost(R, 0, u) = 1/2 u'Ru (tv_lqr.py:110).  Unconstrained optimum first: it is the
        # QP's solution whenever it respects the bounds (u_bound_rel / x_bound_rel bind nothing in this branch)
        K, k, info = dev.tvlqr_riccati(At_d, Bt_d, ct_d, Q_d, Qd_d, R_d, xd_d, alpha_R=0.5)
        xs, us = dev.tvlqr_linear_rollout(At_d, Bt_d, ct_d, K, k, x0_d)
        if int(info.item()) != 0:
            raise ValueError("TV_LQR failed. Optimization problem is not solved.")
        xs, us = xs.cpu().numpy(), us.cpu().numpy()

        def inside(val, bnd, rows, skip=0):
            if bnd is None:
                return True
            b = np.asarray(bnd, float)
            lo = np.broadcast_to(b[0], (rows,) + val.shape[1:]) if b[0].ndim == 1 else b[0][:rows]
            hi = np.broadcast_to(b[1], (rows,) + val.shape[1:]) if b[1].ndim == 1 else b[1][:rows]
            return bool((val[skip:rows] >= lo[skip:rows]).all() and (val[skip:rows] <= hi[skip:rows]).all())

        if inside(xs, x_bound_abs, T + 1, skip=1) and inside(us, u_bound_abs, T):
            return xs, us
    dm = _model_for(n, m, indices_u_into_x)
    x_lo, x_hi = _rows(x_bound_abs, T + 1, n, dev)
    u_lo, u_hi = _rows(u_bound_abs, T, m, dev)
    du_lo, du_hi = _rows(u_bound_rel, T, m, dev) if position else (None, None)
    x_star = dev.to_dev(np.zeros((T + 1, n)))
    u_star = dev.to_dev(np.zeros((T, m)))
    import torch
    info = torch.full((3,), -1, dtype=torch.int32, device=x_star.device)
    lib = load()
    ptr = dev._ptr
    check(lib.irs_tvlqr_box_solve(dm.model_id, dm._p, dm._np, T, ptr(At_d, dev.F64), ptr(Bt_d, dev.F64), ptr(ct_d, dev.F64),
                                  ptr(Q_d, dev.F64), ptr(Qd_d, dev.F64), ptr(R_d, dev.F64), 1.0 if position else 0.5,
                                  ptr(xd_d, dev.F64), ptr(x0_d, dev.F64), 1 if position else 0,
                                  ptr(x_lo, dev.F64), ptr(x_hi, dev.F64), ptr(u_lo, dev.F64), ptr(u_hi, dev.F64),
                                  ptr(du_lo, dev.F64), ptr(du_hi, dev.F64), float(rho), 1.6, int(max_iter), float(eps),
                                  ptr(x_star, dev.F64), ptr(u_star, dev.F64), info.data_ptr(), dev._stream()),
          "irs_tvlqr_box_solve")
    i = info.cpu().numpy()
    if i[0] != 0 or i[2] != 0:
        raise ValueError("TV_LQR failed. Optimization problem is not solved.")
    return x_star.cpu().numpy(), u_star.cpu().numpy()
