"""Host mirror of irs_lqr/tv_lqr.py: get_solver (:11-27) and solve_tvlqr (:30-145).

The reference builds a Drake MathematicalProgram and calls OSQP/Gurobi.  This stand-alone
function solves ONE such QP: exactly, by a backward Riccati pass on the GPU (irs_tvlqr_riccati)
followed by the linear-model rollout of the resulting affine policy (irs_tvlqr_linear_rollout).
That is the QP's solution whenever no box bound is active; the result is checked against the
bounds and an active bound raises.  The reference's only callers are the MPC loops of
`local_descent`, and those run as whole-descent kernels that DO handle active bounds and the
position-controlled (du) cost: irs_tvlqr_box_descent (IrsLqr) and irs_quasistatic_box_descent
(IrsLqrQuasistatic); a single bounded QP is not exposed as an entry point of its own.
"""
import numpy as np

from . import device as dev

_SOLVERS = ("osqp", "snopt", "clp", "gurobi")


class RiccatiSolver:
    """Stand-in for the Drake solver handle get_solver returns (tv_lqr.py:11-27)."""

    def __init__(self, name):
        self.name = name


def get_solver(solver_name: str):
    if solver_name in _SOLVERS:
        return RiccatiSolver(solver_name)
    raise ValueError("Do not recognize solver.")


def solve_tvlqr(At, Bt, ct, Q, Qd, R, x0, x_trj_d, solver=None, indices_u_into_x=None,
                x_bound_abs=None, u_bound_abs=None, x_bound_rel=None, u_bound_rel=None,
                xinit=None, uinit=None):
    """Same signature and return value (xt_star (T+1,n), ut_star (T,m)) as tv_lqr.py:30."""
    if indices_u_into_x is not None:
        raise NotImplementedError("a single position-controlled QP (indices_u_into_x, tv_lqr.py:93-107) is not exposed; "
                                  "IrsLqrQuasistatic.local_descent solves all T of them on the device")
    At_d, Bt_d, ct_d = dev.to_dev(np.asarray(At, float)), dev.to_dev(np.asarray(Bt, float)), dev.to_dev(
        np.asarray(ct, float).reshape(At.shape[0], -1))
    Q_d, Qd_d, R_d = dev.to_dev(np.asarray(Q, float)), dev.to_dev(np.asarray(Qd, float)), dev.to_dev(
        np.asarray(R, float))
    xd_d = dev.to_dev(np.asarray(x_trj_d, float))
    x0_d = dev.to_dev(np.asarray(x0, float))
    # Drake: AddQuadraticCost(R, 0, u) = 1/2 u'Ru (tv_lqr.py:110)
    K, k, info = dev.tvlqr_riccati(At_d, Bt_d, ct_d, Q_d, Qd_d, R_d, xd_d, alpha_R=0.5)
    xs, us = dev.tvlqr_linear_rollout(At_d, Bt_d, ct_d, K, k, x0_d)
    if int(info.item()) != 0:
        raise ValueError("TV_LQR failed. Optimization problem is not solved.")
    xs, us = xs.cpu().numpy(), us.cpu().numpy()
    T = us.shape[0]
    tol = 1e-9

    def _active(val, bnd, rows):
        if bnd is None:
            return False
        b = np.asarray(bnd, float)
        return bool((val < b[0][:rows] - tol).any() or (val > b[1][:rows] + tol).any())

    if (_active(xs, x_bound_abs, T + 1) or _active(us, u_bound_abs, T) or
            _active(np.diff(xs, axis=0), x_bound_rel, T) or
            (u_bound_rel is not None and T > 1 and _active(np.diff(us, axis=0), np.asarray(u_bound_rel)[:, 1:], T - 1))):
        raise NotImplementedError("a box bound is active: a single bounded QP (tv_lqr.py:112-123) is not exposed; "
                                  "IrsLqr.local_descent / IrsLqrQuasistatic.local_descent solve them on the device")
    return xs, us
