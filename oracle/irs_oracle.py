"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not measured as the product.

CPU (NumPy, float64) restatement of the iRS-LQR hot path of hjsuh94/irs_mpc.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the reported CPU baseline.  The product
(irs_mpc_amd) never imports it and fails loudly when its HIP library is absent.

Every function cites the reference lines it follows (paths relative to the
reference repo root).  Parity status: PINNED -- see tests/test_oracle_golden.py:
  * rollout + cost + exact linearisation + TV-LQR + closed-loop forward pass are
    pinned by the reference's own result files examples/pendulum/analysis/
    pendulum_exact.csv and examples/quadrotor/analysis/quadrotor_exact.csv;
  * dynamics / dynamics_batch / zero-order and first-order get_TV_matrices /
    compute_least_squares / CEM local_descent are pinned by fixtures produced by
    running the reference's own source in the build container
    (tests/golden/make_fixtures.py, committed with its outputs).
The QP solver itself (Drake + OSQP, irs_lqr/tv_lqr.py:69-137) is a third-party
dependency that is not vendored and not installed; solve_tvlqr_qp() below
restates the QP exactly as posed and solves its KKT system directly (valid while
the box bounds are inactive), and the two *_exact.csv files anchor it.

Quasistatic part -- the contact STEP and its DERIVATIVE (jacobian_xu: the simulator's Dq_nextDq |
Dq_nextDqa_cmd, i.e. gradient modes "exact" / "first_order") are PINNED by simulator data the reference
ships (examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy -> BoxPushOracle: the 80-step trajectory to
3e-8, all 80 Jacobians to 5e-7 but the contact-onset one) and by the closed form of its 1-D case
(box_on_box.py:11-20 -> BoxOnBoxOracle).  The step QP's dual is solved EXACTLY by default (pgs_iters = 0: dual
active-set method, certified against the QP's KKT conditions -- the reference's simulator hands every step
QP to Gurobi, quasistatic_dynamics.py:146-164; the device default, contact_solver="exact"), or, opt-in, by
`pgs_iters` over-relaxed projected sweeps + an active-set polish (the device's contact_solver="pgs").
PARITY UNPINNED for the planar-hand / box-pivoting geometry and parameters (PlanarHandOracle,
BoxPivotOracle) and for the *_quasistatic / ctrlbox_* functions: the reference steps pangtao22/quasistatic_simulator (external, not vendored, model
files absent; plus Drake and Gurobi), so the contact step restates the published scheme
(Anitescu's convex quasi-dynamic step) on the constants the reference does state, and the
bounded du-cost QPs of irs_lqr/tv_lqr.py:96-127 are certified against their own KKT conditions
instead of a reference run.  The scheme itself (and the shared PGS code) IS pinned by the closed
form the reference prints for its 1-D case, examples/box_pushing/analysis/box_on_box.py:11-20
(BoxOnBoxOracle).  See DESIGN.md section 3.
"""
import numpy as np


# --------------------------------------------------------------------------
# Dynamics plugins (irs_lqr/dynamical_system.py:1-66 is the abstract surface)
# --------------------------------------------------------------------------
class PendulumOracle:
    """examples/pendulum/pendulum_dynamics.py:8-127."""

    name = "pendulum"

    def __init__(self, h):
        self.h = h
        self.dim_x = 2
        self.dim_u = 1

    def dynamics(self, x, u):
        # pendulum_dynamics.py:46-60, semi-implicit Euler
        angle, speed = x[0], x[1]
        next_speed = speed + self.h * (-np.sin(angle) + u[0])
        next_angle = angle + self.h * next_speed
        return np.array([next_angle, next_speed])

    def dynamics_batch(self, x, u):
        # pendulum_dynamics.py:62-81
        angle, speed, torque = x[:, 0], x[:, 1], u[:, 0]
        next_speed = speed + self.h * (-np.sin(angle) + torque)
        next_angle = angle + self.h * next_speed
        return np.vstack((next_angle, next_speed)).transpose()

    def jacobian_xu(self, x, u):
        # pendulum_dynamics.py:110-117 evaluates the symbolic Jacobian of
        # dynamics_sym (:28-43); this is that Jacobian written out.
        h, c = self.h, np.cos(x[0])
        return np.array([[1.0 - h * h * c, h, h * h],
                         [-h * c, 1.0, h]])

    def jacobian_xu_batch(self, x, u):
        # pendulum_dynamics.py:119-127
        return np.stack([self.jacobian_xu(x[i], u[i]) for i in range(x.shape[0])])


class QuadrotorOracle:
    """examples/quadrotor/quadrotor_dynamics.py:15-231."""

    name = "quadrotor"

    def __init__(self, h):
        self.h = h
        self.dim_x = 12
        self.dim_u = 4
        self.m = 0.775
        self.L = 0.15
        self.g = 9.81
        self.I = np.diag([0.0015, 0.0025, 0.0035])
        self.I_inv = np.linalg.inv(self.I)
        self.kF = 1.0
        self.kM = 0.0245

    # quadrotor_dynamics.py:150-186
    @staticmethod
    def _R_WB(rpy):
        cr, sr = np.cos(rpy[0]), np.sin(rpy[0])
        cp, sp = np.cos(rpy[1]), np.sin(rpy[1])
        cy, sy = np.cos(rpy[2]), np.sin(rpy[2])
        Rx = np.array([[1., 0., 0.], [0, cr, -sr], [0, sr, cr]], dtype=cr.dtype)
        Ry = np.array([[cp, 0., sp], [0, 1., 0], [-sp, 0., cp]], dtype=cr.dtype)
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0., 0., 1]], dtype=cr.dtype)
        return Rz.dot(Ry.dot(Rx))

    # quadrotor_dynamics.py:188-199
    @staticmethod
    def _PhiInv(rpy):
        sr, cr = np.sin(rpy[0]), np.cos(rpy[0])
        sp, cp = np.sin(rpy[1]), np.cos(rpy[1])
        return np.array([[1, 0, -sp], [0, cr, sr * cp], [0, -sr, cr * cp]],
                        dtype=sr.dtype)

    # quadrotor_dynamics.py:201-212
    @staticmethod
    def _Phi(rpy):
        sr, cr = np.sin(rpy[0]), np.cos(rpy[0])
        sp, cp = np.sin(rpy[1]), np.cos(rpy[1])
        return np.array([[1, sr * sp / cp, cr * sp / cp],
                         [0, cr, -sr],
                         [0, sr / cp, cr / cp]], dtype=sr.dtype)

    # quadrotor_dynamics.py:215-231
    @staticmethod
    def _PhiD(rpy):
        sr, cr = np.sin(rpy[0]), np.cos(rpy[0])
        sp, cp = np.sin(rpy[1]), np.cos(rpy[1])
        cp2 = cp ** 2
        tp = sp / cp
        D = np.zeros((3, 3, 3), dtype=sr.dtype)
        D[0, 1] = [cr * tp, sr / cp2, 0]
        D[0, 2] = [-sr * tp, cr / cp2, 0]
        D[1, 1] = [-sr, 0, 0]
        D[1, 2] = [-cr, 0, 0]
        D[2, 1] = [cr / cp, sr * sp / cp2, 0]
        D[2, 2] = [-sr / cp, cr * sp / cp2, 0]
        return D

    def dynamics(self, x, u):
        # quadrotor_dynamics.py:40-77 (explicit Euler on an rpy rigid body)
        x = np.asarray(x)
        u = np.asarray(u)
        dt = np.result_type(x.dtype, u.dtype, np.float64)
        x = x.astype(dt)
        u = u.astype(dt)
        xdot = np.empty(x.shape, dtype=dt)
        uF = self.kF * u
        uM = self.kM * u
        Fg = np.array([0., 0., -self.m * self.g])
        F = np.array([0., 0., uF.sum()])
        M = np.array([self.L * (-uF[0] - uF[1] + uF[2] + uF[3]),
                      self.L * (-uF[0] - uF[3] + uF[1] + uF[2]),
                      -uM[0] + uM[1] - uM[2] + uM[3]])
        rpy = x[3:6]
        rpy_d = x[9:12]
        R_WB = self._R_WB(rpy)
        xyz_dd = 1. / self.m * (R_WB.dot(F) + Fg)
        pqr = self._PhiInv(rpy).dot(rpy_d)
        pqr_d = self.I_inv.dot(M - np.cross(pqr, self.I.dot(pqr)))
        Phi_d = self._PhiD(rpy)
        Phi = self._Phi(rpy)
        rpy_dd = Phi.dot(pqr_d) + (Phi_d.dot(rpy_d)).dot(pqr)
        xdot[0:6] = x[6:12]
        xdot[6:9] = xyz_dd
        xdot[9:12] = rpy_dd
        return x + self.h * xdot

    def dynamics_batch(self, x, u):
        # quadrotor_dynamics.py:79-91 (Python loop over the batch)
        return np.stack([self.dynamics(x[b], u[b]) for b in range(x.shape[0])])

    def jacobian_xu(self, x, u):
        # quadrotor_dynamics.py:132-138: pydrake.forwarddiff.jacobian (forward-
        # mode AD, i.e. the exact derivative) of dynamics_xu.  pydrake is not
        # installed; the complex-step derivative of the same analytic map is
        # exact to machine precision (no subtractive cancellation).
        xu = np.hstack((x, u)).astype(np.complex128)
        n, d = self.dim_x, self.dim_x + self.dim_u
        J = np.zeros((n, d))
        eps = 1e-30
        for j in range(d):
            z = xu.copy()
            z[j] += 1j * eps
            J[:, j] = self.dynamics(z[:n], z[n:]).imag / eps
        return J

    def jacobian_xu_batch(self, x, u):
        # quadrotor_dynamics.py:140-148
        return np.stack([self.jacobian_xu(x[i], u[i]) for i in range(x.shape[0])])


class BicycleOracle:
    """examples/bicycle/bicycle_dynamics.py:8-132."""

    name = "bicycle"

    def __init__(self, h):
        self.h = h
        self.dim_x = 5
        self.dim_u = 2

    def dynamics(self, x, u):
        # bicycle_dynamics.py:47-64
        heading, v, steer = x[2], x[3], x[4]
        dxdt = np.array([v * np.cos(heading), v * np.sin(heading), v * np.tan(steer), u[0], u[1]])
        return x + self.h * dxdt

    def dynamics_batch(self, x, u):
        # bicycle_dynamics.py:66-87
        heading, v, steer = x[:, 2], x[:, 3], x[:, 4]
        dxdt = np.vstack((v * np.cos(heading), v * np.sin(heading), v * np.tan(steer),
                          u[:, 0], u[:, 1])).transpose()
        return x + self.h * dxdt

    def jacobian_xu(self, x, u):
        # bicycle_dynamics.py:115-122 evaluates the symbolic Jacobian of dynamics_sym (:26-44)
        h, th, v, st = self.h, x[2], x[3], x[4]
        J = np.zeros((5, 7))
        J[:, :5] = np.eye(5)
        J[0, 2], J[0, 3] = -h * v * np.sin(th), h * np.cos(th)
        J[1, 2], J[1, 3] = h * v * np.cos(th), h * np.sin(th)
        J[2, 3], J[2, 4] = h * np.tan(st), h * v / np.cos(st) ** 2
        J[3, 5] = h
        J[4, 6] = h
        return J

    def jacobian_xu_batch(self, x, u):
        return np.stack([self.jacobian_xu(x[i], u[i]) for i in range(x.shape[0])])


class ThreeCartOracle:
    """examples/three_cart/three_cart_dynamics.py:8-107 (the scalar `dynamics`)."""

    name = "three_cart"

    def __init__(self, h):
        self.h = h
        self.dim_x = 6
        self.dim_u = 2
        self.d = 0.2

    def dynamics(self, x, u):
        # three_cart_dynamics.py:22-107
        q1, q2, q3, v1, v2, v3 = x
        u1, u3 = u
        v1s, v2s, v3s = v1 + self.h * u1, v2, v3 + self.h * u3
        q1s, q2s, q3s = q1 + self.h * v1s, q2 + self.h * v2s, q3 + self.h * v3s
        if (q2s - q1s < self.d) and (q3s - q2s < self.d):
            q2n = (1. / 3.) * (q1s + q2s + q3s)
            q1n, q3n = q2n - self.d, q2n + self.d
            v1n = v2n = v3n = (1. / 3.) * (v1s + v2s + v3s)
        elif q2s - q1s < self.d:
            pen = self.d - (q2s - q1s)
            q2n, q1n = q2s + 0.5 * pen, q1s - 0.5 * pen
            v1n = v2n = 0.5 * (v1s + v2s)
            q3n, v3n = q3s, v3s
        elif q3s - q2s < self.d:
            pen = self.d - (q3s - q2s)
            q3n, q2n = q3s + 0.5 * pen, q2s - 0.5 * pen
            v2n = v3n = 0.5 * (v2s + v3s)
            q1n, v1n = q1s, v1s
        else:
            q1n, q2n, q3n, v1n, v2n, v3n = q1s, q2s, q3s, v1s, v2s, v3s
        return np.array([q1n, q2n, q3n, v1n, v2n, v3n])

    def dynamics_batch(self, x, u):
        """Row-wise scalar dynamics (what the device functor implements; the reference's
        own dynamics_batch, :109-203, resolves penetration by the full depth instead)."""
        return np.stack([self.dynamics(x[b], u[b]) for b in range(x.shape[0])])

    def jacobian_xu(self, x, u):
        """Derivative of the active branch (central differences are exact for a
        piecewise-linear map away from the switching surfaces)."""
        n, d = 6, 8
        xu = np.hstack((x, u))
        J = np.zeros((n, d))
        for j in range(d):
            e = np.zeros(d)
            e[j] = 1e-7
            J[:, j] = (self.dynamics((xu + e)[:n], (xu + e)[n:]) - self.dynamics((xu - e)[:n], (xu - e)[n:])) / 2e-7
        return J

    def jacobian_xu_batch(self, x, u):
        return np.stack([self.jacobian_xu(x[i], u[i]) for i in range(x.shape[0])])


class _ContactQPOracle:
    """Shared tail of the contact oracles: subclasses provide `_qp(x, u)` -> (Dinv, b, J, phi) in
    their INTERNAL coordinate order and `PERM` (internal index -> index in the reference's x)."""

    PGS_OMEGA = 1.5         # over-relaxation of the projected sweeps (csrc/contact_models.hpp, kContactPgsOmega)

    def dynamics_batch(self, x, u):
        Dinv, b, J, W, lam = self._pgs(np.atleast_2d(x), np.atleast_2d(u))
        out = np.array(np.atleast_2d(x), dtype=float)
        out[:, self.PERM] += (np.einsum("bik,bi->bk", J, lam) - b) * Dinv
        return out

    def dynamics(self, x, u):
        return self.dynamics_batch(x[None], u[None])[0]

    def dynamics_exact(self, x, u):
        """The same QP solved to optimality by an independent method (L-BFGS-B on the dual): the physics
        check of the sweeps.  It stalls on a few degenerate samples (rank-deficient W with many active rows);
        `_dual_exact` (pgs_iters = 0) is the exact solver, certified through the QP's KKT conditions."""
        from scipy.optimize import minimize
        Dinv, b, J, phi = self._qp(x, u)
        b, J, phi = b[0], J[0], phi[0]
        W = (J * Dinv).dot(J.T)
        r = phi - (J * Dinv).dot(b)
        res = minimize(lambda l: 0.5 * l.dot(W).dot(l) + r.dot(l), np.zeros(len(r)), jac=lambda l: W.dot(l) + r,
                       bounds=[(0, None)] * len(r), method="L-BFGS-B",
                       options={"ftol": 1e-15, "gtol": 1e-12, "maxiter": 10000})
        out = np.array(x, dtype=float)
        out[self.PERM] += (J.T.dot(res.x) - b) * Dinv
        return out

    ACTIVE_TOL = 1e-7       # a contact row is active when lam_i W_ii (a length) exceeds this
    PIVOT_TOL = 1e-5        # an active row whose pivot falls below PIVOT_TOL * W_ii is dependent: dropped

    # `pgs_iters <= 0` selects the EXACT dual solve (device: the *_EXACT model ids) instead of sweeps
    EXACT_TOL = 1e-10       # constraint violation (relative to max |r|) below which the dual solve stops
    EXACT_PIVOT = 1e-7      # a candidate row whose Schur complement is below this (relative) is dependent

    @staticmethod
    def _masked_solve(W, rhs, A, piv):
        """W_AA y_A = rhs_A by the masked LDL' in row order (y = 0 outside A); batched."""
        Bn, nc, _ = W.shape
        Wm, L, inv = W.copy(), np.zeros_like(W), np.zeros((Bn, nc))
        Wd = np.einsum("bii->bi", W)
        for j in range(nc):
            dj = Wm[:, j, j]
            ok = A[:, j] & (dj > piv * Wd[:, j])
            inv[:, j] = np.where(ok, 1.0 / np.where(ok, dj, 1.0), 0.0)
            for i in range(j + 1, nc):
                L[:, i, j] = Wm[:, i, j] * inv[:, j]
            for i in range(j + 1, nc):
                for k in range(j + 1, i + 1):
                    Wm[:, i, k] -= L[:, i, j] * Wm[:, k, j]
        y = np.where(A, rhs, 0.0)
        for j in range(nc):
            for k in range(j):
                y[:, j] -= L[:, j, k] * y[:, k]
        y *= inv
        for j in range(nc - 1, -1, -1):
            for i in range(j + 1, nc):
                y[:, j] -= L[:, i, j] * y[:, i]
        return y

    def _dual_exact(self, W, r):
        """min 1/2 lam'W lam + r'lam, lam >= 0, solved EXACTLY by the Goldfarb-Idnani dual active-set method
        written in the dual variables (the primal QP has the diagonal Hessian D, so its active-set Schur
        complement is W_AA): start from lam = 0 (the unconstrained primal optimum); repeat: p = the most
        violated row (slack g_p = (r + W lam)_p < 0) outside the active set A; step along
        d lam_A = -rho, d lam_p = +1 with rho = W_AA^-1 W_Ap -- this keeps the active slacks at zero and
        raises g_p at rate z = W_pp - W_pA rho -- until g_p = 0 (full step: p joins A) or some lam_i in A
        reaches zero first (partial step: i leaves A; p stays the candidate).  z = 0 marks a row that
        depends on A: only partial steps are possible.  Every step increases the dual objective: finite, no
        cycling, no regularisation; dependent rows never enter A, so W_AA stays positive definite.
        What the device runs for the *_EXACT contact models (csrc/contact_models.hpp) -- there from a warm
        start (a few projected sweeps guess A, rows with negative restricted multipliers are released until
        the pair is valid), which shortens the path and leaves the unique primal solution unchanged."""
        Bn, nc, _ = W.shape
        ar = np.arange(Bn)
        Wd = np.einsum("bii->bi", W)
        A = np.zeros((Bn, nc), bool)
        lam, g = np.zeros((Bn, nc)), r.copy()
        p = np.full(Bn, -1)
        done = np.zeros(Bn, bool)
        tolv = self.EXACT_TOL * (np.abs(r).max(1) + 1e-300)
        for _ in range(4 * nc):
            need = (p < 0) & ~done
            viol = np.where(A, np.inf, g)
            cand = np.argmin(viol, 1)
            fin = need & (viol[ar, cand] >= -tolv)
            done |= fin
            p = np.where(need & ~fin, cand, p)
            live = ~done
            if not live.any():
                break
            pp = np.where(p >= 0, p, 0)
            Wp = W[ar, :, pp]
            rho = self._masked_solve(W, Wp, A, self.EXACT_PIVOT)
            zp = Wd[ar, pp] - np.einsum("bi,bi->b", np.where(A, Wp, 0.0), rho)
            full_ok = zp > self.EXACT_PIVOT * Wd[ar, pp]
            t2 = np.where(full_ok, -g[ar, pp] / np.where(full_ok, zp, 1.0), np.inf)
            ratio = np.where(A & (rho > 0), lam / np.where(rho > 0, rho, 1.0), np.inf)
            k = np.argmin(ratio, 1)
            t1 = ratio[ar, k]
            t = np.minimum(t1, t2)
            stuck = live & ~np.isfinite(t)            # infeasible primal: keep the current multipliers
            done |= stuck
            live &= ~stuck
            tt = np.where(live & np.isfinite(t), t, 0.0)
            lam = lam - tt[:, None] * rho
            lam[ar, pp] += tt
            g = g + tt[:, None] * (Wp - np.einsum("bij,bj->bi", W, rho))
            full = live & (t2 <= t1)
            part = live & ~(t2 <= t1)
            A[ar[full], pp[full]] = True
            p = np.where(full, -1, p)
            A[ar[part], k[part]] = False
            lam[ar[part], k[part]] = 0.0
            lam = np.maximum(lam, 0.0)
        return lam

    def _pgs(self, x, u):
        Dinv, b, J, phi = self._qp(x, u)
        nc = J.shape[1]
        W = np.einsum("bik,k,bjk->bij", J, Dinv, J)
        r = phi - np.einsum("bik,k,bk->bi", J, Dinv, b)
        if int(self.pgs_iters) <= 0:
            return Dinv, b, J, W, self._dual_exact(W, r)
        lam = np.zeros_like(r)
        invW = self.PGS_OMEGA / np.einsum("bii->bi", W)
        g = r.copy()
        for _ in range(int(self.pgs_iters)):
            for i in range(nc):
                new = np.maximum(lam[:, i] - g[:, i] * invW[:, i], 0.0)
                g += W[:, :, i] * (new - lam[:, i])[:, None]
                lam[:, i] = new
        return Dinv, b, J, W, self._polish(W, r, g, lam)

    POLISH_TOL = 1e-6

    def _polish(self, W, r, g, lam):
        """Active-set polish after the sweeps (csrc/contact_models.hpp, irs_contact_qp_polish): one exact solve
        on the swept active set I = {lam_i > 0}, lam_I += -W_II^-1 g_I, accepted only if it IS the optimum
        (multipliers >= 0 on I, slacks >= 0 off I, active slacks solved to zero, no dependent row in I)."""
        Bn, nc, _ = W.shape
        I = lam > 0
        Wd = np.einsum("bii->bi", W)
        # dependent rows in I (pivot below EXACT_PIVOT W_ii): reject
        Wm = W.copy()
        dep = np.zeros(Bn, bool)
        L = np.zeros_like(W)
        for j in range(nc):
            dj = Wm[:, j, j]
            piv = dj > self.EXACT_PIVOT * Wd[:, j]
            dep |= I[:, j] & ~piv
            invj = np.where(I[:, j] & piv, 1.0 / np.where(piv, dj, 1.0), 0.0)
            for i in range(j + 1, nc):
                L[:, i, j] = Wm[:, i, j] * invj
            for i in range(j + 1, nc):
                for k in range(j + 1, i + 1):
                    Wm[:, i, k] -= L[:, i, j] * Wm[:, k, j]
        dl = self._masked_solve(W, -g, I, self.EXACT_PIVOT)
        gn = g + np.einsum("bij,bj->bi", W, dl)
        tolv = self.POLISH_TOL * (np.abs(r).max(1, keepdims=True) + 1e-30)
        ok = ~dep & np.where(I, (lam + dl >= 0) & (np.abs(gn) <= tolv), gn >= -tolv).all(1)
        return np.where(ok[:, None], np.where(I, lam + dl, 0.0), lam)

    def jacobian_xu_batch(self, x, u):
        """The simulator's `Dq_nextDq | Dq_nextDqa_cmd` (irs_lqr/quasistatic_dynamics.py:184-191,
        `grad_from_active_constraints=True`): the derivative of the step QP's solution through its
        ACTIVE constraints, contact geometry (J) held fixed.  With I the active rows, W_II = J_I D^-1 J_I',
            S  = J_I' W_II^+ J_I       (b enters the QP linearly:   d dq / d b   = -D^-1 + D^-1 S D^-1)
            Sn = J_I' W_II^+ Jn_I      (phi_i is a gap: d phi_i / d q = the NORMAL row Jn_i, the mean of the
                                        contact's two generators;   d dq / d phi_I = -D^-1 J_I' W_II^+)
        and b_a = K (q_a - u), D_aa = K on the actuated dofs, so in the internal order
            B = E_a - D^-1 S[:, a],     A[:, l] = e_l - [l = a_j] B[:, j] - D^-1 Sn[:, l].
        W_II is factorised by a masked LDL' in row order; a dependent active row (pivot below PIVOT_TOL W_ii)
        is dropped -- the projector S does not depend on which one.  PINNED by the simulator's own Jacobians,
        examples/box_pushing/analysis/dxdu_quasistatic.npy (tests/test_oracle_golden.py).  Returns (B, n, n+m)
        in the reference's x order."""
        x, u = np.atleast_2d(x), np.atleast_2d(u)
        Dinv, b, J, W, lam = self._pgs(x, u)
        Bn, nc, nq = J.shape
        act_cols = np.asarray(self.ACT)
        na = len(act_cols)
        Jn = J.copy()
        Jn[:, 0::2] = Jn[:, 1::2] = 0.5 * (J[:, 0::2] + J[:, 1::2])
        Wd = np.einsum("bii->bi", W).copy()
        active = lam * Wd > self.ACTIVE_TOL
        Wm = W.copy()
        Lm = np.zeros_like(W)
        inv = np.zeros((Bn, nc))
        for j in range(nc):
            dj = Wm[:, j, j]
            ok = active[:, j] & (dj > self.PIVOT_TOL * Wd[:, j])
            inv[:, j] = np.where(ok, 1.0 / np.where(ok, dj, 1.0), 0.0)
            for i in range(j + 1, nc):
                Lm[:, i, j] = Wm[:, i, j] * inv[:, j]
            for i in range(j + 1, nc):
                for k in range(j + 1, i + 1):
                    Wm[:, i, k] -= Lm[:, i, j] * Wm[:, k, j]
        R = np.concatenate([J[:, :, act_cols], Jn], axis=2)          # (B, nc, na + nq) right-hand sides
        Y = R.copy()
        for j in range(nc):
            for k in range(j):
                Y[:, j] -= Lm[:, j, k, None] * Y[:, k]
        Y *= inv[:, :, None]
        for j in range(nc - 1, -1, -1):
            for i in range(j + 1, nc):
                Y[:, j] -= Lm[:, i, j, None] * Y[:, i]
        SS = np.einsum("bik,bic->bkc", J, Y)                         # J' Y: (B, nq, na + nq)
        B_int = -Dinv[None, :, None] * SS[:, :, :na]
        B_int[:, act_cols, np.arange(na)] += 1.0
        A_int = -Dinv[None, :, None] * SS[:, :, na:]
        A_int[:, np.arange(nq), np.arange(nq)] += 1.0
        for j, a in enumerate(act_cols):
            A_int[:, :, a] -= B_int[:, :, j]
        P = np.asarray(self.PERM)
        n, m = nq, na
        out = np.zeros((Bn, n, n + m))
        out[:, P[:, None], P[None, :]] = A_int
        out[:, P, n:] = B_int
        return out

    def jacobian_xu(self, x, u):
        return self.jacobian_xu_batch(x[None], u[None])[0]

    def active_mask_batch(self, x, u):
        """Bit i set = contact row i is in the active set `jacobian_xu_batch` differentiates through
        (lam_i W_ii > ACTIVE_TOL); the device's per-sample view is irs_contact_samples_f32."""
        _, _, _, W, lam = self._pgs(np.atleast_2d(x), np.atleast_2d(u))
        act = lam * np.einsum("bii->bi", W) > self.ACTIVE_TOL
        return (act * (1 << np.arange(act.shape[1]))).sum(1).astype(np.int64)


class BoxOnBoxOracle(_ContactQPOracle):
    """The reference's own 1-D instance of the quasi-dynamic step, examples/box_pushing/analysis/
    box_on_box.py:11-20: a stiffness-controlled point (k = 100) commanded to u pushes a unit mass that
    sits at 1; h = 0.1.  x = [x_a, x_u].  Same QP as the planar models (D = diag(K_a, M_u/h^2),
    b = (K_a (x_a - u), 0), one frictionless contact x_u - x_a >= 0): the closed form stated there,
    x+ = w1 * 1 + w2 * u with w1 = m/(m + h^2 k), w2 = h^2 k/(m + h^2 k) once u > 1, is the scheme's
    known answer (tests/test_oracle_golden.py)."""

    PERM = np.array([0, 1])
    ACT = np.array([0])                          # internal indices of the actuated dofs

    def __init__(self, h=0.1, m=1.0, k=100.0, pgs_iters=50):
        self.h, self.m, self.k, self.pgs_iters = h, m, k, pgs_iters
        self.dim_x, self.dim_u = 2, 1
        self.indices_u_into_x = np.array([0])

    def _qp(self, q, u):
        q, u = np.atleast_2d(q), np.atleast_2d(u)
        B = q.shape[0]
        Dinv = np.array([1.0 / self.k, self.h ** 2 / self.m])
        b = np.zeros((B, 2))
        b[:, 0] = self.k * (q[:, 0] - u[:, 0])
        J = np.tile(np.array([[[-1.0, 1.0]]]), (B, 1, 1))
        phi = (q[:, 1] - q[:, 0])[:, None]
        return Dinv, b, J, phi


class PlanarHandOracle(_ContactQPOracle):
    """Planar quasi-dynamic contact step for examples/planar_hand (call sites
    irs_lqr/quasistatic_dynamics.py:136-164; set-up examples/planar_hand/planar_hand_setup.py:8-27;
    geometry examples/planar_hand/analysis/planar_hand_analysis.py:33-101).

    PARITY UNPINNED: the reference steps pangtao22/quasistatic_simulator (external, not vendored,
    its SDF/YAML model files are absent), so this restates the PUBLISHED scheme -- Anitescu's convex
    quasi-dynamic step (Pang & Tedrake 2021), the in-tree 1-D instance of which is
    examples/box_pushing/analysis/box_on_box.py:11-20 -- on the plotters' geometry:

        min_dq 1/2 dq' D dq + b' dq   s.t.  phi_i + J_i dq >= 0,    q+ = q + dq
        D = diag(M_u / h^2, K_a),  b = (-tau_u, K_a (q_a - u))

    State in the REFERENCE's order (Drake's velocity indices of the plant; read off
    examples/planar_hand/analysis/planar_hand_analysis.py:61-67): x = [xo, ql1, qr1, yo, ql2, qr2, th];
    u = commanded joint angles [ql1, ql2, qr1, qr2], so indices_u_into_x = [1, 4, 2, 5].  Internally
    q = x[PERM] = [xo, yo, th, ql1, ql2, qr1, qr2].  `dynamics` solves the dual
    by `pgs_iters` projected Gauss-Seidel sweeps (what the device functor does, same order);
    `dynamics_exact` solves the same QP to optimality (active-set via NNLS) as the physics check.
    """

    def __init__(self, h, mass=1.0, mu=0.5, pgs_iters=0):
        self.h = h
        self.dim_x, self.dim_u = 7, 4
        self.g = 10.0            # planar_hand_setup.py:23
        self.mass = mass
        self.R = 0.25            # sphere_yz_rotation_r_0.25m (planar_hand_setup.py:8)
        self.mu = mu
        self.kp = (50.0, 25.0)   # planar_hand_setup.py:12
        self.l1, self.l2 = 0.3, 0.2
        self.r_link = 0.05
        self.base_x = 0.1
        self.pgs_iters = pgs_iters
        self.indices_u_into_x = self.PERM[3:].copy()

    PERM = np.array([0, 3, 6, 1, 4, 2, 5])       # internal q index -> index in the reference's x
    ACT = np.array([3, 4, 5, 6])

    @classmethod
    def pack(cls, obj, left, right):
        """x in the reference's order from the object pose (xo, yo, th) and the joint angles."""
        x = np.zeros(7)
        x[cls.PERM] = np.concatenate([obj, left, right])
        return x

    def params(self):
        return [self.h, self.g, self.mass, self.R, self.mu, self.kp[0], self.kp[1], self.l1, self.l2,
                self.r_link, self.base_x, self.pgs_iters]

    def _qp(self, q, u):
        """Batched QP data in the INTERNAL order: Dinv (7,), b (B,7), J (B,8,7), phi (B,8)."""
        q = np.atleast_2d(q)[:, self.PERM]
        u = np.atleast_2d(u)
        B = q.shape[0]
        h, m, R, mu = self.h, self.mass, self.R, self.mu
        kp1, kp2 = self.kp
        Dinv = np.array([h * h / m, h * h / m, h * h / (0.5 * m * R * R), 1 / kp1, 1 / kp2, 1 / kp1, 1 / kp2])
        b = np.zeros((B, 7))
        b[:, 1] = m * self.g
        b[:, 3] = kp1 * (q[:, 3] - u[:, 0])
        b[:, 4] = kp2 * (q[:, 4] - u[:, 1])
        b[:, 5] = kp1 * (q[:, 5] - u[:, 2])
        b[:, 6] = kp2 * (q[:, 6] - u[:, 3])
        J = np.zeros((B, 8, 7))
        phi = np.zeros((B, 8))
        for arm in range(2):
            base = -self.base_x if arm == 0 else self.base_x
            a1 = q[:, 3] + np.pi if arm == 0 else q[:, 5]
            a2 = a1 + (q[:, 4] if arm == 0 else q[:, 6])
            s1, c1, s2, c2 = np.sin(a1), np.cos(a1), np.sin(a2), np.cos(a2)
            p1x, p1y = c1 * self.l1 + base, s1 * self.l1
            for link in range(2):
                ax, ay = (np.full(B, base), np.zeros(B)) if link == 0 else (p1x, p1y)
                dx, dy = (c1, s1) if link == 0 else (c2, s2)
                L = self.l1 if link == 0 else self.l2
                sp = np.clip((q[:, 0] - ax) * dx + (q[:, 1] - ay) * dy, 0.0, L)
                wx, wy = ax + dx * sp, ay + dy * sp
                nx, ny = q[:, 0] - wx, q[:, 1] - wy
                dist = np.sqrt(nx * nx + ny * ny)
                nx, ny = nx / dist, ny / dist
                gap = dist - (R + self.r_link)
                cx, cy = wx + nx * self.r_link, wy + ny * self.r_link
                r1x, r1y = cx - base, cy
                r2x, r2y = cx - p1x, cy - p1y
                tx, ty = -ny, nx
                for gen in range(2):
                    row = (arm * 2 + link) * 2 + gen
                    sg = mu if gen == 0 else -mu
                    ex, ey = nx + tx * sg, ny + ty * sg
                    phi[:, row] = gap
                    J[:, row, 0] = ex
                    J[:, row, 1] = ey
                    J[:, row, 2] = (ex * ny - ey * nx) * R
                    j1 = -(ey * r1x - ex * r1y)
                    j2 = -(ey * r2x - ex * r2y) if link == 1 else np.zeros(B)
                    J[:, row, 3 + 2 * arm] = j1
                    J[:, row, 4 + 2 * arm] = j2
        return Dinv, b, J, phi



class BoxPivotOracle(_ContactQPOracle):
    """examples/box_pivoting (box_pivoting_setup.py:6-19, run_box_pivoting.py:20-75): a 1 m square box
    on the ground y = 0, pivoted by a position-controlled disc of radius 0.1 (analysis/
    box_pivoting_analysis.py:34-72).  Reference state order x = [x_h, x_b, y_h, y_b, th_b]
    (box_pivoting_analysis.py:53-64), u = commanded hand position, indices_u_into_x = [0, 2];
    internally q = x[PERM] = [xb, yb, th, xh, yh].  Contacts (2 friction generators each): the 4 box
    corners vs the ground, the hand vs the box (closest boundary point; inside or on the boundary:
    the nearest face), the hand vs the ground.  Same scheme as PlanarHandOracle; PARITY UNPINNED
    (box mass and friction live in the absent box_1m_rotation.sdf / box_pivoting.yml)."""

    PERM = np.array([1, 3, 4, 0, 2])
    ACT = np.array([3, 4])

    ground = True            # contacts with the ground y = 0 (and gravity) present

    def __init__(self, h, mass=1.0, mu=0.5, pgs_iters=0):
        self.h = h
        self.dim_x, self.dim_u = 5, 2
        self.g = 9.81
        self.mass, self.half, self.mu = mass, 0.5, mu
        self.inertia = mass * (2 * self.half) ** 2 / 6.0      # square plate
        self.kp = 50000.0
        self.r_hand = 0.1
        self.pgs_iters = pgs_iters
        self.indices_u_into_x = self.PERM[3:].copy()

    @classmethod
    def pack(cls, box, hand):
        x = np.zeros(5)
        x[cls.PERM] = np.concatenate([box, hand])
        return x

    def params(self):
        return [self.h, self.g, self.mass, self.half, self.mu, self.kp, self.r_hand, self.pgs_iters]

    def _qp(self, q, u):
        q = np.atleast_2d(q)[:, self.PERM]
        u = np.atleast_2d(u)
        B = q.shape[0]
        h, m, a, mu, kp, rh = self.h, self.mass, self.half, self.mu, self.kp, self.r_hand
        Dinv = np.array([h * h / m, h * h / m, h * h / self.inertia, 1 / kp, 1 / kp])
        b = np.zeros((B, 5))
        b[:, 1] = m * self.g
        b[:, 3] = kp * (q[:, 3] - u[:, 0])
        b[:, 4] = kp * (q[:, 4] - u[:, 1])
        J, phi = np.zeros((B, 12, 5)), np.zeros((B, 12))
        sn, cs = np.sin(q[:, 2]), np.cos(q[:, 2])
        for c in range(4):
            lx, ly = (a if c & 1 else -a), (a if c & 2 else -a)
            rx, ry = cs * lx - sn * ly, sn * lx + cs * ly
            for gen in range(2):
                row, sg = 2 * c + gen, (mu if gen == 0 else -mu)
                phi[:, row] = q[:, 1] + ry
                J[:, row, 0], J[:, row, 1], J[:, row, 2] = sg, 1.0, rx - ry * sg
        dx, dy = q[:, 3] - q[:, 0], q[:, 4] - q[:, 1]
        px, py = cs * dx + sn * dy, -sn * dx + cs * dy
        outside = (np.abs(px) > a) | (np.abs(py) > a)
        cxq, cyq = np.clip(px, -a, a), np.clip(py, -a, a)
        ddx, ddy = a - np.abs(px), a - np.abs(py)
        facex = ddx <= ddy
        sgx, sgy = np.where(px >= 0, 1.0, -1.0), np.where(py >= 0, 1.0, -1.0)
        ex, ey = px - cxq, py - cyq
        d_out = np.sqrt(ex * ex + ey * ey)
        with np.errstate(divide="ignore", invalid="ignore"):
            nlx = np.where(outside, ex / d_out, np.where(facex, sgx, 0.0))
            nly = np.where(outside, ey / d_out, np.where(facex, 0.0, sgy))
        qlx = np.where(outside, cxq, np.where(facex, sgx * a, px))
        qly = np.where(outside, cyq, np.where(facex, py, sgy * a))
        dist = np.where(outside, d_out, -np.where(facex, ddx, ddy))
        nx, ny = cs * nlx - sn * nly, sn * nlx + cs * nly
        rx, ry = cs * qlx - sn * qly, sn * qlx + cs * qly
        for gen in range(2):
            row, sg = 8 + gen, (mu if gen == 0 else -mu)
            e_x, e_y = nx - ny * sg, ny + nx * sg
            phi[:, row] = dist - rh
            J[:, row, 0], J[:, row, 1], J[:, row, 2] = -e_x, -e_y, -(e_y * rx - e_x * ry)
            J[:, row, 3], J[:, row, 4] = e_x, e_y
        for gen in range(2):
            row, sg = 10 + gen, (mu if gen == 0 else -mu)
            phi[:, row] = q[:, 4] - rh
            J[:, row, 3], J[:, row, 4] = sg, 1.0
        if not self.ground:
            J, phi = J[:, 8:10], phi[:, 8:10]        # only the hand-box pair
        return Dinv, b, J, phi


class BoxPushOracle(BoxPivotOracle):
    """examples/box_pushing (box_pushing_setup.py:6-19, run_box_pushing.py:20-75): the same square box
    and disc hand seen from above -- no gravity (setup :18), no ground, Kp = 500 (:10).  PINNED by the
    reference's own simulator data, examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy (committed
    as tests/golden/box_pushing_*.npy): an 80-step straight push recorded from the quasistatic
    simulator, row t+1 = (step(row t's state, row t+1's command), that command), plus the simulator's
    Jacobians [Dq_next/Dq | Dq_next/Dq_a_cmd] at every row.  The data identify the parameters the absent
    SDF/YAML would hold: box mass 5 (= Kp h^2: hand and box share the first contact displacement
    equally), touching distance 0.5995, and from the sticking-contact Jacobians (lateral entries
    0.105430 / 0.894570, turning entry 1.579862 = 14.985 x the lateral one) the lever arm of the contact
    point, 0.4995 = the box's half side, with inertia 1/6; so r_hand = 0.1.  With these the analytic
    active-set derivative reproduces 79 of the simulator's 80 Jacobians to 5e-7 (the remaining one is
    the contact-onset step, where the simulator's interior-point multiplier is 1e-5 instead of 0)."""

    ground = False

    def __init__(self, h=0.1, mass=5.0, inertia=1.0 / 6.0, mu=0.5, pgs_iters=0):
        super().__init__(h, mass, mu, pgs_iters)
        self.g = 0.0
        self.inertia = inertia
        self.kp = 500.0
        self.half = 0.4995
        self.r_hand = 0.1

    def params(self):
        return [self.h, self.mass, self.inertia, self.half, self.mu, self.kp, self.r_hand, self.pgs_iters]


def zero_order_B_decoupled(system, x_trj, u_trj, du, decouple=True):
    """calc_B_zero_order (irs_lqr/quasistatic_dynamics.py:242-266, u-only noise: B by least squares, A =
    the simulator's Dq_nextDq at the nominal point) followed -- `decouple` -- by decouple_AB_matrices
    (irs_lqr/irs_lqr_quasistatic.py:275-284): A = I with the actuated columns zeroed, the actuated rows of
    B = I; c = f - A x - B u (irs_lqr_quasistatic.py:313-316)."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    idx = system.indices_u_into_x
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        ft = system.dynamics(x_trj[t], u_trj[t])
        fdt = system.dynamics_batch(np.tile(x_trj[t], (du.shape[1], 1)), u_trj[t] + du[t])
        Bt[t] = zero_order_B_fit(du[t], fdt - ft)
        if decouple:
            Bt[t][idx, :] = np.eye(m)
            At[t] = np.eye(n)
            At[t][:, idx] = 0.0
        else:
            At[t] = system.jacobian_xu(x_trj[t], u_trj[t])[:, :n]
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def _decouple(system, At, Bt):
    """decouple_AB_matrices (irs_lqr/irs_lqr_quasistatic.py:275-284)."""
    idx = system.indices_u_into_x
    Bt[:, idx, :] = np.eye(system.dim_u)
    At[:] = np.eye(system.dim_x)
    At[:, :, idx] = 0.0
    return At, Bt


def first_order_B_decoupled(system, x_trj, u_trj, du, decouple=True):
    """gradient_mode "first_order": calc_AB_first_order (irs_lqr/quasistatic_dynamics.py:193-208, u-only
    noise, mean over the samples of the simulator's [Dq_nextDq | Dq_nextDqa_cmd]) followed by
    decouple_AB_matrices, which keeps only the unactuated rows of the mean B; c = f - A x - B u
    (irs_lqr_quasistatic.py:218-225)."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        AB = system.jacobian_xu_batch(np.tile(x_trj[t], (du.shape[1], 1)), u_trj[t] + du[t])
        Bt[t] = AB[:, :, n:].mean(0)
        At[t] = AB[:, :, :n].mean(0)
    if decouple:
        At, Bt = _decouple(system, At, Bt)
    for t in range(T):
        ft = system.dynamics(x_trj[t], u_trj[t])
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def first_order_sums(system, x_trj, u_trj, du):
    """The (T, n*m) statistics of the contact first-order sample pass (include/irs_hip.h,
    IRS_SMOOTH_FIRST_ORDER on a contact model): per time step the SUM over this shard's samples of the
    n x m block B of the step's active-set derivative, row-major in the reference's x order."""
    T, n = u_trj.shape[0], system.dim_x
    out = np.zeros((T, n * system.dim_u))
    for t in range(T):
        AB = system.jacobian_xu_batch(np.tile(x_trj[t], (du.shape[1], 1)), u_trj[t] + du[t])
        out[t] = AB[:, :, n:].sum(0).reshape(-1)
    return out


def first_order_from_sums(system, x_trj, u_trj, sums, n_total):
    """The solve launch of that mode: mean B inside the decoupled structure, c = f - A x - B u."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    At, Bt, ct = np.zeros((T, n, n)), (np.asarray(sums) / float(n_total)).reshape(T, n, m), np.zeros((T, n))
    At, Bt = _decouple(system, At, Bt)
    for t in range(T):
        ft = system.dynamics(x_trj[t], u_trj[t])
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def exact_contact_TV(system, x_trj, u_trj, decouple=True):
    """gradient_mode "exact": calc_AB_exact (irs_lqr/quasistatic_dynamics.py:189-191) at every nominal
    point, optional decouple_AB_matrices, c = f - A x - B u."""
    T, n = u_trj.shape[0], system.dim_x
    AB = system.jacobian_xu_batch(x_trj[:T], u_trj)
    At, Bt = AB[:, :, :n].copy(), AB[:, :, n:].copy()
    if decouple:
        At, Bt = _decouple(system, At, Bt)
    ct = np.zeros((T, n))
    for t in range(T):
        ft = system.dynamics(x_trj[t], u_trj[t])
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


SYSTEMS = {"pendulum": PendulumOracle, "quadrotor": QuadrotorOracle, "bicycle": BicycleOracle,
           "three_cart": ThreeCartOracle, "planar_hand": PlanarHandOracle, "box_pivoting": BoxPivotOracle, "box_pushing": BoxPushOracle}


# --------------------------------------------------------------------------
# Trajectory helpers (irs_lqr/irs_lqr.py)
# --------------------------------------------------------------------------
def rollout(system, x0, u_trj):
    """irs_lqr/irs_lqr.py:105-119."""
    T = u_trj.shape[0]
    x_trj = np.zeros((T + 1, system.dim_x))
    x_trj[0] = x0
    for t in range(T):
        x_trj[t + 1] = system.dynamics(x_trj[t], u_trj[t])
    return x_trj


def evaluate_cost(x_trj, u_trj, xd_trj, Q, R):
    """irs_lqr/irs_lqr.py:121-137.  NB the terminal term uses Q, not Qd (:135-136)."""
    T = u_trj.shape[0]
    cost = 0.0
    for t in range(T):
        et = x_trj[t] - xd_trj[t]
        cost += et.dot(Q).dot(et)
        cost += u_trj[t].dot(R).dot(u_trj[t])
    et = x_trj[T] - xd_trj[T]
    cost += et.dot(Q).dot(et)
    return cost


# --------------------------------------------------------------------------
# Smoothed linearisations
# --------------------------------------------------------------------------
def compute_least_squares(dxdu, deltaf, dim_x, dim_u):
    """irs_lqr/irs_lqr_zero_order.py:27-36 (SVD lstsq, no intercept)."""
    ABhat = np.linalg.lstsq(dxdu, deltaf, rcond=None)[0].transpose()
    return ABhat[:, :dim_x], ABhat[:, dim_x:dim_x + dim_u]


def zero_order_TV(system, x_trj, u_trj, dx, du):
    """irs_lqr/irs_lqr_zero_order.py:38-63 with the samples SUPPLIED:
    dx (T,N,n), du (T,N,m) are what `sampling(x_t,u_t,iter)` returned at each t."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        fdt = system.dynamics_batch(x_trj[t] + dx[t], u_trj[t] + du[t])
        ft = system.dynamics(x_trj[t], u_trj[t])
        deltaf = fdt - ft
        dxdu = np.hstack((dx[t], du[t]))
        At[t], Bt[t] = compute_least_squares(dxdu, deltaf, n, m)
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def first_order_TV(system, x_trj, u_trj, dx, du):
    """irs_lqr/irs_lqr_first_order.py:28-54 with the samples supplied."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        AB_batch = system.jacobian_xu_batch(x_trj[t] + dx[t], u_trj[t] + du[t])
        ABhat = np.mean(AB_batch, axis=0)
        At[t] = ABhat[:, :n]
        Bt[t] = ABhat[:, n:n + m]
        ct[t] = system.dynamics(x_trj[t], u_trj[t]) - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def exact_TV(system, x_trj, u_trj):
    """irs_lqr/irs_lqr_exact.py:15-31."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        AB = system.jacobian_xu(x_trj[t], u_trj[t])
        At[t] = AB[:, :n]
        Bt[t] = AB[:, n:n + m]
        ct[t] = system.dynamics(x_trj[t], u_trj[t]) - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def zero_order_B_fit(du, dx_next):
    """irs_lqr/quasistatic_dynamics.py:242-266: B-only least squares on
    u-perturbations (`np.linalg.lstsq(du, dx_next)[0].T`)."""
    return np.linalg.lstsq(du, dx_next, rcond=None)[0].transpose()


def zero_order_AB_damped_decoupled(system, x_trj, u_trj, dx, du, damp=1e-2, decouple=True):
    """calc_AB_zero_order (irs_lqr/quasistatic_dynamics.py:268-300): least squares of the one-step
    responses on [dx | du] with `damp`-weighted identity rows appended (Tikhonov), followed by
    decouple_AB_matrices (irs_lqr_quasistatic.py:275-284) and c = f - A x - B u."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    d = n + m
    idx = system.indices_u_into_x
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        ft = system.dynamics(x_trj[t], u_trj[t])
        dfn = system.dynamics_batch(x_trj[t] + dx[t], u_trj[t] + du[t]) - ft
        lhs = np.vstack([np.hstack([dx[t], du[t]]), damp * np.eye(d)])
        rhs = np.vstack([dfn, np.zeros((d, n))])
        AB = np.linalg.lstsq(lhs, rhs, rcond=None)[0].T
        Bt[t] = AB[:, n:]
        if decouple:
            Bt[t][idx, :] = np.eye(m)
            At[t] = np.eye(n)
            At[t][:, idx] = 0.0
        else:
            At[t] = AB[:, :n]
        ct[t] = ft - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


def gaussian_samples(T, N, std_x, std_u, it, power=0.5):
    """The sampling closure of the example scripts, e.g.
    examples/pendulum/pendulum_zero_order.py:38-43: per t, dx~N(0,std_x/iter^p)
    (N,n) then du~N(0,std_u/iter^p) (N,m), drawn from the global legacy RNG."""
    n, m = len(std_x), len(std_u)
    dx = np.zeros((T, N, n))
    du = np.zeros((T, N, m))
    for t in range(T):
        dx[t] = np.random.normal(0.0, np.asarray(std_x) / (it ** power), size=(N, n))
        du[t] = np.random.normal(0.0, np.asarray(std_u) / (it ** power), size=(N, m))
    return dx, du


# --------------------------------------------------------------------------
# TV-LQR
# --------------------------------------------------------------------------
def solve_tvlqr_qp(At, Bt, ct, Q, Qd, R, x0, x_trj_d):
    """irs_lqr/tv_lqr.py:30-145 for indices_u_into_x=None and inactive bounds,
    restated literally as the QP it poses and solved through its KKT system:

      min  sum_t (x_t-xd_t)'Q(x_t-xd_t) + 1/2 u_t'R u_t  + (x_T-xd_T)'Qd(x_T-xd_T)
      s.t. x_0 = x0 (:82),  A_t x_t + B_t u_t - x_{t+1} = -c_t (:87-90)

    The 1/2 on R is Drake's AddQuadraticCost(Q,b,x) = 1/2 x'Qx + b'x (:110);
    AddQuadraticErrorCost (:127,:130) is the full quadratic form."""
    T = At.shape[0]
    n, m = Q.shape[0], R.shape[0]
    nz = (T + 1) * n + T * m
    ix = lambda t: slice(t * n, (t + 1) * n)
    iu = lambda t: slice((T + 1) * n + t * m, (T + 1) * n + (t + 1) * m)
    H = np.zeros((nz, nz))      # objective = 1/2 z'Hz + g'z
    g = np.zeros(nz)
    for t in range(T):
        H[ix(t), ix(t)] += 2.0 * Q
        g[ix(t)] += -2.0 * Q.dot(x_trj_d[t])
        H[iu(t), iu(t)] += R
    H[ix(T), ix(T)] += 2.0 * Qd
    g[ix(T)] += -2.0 * Qd.dot(x_trj_d[T])
    ne = (T + 1) * n
    E = np.zeros((ne, nz))
    b = np.zeros(ne)
    E[0:n, ix(0)] = np.eye(n)
    b[0:n] = x0
    for t in range(T):
        r = slice((t + 1) * n, (t + 2) * n)
        E[r, ix(t)] = At[t]
        E[r, iu(t)] = Bt[t]
        E[r, ix(t + 1)] = -np.eye(n)
        b[r] = -ct[t]
    KKT = np.block([[H, E.T], [E, np.zeros((ne, ne))]])
    sol = np.linalg.solve(KKT, np.concatenate([-g, b]))
    z = sol[:nz]
    return z[:(T + 1) * n].reshape(T + 1, n), z[(T + 1) * n:].reshape(T, m)


def tvlqr_riccati(At, Bt, ct, Q, Qd, R, x_trj_d, alpha_R=0.5):
    """Backward Riccati pass for the QP of solve_tvlqr_qp (the gains the
    reference never materialises).  Value function
    V_t(x) = x'P_t x + 2 p_t'x + const, P_T=Qd, p_T=-Qd xd_T:
        H = alpha_R R + B'PB,  K = -H^-1 B'PA,  k = -H^-1 B'(Pc+p)
        P <- Q + A'P(A+BK),    p <- -Q xd_t + (A+BK)'(Pc+p)
    Returns K (T,m,n), k (T,m)."""
    T = At.shape[0]
    n, m = Q.shape[0], R.shape[0]
    K = np.zeros((T, m, n))
    k = np.zeros((T, m))
    P = Qd.copy()
    p = -Qd.dot(x_trj_d[T])
    for t in range(T - 1, -1, -1):
        A, B, c = At[t], Bt[t], ct[t]
        PB = P.dot(B)
        H = alpha_R * R + B.T.dot(PB)
        q = P.dot(c) + p
        K[t] = -np.linalg.solve(H, PB.T.dot(A))
        k[t] = -np.linalg.solve(H, B.T.dot(q))
        Acl = A + B.dot(K[t])
        P = Q + A.T.dot(P).dot(Acl)
        P = 0.5 * (P + P.T)
        p = -Q.dot(x_trj_d[t]) + Acl.T.dot(q)
    return K, k


def closed_loop_rollout(system, K, k, x0):
    """irs_lqr/irs_lqr.py:169-184 with the re-solved QP's first control written
    as the affine policy u_t = K_t x_t + k_t, and the TRUE dynamics (:184)."""
    T, m, n = K.shape
    x_new = np.zeros((T + 1, n))
    u_new = np.zeros((T, m))
    x_new[0] = x0
    for t in range(T):
        u_new[t] = K[t].dot(x_new[t]) + k[t]
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new


def local_descent_qp(system, At, Bt, ct, Q, Qd, R, x0, xd_trj):
    """irs_lqr/irs_lqr.py:148-186 literally: T tail re-solves, keep u*[0]."""
    T = At.shape[0]
    n, m = system.dim_x, system.dim_u
    x_new = np.zeros((T + 1, n))
    u_new = np.zeros((T, m))
    x_new[0] = x0
    for t in range(T):
        _, u_star = solve_tvlqr_qp(At[t:T], Bt[t:T], ct[t:T], Q, Qd, R,
                                   x_new[t], xd_trj[t:T + 1])
        u_new[t] = u_star[0]
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new


def local_descent(system, At, Bt, ct, Q, Qd, R, x0, xd_trj):
    """Riccati form of local_descent_qp (identical result, O(T) instead of O(T^2))."""
    K, k = tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd_trj, alpha_R=0.5)
    x_new, u_new = closed_loop_rollout(system, K, k, x0)
    return x_new, u_new, K, k


def iterate(system, Q, Qd, R, x0, xd_trj, u_trj_initial, max_iterations,
            tv_fn):
    """irs_lqr/irs_lqr.py:35-71 (ctor) + :188-218 (iterate).  `tv_fn(x_trj,
    u_trj, iter) -> At,Bt,ct` is get_TV_matrices.  Performs max_iterations+1
    descents; the last is logged but not adopted (:205-216)."""
    u_trj = u_trj_initial
    x_trj = rollout(system, x0, u_trj)
    cost = evaluate_cost(x_trj, u_trj, xd_trj, Q, R)
    cost_lst, x_lst, u_lst = [cost], [x_trj], [u_trj]
    it = 1
    while True:
        At, Bt, ct = tv_fn(x_trj, u_trj, it)
        x_new, u_new, _, _ = local_descent(system, At, Bt, ct, Q, Qd, R,
                                           x_trj[0], xd_trj)
        cost_new = evaluate_cost(x_new, u_new, xd_trj, Q, R)
        x_lst.append(x_new)
        u_lst.append(u_new)
        cost_lst.append(cost_new)
        if it > max_iterations:
            break
        cost, x_trj, u_trj = cost_new, x_new, u_new
        it += 1
    return x_trj, u_trj, cost, cost_lst, x_lst, u_lst


# --------------------------------------------------------------------------
# CEM baseline (irs_lqr/cem.py)
# --------------------------------------------------------------------------
def cem_local_descent(system, x0, u_trj, std_trj, xd_trj, Q, R, n_elite,
                      u_trj_candidates):
    """irs_lqr/cem.py:151-184 with the candidates supplied
    (u_cand ~ N(u_trj, std_trj), shape (B,T,m), drawn at :159-161)."""
    B = u_trj_candidates.shape[0]
    cost_array = np.zeros(B)
    for b in range(B):
        cost_array[b] = evaluate_cost(rollout(system, x0, u_trj_candidates[b]),
                                      u_trj_candidates[b], xd_trj, Q, R)
    best_idx = np.argpartition(cost_array, n_elite)[:n_elite]
    best = u_trj_candidates[best_idx]
    u_new = np.mean(best, axis=0)
    std_new = np.std(best, axis=0)
    x_new = rollout(system, x0, u_new)
    return x_new, u_new, std_new, cost_array


# --------------------------------------------------------------------------
# Device-RNG specification (mode G).  Not reference arithmetic: the reference
# draws from NumPy's global MT19937 (e.g. pendulum_zero_order.py:38-43), which a
# GPU cannot reproduce in parallel.  This restates the counter-based generator
# the HIP library implements so its on-device samples can be checked exactly.
# --------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al. 2011).  ctr: (...,4) uint32, key: (2,) ints."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PHILOX_M0 * c[0]
        p1 = _PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return np.stack([v.astype(np.uint32) for v in c], axis=-1)


def device_gaussian_samples(T, N, n, m, std_x, std_u, seed, it,
                            sample_offset=0, dtype=np.float64):
    """Samples exactly as irs_mpc_amd/csrc draws them in mode G:
    counter = (global sample index, t, block j, iter), key = (seed lo, seed hi);
    each Philox call yields 4 uint32 -> 2 Box-Muller pairs -> 4 normals, which
    fill components 4j..4j+3 of z=[dx|du]; component c is scaled by std[c]."""
    d = n + m
    nblk = (d + 3) // 4
    std = np.concatenate([np.asarray(std_x, float), np.asarray(std_u, float)])
    s_idx = (np.arange(N, dtype=np.uint64) + np.uint64(sample_offset))
    z = np.zeros((T, N, nblk * 4))
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    for t in range(T):
        for j in range(nblk):
            ctr = np.zeros((N, 4), dtype=np.uint32)
            ctr[:, 0] = (s_idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            ctr[:, 1] = t
            ctr[:, 2] = j | ((s_idx >> np.uint64(32)).astype(np.uint32) << np.uint32(8))
            ctr[:, 3] = it
            r = philox4x32_10(ctr, key).astype(np.float64)
            u = (r + 0.5) * (1.0 / 4294967296.0)        # (0,1)
            for pair in range(2):
                rad = np.sqrt(-2.0 * np.log(u[:, 2 * pair]))
                ang = 2.0 * np.pi * u[:, 2 * pair + 1]
                z[t, :, 4 * j + 2 * pair] = rad * np.cos(ang)
                z[t, :, 4 * j + 2 * pair + 1] = rad * np.sin(ang)
    z = z[:, :, :d] * std
    return z[:, :, :n].astype(dtype), z[:, :, n:].astype(dtype)


# --------------------------------------------------------------------------
# Sufficient-statistics ("Gram") form of the zero-order fit.  Algebraically the
# same estimator as compute_least_squares (normal equations of the lstsq at
# irs_lqr_zero_order.py:33); this is the form that shards over devices: the
# sums of disjoint sample subsets ADD.  Used by tests to check the sharding /
# all-reduce logic and the device `sums` buffers.
# --------------------------------------------------------------------------
def zero_order_sums(system, x_trj, u_trj, dx, du, sum_z=False):
    """(T,P), the layout of irs_hip.h `sums`: upper-triangular Gram of z=[dx|du] followed by z df'
    row-major, df = f(x+dx,u+du) - f(x,u).  sum_z=True (the contact models' layout): df = f(..) - xb,
    xb = the nominal state rounded to f32, and sum(z) is appended."""
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    d = n + m
    iu = np.triu_indices(d)
    ng = len(iu[0])
    out = np.zeros((T, ng + d * n + (d if sum_z else 0)))
    for t in range(T):
        z = np.hstack((dx[t], du[t]))
        ref = x_trj[t].astype(np.float32).astype(np.float64) if sum_z else system.dynamics(x_trj[t], u_trj[t])
        df = system.dynamics_batch(x_trj[t] + dx[t], u_trj[t] + du[t]) - ref
        out[t, :ng] = (z.T @ z)[iu]
        out[t, ng:ng + d * n] = (z.T @ df).ravel()
        if sum_z:
            out[t, ng + d * n:] = z.sum(axis=0)
    return out


def zero_order_from_sums(system, x_trj, u_trj, sums):
    T, n, m = u_trj.shape[0], system.dim_x, system.dim_u
    d = n + m
    iu = np.triu_indices(d)
    ng = len(iu[0])
    sum_z = sums.shape[1] == ng + d * n + d
    At, Bt, ct = np.zeros((T, n, n)), np.zeros((T, n, m)), np.zeros((T, n))
    for t in range(T):
        G = np.zeros((d, d))
        G[iu] = sums[t, :ng]
        G = G + np.triu(G, 1).T
        f = system.dynamics(x_trj[t], u_trj[t])
        H = sums[t, ng:ng + d * n].reshape(d, n)
        if sum_z:
            H = H - np.outer(sums[t, ng + d * n:], f - x_trj[t].astype(np.float32).astype(np.float64))
        AB = np.linalg.solve(G, H).T
        At[t], Bt[t] = AB[:, :n], AB[:, n:]
        ct[t] = f - At[t].dot(x_trj[t]) - Bt[t].dot(u_trj[t])
    return At, Bt, ct


# --------------------------------------------------------------------------
# Box-constrained TV-LQR (irs_lqr/tv_lqr.py:112-123 with ACTIVE bounds) -- the QP the
# reference hands to OSQP, solved here by ADMM on the box with the equality-
# constrained (LQR) sub-problem solved exactly by a Riccati sweep:
#     min f(z) + I_box(w)  s.t. z = w,   f = quadratic cost + dynamics,
#     z <- argmin f(z) + rho/2 |z - w + y|^2 ; w <- clip(z + y) ; y <- y + z - w.
# Only BOUNDED components carry a rho term.  The Riccati matrices depend on
# (A,B,Q,R,rho) only: one factorisation serves every ADMM iteration AND every one of
# the T MPC tail re-solves of IrsLqr.local_descent (irs_lqr.py:169-184).
# OSQP's default accuracy is 1e-3 (Drake OsqpSolver), so the reference's own curves
# carry that much solver noise; this restatement converges to ~1e-9.
# --------------------------------------------------------------------------
def _box_masks(lo, hi):
    """A component carries a rho term if it has a finite bound at ANY time (bounds may be one
    row or one row per time step)."""
    lo, hi = np.atleast_2d(np.asarray(lo, float)), np.atleast_2d(np.asarray(hi, float))
    return lo, hi, (np.isfinite(lo) | np.isfinite(hi)).any(axis=0).astype(float)


def _rows(b, rows, width):
    """Bounds as one row per time step: (width,) is repeated, (rows,width) passes through."""
    return np.broadcast_to(np.asarray(b, float), (rows, width))


def tvlqr_box_factor(At, Bt, ct, Q, Qd, R, xlo, xhi, ulo, uhi, rho, alpha_R=0.5):
    """Backward sweep with Q^ = Q + rho/2 Mx, R^ = alpha R + rho/2 Mu (M = bounded mask)."""
    T, n, m = At.shape[0], Q.shape[0], R.shape[0]
    _, _, mx = _box_masks(xlo, xhi)
    _, _, mu = _box_masks(ulo, uhi)
    Qh, Qdh = Q + 0.5 * rho * np.diag(mx), Qd + 0.5 * rho * np.diag(mx)
    Rh = alpha_R * R + 0.5 * rho * np.diag(mu)
    F = dict(K=np.zeros((T, m, n)), Acl=np.zeros((T, n, n)), Hinv=np.zeros((T, m, m)),
             Minv=np.zeros((T, m, n)), d=np.zeros((T, n)), mx=mx, mu=mu, rho=rho)
    P = Qdh.copy()
    for t in range(T - 1, -1, -1):
        A, B = At[t], Bt[t]
        H = Rh + B.T @ P @ B
        Hinv = np.linalg.inv(H)
        F["Hinv"][t] = Hinv
        F["Minv"][t] = Hinv @ B.T
        F["K"][t] = -Hinv @ B.T @ P @ A
        F["Acl"][t] = A + B @ F["K"][t]
        F["d"][t] = P @ ct[t]
        P = Qh + A.T @ P @ F["Acl"][t]
        P = 0.5 * (P + P.T)
    return F


def tvlqr_box_solve(F, At, Bt, ct, Q, Qd, xd, x_start, t0, xlo, xhi, ulo, uhi, state=None,
                    max_iter=2000, eps=1e-9, relax=1.0):
    """ADMM for the tail problem t0..T from the fixed state x_start.  `state` = (wx, yx,
    wu, yu) warm start (arrays over the FULL horizon).  Returns x (T+1,n), u (T,m) (rows
    < t0 untouched/zero), state, iterations."""
    T, n = At.shape[0], Q.shape[0]
    m = Bt.shape[2]
    rho, mx, mu = F["rho"], F["mx"], F["mu"]
    xlo, xhi = _rows(xlo, T + 1, n), _rows(xhi, T + 1, n)       # row t bounds x_t
    ulo, uhi = _rows(ulo, T, m), _rows(uhi, T, m)               # row t bounds u_t
    if state is None:
        state = (np.zeros((T + 1, n)), np.zeros((T + 1, n)), np.zeros((T, m)), np.zeros((T, m)))
    wx, yx, wu, yu = state
    zx, zu, k = np.zeros((T + 1, n)), np.zeros((T, m)), np.zeros((T, m))
    it = 0
    for it in range(1, max_iter + 1):
        # backward affine sweep
        p = -(Qd @ xd[T] + 0.5 * rho * mx * (wx[T] - yx[T]))
        for t in range(T - 1, t0 - 1, -1):
            g = F["d"][t] + p
            s = -0.5 * rho * mu * (wu[t] - yu[t])
            k[t] = -(F["Minv"][t] @ g + F["Hinv"][t] @ s)
            q = -(Q @ xd[t] + 0.5 * rho * mx * (wx[t] - yx[t]))
            p = q + F["Acl"][t].T @ g + F["K"][t].T @ s
        # forward sweep on the linear model
        zx[t0] = x_start
        for t in range(t0, T):
            zu[t] = F["K"][t] @ zx[t] + k[t]
            zx[t + 1] = At[t] @ zx[t] + Bt[t] @ zu[t] + ct[t]
        # box projection + dual update (x_t0 is fixed, not a variable)
        zrx = relax * zx[t0 + 1:] + (1.0 - relax) * wx[t0 + 1:]      # over-relaxation (OSQP's alpha)
        zru = relax * zu[t0:] + (1.0 - relax) * wu[t0:]
        wxn = np.clip(zrx + yx[t0 + 1:], xlo[t0 + 1:], xhi[t0 + 1:])
        wun = np.clip(zru + yu[t0:], ulo[t0:], uhi[t0:])
        rp = max(np.abs(mx * (zx[t0 + 1:] - wxn)).max(), np.abs(mu * (zu[t0:] - wun)).max())
        rd = rho * max(np.abs(mx * (wxn - wx[t0 + 1:])).max(), np.abs(mu * (wun - wu[t0:])).max())
        yx[t0 + 1:] += mx * (zrx - wxn)
        yu[t0:] += mu * (zru - wun)
        wx[t0 + 1:], wu[t0:] = wxn, wun
        if max(rp, rd) < eps:
            break
    return zx, zu, (wx, yx, wu, yu), it


def local_descent_box(system, At, Bt, ct, Q, Qd, R, x0, xd_trj, xlo, xhi, ulo, uhi, rho=1.0,
                      max_iter=2000, eps=1e-9):
    """irs_lqr/irs_lqr.py:148-186 with active abs bounds: T tail re-solves (warm started),
    first control applied to the TRUE dynamics."""
    T = At.shape[0]
    n, m = system.dim_x, system.dim_u
    F = tvlqr_box_factor(At, Bt, ct, Q, Qd, R, xlo, xhi, ulo, uhi, rho)
    x_new, u_new = np.zeros((T + 1, n)), np.zeros((T, m))
    x_new[0] = x0
    state, iters = None, []
    ulo_t, uhi_t = _rows(ulo, T, m), _rows(uhi, T, m)
    for t in range(T):
        zx, zu, state, it = tvlqr_box_solve(F, At, Bt, ct, Q, Qd, xd_trj, x_new[t], t, xlo, xhi, ulo, uhi,
                                            state, max_iter, eps)
        iters.append(it)
        u_new[t] = np.clip(zu[t], ulo_t[t], uhi_t[t])
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new, iters


def qp_box_kkt_residuals(At, Bt, ct, Q, Qd, R, x0, xd, xlo, xhi, ulo, uhi, x, u, alpha_R=0.5, tol=1e-6):
    """Optimality certificate of (x,u) for the box-constrained QP of tv_lqr.py:69-137,
    independent of how it was solved.  With z = (x_1..x_T, u_0..u_{T-1}) (x_0 is fixed),
    objective 1/2 z'Hz + g'z, dynamics E z = b and box l <= z <= h, KKT reads
        H z + g + E'nu + mu = 0,  mu_i >= 0 on an active upper bound, <= 0 on an active
        lower bound, = 0 elsewhere.
    (nu, mu_active) are recovered by least squares; returns (dynamics residual, bound
    violation, stationarity residual, worst multiplier sign violation) -- all ~0 at the
    solution."""
    T, n = At.shape[0], Q.shape[0]
    m = R.shape[0]
    xlo, xhi = _rows(xlo, T + 1, n), _rows(xhi, T + 1, n)
    ulo, uhi = _rows(ulo, T, m), _rows(uhi, T, m)
    nz = T * n + T * m
    ix = lambda t: slice((t - 1) * n, t * n)                 # x_t, t = 1..T
    iu = lambda t: slice(T * n + t * m, T * n + (t + 1) * m)
    H, g = np.zeros((nz, nz)), np.zeros(nz)
    for t in range(1, T + 1):
        W = Qd if t == T else Q
        H[ix(t), ix(t)] = 2.0 * W
        g[ix(t)] = -2.0 * W @ xd[t]
    for t in range(T):
        H[iu(t), iu(t)] = 2.0 * alpha_R * R
    E, b = np.zeros((T * n, nz)), np.zeros(T * n)
    for t in range(T):                                       # A x_t + B u_t - x_{t+1} = -c_t
        r = slice(t * n, (t + 1) * n)
        if t > 0:
            E[r, ix(t)] = At[t]
        else:
            b[r] -= At[0] @ x0
        E[r, iu(t)] = Bt[t]
        E[r, ix(t + 1)] = -np.eye(n)
        b[r] -= ct[t]
    z = np.concatenate([x[1:].ravel(), u.ravel()])
    lo = np.concatenate([xlo[1:].ravel(), ulo.ravel()])
    hi = np.concatenate([xhi[1:].ravel(), uhi.ravel()])
    r_dyn = np.abs(E @ z - b).max()
    r_box = max(0.0, (lo - z).max(), (z - hi).max())
    at_hi, at_lo = z >= hi - tol, z <= lo + tol
    act = np.where(at_hi | at_lo)[0]
    Msys = np.hstack([E.T, np.eye(nz)[:, act]])
    sol = np.linalg.lstsq(Msys, -(H @ z + g), rcond=None)[0]
    r_stat = np.abs(Msys @ sol + H @ z + g).max()
    mu = sol[T * n:]
    sign_bad = 0.0
    for j, i in enumerate(act):
        if at_hi[i] and not at_lo[i]:
            sign_bad = max(sign_bad, -mu[j])
        elif at_lo[i] and not at_hi[i]:
            sign_bad = max(sign_bad, mu[j])
    return r_dyn, r_box, r_stat, sign_bad


# --------------------------------------------------------------------------
# IrsLqrQuasistatic (irs_lqr/irs_lqr_quasistatic.py): position-controlled systems
# --------------------------------------------------------------------------
def quasistatic_augment(At, Bt, ct, Q, Qd, xd_trj):
    """solve_tvlqr with indices_u_into_x (irs_lqr/tv_lqr.py:96-108) penalises du_t = u_t - u_{t-1}
    (du_0 = u_0 - x_0[idx]).  With z = [x; u_prev] and v = du it is a plain TV-LQR:
        z+ = [[A, B], [0, I]] z + [B; I] v + [c; 0],   cost (x-xd)'Q(x-xd) + v'R v."""
    T, n, m = Bt.shape
    N = n + m
    Ab, Bb, cb = np.zeros((T, N, N)), np.zeros((T, N, m)), np.zeros((T, N))
    Ab[:, :n, :n], Ab[:, :n, n:], Ab[:, n:, n:] = At, Bt, np.eye(m)
    Bb[:, :n], Bb[:, n:] = Bt, np.eye(m)
    cb[:, :n] = ct
    Qb, Qdb = np.zeros((N, N)), np.zeros((N, N))
    Qb[:n, :n], Qdb[:n, :n] = Q, Qd
    xdb = np.hstack([xd_trj, np.zeros((xd_trj.shape[0], m))])
    return Ab, Bb, cb, Qb, Qdb, xdb


def quasistatic_bounds(x_trj, idx, x_bounds_abs=None, u_bounds_abs=None, u_bounds_rel=None):
    """irs_lqr_quasistatic.py:303-325: abs bounds are trust-region OFFSETS around the nominal
    trajectory (u's around the nominal actuated positions x_trj[:-1, idx]), rel bounds limit
    u_t - u_{t-1}.  Returns absolute per-time rows (x_lo, x_hi (T+1,n); u_lo, u_hi, du_lo, du_hi
    (T,m)), +-inf where absent."""
    T, n, m = x_trj.shape[0] - 1, x_trj.shape[1], len(idx)
    inf = np.inf
    x_lo, x_hi = np.full((T + 1, n), -inf), np.full((T + 1, n), inf)
    u_lo, u_hi = np.full((T, m), -inf), np.full((T, m), inf)
    du_lo, du_hi = np.full((T, m), -inf), np.full((T, m), inf)
    if x_bounds_abs is not None:
        x_lo, x_hi = x_trj + x_bounds_abs[0], x_trj + x_bounds_abs[1]
    if u_bounds_abs is not None:
        u_lo, u_hi = x_trj[:-1, idx] + u_bounds_abs[0], x_trj[:-1, idx] + u_bounds_abs[1]
    if u_bounds_rel is not None:
        du_lo = np.tile(np.asarray(u_bounds_rel[0], float), (T, 1))
        du_hi = np.tile(np.asarray(u_bounds_rel[1], float), (T, 1))
    return x_lo, x_hi, u_lo, u_hi, du_lo, du_hi


def eval_cost_quasistatic(x_trj, u_trj, xd_trj, Q, Qd, R, idx):
    """irs_lqr_quasistatic.py:153-194: state error with Q (terminal: Qd, unlike IrsLqr), input cost
    on u_t - u_{t-1} with du_0 = u_0 - x_0[idx]."""
    T = u_trj.shape[0]
    cost = 0.0
    for t in range(T):
        e = x_trj[t] - xd_trj[t]
        du = u_trj[t] - (x_trj[0, idx] if t == 0 else u_trj[t - 1])
        cost += e.dot(Q).dot(e) + du.dot(R).dot(du)
    e = x_trj[T] - xd_trj[T]
    return cost + e.dot(Qd).dot(e)


def local_descent_quasistatic(system, At, Bt, ct, Q, Qd, R, x0, xd_trj, x_lo, x_hi, u_lo, u_hi, du_lo, du_hi,
                              rho=1.0, max_iter=4000, eps=1e-9, relax=1.0):
    """irs_lqr_quasistatic.py:326-345: for every t the tail QP is re-solved from the realised
    state -- whose local du_0 is measured from the realised actuated position x_t[idx]
    (tv_lqr.py:99-100) -- and its first control goes to the TRUE dynamics.  Bounds as returned by
    quasistatic_bounds.  The QPs are solved by the box ADMM above on the augmented problem."""
    T, n, m = Bt.shape
    idx = system.indices_u_into_x
    Ab, Bb, cb, Qb, Qdb, xdb = quasistatic_augment(At, Bt, ct, Q, Qd, xd_trj)
    # z_t = [x_t; u_{t-1}]: the u bounds at time t sit on the u_prev block of z_{t+1}
    zlo = np.hstack([x_lo, np.vstack([np.full((1, m), -np.inf), u_lo])])
    zhi = np.hstack([x_hi, np.vstack([np.full((1, m), np.inf), u_hi])])
    F = tvlqr_box_factor(Ab, Bb, cb, Qb, Qdb, R, zlo, zhi, du_lo, du_hi, rho, alpha_R=1.0)
    x_new, u_new = np.zeros((T + 1, n)), np.zeros((T, m))
    x_new[0] = x0
    state, iters = None, []
    for t in range(T):
        z_t = np.concatenate([x_new[t], x_new[t][idx]])
        zx, zu, state, it = tvlqr_box_solve(F, Ab, Bb, cb, Qb, Qdb, xdb, z_t, t, zlo, zhi, du_lo, du_hi,
                                            state, max_iter, eps, relax)
        iters.append(it)
        v = np.clip(zu[t], du_lo[t], du_hi[t])
        u_new[t] = np.clip(z_t[n:] + v, u_lo[t], u_hi[t])
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new, iters


# --------------------------------------------------------------------------
# Control-box LQR by a primal-dual active-set method (the fast path of the device's quasistatic
# descent when only ONE of u_bounds_abs / u_bounds_rel is given -- all the reference's examples)
# --------------------------------------------------------------------------
def quasistatic_ctrl_problem(At, Bt, ct, Q, Qd, R, xd_trj, kind):
    """The QP of tv_lqr.py:69-137 (position-controlled branch) written so that the BOUNDED
    quantity is the control of an LQR with state s = [x; u_prev]:
      kind "rel": control v_t = u_t - u_{t-1};  s+ = [[A,B],[0,I]] s + [B;I] v + [c;0];
                  stage cost (x-xd)'Q(x-xd) + v'Rv
      kind "abs": control u_t;                  s+ = [[A,0],[0,0]] s + [B;I] u + [c;0];
                  stage cost (x-xd)'Q(x-xd) + (u-w)'R(u-w)  =  (s-sd)'diag(Q,R)(s-sd) + u'Ru + 2 s'Nc u,
                  Nc = [0; -R]
    Returns dict(A,B,c (T,..), Qs, Qsd, Nc, Ru, sd (T+1, n+m))."""
    T, n, m = Bt.shape
    Ns = n + m
    A, B, c = np.zeros((T, Ns, Ns)), np.zeros((T, Ns, m)), np.zeros((T, Ns))
    A[:, :n, :n] = At
    B[:, :n], B[:, n:] = Bt, np.eye(m)
    c[:, :n] = ct
    Qs, Qsd, Nc = np.zeros((Ns, Ns)), np.zeros((Ns, Ns)), np.zeros((Ns, m))
    Qs[:n, :n], Qsd[:n, :n] = Q, Qd
    if kind == "rel":
        A[:, :n, n:] = Bt
        A[:, n:, n:] = np.eye(m)
    elif kind == "abs":
        Qs[n:, n:] = R
        Nc[n:] = -R
    else:
        raise ValueError(kind)
    sd = np.hstack([xd_trj, np.zeros((T + 1, m))])
    return dict(A=A, B=B, c=c, Qs=Qs, Qsd=Qsd, Nc=Nc, Ru=0.5 * (R + R.T), sd=sd, n=n, m=m, kind=kind)


def ctrlbox_backward(prob, act, lo, hi, t_hi, t_lo, W):
    """Policy evaluation/improvement sweep t = t_hi .. t_lo (inclusive, descending) for the
    active sets act (T,m) in {-1: at lo, 0: free, +1: at hi}: free components are minimised, pinned
    ones are constants.  W caches per-time (P, p, K, k, H, G, g); row t_hi+1 of P,p must be valid.
    V_t(s) = s'P s + 2 p's + const."""
    A, B, c, Qs, Nc, Ru, sd = (prob[k] for k in ("A", "B", "c", "Qs", "Nc", "Ru", "sd"))
    m = prob["m"]
    for t in range(t_hi, t_lo - 1, -1):
        P, p = W["P"][t + 1], W["p"][t + 1]
        PB = P @ B[t]
        H = Ru + B[t].T @ PB
        G = PB.T @ A[t] + Nc.T
        q = P @ c[t] + p
        g = B[t].T @ q
        free = act[t] == 0
        ubar = np.where(act[t] < 0, lo[t], hi[t])
        K, k = np.zeros((m, A.shape[1])), np.where(free, 0.0, ubar)
        if free.any():
            Hff = H[np.ix_(free, free)]
            rhs_k = g[free] + H[np.ix_(free, ~free)] @ ubar[~free]
            K[free] = -np.linalg.solve(Hff, G[free])
            k[free] = -np.linalg.solve(Hff, rhs_k)
        Pn = Qs + A[t].T @ P @ A[t] + K.T @ H @ K + K.T @ G + G.T @ K
        W["P"][t] = 0.5 * (Pn + Pn.T)
        W["p"][t] = -Qs @ sd[t] + A[t].T @ q + K.T @ (H @ k + g) + G.T @ k
        W["K"][t], W["k"][t], W["H"][t], W["G"][t], W["g"][t] = K, k, H, G, g


def ctrlbox_solve(prob, s_start, t0, lo, hi, u, act, W, valid_from, pdas_iter=10, max_iter=2000, tol=1e-10,
                  single_iter=50):
    """Active-set solve of the tail t0..T-1 from s_start.  `act` (T,m) in {-1 at lo, 0 free, +1 at hi}
    is the warm start (updated in place), W the backward-pass cache valid for t >= valid_from
    (T = nothing valid), `u` (T,m) receives the solution.
      phase 1: primal-dual active-set iterations (Hintermueller-Ito-Kunisch): all violated bounds
               are pinned and all wrong-signed multipliers released at once; converges in a handful
               of iterations when it converges, but may cycle;
      phase 1b (iterations pdas_iter+1 .. pdas_iter+single_iter, round 3; csrc/ctrlbox_mfma.hip kPdasSingleM): the
               same with a damped release rule -- all violated bounds pinned, only the ONE pinned component with
               the worst multiplier released -- which ends the cycles of rate-limited (bang-bang) tails;
      phase 2 (after that): classic primal active-set from the clipped iterate --
               one constraint added (blocking step) or dropped (worst multiplier) per iteration;
               finite and monotone for a strictly convex QP.
    Every iteration is a partial backward sweep (from the latest changed time step) plus vector
    sweeps.  Returns s, u, mu (half-gradient wrt u), (pdas iterations, primal iterations, backward
    time steps swept), valid_from."""
    T, m = prob["B"].shape[0], prob["m"]
    A, B, c = prob["A"], prob["B"], prob["c"]
    s = np.zeros((T + 1, A.shape[1]))
    mu = np.zeros((T, m))
    swept = 0

    def backward(t_hi):
        nonlocal swept
        if t_hi >= t0:
            ctrlbox_backward(prob, act, lo, hi, t_hi, t0, W)
            swept += t_hi - t0 + 1

    def policy_rollout(us, clip=False):
        # clip (csrc/ctrlbox_mfma.hip, round 3; trust-region problems, phase 1): the rollout applies the controls
        # CLIPPED to their boxes, so that the violations a round reports are those of a trajectory that stays in the
        # box upstream.  Only an update rule: the accepted round clips nothing.
        s[t0] = s_start
        for t in range(t0, T):
            us[t] = W["K"][t] @ s[t] + W["k"][t]
            mu[t] = W["H"][t] @ us[t] + W["G"][t] @ s[t] + W["g"][t]
            s[t + 1] = A[t] @ s[t] + B[t] @ (np.clip(us[t], lo[t], hi[t]) if clip else us[t]) + c[t]

    t_dirty = T - 1 if valid_from >= T else (valid_from - 1 if valid_from > t0 else t0 - 1)
    # ---- phase 1: primal-dual active set
    for it in range(1, pdas_iter + single_iter + 1):
        backward(t_dirty)
        policy_rollout(u, clip=prob.get("kind") == "abs")
        a, uu, mm = act[t0:], u[t0:], mu[t0:]
        new = a.copy()
        new[(a == 0) & (uu < lo[t0:] - tol)] = -1
        new[(a == 0) & (uu > hi[t0:] + tol)] = 1
        rel = ((a < 0) & (mm < -tol)) | ((a > 0) & (mm > tol))
        if it > pdas_iter and rel.any():
            worst = np.unravel_index(np.argmax(np.where(rel, np.abs(mm), 0.0)), rel.shape)
            rel = np.zeros_like(rel)
            rel[worst] = True
        new[rel] = 0
        changed = np.nonzero((new != a).any(axis=1))[0]
        if changed.size == 0:
            return s, u, mu, (it, 0, swept), t0
        act[t0:] = new
        t_dirty = t0 + int(changed.max())
    # ---- phase 2: primal active set from the clipped iterate
    u[t0:] = np.clip(u[t0:], lo[t0:], hi[t0:])
    new = np.where(u[t0:] <= lo[t0:], -1, np.where(u[t0:] >= hi[t0:], 1, 0))
    changed = np.nonzero((new != act[t0:]).any(axis=1))[0]
    act[t0:] = new
    t_dirty = max(t_dirty, t0 + int(changed.max())) if changed.size else t_dirty
    us = np.zeros_like(u)
    for it2 in range(1, max_iter + 1):
        backward(t_dirty)
        t_dirty = t0 - 1
        policy_rollout(us)
        d = us[t0:] - u[t0:]
        free = act[t0:] == 0
        # largest feasible step along d
        with np.errstate(divide="ignore", invalid="ignore"):
            room = np.where(d > 0, (hi[t0:] - u[t0:]) / d, np.where(d < 0, (lo[t0:] - u[t0:]) / d, np.inf))
        room = np.where(free, room, np.inf)
        alpha = min(1.0, room.min())
        if alpha < 1.0:
            r, j = np.unravel_index(np.argmin(room), room.shape)
            u[t0:] += alpha * d
            act[t0 + r, j] = 1 if d[r, j] > 0 else -1
            u[t0 + r, j] = hi[t0 + r, j] if d[r, j] > 0 else lo[t0 + r, j]
            t_dirty = t0 + r
            continue
        u[t0:] = us[t0:]
        viol = np.where(act[t0:] < 0, -mu[t0:], np.where(act[t0:] > 0, mu[t0:], 0.0))   # > 0 = wrong sign
        if viol.max() <= tol:
            return s, u, mu, (pdas_iter + single_iter, it2, swept), t0
        r, j = np.unravel_index(np.argmax(viol), viol.shape)
        act[t0 + r, j] = 0
        t_dirty = t0 + r
    return s, u, mu, (pdas_iter + single_iter, -max_iter, swept), t0


def ctrlbox_workspace(prob):
    T, Ns, m = prob["B"].shape[0], prob["A"].shape[1], prob["m"]
    W = dict(P=np.zeros((T + 1, Ns, Ns)), p=np.zeros((T + 1, Ns)), K=np.zeros((T, m, Ns)), k=np.zeros((T, m)),
             H=np.zeros((T, m, m)), G=np.zeros((T, m, Ns)), g=np.zeros((T, m)))
    W["P"][T] = prob["Qsd"]
    W["p"][T] = -prob["Qsd"] @ prob["sd"][T]
    return W


def ctrlbox_saturated_start(prob, s_start, lo, hi, act, W, tol=1e-10):
    """Cold start of the first tail (csrc/ctrlbox_mfma.hip, saturated_start): sweep with every component free
    (the plain Riccati pass), then roll that policy out on the linear model with the controls CLIPPED to their
    boxes and pin what the rollout saturates.  Returns the number of time steps swept.  The starting set does
    not change the QP's solution, only how many sweeps the active-set iterations need to reach it."""
    T = prob["B"].shape[0]
    act[:] = 0
    ctrlbox_backward(prob, act, lo, hi, T - 1, 0, W)
    s = np.array(s_start, dtype=float)
    for t in range(T):
        v = W["K"][t] @ s + W["k"][t]
        act[t] = np.where(v < lo[t] - tol, -1, np.where(v > hi[t] + tol, 1, 0))
        s = prob["A"][t] @ s + prob["B"][t] @ np.clip(v, lo[t], hi[t]) + prob["c"][t]
    return T


def local_descent_quasistatic_as(system, At, Bt, ct, Q, Qd, R, x0, xd_trj, lo, hi, kind, pdas_iter=10,
                                 max_iter=2000, tol=1e-10, act_io=None, sat_start=False):
    """irs_lqr_quasistatic.py:326-345 with ONE control box (kind "abs": lo,hi (T,m) absolute rows
    on u_t; kind "rel": rows on u_t - u_{t-1}): T tail QPs solved EXACTLY by the active-set method,
    the active set and the backward pass carried from tail to tail.  `act_io` (T,m) in {-1,0,+1}, in/out
    (include/irs_hip.h, irs_quasistatic_box_descent_ws): the set the first tail starts from (a previous
    iteration's) and, on return, the set it converged to.  `sat_start`: without a warm start, the first tail
    starts from the saturated unconstrained policy (ctrlbox_saturated_start; the matrix-core kernel does)."""
    T, n, m = Bt.shape
    idx = system.indices_u_into_x
    prob = quasistatic_ctrl_problem(At, Bt, ct, Q, Qd, R, xd_trj, kind)
    W = ctrlbox_workspace(prob)
    act = np.zeros((T, m), dtype=int)
    if act_io is not None:
        a0 = np.sign(np.asarray(act_io)).astype(int)
        act[:] = np.where((a0 < 0) & np.isfinite(lo), -1, np.where((a0 > 0) & np.isfinite(hi), 1, 0))
    x_new, u_new = np.zeros((T + 1, n)), np.zeros((T, m))
    x_new[0] = x0
    stats, valid_from = [], T
    u = np.zeros((T, m))
    for t in range(T):
        s_t = np.concatenate([x_new[t], x_new[t][idx]])
        extra = 0
        if t == 0 and sat_start and not act.any():
            extra = ctrlbox_saturated_start(prob, s_t, lo, hi, act, W, tol)
        s, u, mu, st, valid_from = ctrlbox_solve(prob, s_t, t, lo, hi, u, act, W, valid_from, pdas_iter,
                                                 max_iter, tol)
        st = (st[0], st[1], st[2] + extra)
        stats.append(st)
        if t == 0 and act_io is not None:
            act_io[:] = act
        ctl = np.clip(u[t], lo[t], hi[t])
        u_new[t] = ctl if kind == "abs" else s_t[n:] + ctl
        x_new[t + 1] = system.dynamics(x_new[t], u_new[t])
    return x_new, u_new, stats


def cem_quasistatic_local_descent(system, x0, u_trj, std_trj, xd_trj, Q, Qd, R, n_elite, batch_size):
    """irs_lqr/cem_quasistatic.py:168-211: candidates from np.random.normal (global RNG, like the
    reference), each priced by the quasistatic eval_cost (:124-165), elites by argpartition, refit."""
    T, m = u_trj.shape
    idx = system.indices_u_into_x
    cand = np.random.normal(u_trj, std_trj, (batch_size, T, m))
    costs = np.zeros(batch_size)
    for k in range(batch_size):
        costs[k] = eval_cost_quasistatic(rollout(system, x0, cand[k]), cand[k], xd_trj, Q, Qd, R, idx)
    best = np.argpartition(costs, n_elite)[:n_elite]
    u_new = np.mean(cand[best], axis=0)
    std_new = np.std(cand[best], axis=0)
    return rollout(system, x0, u_new), u_new, std_new, costs, cand
