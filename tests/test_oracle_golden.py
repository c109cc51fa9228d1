"""Pins oracle/irs_oracle.py against (a) the reference's own result files and
(b) fixtures produced by running the reference source (tests/golden/make_fixtures.py)."""
import os

import numpy as np
import pytest

from oracle import irs_oracle as orc


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def pend_problem(T):
    Q, Qd, R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
    x0 = np.array([0., 0.])
    xd = np.tile(np.array([np.pi, 0.]), (T + 1, 1))
    u0 = np.tile(np.array([0.1]), (T, 1))
    return Q, Qd, R, x0, xd, u0


def quad_problem(T):
    Q = np.diag([10., 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0])
    Qd = 10.0 * np.diag([10., 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1])
    R = np.eye(4)
    x0 = np.zeros(12)
    xd = np.zeros((T + 1, 12))
    for i in range(T + 1):
        xd[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
    u0 = np.tile(np.array([2.0, 2.0, 2.0, 2.0]), (T, 1))
    return Q, Qd, R, x0, xd, u0


# ---- (b) fixtures from the reference's own source -------------------------
def test_pendulum_dynamics_fixture(golden_dir):
    f = load(golden_dir, "pendulum_dynamics")
    s = orc.PendulumOracle(float(f["h"]))
    assert np.array_equal(s.dynamics_batch(f["X"], f["U"]), f["Xn"])
    for i in range(f["X"].shape[0]):
        assert np.array_equal(s.dynamics(f["X"][i], f["U"][i]), f["Xn_scalar"][i])


def test_quadrotor_dynamics_fixture(golden_dir):
    f = load(golden_dir, "quadrotor_dynamics")
    s = orc.QuadrotorOracle(float(f["h"]))
    np.testing.assert_allclose(s.dynamics_batch(f["X"], f["U"]), f["Xn"], rtol=0, atol=1e-13)


def test_pendulum_rollout_cost_fixture(golden_dir):
    f = load(golden_dir, "pendulum_T200_init")
    Q, Qd, R, x0, xd, u0 = pend_problem(200)
    s = orc.PendulumOracle(0.05)
    x = orc.rollout(s, x0, u0)
    assert np.array_equal(x, f["x_trj"])
    assert orc.evaluate_cost(x, u0, xd, Q, R) == float(f["cost0"])


def test_quadrotor_rollout_cost_fixture(golden_dir):
    f = load(golden_dir, "quadrotor_T200_init")
    Q, Qd, R, x0, xd, u0 = quad_problem(200)
    s = orc.QuadrotorOracle(0.05)
    x = orc.rollout(s, x0, u0)
    np.testing.assert_allclose(x, f["x_trj"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(orc.evaluate_cost(x, u0, xd, Q, R), float(f["cost0"]), rtol=1e-13)


@pytest.mark.parametrize("name,sysname", [("pendulum_zero_T30_N100", "pendulum"),
                                          ("quadrotor_zero_T6_N64", "quadrotor")])
def test_zero_order_fixture(golden_dir, name, sysname):
    f = load(golden_dir, name)
    s = orc.SYSTEMS[sysname](float(f["h"]))
    At, Bt, ct = orc.zero_order_TV(s, f["x_trj"], f["u_trj"], f["dx"], f["du"])
    np.testing.assert_allclose(At, f["At"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(Bt, f["Bt"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(ct, f["ct"], rtol=0, atol=1e-11)


def test_zero_order_fixture_sampling_replay(golden_dir):
    """np.random.seed + the script's sampling closure reproduces the recorded samples."""
    f = load(golden_dir, "pendulum_zero_T30_N100")
    np.random.seed(int(f["seed"]))
    dx, du = orc.gaussian_samples(30, 100, [1.0, 1.0], [1.0], 1)
    assert np.array_equal(dx, f["dx"]) and np.array_equal(du, f["du"])


def test_first_order_fixture(golden_dir):
    f = load(golden_dir, "pendulum_first_T30_N100")
    s = orc.PendulumOracle(float(f["h"]))
    At, Bt, ct = orc.first_order_TV(s, f["x_trj"], f["u_trj"], f["dx"], f["du"])
    np.testing.assert_allclose(At, f["At"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(Bt, f["Bt"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(ct, f["ct"], rtol=0, atol=1e-13)


def test_cem_fixture(golden_dir):
    f = load(golden_dir, "pendulum_cem_T30_B50")
    Q, Qd, R, x0, xd, _ = pend_problem(30)
    s = orc.PendulumOracle(float(f["h"]))
    x_new, u_new, std_new, _ = orc.cem_local_descent(
        s, x0, f["u_trj"], f["std0"], xd, Q, R, int(f["n_elite"]), f["cand"])
    np.testing.assert_allclose(u_new, f["u_new"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(std_new, f["std_new"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(x_new, f["x_new"], rtol=0, atol=1e-13)


def test_bicycle_fixtures(golden_dir):
    f = load(golden_dir, "bicycle_dynamics")
    s = orc.BicycleOracle(float(f["h"]))
    assert np.array_equal(s.dynamics_batch(f["X"], f["U"]), f["Xn"])
    for i in range(f["X"].shape[0]):
        assert np.array_equal(s.dynamics(f["X"][i], f["U"][i]), f["Xn_scalar"][i])
    # examples/bicycle/bicycle_zero_order.py:11-31 initial trajectory cost
    # (== first line of examples/bicycle/analysis/bicycle_easy_*.csv: 3302.0894)
    g = load(golden_dir, "bicycle_T100_init")
    Q, R = np.diag([5, 5, 3, 0.1, 0.1]), np.diag([1, 0.1])
    xd = np.tile(np.array([3.0, 1.0, np.pi / 2, 0, 0]), (101, 1))
    u0 = np.tile(np.array([0.1, 0.0]), (100, 1))
    x = orc.rollout(s, np.zeros(5), u0)
    np.testing.assert_allclose(x, g["x_trj"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(orc.evaluate_cost(x, u0, xd, Q, R), float(g["cost0"]), rtol=1e-14)
    assert abs(float(g["cost0"]) - 3302.0894) < 1e-3
    z = load(golden_dir, "bicycle_zero_T8_N200")
    At, Bt, ct = orc.zero_order_TV(s, z["x_trj"], z["u_trj"], z["dx"], z["du"])
    np.testing.assert_allclose(At, z["At"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(Bt, z["Bt"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(ct, z["ct"], rtol=0, atol=1e-10)


def test_three_cart_fixture(golden_dir):
    """All four contact branches of three_cart_dynamics.py:44-104."""
    f = load(golden_dir, "three_cart_dynamics")
    s = orc.ThreeCartOracle(float(f["h"]))
    for i in range(f["X"].shape[0]):
        assert np.array_equal(s.dynamics(f["X"][i], f["U"][i]), f["Xn_scalar"][i])


# ---- box-constrained TV-LQR (tv_lqr.py:112-123 with active bounds) -------------------
def bike_problem(T):
    Q, Qd, R = np.diag([5, 5, 3, 0.1, 0.1]), np.diag([50., 50, 30, 1, 1]), np.diag([1, 0.1])
    x0 = np.zeros(5)
    xd = np.tile(np.array([3.0, 1.0, np.pi / 2, 0, 0]), (T + 1, 1))
    u0 = np.tile(np.array([0.1, 0.0]), (T, 1))
    return Q, Qd, R, x0, xd, u0


def test_box_qp_solution_satisfies_kkt():
    """The ADMM/Riccati QP solution is certified by the QP's own KKT conditions
    (multipliers recovered by least squares), with state AND input bounds active."""
    T = 25
    s = orc.BicycleOracle(0.1)
    Q, Qd, R, x0, xd, u0 = bike_problem(T)
    xlo = np.array([-np.inf] * 4 + [-0.3])
    ulo = np.array([-2.0, -np.inf])
    x = orc.rollout(s, x0, u0)
    At, Bt, ct = orc.exact_TV(s, x, u0)
    F = orc.tvlqr_box_factor(At, Bt, ct, Q, Qd, R, xlo, -xlo, ulo, -ulo, 10.0)
    zx, zu, _, it = orc.tvlqr_box_solve(F, At, Bt, ct, Q, Qd, xd, x0, 0, xlo, -xlo, ulo, -ulo, None, 20000, 1e-10)
    assert it < 20000
    assert (np.abs(zx[:, 4]) > 0.3 - 1e-6).sum() > 5 and (np.abs(zu[:, 0]) > 2 - 1e-6).sum() > 2
    r_dyn, r_box, r_stat, sign_bad = orc.qp_box_kkt_residuals(At, Bt, ct, Q, Qd, R, x0, xd, xlo, -xlo, ulo, -ulo, zx, zu)
    assert r_dyn < 1e-10 and r_box < 1e-8 and r_stat < 1e-7 and sign_bad < 1e-7
    # with every bound infinite the same code returns the equality-constrained QP solution
    inf5, inf2 = np.full(5, np.inf), np.full(2, np.inf)
    F2 = orc.tvlqr_box_factor(At, Bt, ct, Q, Qd, R, -inf5, inf5, -inf2, inf2, 10.0)
    zx2, zu2, _, it2 = orc.tvlqr_box_solve(F2, At, Bt, ct, Q, Qd, xd, x0, 0, -inf5, inf5, -inf2, inf2)
    xq, uq = orc.solve_tvlqr_qp(At, Bt, ct, Q, Qd, R, x0, xd)
    assert it2 == 1
    np.testing.assert_allclose(zu2, uq, rtol=1e-9, atol=1e-10)


def test_bicycle_exact_csv_first_descent(golden_dir):
    """examples/bicycle/bicycle_exact.py (steer bound +-pi/4 ACTIVE) vs the first entries of
    examples/bicycle/analysis/bicycle_easy_exact.csv.  The reference's curve carries OSQP's
    default 1e-3 accuracy on each of the 100 tail QPs of a descent, so it is matched to ~1 %
    on the first descent only (the iLQR map amplifies solver noise afterwards)."""
    gold = np.loadtxt(os.path.join(golden_dir, "bicycle_easy_exact.csv"))
    T = 100
    s = orc.BicycleOracle(0.1)
    Q, Qd, R, x0, xd, u0 = bike_problem(T)
    xlo = np.array([-np.inf] * 4 + [-np.pi / 4])
    inf2 = np.full(2, np.inf)
    x = orc.rollout(s, x0, u0)
    assert orc.evaluate_cost(x, u0, xd, Q, R) == pytest.approx(gold[0], rel=1e-13)
    At, Bt, ct = orc.exact_TV(s, x, u0)
    xn, un, iters = orc.local_descent_box(s, At, Bt, ct, Q, Qd, R, x0, xd, xlo, -xlo, -inf2, inf2, rho=10.0,
                                          max_iter=5000, eps=1e-8)
    assert max(iters) < 5000
    c1 = orc.evaluate_cost(xn, un, xd, Q, R)
    assert abs(c1 - gold[1]) / gold[1] < 0.012
    assert np.abs(xn[:, 4]).max() < np.pi / 4 + 1e-3          # the bound shapes the plan
    # without the bound the Riccati descent leaves the feasible set by a wide margin
    xr, ur, _, _ = orc.local_descent(s, At, Bt, ct, Q, Qd, R, x0, xd)
    assert np.abs(xr[:, 4]).max() > np.pi / 4 + 0.1


# ---- Jacobians: exact derivative vs central differences of the pinned dynamics
@pytest.mark.parametrize("sysname", ["pendulum", "quadrotor"])
def test_jacobian_vs_finite_difference(sysname):
    s = orc.SYSTEMS[sysname](0.05)
    rng = np.random.default_rng(5)
    n, m = s.dim_x, s.dim_u
    for _ in range(5):
        x = rng.normal(size=n) * 0.4
        u = 2.0 + rng.normal(size=m) * 0.4
        J = s.jacobian_xu(x, u)
        xu = np.hstack((x, u))
        Jfd = np.zeros((n, n + m))
        for j in range(n + m):
            e = np.zeros(n + m)
            e[j] = 1e-6
            Jfd[:, j] = (s.dynamics((xu + e)[:n], (xu + e)[n:]) -
                         s.dynamics((xu - e)[:n], (xu - e)[n:])) / 2e-6
        np.testing.assert_allclose(J, Jfd, rtol=0, atol=2e-8)


# ---- Riccati form == literal QP re-solves (irs_lqr.py:169-184) -------------
@pytest.mark.parametrize("sysname,T", [("pendulum", 12), ("quadrotor", 6)])
def test_riccati_equals_qp_resolves(sysname, T):
    s = orc.SYSTEMS[sysname](0.05)
    Q, Qd, R, x0, xd, u0 = (pend_problem if sysname == "pendulum" else quad_problem)(T)
    x = orc.rollout(s, x0, u0)
    At, Bt, ct = orc.exact_TV(s, x, u0)
    xq, uq = orc.local_descent_qp(s, At, Bt, ct, Q, Qd, R, x0, xd)
    xr, ur, K, k = orc.local_descent(s, At, Bt, ct, Q, Qd, R, x0, xd)
    np.testing.assert_allclose(ur, uq, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(xr, xq, rtol=1e-7, atol=1e-8)
    # open-loop QP solution from t=0 == linear-model rollout of the policy
    xs, us = orc.solve_tvlqr_qp(At, Bt, ct, Q, Qd, R, x0, xd)
    xl = x0.copy()
    for t in range(T):
        ul = K[t].dot(xl) + k[t]
        np.testing.assert_allclose(ul, us[t], rtol=1e-6, atol=1e-7)
        xl = At[t].dot(xl) + Bt[t].dot(ul) + ct[t]


# ---- (a) the reference's own result files ---------------------------------
def test_pendulum_exact_csv(golden_dir):
    """examples/pendulum/pendulum_exact.py (T=200, iterate(10) logs costs) vs
    examples/pendulum/analysis/pendulum_exact.csv."""
    gold = np.loadtxt(os.path.join(golden_dir, "pendulum_exact.csv"))
    Q, Qd, R, x0, xd, u0 = pend_problem(200)
    s = orc.PendulumOracle(0.05)
    tv = lambda x, u, it: orc.exact_TV(s, x, u)
    *_, cost_lst, _, _ = orc.iterate(s, Q, Qd, R, x0, xd, u0, len(gold) - 2, tv)
    np.testing.assert_allclose(cost_lst, gold, rtol=2e-9)


def test_quadrotor_exact_csv(golden_dir):
    """examples/quadrotor/quadrotor_exact.py vs analysis/quadrotor_exact.csv.
    Entries 0-4: the rpy box (quadrotor_first_order.py:29-34) is inactive there."""
    gold = np.loadtxt(os.path.join(golden_dir, "quadrotor_exact.csv"))[:5]
    Q, Qd, R, x0, xd, u0 = quad_problem(200)
    s = orc.QuadrotorOracle(0.05)
    tv = lambda x, u, it: orc.exact_TV(s, x, u)
    *_, cost_lst, x_lst, _ = orc.iterate(s, Q, Qd, R, x0, xd, u0, 3, tv)
    np.testing.assert_allclose(cost_lst, gold, rtol=1e-7)
    assert max(np.abs(x[:, 4]).max() for x in x_lst) < np.pi / 2


def test_alpha_R_one_does_not_match(golden_dir):
    """Guards the Drake 1/2-R convention (tv_lqr.py:110): full-weight R misses."""
    gold = np.loadtxt(os.path.join(golden_dir, "pendulum_exact.csv"))
    Q, Qd, R, x0, xd, u0 = pend_problem(200)
    s = orc.PendulumOracle(0.05)
    x = orc.rollout(s, x0, u0)
    At, Bt, ct = orc.exact_TV(s, x, u0)
    K, k = orc.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd, alpha_R=1.0)
    xn, un = orc.closed_loop_rollout(s, K, k, x0)
    assert abs(orc.evaluate_cost(xn, un, xd, Q, R) - gold[1]) > 10.0


def test_stochastic_band(golden_dir):
    """Zero-order N=1000 iteration-1 cost lands in the band of the reference's
    own two unseeded runs (943.04 / 944.37), SURVEY 4."""
    gold = np.loadtxt(os.path.join(golden_dir, "pendulum_zero_order.csv"))
    Q, Qd, R, x0, xd, u0 = pend_problem(200)
    s = orc.PendulumOracle(0.05)
    np.random.seed(11)
    x = orc.rollout(s, x0, u0)
    dx, du = orc.gaussian_samples(200, 1000, [1., 1.], [1.], 1)
    At, Bt, ct = orc.zero_order_TV(s, x, u0, dx, du)
    xn, un, _, _ = orc.local_descent(s, At, Bt, ct, Q, Qd, R, x0, xd)
    c1 = orc.evaluate_cost(xn, un, xd, Q, R)
    assert abs(c1 - gold[1]) / gold[1] < 0.01


# ---- device RNG spec ------------------------------------------------------
def test_philox_known_answer():
    """Random123 kat_vectors: philox4x32-10."""
    z = orc.philox4x32_10(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(v) for v in z] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    f = np.full((1, 4), 0xFFFFFFFF, np.uint32)
    z = orc.philox4x32_10(f, (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(v) for v in z] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    c = np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32)
    z = orc.philox4x32_10(c, (0xa4093822, 0x299f31d0))[0]
    assert [hex(v) for v in z] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_device_gaussian_moments():
    dx, du = orc.device_gaussian_samples(2, 20000, 2, 1, [1.0, 0.5], [2.0], seed=7, it=1)
    z = np.concatenate([dx, du], axis=2).reshape(-1, 3)
    assert np.abs(z.mean(0)).max() < 0.02
    np.testing.assert_allclose(z.std(0), [1.0, 0.5, 2.0], rtol=0.02)
    assert np.abs(np.corrcoef(z.T) - np.eye(3)).max() < 0.02


# ---------------------------------------------------------------- planar quasi-dynamic contact (unpinned)
def _hand_x0():
    # examples/planar_hand/run_planar_hand.py:31-44
    return orc.PlanarHandOracle.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])


def test_planar_hand_free_fall_and_servo():
    """No contact: the disc moves by -g h^2 (quasi-dynamic, zero initial velocity) and the joints
    reach their commands exactly (stiffness-only actuated rows)."""
    o = orc.PlanarHandOracle(0.1)
    x = orc.PlanarHandOracle.pack([0.0, 2.0, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])   # far above
    idx, obj = o.indices_u_into_x, o.PERM[:3]
    u = x[idx] + np.array([0.05, -0.02, 0.03, 0.01])
    xn = o.dynamics(x, u)
    np.testing.assert_allclose(xn[obj], [0.0, 2.0 - 10.0 * 0.1 ** 2, 0.0], atol=1e-14)
    np.testing.assert_allclose(xn[idx], u, atol=1e-14)


def test_planar_hand_pgs_converges_to_exact_qp_and_stays_feasible():
    rng = np.random.default_rng(3)
    o = orc.PlanarHandOracle(0.1, pgs_iters=2000)
    x = _hand_x0()
    u = x[o.indices_u_into_x].copy()
    for _ in range(5):
        x = o.dynamics(x, u)                    # settle into the cradle
    for _ in range(10):
        xs = x + 0.02 * rng.normal(size=7)
        us = u + 0.1 * rng.normal(size=4)
        xn = o.dynamics(xs, us)
        np.testing.assert_allclose(xn, o.dynamics_exact(xs, us), atol=5e-7)
        # linearised non-penetration holds at the optimum
        Dinv, b, J, phi = o._qp(xs, us)
        assert np.all(phi[0] + J[0].dot((xn - xs)[o.PERM]) > -1e-8)      # J is in the internal order


def test_planar_hand_exact_dual_solver_is_the_qp_optimum():
    """The exact dual active-set solve (`pgs_iters = 0`; the device's *_EXACT contact models) on the
    benchmark's hardest samples -- the settled grasp with u-noise of std 0.3 and 0.05, where 50 projected
    sweeps are off by up to 5e-2: every sample satisfies the QP's KKT conditions to 1e-9 (primal feasible
    linearised gaps, multipliers >= 0, complementarity, stationarity by construction), its dual objective is
    never above a long run of sweeps', and it matches 3000 over-relaxed sweeps to 1e-7."""
    o = orc.PlanarHandOracle(0.1, pgs_iters=0)
    ref = orc.PlanarHandOracle(0.1, pgs_iters=3000)
    x0 = _hand_x0()
    T = 30
    u = np.tile(x0[o.indices_u_into_x], (T, 1))
    xt = orc.rollout(o, x0, u)
    rng = np.random.default_rng(8)
    for t, std in ((0, 0.3), (25, 0.3), (25, 0.05), (29, 0.05)):
        N = 150
        X, U = np.tile(xt[t], (N, 1)), u[t] + std * rng.normal(size=(N, 4))
        Dinv, b, J, W, lam = o._pgs(X, U)
        _, _, _, phi = o._qp(X, U)
        r = phi - np.einsum("bik,k,bk->bi", J, Dinv, b)
        g = r + np.einsum("bij,bj->bi", W, lam)
        scale = np.abs(r).max(1, keepdims=True)
        assert (lam >= 0).all() and (g >= -1e-9 * scale).all()
        assert np.abs(g * lam).max() < 1e-9 * (scale ** 2).max() * 1e3
        _, _, _, _, lam_ref = ref._pgs(X, U)
        obj = lambda l: 0.5 * np.einsum("bi,bij,bj->b", l, W, l) + np.einsum("bi,bi->b", r, l)
        assert (obj(lam) <= obj(lam_ref) + 1e-12).all()
        np.testing.assert_allclose(o.dynamics_batch(X, U), ref.dynamics_batch(X, U), rtol=0, atol=1e-7)
        assert (lam > 0).sum(1).max() >= 5                         # many rows active at once
    # the pinned model: same trajectory, same Jacobians with the exact solver
    assert np.abs(orc.PlanarHandOracle(0.1, pgs_iters=50).dynamics_batch(X, U) - o.dynamics_batch(X, U)).max() > 1e-5


@pytest.mark.parametrize("name", ["planar_hand", "box_pivoting", "box_pushing"])
def test_exact_dual_solver_kkt_on_random_states(name):
    """The exact dual active-set solve on 4000 RANDOM states and commands per model -- separated, touching,
    deeply penetrating (gaps down to -0.3 / -0.6), up to 7 rows active at once: no NaN, and every answer
    satisfies the QP's KKT conditions (multipliers >= 0, linearised gaps >= 0, complementarity; stationarity
    holds by construction of the primal step) to 1e-7 relative."""
    rng = np.random.default_rng(123)
    cls = orc.SYSTEMS[name]
    ex = cls(0.1)
    ex.pgs_iters = 0
    N = 4000
    if name == "planar_hand":
        obj = np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(0.1, 0.7, N), rng.uniform(-1, 1, N)], 1)
        left = np.stack([rng.uniform(-2.2, -0.2, N), rng.uniform(-1.5, 0.5, N)], 1)
        right = np.stack([rng.uniform(0.2, 2.2, N), rng.uniform(-0.5, 1.5, N)], 1)
        q, std = np.hstack([obj, left, right]), 0.3
    else:
        pivot = name == "box_pivoting"
        box = np.stack([rng.uniform(-0.5, 0.5, N), rng.uniform(0.45, 0.9, N) if pivot else rng.uniform(-0.5, 0.5, N),
                        rng.uniform(-1, 1, N)], 1)
        hand = np.stack([rng.uniform(-1.2, 1.2, N), rng.uniform(0.05, 1.5, N) if pivot else rng.uniform(-1.2, 1.2, N)], 1)
        q, std = np.hstack([box, hand]), 0.2
    X = np.zeros((N, ex.dim_x))
    X[:, cls.PERM] = q
    U = X[:, ex.indices_u_into_x] + rng.normal(0, std, (N, ex.dim_u))
    Dinv, b, J, W, lam = ex._pgs(X, U)
    _, _, _, phi = ex._qp(X, U)
    r = phi - np.einsum("bik,k,bk->bi", J, Dinv, b)
    g = r + np.einsum("bij,bj->bi", W, lam)
    scale = np.abs(r).max(1, keepdims=True) + 1e-30
    assert np.isfinite(lam).all() and (lam >= 0).all()
    assert (g >= -1e-7 * scale).all()
    assert (np.abs(g * lam) <= 1e-7 * scale * np.abs(lam).max(1, keepdims=True).clip(1e-30)).all()
    assert phi.min() < -0.2 and (lam > 0).sum(1).max() >= (2 if name == "box_pushing" else 5)
    assert np.isfinite(ex.dynamics_batch(X, U)).all()


def test_planar_hand_symmetric_grasp_stays_symmetric():
    o = orc.PlanarHandOracle(0.1, pgs_iters=2000)
    x = _hand_x0()
    for _ in range(3):
        x = o.dynamics(x, _hand_x0()[o.indices_u_into_x])
    xo, yo, th, l1, l2, r1, r2 = x[o.PERM]
    assert abs(xo) < 1e-9 and abs(th) < 1e-9
    np.testing.assert_allclose([l1, l2], [-r1, -r2], atol=1e-9)
    assert 0.30 < yo < 0.35                     # resting in the cradle, not through it


def test_quasistatic_tail_qp_kkt_certificate():
    """The du-cost, per-time-bounded tail QP of IrsLqrQuasistatic.local_descent
    (irs_lqr_quasistatic.py:326-345 -> tv_lqr.py:30-137), solved by the box ADMM on the
    [x; u_prev] augmentation, satisfies the ORIGINAL QP's KKT conditions."""
    T, N = 6, 300
    o = orc.PlanarHandOracle(0.1)
    x0 = _hand_x0()
    for _ in range(4):
        x0 = o.dynamics(x0, _hand_x0()[o.indices_u_into_x])
    u_trj = np.tile(x0[o.indices_u_into_x], (T, 1))
    x_trj = orc.rollout(o, x0, u_trj)
    du = 0.1 * np.random.default_rng(5).normal(size=(T, N, 4))
    At, Bt, ct = orc.zero_order_B_decoupled(o, x_trj, u_trj, du)
    q = orc.PlanarHandOracle.pack([1e-3, 1e-3, 10], [1e-3, 1e-3], [1e-3, 1e-3])
    Q, Qd, R = np.diag(q), np.diag(100 * q), 5 * np.eye(4)
    xd = np.tile(x0 + orc.PlanarHandOracle.pack([0.3, -0.1, 0.5], [0, 0], [0, 0]), (T + 1, 1))
    idx, m = o.indices_u_into_x, 4
    rows = orc.quasistatic_bounds(x_trj, idx, None, np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]),
                                  np.array([-np.ones(4) * 0.03, np.ones(4) * 0.03]))
    Ab, Bb, cb, Qb, Qdb, xdb = orc.quasistatic_augment(At, Bt, ct, Q, Qd, xd)
    zlo = np.hstack([rows[0], np.vstack([np.full((1, m), -np.inf), rows[2]])])
    zhi = np.hstack([rows[1], np.vstack([np.full((1, m), np.inf), rows[3]])])
    F = orc.tvlqr_box_factor(Ab, Bb, cb, Qb, Qdb, R, zlo, zhi, rows[4], rows[5], 100.0, alpha_R=1.0)
    z0 = np.concatenate([x0, x0[idx]])
    zx, zu, _, it = orc.tvlqr_box_solve(F, Ab, Bb, cb, Qb, Qdb, xdb, z0, 0, zlo, zhi, rows[4], rows[5], None,
                                        40000, 1e-11, 1.6)
    assert it < 40000
    r_dyn, r_box, r_stat, sign_bad = orc.qp_box_kkt_residuals(Ab, Bb, cb, Qb, Qdb, R, z0, xdb, zlo, zhi,
                                                              rows[4], rows[5], zx, zu, alpha_R=1.0)
    assert r_dyn < 1e-10 and r_box < 1e-9 and r_stat < 1e-7 and sign_bad < 1e-7
    # the augmentation reproduces the reference's variables: u_t = u_prev block one step later
    u = zx[1:, 7:]
    np.testing.assert_allclose(np.diff(np.vstack([x0[idx][None], u]), axis=0), zu, atol=1e-12)
    assert np.abs(zu).max() > 0.03 - 1e-9 and np.abs(u - x_trj[:-1, idx]).max() > 0.05 - 1e-9


def test_box_pivot_oracle_physics():
    """BoxPivotOracle: the fixed-sweep PGS tracks the exact QP optimum along the reference's hand
    sweep (run_box_pivoting.py:20-43); the box neither sinks into the ground nor into the hand."""
    o = orc.BoxPivotOracle(0.1, pgs_iters=3000)
    x = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.5, 0.5])
    T = 30
    for t in range(T):
        u = np.array([-0.5 + 0.5 * (t + 1) / T, 0.5])
        xn = o.dynamics(x, u)
        # 12 dual variables over a 5-dof primal: the dual is degenerate and PGS crawls along its flat
        # directions; the primal step is what is compared
        np.testing.assert_allclose(xn, o.dynamics_exact(x, u), atol=3e-5)
        Dinv, b, J, phi = o._qp(x, u)
        assert np.all(phi[0] + J[0].dot((xn - x)[o.PERM]) > -3e-5)        # linearised non-penetration
        x = xn
    xb, yb, th, xh, yh = x[o.PERM]
    assert xb > 0.3 and yb > 0.499 and abs(yh - 0.5) < 1e-3               # pushed along, still on the ground
    # far from everything: the box falls by g h^2, the hand reaches its command
    x = orc.BoxPivotOracle.pack([0.0, 3.0, 0.3], [-2.0, 1.0])
    xn = o.dynamics(x, np.array([-1.9, 1.1]))
    np.testing.assert_allclose(xn[o.PERM], [0.0, 3.0 - 9.81 * 0.01, 0.3, -1.9, 1.1], atol=1e-12)


@pytest.mark.parametrize("kind", ["abs", "rel"])
def test_ctrlbox_active_set_equals_admm_on_random_problems(kind):
    """The active-set solver's oracle twin (ctrlbox_solve: primal-dual active set + primal safeguard on
    the control-box form) against the ADMM solution of the SAME QP posed on the [x; u_prev]
    augmentation -- two independent algorithms, random dynamics / costs / bounds."""
    rng = np.random.default_rng(7 if kind == "abs" else 8)
    worst, used_fallback = 0.0, 0
    for _ in range(8):
        n, m, T = int(rng.integers(2, 6)), int(rng.integers(1, 4)), int(rng.integers(5, 13))
        At = np.eye(n) + 0.2 * rng.normal(size=(T, n, n))
        Bt, ct = rng.normal(size=(T, n, m)), 0.1 * rng.normal(size=(T, n))
        Q = np.diag(rng.uniform(0.01, 5, n))
        Qd, R = 10 * Q, np.diag(rng.uniform(0.1, 5, m))
        xd, x0, w0 = rng.normal(size=(T + 1, n)), rng.normal(size=n), 0.1 * rng.normal(size=m)
        width, centre = rng.uniform(0.02, 0.5), 0.1 * rng.normal(size=(T, m))
        lo, hi = centre - width, centre + width
        prob = orc.quasistatic_ctrl_problem(At, Bt, ct, Q, Qd, R, xd, kind)
        W, act, u = orc.ctrlbox_workspace(prob), np.zeros((T, m), dtype=int), np.zeros((T, m))
        s0 = np.concatenate([x0, w0])
        _, u, _, st, _ = orc.ctrlbox_solve(prob, s0, 0, lo, hi, u, act, W, T)
        assert st[1] >= 0                                        # converged
        used_fallback += st[1] > 0
        Ab, Bb, cb, Qb, Qdb, xdb = orc.quasistatic_augment(At, Bt, ct, Q, Qd, xd)
        inf = np.inf
        if kind == "rel":
            zlo, zhi, vlo, vhi = np.full((T + 1, n + m), -inf), np.full((T + 1, n + m), inf), lo, hi
        else:
            zlo = np.hstack([np.full((T + 1, n), -inf), np.vstack([np.full((1, m), -inf), lo])])
            zhi = np.hstack([np.full((T + 1, n), inf), np.vstack([np.full((1, m), inf), hi])])
            vlo, vhi = np.full((T, m), -inf), np.full((T, m), inf)
        F = orc.tvlqr_box_factor(Ab, Bb, cb, Qb, Qdb, R, zlo, zhi, vlo, vhi, 10.0, alpha_R=1.0)
        zx, _, _, it = orc.tvlqr_box_solve(F, Ab, Bb, cb, Qb, Qdb, xdb, s0, 0, zlo, zhi, vlo, vhi, None, 100000,
                                           1e-11, 1.6)
        assert it < 100000
        u_abs = u if kind == "abs" else s0[n:] + np.cumsum(u, axis=0)
        worst = max(worst, np.abs(zx[1:, n:] - u_abs).max())
    assert worst < 1e-8


def test_quasistatic_descent_two_solvers_agree():
    """local_descent_quasistatic (ADMM) == local_descent_quasistatic_as (active set) on the planar hand,
    trust-region and rate-limit bounds: the whole MPC loop with the contact step in it."""
    T = 8
    o = orc.PlanarHandOracle(0.1)
    x0 = _hand_x0()
    for _ in range(4):
        x0 = o.dynamics(x0, _hand_x0()[o.indices_u_into_x])
    idx = o.indices_u_into_x
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(o, x0, u_trj)
    At, Bt, ct = orc.zero_order_B_decoupled(o, x_trj, u_trj, 0.1 * np.random.default_rng(2).normal(size=(T, 300, 4)))
    q = orc.PlanarHandOracle.pack([1e-3, 1e-3, 10], [1e-3, 1e-3], [1e-3, 1e-3])
    Q, Qd, R = np.diag(q), np.diag(100 * q), 5 * np.eye(4)
    xd = np.tile(x0 + orc.PlanarHandOracle.pack([0.3, -0.1, 0.5], [0, 0], [0, 0]), (T + 1, 1))
    for kind, ub, rb in (("abs", np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]), None),
                         ("rel", None, np.array([-np.ones(4) * 0.03, np.ones(4) * 0.03]))):
        rows = orc.quasistatic_bounds(x_trj, idx, None, ub, rb)
        lo, hi = (rows[2], rows[3]) if kind == "abs" else (rows[4], rows[5])
        xa, ua, stats = orc.local_descent_quasistatic_as(o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, kind)
        xb, ub_, iters = orc.local_descent_quasistatic(o, At, Bt, ct, Q, Qd, R, x0, xd, *rows, rho=100.0,
                                                       max_iter=40000, eps=1e-11, relax=1.6)
        assert max(iters) < 40000 and all(st[1] >= 0 for st in stats)
        np.testing.assert_allclose(ua, ub_, rtol=0, atol=1e-8)
        np.testing.assert_allclose(xa, xb, rtol=0, atol=1e-8)


def test_contact_scheme_reproduces_reference_box_on_box_closed_form():
    """The quasi-dynamic step shared by all contact oracles / device functors, on the reference's own
    1-D example: examples/box_pushing/analysis/box_on_box.py:11-20 states its result in closed form
    (m = 1, k = 100, h = 0.1, pusher at 0, box at 1).  The one contact enters twice (generator pairs);
    the over-relaxed sweeps (omega = 1.5) contract the residual by 4 per sweep on that pair."""
    m, k, h = 1.0, 100.0, 0.1
    o = orc.BoxOnBoxOracle(h, m, k, pgs_iters=50)
    w1 = m / (m + h ** 2.0 * k)                 # box_on_box.py:16
    w2 = h ** 2.0 * k / (m + h ** 2.0 * k)      # box_on_box.py:17
    for u in np.linspace(-2.0, 2.0, 81):
        got = o.dynamics(np.array([0.0, 1.0]), np.array([u]))
        want = np.array([w1 * 1.0 + w2 * u] * 2) if u > 1 else np.array([u, 1.0])   # :18 / :20
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)
        np.testing.assert_allclose(o.dynamics_exact(np.array([0.0, 1.0]), np.array([u])), want, atol=1e-9)


def _box_pushing_data(golden_dir):
    xu = np.load(os.path.join(golden_dir, "box_pushing_xu_quasistatic.npy"))
    J = np.load(os.path.join(golden_dir, "box_pushing_dxdu_quasistatic.npy"))
    return xu[:, :5], xu[:, 5:], J


def test_box_pushing_step_reproduces_simulator_trajectory(golden_dir):
    """PIN of the contact step: the 80-step push the reference recorded from the quasistatic simulator
    (examples/box_pushing/analysis/xu_quasistatic.npy; row t = [step(x_{t-1}, u_t), u_t]) is reproduced
    step by step -- free motion, contact onset (hand and box share the displacement: M/h^2 = Kp) and
    steady pushing -- to the simulator's own QP tolerance."""
    x, u, _ = _box_pushing_data(golden_dir)
    o = orc.BoxPushOracle(0.1)
    assert np.abs(x[:, [0, 1, 4]]).max() < 1e-12 and np.abs(u[:, 0]).max() == 0      # a straight push
    onset = int(np.argmax(x[:, 3] > 0.5 + 1e-6))
    assert 10 < onset < 30
    for t in range(len(x) - 1):
        np.testing.assert_allclose(o.dynamics(x[t], u[t + 1]), x[t + 1], rtol=0, atol=3e-8)
        np.testing.assert_allclose(o.dynamics_exact(x[t], u[t + 1]), x[t + 1], rtol=0, atol=3e-8)
    # the whole trajectory as an open-loop rollout from the first state
    xr = orc.rollout(o, x[0], u[1:])
    np.testing.assert_allclose(xr, x, rtol=0, atol=2e-7)


def _box_pushing_script_problem(T):
    """examples/box_pushing/run_box_pushing.py:20-131 (x = [x_h, x_b, y_h, y_b, th_b])."""
    BOX = orc.BoxPivotOracle
    q_u0, qa0 = np.array([0.0, 0.5, 0.0]), np.array([0.0, -0.2])
    x0 = BOX.pack(q_u0, qa0)
    Q = np.diag(BOX.pack([3.0, 3.0, 1.2], [0.0, 0.0]))                 # :101-103
    Qd = 0.0 * Q                                                      # :104
    R = 1e1 * np.eye(2)                                               # :105
    xd = np.tile(BOX.pack(q_u0 + np.array([0.5, 0.5, -np.pi / 4]), qa0), (T + 1, 1))     # :107-110
    return x0, np.tile(qa0, (T, 1)), Q, Qd, R, xd


def test_box_pushing_exact_csv_pins_cost_bookkeeping_and_exact_mode(golden_dir):
    """The reference's result file examples/box_pushing/analysis/box_pushing_exact.csv (run_box_pushing.py,
    gradient_mode "exact"): from its third entry on, 112.0110165024113, twenty times over.  That number is
    50 (3 0.5^2 + 3 0.5^2 + 1.2 (pi/4)^2) to 1.5e-11: the box at rest for T = 50 steps (the script in the tree says
    T = 60: it has drifted from the file), running rows weighted by Q_dict, NO terminal row (Qd = 0 Q,
    :104), zero input-rate cost.  It pins (1) IrsLqrQuasistatic.eval_cost's bookkeeping
    (irs_lqr_quasistatic.py:153-194): T running rows 0..T-1, the terminal row with Qd, du measured from
    x_0[idx]; (2) the behaviour of gradient_mode "exact" out of contact: the hand (0.1 from the box) sees a
    zero derivative of the box with respect to its command, so the bounded QP's optimum is "do not move" and
    the cost stays put -- iteration after iteration, as in the file.  Its first two entries (112.0360...,
    112.0167...) come from an initial hand motion the scripts in the tree no longer contain (10 |du|^2 =
    0.025): not reproducible."""
    gold = np.loadtxt(os.path.join(golden_dir, "box_pushing_exact.csv"))
    assert len(gold) == 22 and np.ptp(gold[3:]) == 0.0
    T = 50
    x0, u0, Q, Qd, R, xd = _box_pushing_script_problem(T)
    o = orc.BoxPushOracle(0.1)
    idx = o.indices_u_into_x
    x_trj = orc.rollout(o, x0, u0)
    assert np.abs(x_trj - x0).max() == 0.0                                   # nothing moves
    closed_form = 50 * (3 * 0.25 + 3 * 0.25 + 1.2 * (np.pi / 4) ** 2)
    c = orc.eval_cost_quasistatic(x_trj, u0, xd, Q, Qd, R, idx)
    np.testing.assert_allclose(c, closed_form, rtol=1e-15)
    # (the file sits 1.5e-11 below the closed form: the simulator's interior-point QP leaves the resting
    # box a hair off its pose)
    np.testing.assert_allclose(gold[2:], c, rtol=1e-10)
    # T = 60 (the script as it stands) or a terminal row would NOT give the file's number
    x60 = orc.rollout(o, x0, np.tile(u0[0], (60, 1)))
    assert abs(orc.eval_cost_quasistatic(x60, np.tile(u0[0], (60, 1)), np.tile(xd[0], (61, 1)), Q, Qd, R, idx) - gold[-1]) > 20
    assert abs(orc.eval_cost_quasistatic(x_trj, u0, xd, Q, Q, R, idx) - gold[-1]) > 2
    # the host twin's eval_cost (five terms, per-model dicts) gives the same number
    import types
    from irs_mpc_amd.quasistatic_base import quasistatic_eval_cost
    qd = types.SimpleNamespace(position_indices={"box": np.array([1, 3, 4]), "hand": np.array([0, 2])},
                               models_unactuated=["box"], models_actuated=["hand"],
                               get_u_indices_into_x=lambda: np.array([0, 2]))
    terms = quasistatic_eval_cost(qd, x_trj, u0, xd, {"box": np.array([3.0, 3.0, 1.2]), "hand": np.zeros(2)},
                                  {"box": np.zeros(3), "hand": np.zeros(2)}, R)
    np.testing.assert_allclose(sum(terms), c, rtol=1e-15)
    # gradient_mode "exact" + the script's rate bound: the descent keeps the hand where it is
    At, Bt, ct = orc.exact_contact_TV(o, x_trj, u0, decouple=True)
    assert np.abs(Bt[:, [1, 3, 4], :]).max() == 0.0                          # no contact: the box does not see u
    lo, hi = np.full((T, 2), -0.04), np.full((T, 2), 0.04)                   # :117-118: +-0.4 h
    xa, ua, _ = orc.local_descent_quasistatic_as(o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, "rel")
    np.testing.assert_allclose(orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx), c, rtol=1e-13)
    np.testing.assert_allclose(orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx), gold[-1], rtol=1e-10)


def test_box_pushing_active_set_jacobian_matches_simulator(golden_dir):
    """PIN of the contact step's DERIVATIVE (gradient modes "exact" / "first_order"): the analytic
    active-set Jacobian of the restated step against all 80 of the simulator's own
    [Dq_next/Dq | Dq_next/Dq_a_cmd] (dxdu_quasistatic.npy), every row and column -- free flight,
    contact onset, sticking contact with the friction-cone coupling that turns the box.  79 agree to
    5e-7; the remaining one is the onset step (hand 0.5 mm from the box), where the simulator's
    interior-point multiplier is 1e-5 rather than 0 and its Jacobian is that much off the active-set one."""
    x, u, J = _box_pushing_data(golden_dir)
    o = orc.BoxPushOracle(0.1)
    G = o.jacobian_xu_batch(x, u)
    err = np.abs(G - J).reshape(len(x), -1).max(1)
    onset = int(np.argmax(err))
    assert 0.0 < x[onset, 3] - x[onset, 2] - 0.5995 < 1e-3          # the one at the contact onset
    assert err[onset] < 2e-4
    assert np.delete(err, onset).max() < 5e-7
    assert abs(G[40][4, 5] - 1.579862) < 1e-6 and abs(G[40][1, 5] - 0.1054295) < 1e-6   # sticking: drag + turn
    np.testing.assert_allclose(o.jacobian_xu(x[40], u[40]), G[40], rtol=0, atol=0)


def test_contact_jacobian_is_the_projector_formula():
    """The masked-LDL' evaluation of the active-set derivative == its pseudo-inverse statement
    B = E_a - D^-1 J_I' (J_I D^-1 J_I')^+ J_I[:, a] on planar-hand and box-pivoting samples with several active
    rows, and == central differences of the exactly solved step with respect to u where the PGS active
    set is the QP's (u enters the QP through b only, so that derivative has no geometry term)."""
    hand = orc.PlanarHandOracle(0.1)
    box = orc.BoxPivotOracle(0.1)
    cases = [(hand, _hand_x0(), 0.1), (box, orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.6, 0.3]), 0.05)]
    rng = np.random.default_rng(5)
    for o, x0, std in cases:
        n, m = o.dim_x, o.dim_u
        u0 = x0[o.indices_u_into_x]
        N = 120
        X, U = np.tile(x0, (N, 1)), u0 + rng.normal(0, std, (N, m))
        G = o.jacobian_xu_batch(X, U)
        counts = set()
        n_fd = 0
        for i in range(N):
            Dinv, b, J, W, lam = o._pgs(X[i:i + 1], U[i:i + 1])
            J, lam, W = J[0], lam[0], W[0]
            act = lam * np.diag(W) > o.ACTIVE_TOL
            counts.add(int(act.sum()))
            JI = J[act]
            S = JI.T.dot(np.linalg.pinv((JI * Dinv).dot(JI.T), rcond=1e-9)).dot(JI) if act.any() else np.zeros((n, n))
            Bint = -(Dinv[:, None] * S)[:, o.ACT]
            Bint[o.ACT, np.arange(m)] += 1.0
            Bext = np.zeros((n, m))
            Bext[o.PERM] = Bint
            np.testing.assert_allclose(G[i][:, n:], Bext, rtol=0, atol=1e-10)
            if i < 25:
                xe = o.dynamics_exact(X[i], U[i])
                if np.abs(xe - o.dynamics(X[i], U[i])).max() > 1e-7:
                    continue                                  # PGS not converged here: another active set
                fd = np.zeros((n, m))
                for j in range(m):
                    e = np.zeros(m)
                    e[j] = 1e-6
                    fd[:, j] = (o.dynamics_exact(X[i], U[i] + e) - o.dynamics_exact(X[i], U[i] - e)) / 2e-6
                if np.abs(fd - G[i][:, n:]).max() < 2e-3:      # L-BFGS-B noise / step
                    n_fd += 1
        assert len(counts) >= 2 and max(counts) >= 3     # (the exact solve keeps an INDEPENDENT active set)
        assert n_fd >= 10


def test_box_pushing_input_jacobian_matches_simulator(golden_dir):
    """The simulator's Dq_next/Dq_a_cmd (dxdu_quasistatic.npy[:, :, 5:]) against central differences of
    the restated step: free flight (identity rows), and sticking contact, where the hand drags and turns
    the box through the friction-cone rows (0.8946 / 0.1054 / 1.58: they fix inertia/mass = 1/30).  The
    simulator's state Jacobian holds the contact geometry fixed, so only its geometry-free part (the y
    rows) is compared."""
    x, u, J = _box_pushing_data(golden_dir)
    o = orc.BoxPushOracle(0.1)

    def fd(xs, us, eps=1e-6):
        out, z = np.zeros((5, 7)), np.concatenate([xs, us])
        for j in range(7):
            e = np.zeros(7)
            e[j] = eps
            out[:, j] = (o.dynamics_exact((z + e)[:5], (z + e)[5:]) - o.dynamics_exact((z - e)[:5], (z - e)[5:])) / (2 * eps)
        return out

    for t in (5, 12, 30, 40, 60, 79):
        Jf = fd(x[t], u[t])
        np.testing.assert_allclose(Jf[:, 5:], J[t][:, 5:], rtol=0, atol=1.5e-3)       # B, all rows
        np.testing.assert_allclose(Jf[[2, 3]], J[t][[2, 3]], rtol=0, atol=1e-4)        # y rows of [A | B]
    assert abs(J[40][4, 5] - 1.5799) < 1e-3 and abs(J[40][2, 6] - 0.5) < 1e-6          # the data really are in contact


def test_saturated_start_same_solution_fewer_sweeps():
    """The cold start of the matrix-core descent kernel (saturated unconstrained policy) restated: the T
    re-solved tail QPs have the same solutions from either starting set, and on the benchmark-like problem
    (planar hand, trust region binding on about half the components) it costs no more backward time steps."""
    T = 30
    o = orc.PlanarHandOracle(0.1)
    idx = o.indices_u_into_x
    x0 = _hand_x0()
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(o, x0, u_trj)
    du = 0.3 * np.random.default_rng(3).normal(size=(T, 300, 4))
    At, Bt, ct = orc.zero_order_B_decoupled(o, x_trj, u_trj, du)
    q = orc.PlanarHandOracle.pack([1e-3, 1e-3, 10.0], [1e-3, 1e-3], [1e-3, 1e-3])
    Q, Qd, R = np.diag(q), np.diag(100 * q), 5.0 * np.eye(4)
    xd = np.tile(x0 + orc.PlanarHandOracle.pack([0.3, -0.1, 0.5], [0, 0], [0, 0]), (T + 1, 1))
    for kind, (lo, hi) in (("abs", (x_trj[:-1, idx] - 0.05, x_trj[:-1, idx] + 0.05)),
                           ("rel", (np.full((T, 4), -0.03), np.full((T, 4), 0.03)))):
        xa, ua, st_a = orc.local_descent_quasistatic_as(o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, kind)
        xb, ub, st_b = orc.local_descent_quasistatic_as(o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, kind, sat_start=True)
        np.testing.assert_allclose(ub, ua, rtol=0, atol=1e-9)
        np.testing.assert_allclose(xb, xa, rtol=0, atol=1e-9)
        if kind == "abs":
            # (until the primal-dual rounds themselves rolled out CLIPPED controls -- round 3 -- the saturated start
            # needed a third of the backward steps of the cold one here; now every round is such a rollout and the
            # two starts cost the same)
            assert sum(s[2] for s in st_b) < 1.1 * sum(s[2] for s in st_a)


# ---- the metric's own contact model, pinned by the reference's result files ---------------------------------
def _planar_hand_spin_problem(T=30, h=0.1, **kw):
    """examples/planar_hand/run_planar_hand_spin.py:21-143: T = 3 / h, disc at (0, 0.6), both arms held at
    -+(pi/2 - 0.5) for the whole horizon (ZeroOrderHold, :28-41 -> u_traj_0 :88-113), goal = the disc lowered
    0.2 and turned -pi/4 (:131-136), Q = (10, 1, 10 | 1e-3 x 4), Qd = 10 Q, R = 100 (:121-129)."""
    o = orc.PlanarHandOracle(h, **kw)
    ql, qr = np.full(2, -np.pi / 2 + 0.5), np.full(2, np.pi / 2 - 0.5)
    q_u0 = np.array([0.0, 0.6, 0.0])
    x0 = o.pack(q_u0, ql, qr)
    u0 = np.tile(np.concatenate([ql, qr]), (T, 1))
    xd = np.tile(o.pack(q_u0 + np.array([0.0, -0.2, -np.pi / 4]), ql, qr), (T + 1, 1))
    Q = np.diag(o.pack(np.array([10.0, 1.0, 10.0]), np.full(2, 1e-3), np.full(2, 1e-3)))
    return o, x0, u0, xd, Q, 10.0 * Q, 100.0 * np.eye(4)


def _spin_initial_cost(**kw):
    o, x0, u0, xd, Q, Qd, R = _planar_hand_spin_problem(**kw)
    return orc.eval_cost_quasistatic(orc.rollout(o, x0, u0), u0, xd, Q, Qd, R, o.indices_u_into_x)


def test_planar_hand_spin_initial_cost_matches_reference(golden_dir):
    """PIN of the planar hand's contact functor -- the model of the metric's config.  The reference's result
    files examples/planar_hand/analysis/planar_hand_spin_{exact,zero_order_B,first_order}.csv all open with
    249.6305470294...: the cost of the INITIAL rollout of run_planar_hand_spin.py, before any optimiser
    step -- 30 steps of the external simulator on the grasp under gravity: the disc drops from y = 0.6 into
    the two distal links and settles at 0.670 while the joints give way under their finite stiffness.  The
    restated step reproduces it to 2e-10 at the defaults (mass 1, capsule geometry of
    planar_hand_analysis.py:33-101, Kp = (50, 25), g = 10) -- the accuracy of the simulator's own QP solves
    -- and the match is not vacuous: half / double the mass misses by 1.6e-3 / 2e-4, mu = 0.3 (the grasp
    slides) by 8e-4.  Friction at or above 0.5 is not identified by it (the grasp sticks).  R is not
    exercised either (the initial commands hold still)."""
    firsts = [np.loadtxt(os.path.join(golden_dir, "planar_hand_spin_%s.csv" % k))[0]
              for k in ("exact", "zero_order_B", "first_order")]
    assert np.ptp(firsts) < 1e-11                      # three runs of the script, one initial rollout
    gold = firsts[0]
    c = _spin_initial_cost()
    np.testing.assert_allclose(c, gold, rtol=1e-8)
    assert abs(c - gold) / gold < 5e-10                # measured: 2.1e-10
    for kw in (dict(mass=0.5), dict(mass=2.0), dict(mu=0.3)):
        assert abs(_spin_initial_cost(**kw) - gold) / gold > 1e-4, kw
    # the sweep solver (opt-in) lands on the same rollout: the resting grasp is a benign QP
    np.testing.assert_allclose(_spin_initial_cost(pgs_iters=50), gold, rtol=1e-7)
    # the settled pose: symmetric, the disc resting on the distal links
    o, x0, u0, *_ = _planar_hand_spin_problem()
    xT = orc.rollout(o, x0, u0)[-1]
    assert abs(xT[0]) < 1e-12 and abs(xT[6]) < 1e-12 and abs(xT[1] + xT[2]) < 1e-12
    np.testing.assert_allclose(xT[3], 0.670326, atol=1e-6)


def test_planar_hand_spin_exact_mode_second_entry_is_not_reproducible(golden_dir):
    """Recorded, not pinned: entry 2 of planar_hand_spin_exact.csv (269.956..., gradient_mode "exact",
    u_bounds_abs = +-1.0 h) is the cost after the FIRST descent.  The grasp is statically indeterminate (8
    cone generators on 7 dofs, 4 of them loaded), so that descent runs on (i) whichever multipliers Gurobi
    returned and (ii) the simulator's derivative through them, a rank-revealing least-squares solve of the
    active-set KKT system cut at gradient_lstsq_tolerance = 1e-3 (planar_hand_setup.py:25) -- neither is in
    the tree.  With the UNTRUNCATED derivative (what the oracle and the device compute; pinned on box_pushing,
    where the KKT system is well conditioned) the first descent improves the cost to 206.2; cutting the KKT
    system's singular values (or pivoted-QR diagonal) at 1e-3 relative removes every contact row (they are
    3e-5..4e-4 of the largest) and gives 259.9 / 259.7 -- the file's direction (the cost goes UP), not its
    value.  R x {0.01..10}, trust region {0.5 h, 1 h, none} and decouple_AB on/off were tried as well:
    none gives 269.956.  This test keeps the numbers of that attempt honest."""
    gold = np.loadtxt(os.path.join(golden_dir, "planar_hand_spin_exact.csv"))
    o, x0, u0, xd, Q, Qd, R = _planar_hand_spin_problem()
    idx = o.indices_u_into_x
    x = orc.rollout(o, x0, u0)
    At, Bt, ct = orc.exact_contact_TV(o, x, u0, decouple=True)
    lo, hi = x[:-1, idx] - 0.1, x[:-1, idx] + 0.1
    xn, un, _ = orc.local_descent_quasistatic_as(o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, "abs")
    c1 = orc.eval_cost_quasistatic(xn, un, xd, Q, Qd, R, idx)
    np.testing.assert_allclose(c1, 206.224, atol=2e-3)
    assert c1 < gold[0] < gold[1]
