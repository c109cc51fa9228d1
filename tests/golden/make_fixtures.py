"""
Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN SOURCE in the build
container (where /root/reference is mounted read-only).  Never runs on the GPU
box; only its outputs (inputs + expected outputs, i.e. data) are committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

What executes from the reference (NumPy only):
  irs_lqr/irs_lqr.py            IrsLqr.__init__, rollout, evaluate_cost
  irs_lqr/irs_lqr_zero_order.py get_TV_matrices, compute_least_squares
  irs_lqr/irs_lqr_first_order.py get_TV_matrices  (Jacobians supplied, see below)
  irs_lqr/cem.py                CrossEntropyMethod.local_descent
  examples/pendulum/pendulum_dynamics.py   dynamics, dynamics_batch
  examples/quadrotor/quadrotor_dynamics.py dynamics, dynamics_batch

`pydrake` (Drake) is not installed.  The reference imports it at module scope
(irs_lqr/tv_lqr.py:2-8, pendulum_dynamics.py:2, quadrotor_dynamics.py:2-11), so
inert placeholder modules are registered for the import to resolve; none of the
functions listed above touches a pydrake symbol, and every placeholder raises if
called.  Consequently solve_tvlqr (Drake QP) and the symbolic / autodiff
Jacobians are NOT exercised here: they are pinned by the *_exact.csv cost curves
instead (tests/test_oracle_golden.py).  For the first-order fixture the
reference's averaging loop runs on Jacobians supplied by the oracle.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _inert(name):
    def f(*a, **k):
        raise RuntimeError("pydrake placeholder '%s' called: Drake is not installed" % name)
    f.__name__ = name
    return f


def install_placeholders():
    mods = {}
    for name in ["pydrake", "pydrake.all", "pydrake.symbolic", "pydrake.examples",
                 "pydrake.examples.quadrotor", "pydrake.forwarddiff"]:
        mods[name] = types.ModuleType(name)
    for sym in ["MathematicalProgram", "OsqpSolver", "SnoptSolver", "ClpSolver",
                "GurobiSolver", "eq", "Simulator", "ResetIntegratorFromFlags"]:
        setattr(mods["pydrake.all"], sym, _inert(sym))
    mods["pydrake.examples.quadrotor"].QuadrotorPlant = _inert("QuadrotorPlant")
    mods["pydrake.forwarddiff"].jacobian = _inert("jacobian")
    sys.modules.update(mods)
    # quadrotor_dynamics.py:41,94 use aliases NumPy removed in 1.24
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "object"):
        np.object = object


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "examples", "pendulum"))
    sys.path.insert(0, os.path.join(REF, "examples", "quadrotor"))
    sys.path.insert(0, REPO)

    from irs_lqr.irs_lqr import IrsLqr, IrsLqrParameters
    from irs_lqr.irs_lqr_zero_order import IrsLqrZeroOrder
    from irs_lqr.irs_lqr_first_order import IrsLqrFirstOrder
    import irs_lqr.irs_lqr as irs_mod
    irs_mod.get_solver = lambda name: None          # Drake solver handle, unused here
    from pendulum_dynamics import PendulumDynamics
    from quadrotor_dynamics import QuadrotorDynamics
    from oracle import irs_oracle as orc

    out = {}

    # ---------------- pendulum (BASELINE config 0: T=30, N=100) ----------------
    pend = PendulumDynamics.__new__(PendulumDynamics)   # __init__ needs pydrake.symbolic
    pend.h, pend.dim_x, pend.dim_u = 0.05, 2, 1

    def pend_params(T):
        p = IrsLqrParameters()
        p.Q = np.diag([1., 1.])
        p.Qd = np.diag([20., 20.])
        p.R = np.diag([1.])
        p.x0 = np.array([0., 0.])
        p.xd_trj = np.tile(np.array([np.pi, 0.]), (T + 1, 1))
        p.u_trj_initial = np.tile(np.array([0.1]), (T, 1))
        p.xbound = None
        p.ubound = None
        return p

    class Recorder:
        """The example scripts' sampling closure (pendulum_zero_order.py:38-43),
        recording what it returned so the same samples can be replayed."""
        def __init__(self, N, sx, su, n, m):
            self.N, self.sx, self.su, self.n, self.m = N, sx, su, n, m
            self.dx, self.du = [], []

        def __call__(self, xbar, ubar, it):
            dx = np.random.normal(0.0, self.sx / (it ** 0.5), size=(self.N, self.n))
            du = np.random.normal(0.0, self.su / (it ** 0.5), size=(self.N, self.m))
            self.dx.append(dx)
            self.du.append(du)
            return dx, du

    T, N = 30, 100
    rec = Recorder(N, np.array([1.0, 1.0]), np.array([1.0]), 2, 1)
    np.random.seed(0)
    sol = IrsLqrZeroOrder(pend, pend_params(T), rec)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    out["pendulum_zero_T30_N100"] = dict(
        h=0.05, x_trj=sol.x_trj, u_trj=sol.u_trj, cost0=sol.cost,
        dx=np.stack(rec.dx), du=np.stack(rec.du), At=At, Bt=Bt, ct=ct, seed=0)

    # initial cost of the T=200 script problem (== pendulum_exact.csv line 1)
    sol200 = IrsLqr(pend, pend_params(200))
    out["pendulum_T200_init"] = dict(x_trj=sol200.x_trj, u_trj=sol200.u_trj,
                                     cost0=sol200.cost)

    # dynamics / dynamics_batch vectors
    rng = np.random.default_rng(1)
    X = rng.normal(size=(64, 2)) * 2.0
    U = rng.normal(size=(64, 1)) * 2.0
    out["pendulum_dynamics"] = dict(h=0.05, X=X, U=U, Xn=pend.dynamics_batch(X, U),
                                    Xn_scalar=np.stack([pend.dynamics(X[i], U[i]) for i in range(64)]))

    # first-order averaging loop (Jacobians supplied by the oracle, see header)
    orc_p = orc.PendulumOracle(0.05)
    pend.jacobian_xu_batch = orc_p.jacobian_xu_batch
    rec = Recorder(N, np.array([1.0, 1.0]), np.array([1.0]), 2, 1)
    np.random.seed(1)
    sol = IrsLqrFirstOrder(pend, pend_params(T), rec)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    out["pendulum_first_T30_N100"] = dict(
        h=0.05, x_trj=sol.x_trj, u_trj=sol.u_trj, dx=np.stack(rec.dx),
        du=np.stack(rec.du), At=At, Bt=Bt, ct=ct, seed=1)

    # ---------------- quadrotor ----------------
    quad = QuadrotorDynamics(0.05)

    def quad_params(T):
        p = IrsLqrParameters()
        p.Q = 1.0 * np.diag([10, 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0]).astype(float)
        p.Qd = 10.0 * np.diag([10, 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1]).astype(float)
        p.R = 1.0 * np.diag([1, 1, 1, 1]).astype(float)
        p.x0 = np.zeros(12)
        p.xd_trj = np.zeros((T + 1, 12))
        for i in range(T + 1):
            p.xd_trj[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
        p.u_trj_initial = np.tile(np.array([2.0, 2.0, 2.0, 2.0]), (T, 1))
        p.xbound = None
        p.ubound = None
        return p

    rng = np.random.default_rng(2)
    X = rng.normal(size=(32, 12)) * 0.5
    U = 2.0 + rng.normal(size=(32, 4)) * 0.5
    out["quadrotor_dynamics"] = dict(h=0.05, X=X, U=U, Xn=quad.dynamics_batch(X, U))

    sol200 = IrsLqr(quad, quad_params(200))
    out["quadrotor_T200_init"] = dict(cost0=sol200.cost, x_trj=sol200.x_trj)

    T, N = 6, 64
    rec = Recorder(N, 0.1 * np.ones(12), 0.1 * np.ones(4), 12, 4)
    np.random.seed(2)
    sol = IrsLqrZeroOrder(quad, quad_params(T), rec)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    out["quadrotor_zero_T6_N64"] = dict(
        h=0.05, x_trj=sol.x_trj, u_trj=sol.u_trj, cost0=sol.cost,
        dx=np.stack(rec.dx), du=np.stack(rec.du), At=At, Bt=Bt, ct=ct, seed=2)

    # ---------------- bicycle (examples/bicycle) ----------------
    sys.path.insert(0, os.path.join(REF, "examples", "bicycle"))
    sys.path.insert(0, os.path.join(REF, "examples", "three_cart"))
    from bicycle_dynamics import BicycleDynamics
    from three_cart_dynamics import ThreeCartDynamics
    bike = BicycleDynamics.__new__(BicycleDynamics)      # __init__ needs pydrake.symbolic
    bike.h, bike.dim_x, bike.dim_u = 0.1, 5, 2
    rng = np.random.default_rng(4)
    X = rng.normal(size=(48, 5)) * np.array([2., 2., 1., 2., 0.3])
    U = rng.normal(size=(48, 2))
    out["bicycle_dynamics"] = dict(h=0.1, X=X, U=U, Xn=bike.dynamics_batch(X, U),
                                   Xn_scalar=np.stack([bike.dynamics(X[i], U[i]) for i in range(48)]))

    def bike_params(T, xd):        # bicycle_zero_order.py:16-31 ("easy"); "hard" differs in xd
        p = IrsLqrParameters()
        p.Q = np.diag([5, 5, 3, 0.1, 0.1])
        p.Qd = np.diag([50, 50, 30, 1, 1]).astype(float)
        p.R = np.diag([1, 0.1])
        p.x0 = np.zeros(5)
        p.xd_trj = np.tile(np.asarray(xd, float), (T + 1, 1))
        p.u_trj_initial = np.tile(np.array([0.1, 0.0]), (T, 1))
        p.xbound = None
        p.ubound = None
        return p

    s_b = IrsLqr(bike, bike_params(100, [3.0, 1.0, np.pi / 2, 0, 0]))
    out["bicycle_T100_init"] = dict(cost0=s_b.cost, x_trj=s_b.x_trj)
    rec = Recorder(200, np.array([2.0, 2.0, 1.0, 2.0, 0.01]), np.array([2.0, 1.0]), 5, 2)
    np.random.seed(4)
    sol = IrsLqrZeroOrder(bike, bike_params(8, [3.0, 1.0, np.pi / 2, 0, 0]), rec)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    out["bicycle_zero_T8_N200"] = dict(h=0.1, x_trj=sol.x_trj, u_trj=sol.u_trj, dx=np.stack(rec.dx),
                                       du=np.stack(rec.du), At=At, Bt=Bt, ct=ct, seed=4)

    # ---------------- three_cart: scalar dynamics, all four contact branches ----------------
    carts = ThreeCartDynamics(0.05)
    rng = np.random.default_rng(5)
    X = np.zeros((80, 6))
    X[:, 0] = rng.normal(size=80) * 0.3
    X[:, 1] = X[:, 0] + 0.2 + rng.normal(size=80) * 0.15      # gaps straddle the cart width 0.2
    X[:, 2] = X[:, 1] + 0.2 + rng.normal(size=80) * 0.15
    X[:, 3:] = rng.normal(size=(80, 3))
    U = rng.normal(size=(80, 2))
    Xn = np.stack([carts.dynamics(X[i], U[i]) for i in range(80)])
    out["three_cart_dynamics"] = dict(h=0.05, X=X, U=U, Xn_scalar=Xn)

    # ---------------- CEM (irs_lqr/cem.py) on the pendulum ----------------
    from irs_lqr.cem import CemParameters, CrossEntropyMethod
    T = 30
    cp = CemParameters()
    cp.Q, cp.Qd, cp.R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
    cp.x0 = np.array([0., 0.])
    cp.xd_trj = np.tile(np.array([np.pi, 0.]), (T + 1, 1))
    cp.u_trj_initial = np.tile(np.array([0.1]), (T, 1))
    cp.initial_std = np.array([1.0])
    cp.batch_size = 50
    cp.n_elite = 5
    cem = CrossEntropyMethod(pend, cp)
    std0 = np.array(cem.std_trj)
    np.random.seed(3)
    state = np.random.get_state()
    x_new, u_new = cem.local_descent(cem.x_trj, cem.u_trj)
    np.random.set_state(state)          # replay the draw of cem.py:159-161
    cand = np.random.normal(cem.u_trj, std0, (cp.batch_size, T, 1))
    out["pendulum_cem_T30_B50"] = dict(
        h=0.05, u_trj=cem.u_trj, std0=std0, cand=cand, n_elite=5,
        x_new=x_new, u_new=u_new, std_new=np.array(cem.std_trj), seed=3)

    for name, d in out.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in d.items()})
        print("wrote", path, os.path.getsize(path), "bytes")

    # golden result files of the reference (data): cost curves
    import shutil
    for src, dst in [("examples/pendulum/analysis/pendulum_exact.csv", "pendulum_exact.csv"),
                     ("examples/quadrotor/analysis/quadrotor_exact.csv", "quadrotor_exact.csv"),
                     ("examples/pendulum/analysis/pendulum_zero_order.csv", "pendulum_zero_order.csv"),
                     ("examples/pendulum/analysis/pendulum_first_order.csv", "pendulum_first_order.csv"),
                     ("examples/bicycle/analysis/bicycle_easy_exact.csv", "bicycle_easy_exact.csv"),
                     ("examples/bicycle/analysis/bicycle_hard_exact.csv", "bicycle_hard_exact.csv")]:
        shutil.copyfile(os.path.join(REF, src), os.path.join(HERE, dst))
        print("copied", dst)


# ---- simulator data the reference ships (quasistatic_simulator outputs; data, not source) ------------
# examples/box_pushing/analysis/{xu,dxdu}_quasistatic.npy: an 80-step push recorded from the external
# quasistatic simulator (rows [x_t, u_t] with x_t = step(x_{t-1}, u_t)) and the simulator's
# Jacobians [Dq_next/Dq | Dq_next/Dq_a_cmd] at those points.  Copied verbatim: they pin the contact step.
def copy_box_pushing_data():
    import shutil
    src = os.path.join(REF, "examples", "box_pushing", "analysis")
    for name in ("xu_quasistatic.npy", "dxdu_quasistatic.npy"):
        shutil.copyfile(os.path.join(src, name), os.path.join(HERE, "box_pushing_" + name))
        print("copied", "box_pushing_" + name)
    # the reference's own result file of run_box_pushing.py with gradient_mode "exact": 22 costs, converged to
    # 112.0110165024113 = 50 (3 0.5^2 + 3 0.5^2 + 1.2 (pi/4)^2) -- the box never moves (a result file: data)
    shutil.copyfile(os.path.join(src, "box_pushing_exact.csv"), os.path.join(HERE, "box_pushing_exact.csv"))
    print("copied box_pushing_exact.csv")


if __name__ == "__main__":
    main()
    copy_box_pushing_data()
