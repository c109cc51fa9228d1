"""GPU parity tests: the HIP path (through the C ABI) vs the reference-run golden
fixtures and vs the NumPy oracle on identical inputs.

Tolerances.  Everything computed in f64 on the device (dynamics/Jacobian plugin
calls, rollouts, Riccati, costs) is held to ~1e-10.  The sample pass evaluates the
dynamics in f32 on f32 samples: A_t, B_t, c_t are held to rtol 1e-4 / atol 2e-5 of
the f64 reference ("fp32 tolerance" of BASELINE.json), the resulting trajectories
and costs to rtol 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from oracle import irs_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the stated fp32 tolerance of the sample pass against the f64 oracle on identical samples;
# measured on the device: ~1e-6 for the analytic AND the contact models (tests/tools/contact_tolerance_probe.py)
FP32_TOL = dict(rtol=1e-4, atol=2e-5)


@pytest.fixture(params=[2, 3], ids=["lanes", "mfma"])
def as_solver(request):
    """The two device implementations of the exact active-set descent (include/irs_hip.h, `solver`):
    2 = ctrlbox.hip (lanes + LDS), 3 = ctrlbox_mfma.hip (matrix-core tiles).  Same method, same answers."""
    return request.param

TOL_AB = dict(rtol=1e-4, atol=2e-5)


@pytest.fixture(scope="module")
def amd():
    import irs_mpc_amd
    from irs_mpc_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()     # fails loudly if the HIP library is missing
    return irs_mpc_amd


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def pend_params(amd, T):
    p = amd.IrsLqrParameters()
    p.Q, p.Qd, p.R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
    p.x0 = np.array([0., 0.])
    p.xd_trj = np.tile(np.array([np.pi, 0.]), (T + 1, 1))
    p.u_trj_initial = np.tile(np.array([0.1]), (T, 1))
    p.xbound = [-np.array([1e4, 1e4]), np.array([1e4, 1e4])]
    p.ubound = np.array([-np.array([1e4]), np.array([1e4])])
    return p


def quad_params(amd, T):
    p = amd.IrsLqrParameters()
    p.Q = np.diag([10., 10, 10, 10, 10, 10, 0, 0, 0, 0, 0, 0])
    p.Qd = 10.0 * np.diag([10., 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1])
    p.R = np.eye(4)
    p.x0 = np.zeros(12)
    p.xd_trj = np.zeros((T + 1, 12))
    for i in range(T + 1):
        p.xd_trj[i, :3] = [1.5 * np.cos(0.05 * i), 1.5 * np.sin(0.05 * i), 0.02 * i]
    p.u_trj_initial = np.tile(np.array([2.0, 2.0, 2.0, 2.0]), (T, 1))
    return p


def systems(amd, name, h=0.05):
    if name == "pendulum":
        return amd.PendulumDynamics(h), orc.PendulumOracle(h)
    return amd.QuadrotorDynamics(h), orc.QuadrotorOracle(h)


class Replay:
    """sampling closure that replays recorded draws (one (N,n),(N,m) pair per call)."""

    def __init__(self, dx, du):
        self.dx, self.du, self.i = dx, du, 0

    def __call__(self, x, u, it):
        i = self.i % len(self.dx)       # a second pass replays the same draws
        self.i += 1
        return self.dx[i], self.du[i]


# ---------------------------------------------------------------- plugin surface
@pytest.mark.parametrize("name", ["pendulum", "quadrotor"])
def test_dynamics_batch_vs_reference_fixture(amd, golden_dir, name):
    f = load(golden_dir, name + "_dynamics")
    sys_d, _ = systems(amd, name, float(f["h"]))
    Xn = sys_d.dynamics_batch(f["X"], f["U"])
    np.testing.assert_allclose(Xn, f["Xn"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(sys_d.dynamics(f["X"][3], f["U"][3]), f["Xn"][3], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["pendulum", "quadrotor"])
def test_jacobian_batch_vs_oracle(amd, name):
    sys_d, sys_o = systems(amd, name)
    rng = np.random.default_rng(3)
    X = rng.normal(size=(40, sys_o.dim_x)) * 0.5
    U = 2.0 + rng.normal(size=(40, sys_o.dim_u)) * 0.5
    J = sys_d.jacobian_xu_batch(X, U)
    np.testing.assert_allclose(J, sys_o.jacobian_xu_batch(X, U), rtol=0, atol=1e-11)
    np.testing.assert_allclose(sys_d.jacobian_xu(X[0], U[0]), sys_o.jacobian_xu(X[0], U[0]), rtol=0, atol=1e-11)


def test_rollout_and_cost_vs_reference_fixture(amd, golden_dir):
    f = load(golden_dir, "pendulum_T200_init")
    sol = amd.IrsLqrExact(amd.PendulumDynamics(0.05), pend_params(amd, 200))
    np.testing.assert_allclose(sol.x_trj, f["x_trj"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(sol.cost, float(f["cost0"]), rtol=1e-13)
    f = load(golden_dir, "quadrotor_T200_init")
    sol = amd.IrsLqrExact(amd.QuadrotorDynamics(0.05), quad_params(amd, 200))
    np.testing.assert_allclose(sol.x_trj, f["x_trj"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(sol.cost, float(f["cost0"]), rtol=1e-12)
    # evaluate_cost of a pair that is NOT dynamically consistent
    rng = np.random.default_rng(0)
    x, u = rng.normal(size=(201, 12)), rng.normal(size=(200, 4))
    p = quad_params(amd, 200)
    np.testing.assert_allclose(sol.evaluate_cost(x, u), orc.evaluate_cost(x, u, p.xd_trj, p.Q, p.R), rtol=1e-12)


# ---------------------------------------------------------------- smoothing vs REFERENCE outputs
@pytest.mark.parametrize("fix,name,T", [("pendulum_zero_T30_N100", "pendulum", 30),
                                        ("quadrotor_zero_T6_N64", "quadrotor", 6)])
def test_zero_order_vs_reference_fixture(amd, golden_dir, fix, name, T):
    f = load(golden_dir, fix)
    sys_d, _ = systems(amd, name, float(f["h"]))
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    sol = amd.IrsLqrZeroOrder(sys_d, params, Replay(f["dx"], f["du"]))
    np.testing.assert_allclose(sol.x_trj, f["x_trj"], rtol=0, atol=1e-10)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    np.testing.assert_allclose(At, f["At"], **TOL_AB)
    np.testing.assert_allclose(Bt, f["Bt"], **TOL_AB)
    np.testing.assert_allclose(ct, f["ct"], **TOL_AB)


def test_zero_order_identical_seed_as_reference(amd, golden_dir):
    """np.random.seed + the script's closure: the device path consumes the very
    samples the reference run consumed (fixture recorded with seed 0)."""
    f = load(golden_dir, "pendulum_zero_T30_N100")
    N = 100

    def sampling(xbar, ubar, it):      # pendulum_zero_order.py:38-43
        dx = np.random.normal(0.0, np.array([1.0, 1.0]) / it ** 0.5, size=(N, 2))
        du = np.random.normal(0.0, np.array([1.0]) / it ** 0.5, size=(N, 1))
        return dx, du

    np.random.seed(int(f["seed"]))
    sol = amd.IrsLqrZeroOrder(amd.PendulumDynamics(0.05), pend_params(amd, 30), sampling)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    np.testing.assert_allclose(At, f["At"], **TOL_AB)
    np.testing.assert_allclose(Bt, f["Bt"], **TOL_AB)
    np.testing.assert_allclose(ct, f["ct"], **TOL_AB)


def test_first_order_vs_reference_fixture(amd, golden_dir):
    f = load(golden_dir, "pendulum_first_T30_N100")
    sol = amd.IrsLqrFirstOrder(amd.PendulumDynamics(0.05), pend_params(amd, 30), Replay(f["dx"], f["du"]))
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    np.testing.assert_allclose(At, f["At"], **TOL_AB)
    np.testing.assert_allclose(Bt, f["Bt"], **TOL_AB)
    np.testing.assert_allclose(ct, f["ct"], **TOL_AB)


# ---------------------------------------------------------------- smoothing vs oracle, more shapes
@pytest.mark.parametrize("name,T,N,std", [("pendulum", 30, 10000, 1.0), ("pendulum", 30, 100000, 0.3),
                                          ("pendulum", 7, 1, 1.0), ("pendulum", 3, 257, 1.0),
                                          ("quadrotor", 50, 300, 0.1), ("quadrotor", 4, 2000, 0.1)])
def test_zero_order_vs_oracle(amd, name, T, N, std):
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    rng = np.random.default_rng(T * 1000 + N)
    dx = (rng.normal(size=(T, N, n)) * std).astype(np.float32)
    du = (rng.normal(size=(T, N, m)) * std).astype(np.float32)
    sol = amd.IrsLqrZeroOrder(sys_d, params, Replay(dx, du))
    if N < n + m:
        with pytest.raises(ValueError, match="rank deficient"):
            sol.get_TV_matrices(sol.x_trj, sol.u_trj)
        return
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    Ao, Bo, co = orc.zero_order_TV(sys_o, sol.x_trj, sol.u_trj, dx.astype(np.float64), du.astype(np.float64))
    np.testing.assert_allclose(At, Ao, **TOL_AB)
    np.testing.assert_allclose(Bt, Bo, **TOL_AB)
    np.testing.assert_allclose(ct, co, **TOL_AB)


@pytest.mark.parametrize("name,T,N,std", [("pendulum", 30, 5000, 1.0), ("pendulum", 5, 1, 0.5),
                                          ("quadrotor", 5, 333, 0.1), ("quadrotor", 50, 64, 0.1)])
def test_first_order_vs_oracle(amd, name, T, N, std):
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    rng = np.random.default_rng(T * 1000 + N + 1)
    dx = (rng.normal(size=(T, N, n)) * std).astype(np.float32)
    du = (rng.normal(size=(T, N, m)) * std).astype(np.float32)
    sol = amd.IrsLqrFirstOrder(sys_d, params, Replay(dx, du))
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    Ao, Bo, co = orc.first_order_TV(sys_o, sol.x_trj, sol.u_trj, dx.astype(np.float64), du.astype(np.float64))
    np.testing.assert_allclose(At, Ao, **TOL_AB)
    np.testing.assert_allclose(Bt, Bo, **TOL_AB)
    np.testing.assert_allclose(ct, co, **TOL_AB)


def test_first_order_zero_std_equals_exact(amd):
    """Size-independent property at BASELINE config 2's full size (quadrotor T=50,
    N=10000): with zero perturbations the mean Jacobian is the exact linearisation."""
    T, N = 50, 10000
    sys_d, _ = systems(amd, "quadrotor")
    dx = np.zeros((T, N, 12), np.float32)
    du = np.zeros((T, N, 4), np.float32)
    sol = amd.IrsLqrFirstOrder(sys_d, quad_params(amd, T), Replay(dx, du))
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    ex = amd.IrsLqrExact(sys_d, quad_params(amd, T))
    Ae, Be, ce = ex.get_TV_matrices(ex.x_trj, ex.u_trj)
    np.testing.assert_allclose(At, Ae, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(Bt, Be, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ct, ce, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name,T,N", [("pendulum", 10, 4000), ("quadrotor", 5, 1000)])
def test_zero_order_B_vs_oracle(amd, name, T, N):
    """quasistatic_dynamics.py:242-266 estimator: u-only noise, A exact."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    x_trj = orc.rollout(sys_o, params.x0, params.u_trj_initial)
    rng = np.random.default_rng(9)
    du = (rng.normal(size=(T, N, m)) * 0.2).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(params.u_trj_initial)
    sums = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    At, Bt, ct, info = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, sums)
    assert int(info.abs().sum().item()) == 0
    for t in range(T):
        x, u = x_trj[t], params.u_trj_initial[t]
        fn = sys_o.dynamics_batch(np.tile(x, (N, 1)), u + du[t].astype(np.float64))
        B = orc.zero_order_B_fit(du[t].astype(np.float64), fn - sys_o.dynamics(x, u))
        A = sys_o.jacobian_xu(x, u)[:, :n]
        np.testing.assert_allclose(Bt[t].cpu().numpy(), B, **TOL_AB)
        np.testing.assert_allclose(At[t].cpu().numpy(), A, rtol=0, atol=1e-11)
        np.testing.assert_allclose(ct[t].cpu().numpy(), sys_o.dynamics(x, u) - A.dot(x) - B.dot(u), **TOL_AB)


def test_shard_sum_invariance_full_size(amd):
    """Multi-GPU contract at BASELINE config 3's per-GPU size: the sums of 8 logical
    shards add up to the unsharded sums (what the all-reduce relies on)."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_AB
    from irs_mpc_amd.distributed import shard_range
    T, N = 50, 100000
    sys_d, sys_o = systems(amd, "pendulum")
    params = pend_params(amd, T)
    x_trj = dev.to_dev(orc.rollout(sys_o, params.x0, params.u_trj_initial))
    u_trj = dev.to_dev(params.u_trj_initial)
    g = torch.Generator(device="cuda").manual_seed(0)
    dx = torch.randn((T, N, 2), generator=g, device="cuda", dtype=torch.float32)
    du = torch.randn((T, N, 1), generator=g, device="cuda", dtype=torch.float32)
    dm = sys_d.dm()
    full = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_AB, x_trj, u_trj, dx, du).clone()
    acc = torch.zeros_like(full)
    for r in range(8):
        lo, hi = shard_range(N, r, 8)
        acc += dm.smooth_accumulate(SMOOTH_ZERO_ORDER_AB, x_trj, u_trj, dx[:, lo:hi].contiguous(),
                                    du[:, lo:hi].contiguous())
    np.testing.assert_allclose(acc.cpu().numpy(), full.cpu().numpy(), rtol=2e-6, atol=1e-3)
    A1, B1, c1, _ = dm.smooth_finalize(SMOOTH_ZERO_ORDER_AB, N, x_trj, u_trj, full)
    A2, B2, c2, _ = dm.smooth_finalize(SMOOTH_ZERO_ORDER_AB, N, x_trj, u_trj, acc)
    np.testing.assert_allclose(A1.cpu().numpy(), A2.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # run-to-run determinism (fixed-order reductions, no atomics)
    again = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_AB, x_trj, u_trj, dx, du)
    assert torch.equal(again, full)


@pytest.mark.parametrize("name,mode_name", [("pendulum", "ZERO_ORDER_AB"), ("quadrotor", "ZERO_ORDER_AB"),
                                            ("quadrotor", "FIRST_ORDER")])
def test_fused_launch_equals_two_stage(amd, name, mode_name):
    """irs_smooth (one launch) == irs_smooth_accumulate + irs_smooth_finalize (the
    multi-GPU path), bit for bit, including repeated use of one workspace and a
    change of N in between (arrival counters re-arm)."""
    from irs_mpc_amd import _lib, device as dev
    mode = getattr(_lib, "SMOOTH_" + mode_name)
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    T = 9
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    x_trj = dev.to_dev(orc.rollout(sys_o, params.x0, params.u_trj_initial))
    u_trj = dev.to_dev(params.u_trj_initial)
    dm = sys_d.dm()
    g = torch.Generator(device="cuda").manual_seed(3)
    for N in (5000, 700, 5000, 123457):
        dx = 0.1 * torch.randn((T, N, n), generator=g, device="cuda", dtype=torch.float32)
        du = 0.1 * torch.randn((T, N, m), generator=g, device="cuda", dtype=torch.float32)
        o = dm.smooth(mode, x_trj, u_trj, dx, du)
        sums = dm.smooth_accumulate(mode, x_trj, u_trj, dx, du)
        At, Bt, ct, info = dm.smooth_finalize(mode, N, x_trj, u_trj, sums)
        # two instantiations of one template: the compiler may contract FMAs differently,
        # so the f32 sums agree to rounding rather than bit for bit
        scale = sums.abs().max().item()
        np.testing.assert_allclose(o["sums"].cpu().numpy(), sums.cpu().numpy(), rtol=1e-5, atol=1e-6 * scale)
        np.testing.assert_allclose(o["At"].cpu().numpy(), At.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o["Bt"].cpu().numpy(), Bt.cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o["ct"].cpu().numpy(), ct.cpu().numpy(), rtol=1e-5, atol=1e-6)
        assert int(info.abs().sum().item()) == 0 and int(o["info"].abs().sum().item()) == 0
        # the same launch twice IS bit-identical (fixed-order reductions, no float atomics)
        o2 = dm.smooth(mode, x_trj, u_trj, dx, du)
        assert torch.equal(o2["sums"], o["sums"]) and torch.equal(o2["At"], o["At"])


def test_riccati_generic_sizes_vs_oracle(amd):
    """irs_tvlqr_riccati for (n,m) without a compile-time specialisation."""
    from irs_mpc_amd import device as dev
    rng = np.random.default_rng(4)
    for n, m, T in ((3, 2, 17), (7, 4, 40), (32, 16, 5), (1, 1, 3)):
        At = np.eye(n) + 0.1 * rng.normal(size=(T, n, n))
        Bt = rng.normal(size=(T, n, m))
        ct = 0.1 * rng.normal(size=(T, n))
        Q, Qd, R = np.eye(n) * 2.0, np.eye(n) * 5.0, np.eye(m) * 0.7
        xd = rng.normal(size=(T + 1, n))
        K, k, info = dev.tvlqr_riccati(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd)], alpha_R=0.5)
        Ko, ko = orc.tvlqr_riccati(At, Bt, ct, Q, Qd, R, xd, alpha_R=0.5)
        assert int(info.item()) == 0
        np.testing.assert_allclose(K.cpu().numpy(), Ko, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(k.cpu().numpy(), ko, rtol=1e-8, atol=1e-9)
    # an indefinite Hessian is reported, LAPACK style
    K, k, info = dev.tvlqr_riccati(*[dev.to_dev(a) for a in (At, Bt, ct, -Q * 50, -Qd * 50, R, xd)], alpha_R=0.5)
    assert int(info.item()) != 0


# ---------------------------------------------------------------- device RNG (mode G)
def test_device_rng_matches_specification(amd):
    dm = amd.QuadrotorDynamics(0.05).dm()
    std_x, std_u = 0.1 * np.arange(1, 13), 0.2 * np.arange(1, 5)
    dx, du = dm.rng_samples(3, 1000, std_x, std_u, seed=0x123456789ABC, it=4, sample_offset=77)
    ox, ou = orc.device_gaussian_samples(3, 1000, 12, 4, std_x, std_u, 0x123456789ABC, 4, sample_offset=77)
    np.testing.assert_allclose(dx.cpu().numpy(), ox, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(du.cpu().numpy(), ou, rtol=2e-5, atol=2e-6)
    # the split over devices does not change the stream
    dx2, _ = dm.rng_samples(3, 400, std_x, std_u, seed=0x123456789ABC, it=4, sample_offset=77 + 600)
    assert torch.equal(dx2, dx[:, 600:])


@pytest.mark.parametrize("name,cls,N", [("pendulum", "IrsLqrZeroOrder", 10000), ("quadrotor", "IrsLqrFirstOrder", 500)])
def test_mode_G_smoothing_vs_oracle(amd, name, cls, N):
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    T = 12
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    sm = amd.GaussianSmoothing(0.3 * np.ones(n), 0.2 * np.ones(m), N, seed=42)
    sol = getattr(amd, cls)(sys_d, params, sm)
    sol.iter = 3
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    sx, su = sm.stds(3)
    dx, du = orc.device_gaussian_samples(T, N, n, m, sx, su, 42, 3, dtype=np.float32)
    fn = orc.zero_order_TV if cls == "IrsLqrZeroOrder" else orc.first_order_TV
    Ao, Bo, co = fn(sys_o, sol.x_trj, sol.u_trj, dx.astype(np.float64), du.astype(np.float64))
    np.testing.assert_allclose(At, Ao, rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(Bt, Bo, rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(ct, co, rtol=2e-4, atol=5e-5)


# ---------------------------------------------------------------- TV-LQR
@pytest.mark.parametrize("name,T", [("pendulum", 30), ("quadrotor", 50)])
def test_riccati_gains_and_descent_vs_oracle(amd, name, T):
    sys_d, sys_o = systems(amd, name)
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    sol = amd.IrsLqrExact(sys_d, params)
    x_new, u_new = sol.local_descent(sol.x_trj, sol.u_trj)
    At, Bt, ct = orc.exact_TV(sys_o, sol.x_trj, sol.u_trj)
    np.testing.assert_allclose(sol._last["At"].cpu().numpy(), At, rtol=0, atol=1e-11)
    xo, uo, K, k = orc.local_descent(sys_o, At, Bt, ct, params.Q, params.Qd, params.R, params.x0, params.xd_trj)
    np.testing.assert_allclose(sol._last["K"].cpu().numpy(), K, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sol._last["k"].cpu().numpy(), k, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(u_new, uo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(x_new, xo, rtol=1e-8, atol=1e-9)


def test_solve_tvlqr_matches_qp(amd):
    """tv_lqr.solve_tvlqr drop-in vs the literal QP restatement (KKT solve)."""
    sys_o = orc.QuadrotorOracle(0.05)
    p = quad_params(amd, 8)
    x = orc.rollout(sys_o, p.x0, p.u_trj_initial)
    At, Bt, ct = orc.exact_TV(sys_o, x, p.u_trj_initial)
    xs, us = amd.solve_tvlqr(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj, amd.get_solver("osqp"))
    xq, uq = orc.solve_tvlqr_qp(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj)
    np.testing.assert_allclose(us, uq, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(xs, xq, rtol=1e-6, atol=1e-7)
    # inactive bounds: the Riccati solution stands
    wide = np.stack([np.full((8, 4), -1e3), np.full((8, 4), 1e3)])
    xs2, us2 = amd.solve_tvlqr(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj, None, u_bound_abs=wide)
    np.testing.assert_array_equal(us2, us)
    with pytest.raises(ValueError):
        amd.get_solver("nope")


def test_solve_tvlqr_bounded_stand_alone(amd):
    """solve_tvlqr with ACTIVE bounds as a stand-alone call (tv_lqr.py:112-137): one launch of the bounded
    kernel for a single tail.  Certified against the QP's KKT conditions (oracle, independent of any solver)
    and equal to the oracle's ADMM solution.  Plain form (bicycle: steer bound + input bound, alpha_R = 1/2)
    and position-controlled form (planar hand: indices_u_into_x, cost on du, trust region + rate limit)."""
    # ---- plain form
    sys_o = orc.BicycleOracle(0.1)
    p = bike_params(amd, 25)
    x = orc.rollout(sys_o, p.x0, p.u_trj_initial)
    At, Bt, ct = orc.exact_TV(sys_o, x, p.u_trj_initial)
    T = 25
    xb = np.stack([np.tile([-1e4, -1e4, -1e4, -1e4, -0.3], (T + 1, 1)), np.tile([1e4, 1e4, 1e4, 1e4, 0.3], (T + 1, 1))])
    ub = np.stack([np.full((T, 2), -2.0), np.full((T, 2), 2.0)])
    xs, us = amd.solve_tvlqr(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj, amd.get_solver("osqp"),
                             x_bound_abs=xb, u_bound_abs=ub, eps=1e-9)
    assert np.abs(us).max() > 2.0 - 1e-6 or np.abs(xs[1:, 4]).max() > 0.3 - 1e-6       # something binds
    res = orc.qp_box_kkt_residuals(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj, xb[0][0], xb[1][0], ub[0][0], ub[1][0],
                                   xs, us, alpha_R=0.5)
    assert max(res) < 1e-5, res
    xu, uu = amd.solve_tvlqr(At, Bt, ct, p.Q, p.Qd, p.R, p.x0, p.xd_trj, None)
    assert np.abs(uu - us).max() > 1e-2                                                  # and it matters
    # ---- position-controlled form: the first tail QP of the planar hand's descent
    from irs_mpc_amd import device as dev
    T = 10
    sys_d, sys_o2, x0, u_trj, x_trj, _, (A2, B2, c2), (Q, Qd, R, xd) = _hand_problem(amd, T, 300, 77)
    idx = sys_o2.indices_u_into_x
    rows = orc.quasistatic_bounds(x_trj, idx, None, np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]),
                                  np.array([-np.ones(4) * 0.03, np.ones(4) * 0.03]))
    xs3, us3 = amd.solve_tvlqr(A2, B2, c2, Q, Qd, R, x0, xd, None, indices_u_into_x=idx,
                               u_bound_abs=np.stack([rows[2], rows[3]]), u_bound_rel=np.stack([rows[4], rows[5]]),
                               rho=100.0, eps=1e-10, max_iter=40000)
    # oracle: the same QP on the [x; u_prev] augmentation, solved by its ADMM
    Ab, Bb, cb, Qb, Qdb, xdb = orc.quasistatic_augment(A2, B2, c2, Q, Qd, xd)
    zlo = np.hstack([rows[0], np.vstack([np.full((1, 4), -np.inf), rows[2]])])
    zhi = np.hstack([rows[1], np.vstack([np.full((1, 4), np.inf), rows[3]])])
    F = orc.tvlqr_box_factor(Ab, Bb, cb, Qb, Qdb, R, zlo, zhi, rows[4], rows[5], 100.0, alpha_R=1.0)
    z0 = np.concatenate([x0, x0[idx]])
    zx, zu, _, it = orc.tvlqr_box_solve(F, Ab, Bb, cb, Qb, Qdb, xdb, z0, 0, zlo, zhi, rows[4], rows[5], None, 40000, 1e-10, 1.6)
    np.testing.assert_allclose(xs3, zx[:, :7], rtol=0, atol=1e-7)
    np.testing.assert_allclose(us3, zx[1:, 7:], rtol=0, atol=1e-7)
    assert np.abs(np.diff(np.vstack([x0[idx][None], us3]), axis=0)).max() <= 0.03 + 1e-7
    with pytest.raises(NotImplementedError):
        amd.solve_tvlqr(A2, B2, c2, Q, Qd, R, x0, xd, None, indices_u_into_x=[0, 1, 2, 3])


def test_compute_least_squares_vs_reference_fixture(amd, golden_dir):
    """IrsLqrZeroOrder.compute_least_squares stand-alone (irs_lqr_zero_order.py:27-36) on the device, against
    the A_t, B_t the REFERENCE's own get_TV_matrices produced from the same samples (fixture generated by
    running the reference: tests/golden/make_fixtures.py) and against numpy's SVD lstsq."""
    d = np.load(os.path.join(golden_dir, "pendulum_zero_T30_N100.npz"))
    sys_o = orc.PendulumOracle(float(d["h"]))
    sol = amd.IrsLqrZeroOrder(amd.PendulumDynamics(float(d["h"])), pend_params(amd, 30), sampling=None)
    for t in (0, 7, 29):
        dx, du = d["dx"][t].astype(float), d["du"][t].astype(float)
        X, U = d["x_trj"][t] + dx, d["u_trj"][t] + du
        deltaf = sys_o.dynamics_batch(X, U) - sys_o.dynamics(d["x_trj"][t], d["u_trj"][t])
        A, B = sol.compute_least_squares(np.hstack([dx, du]), deltaf)
        np.testing.assert_allclose(A, d["At"][t], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(B, d["Bt"][t], rtol=1e-9, atol=1e-11)
        ref = np.linalg.lstsq(np.hstack([dx, du]), deltaf, rcond=None)[0].T
        np.testing.assert_allclose(np.hstack([A, B]), ref, rtol=1e-9, atol=1e-11)
    with pytest.raises(ValueError):
        sol.compute_least_squares(np.zeros((10, 3)), np.zeros((10, 2)))                    # rank deficient


# ---------------------------------------------------------------- end to end vs the reference's result files
def test_pendulum_exact_csv_end_to_end(amd, golden_dir):
    gold = np.loadtxt(os.path.join(golden_dir, "pendulum_exact.csv"))
    sol = amd.IrsLqrExact(amd.PendulumDynamics(0.05), pend_params(amd, 200))
    sol.verbose = False
    sol.iterate(len(gold) - 2)
    np.testing.assert_allclose(sol.cost_lst, gold, rtol=2e-9)
    assert len(sol.x_trj_lst) == len(gold) and sol.iter == len(gold) - 1


def test_quadrotor_exact_csv_end_to_end(amd, golden_dir):
    gold = np.loadtxt(os.path.join(golden_dir, "quadrotor_exact.csv"))[:5]
    sol = amd.IrsLqrExact(amd.QuadrotorDynamics(0.05), quad_params(amd, 200))
    sol.verbose = False
    sol.iterate(3)
    np.testing.assert_allclose(sol.cost_lst, gold, rtol=1e-7)


def test_zero_order_iterate_vs_oracle_same_seed(amd):
    """Whole iterate() loop, identical seeds: cost history vs the oracle fed the
    same NumPy RNG stream (f32-rounded samples)."""
    T, N = 30, 1000
    params = pend_params(amd, T)

    def sampling(xbar, ubar, it):
        dx = np.random.normal(0.0, np.array([1.0, 1.0]) / it ** 0.5, size=(N, 2)).astype(np.float32)
        du = np.random.normal(0.0, np.array([1.0]) / it ** 0.5, size=(N, 1)).astype(np.float32)
        return dx, du

    np.random.seed(5)
    sol = amd.IrsLqrZeroOrder(amd.PendulumDynamics(0.05), params, sampling)
    sol.verbose = False
    sol.iterate(4)

    sys_o = orc.PendulumOracle(0.05)

    def tv(x, u, it):
        dx = np.zeros((T, N, 2))
        du = np.zeros((T, N, 1))
        for t in range(T):
            a, b = sampling(x[t], u[t], it)
            dx[t], du[t] = a, b
        return orc.zero_order_TV(sys_o, x, u, dx, du)

    np.random.seed(5)
    *_, cost_lst, x_lst, u_lst = orc.iterate(sys_o, params.Q, params.Qd, params.R, params.x0, params.xd_trj,
                                             params.u_trj_initial, 4, tv)
    np.testing.assert_allclose(sol.cost_lst, cost_lst, rtol=1e-4)
    np.testing.assert_allclose(sol.u_trj_lst[-1], u_lst[-1], rtol=1e-3, atol=1e-3)


def test_stochastic_band_vs_reference_csv(amd, golden_dir):
    """Mode G (device RNG), the script's configuration (T=200, N=1000): iteration
    costs land in the run-to-run band of the reference's own two unseeded runs."""
    gold = np.loadtxt(os.path.join(golden_dir, "pendulum_zero_order.csv"))
    sm = amd.GaussianSmoothing([1.0, 1.0], [1.0], 1000, seed=1)
    sol = amd.IrsLqrZeroOrder(amd.PendulumDynamics(0.05), pend_params(amd, 200), sm)
    sol.verbose = False
    sol.iterate(7)
    assert sol.cost_lst[0] == pytest.approx(gold[0], rel=1e-12)
    np.testing.assert_allclose(sol.cost_lst[1:5], gold[1:5], rtol=0.02)
    assert abs(sol.cost_lst[8] - gold[8]) / gold[8] < 0.005


# ---------------------------------------------------------------- error behaviour (irs_lqr.py:73-103)
def test_error_behaviour(amd):
    p = pend_params(amd, 5)
    p.Q = np.eye(3)
    with pytest.raises(RuntimeError, match="Q matrix"):
        amd.IrsLqrExact(amd.PendulumDynamics(0.05), p)

    class Empty(amd.DynamicalSystem):
        pass

    with pytest.raises(RuntimeError, match="zero states"):
        amd.IrsLqrExact(Empty(), pend_params(amd, 5))

    class NoDevice(amd.DynamicalSystem):
        def __init__(self):
            super().__init__()
            self.dim_x, self.dim_u, self.h = 2, 1, 0.1

    with pytest.raises(RuntimeError, match="Could not evaluate dynamics"):
        amd.IrsLqrExact(NoDevice(), pend_params(amd, 5))


# ---------------------------------------------------------------- bicycle / three_cart device models
def bike_params(amd, T):
    p = amd.IrsLqrParameters()
    p.Q, p.Qd, p.R = np.diag([5, 5, 3, 0.1, 0.1]), np.diag([50., 50, 30, 1, 1]), np.diag([1, 0.1])
    p.x0 = np.zeros(5)
    p.xd_trj = np.tile(np.array([3.0, 1.0, np.pi / 2, 0, 0]), (T + 1, 1))
    p.u_trj_initial = np.tile(np.array([0.1, 0.0]), (T, 1))
    return p


def cart_params(amd, T):
    p = amd.IrsLqrParameters()
    p.Q = 0.01 * np.diag([50., 50, 50, 20, 100, 20])
    p.Qd = np.diag([50., 50, 50, 20, 100, 20])
    p.R = 0.01 * np.eye(2)
    p.x0 = np.array([0., 1, 2, 0, 0, 0])
    p.xd_trj = np.tile(np.array([2., 3, 4, 0, 0, 0]), (T + 1, 1))
    p.u_trj_initial = np.tile(np.array([0.1, -0.1]), (T, 1))
    return p


def test_bicycle_vs_reference_fixtures(amd, golden_dir):
    f = load(golden_dir, "bicycle_dynamics")
    bike = amd.BicycleDynamics(float(f["h"]))
    np.testing.assert_allclose(bike.dynamics_batch(f["X"], f["U"]), f["Xn"], rtol=0, atol=1e-12)
    so = orc.BicycleOracle(0.1)
    np.testing.assert_allclose(bike.jacobian_xu_batch(f["X"], f["U"]), so.jacobian_xu_batch(f["X"], f["U"]),
                               rtol=1e-12, atol=1e-12)
    g = load(golden_dir, "bicycle_T100_init")
    sol = amd.IrsLqrExact(bike, bike_params(amd, 100))
    np.testing.assert_allclose(sol.x_trj, g["x_trj"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(sol.cost, float(g["cost0"]), rtol=1e-13)       # 3302.0894 of the reference CSVs
    z = load(golden_dir, "bicycle_zero_T8_N200")
    sol = amd.IrsLqrZeroOrder(bike, bike_params(amd, 8), Replay(z["dx"], z["du"]))
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    # x_initial_var mixes std 2.0 and 0.01 columns (bicycle_zero_order.py:34): the Jacobi
    # scaling keeps the normal equations well conditioned
    np.testing.assert_allclose(At, z["At"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(Bt, z["Bt"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ct, z["ct"], rtol=2e-4, atol=2e-4)


def test_three_cart_vs_reference_fixture(amd, golden_dir):
    f = load(golden_dir, "three_cart_dynamics")
    carts = amd.ThreeCartDynamics(float(f["h"]))
    np.testing.assert_allclose(carts.dynamics_batch(f["X"], f["U"]), f["Xn_scalar"], rtol=0, atol=1e-13)
    so = orc.ThreeCartOracle(0.05)
    np.testing.assert_allclose(carts.jacobian_xu_batch(f["X"], f["U"]), so.jacobian_xu_batch(f["X"], f["U"]),
                               rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,T,N", [("bicycle", 20, 4000), ("three_cart", 20, 4000)])
def test_new_models_smoothing_and_descent_vs_oracle(amd, name, T, N):
    sys_d = amd.BicycleDynamics(0.1) if name == "bicycle" else amd.ThreeCartDynamics(0.05)
    sys_o = orc.BicycleOracle(0.1) if name == "bicycle" else orc.ThreeCartOracle(0.05)
    params = (bike_params if name == "bicycle" else cart_params)(amd, T)
    n, m = sys_o.dim_x, sys_o.dim_u
    rng = np.random.default_rng(7)
    dx = (rng.normal(size=(T, N, n)) * 0.3).astype(np.float32)
    du = (rng.normal(size=(T, N, m)) * 0.3).astype(np.float32)
    sol = amd.IrsLqrZeroOrder(sys_d, params, Replay(dx, du))
    sol.verbose = False
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    Ao, Bo, co = orc.zero_order_TV(sys_o, sol.x_trj, sol.u_trj, dx.astype(np.float64), du.astype(np.float64))
    np.testing.assert_allclose(At, Ao, rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(Bt, Bo, rtol=2e-4, atol=1e-4)
    np.testing.assert_allclose(ct, co, rtol=2e-4, atol=1e-4)
    if name == "bicycle":       # smooth model: first-order too
        fo = amd.IrsLqrFirstOrder(sys_d, params, Replay(dx, du))
        A1, B1, c1 = fo.get_TV_matrices(fo.x_trj, fo.u_trj)
        A2, B2, c2 = orc.first_order_TV(sys_o, fo.x_trj, fo.u_trj, dx.astype(np.float64), du.astype(np.float64))
        np.testing.assert_allclose(A1, A2, **TOL_AB)
        np.testing.assert_allclose(B1, B2, **TOL_AB)
    # Riccati + closed-loop rollout on the device's own linearisation vs the oracle on the same
    x_new, u_new = sol.local_descent(sol.x_trj, sol.u_trj)
    L = sol._last
    xo, uo, K, k = orc.local_descent(sys_o, L["At"].cpu().numpy(), L["Bt"].cpu().numpy(), L["ct"].cpu().numpy(),
                                     params.Q, params.Qd, params.R, params.x0, params.xd_trj)
    np.testing.assert_allclose(L["K"].cpu().numpy(), K, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(u_new, uo, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(x_new, xo, rtol=1e-7, atol=1e-8)


# ---------------------------------------------------------------- box-constrained TV-LQR
@pytest.mark.parametrize("T,steer,ubnd", [(30, np.pi / 4, None), (25, 0.3, 2.0)])
def test_box_descent_vs_oracle(amd, T, steer, ubnd):
    """irs_tvlqr_box_descent (T warm-started tail QPs, ADMM + shared Riccati factorisation)
    vs the oracle's KKT-certified restatement, state and input bounds active."""
    params = bike_params(amd, T)
    params.xbound = [-np.array([1e4, 1e4, 1e4, 1e4, steer]), np.array([1e4, 1e4, 1e4, 1e4, steer])]
    if ubnd is not None:
        params.ubound = np.array([[-ubnd, -1e4], [ubnd, 1e4]])
    sol = amd.IrsLqrExact(amd.BicycleDynamics(0.1), params)
    x_new, u_new = sol.local_descent(sol.x_trj, sol.u_trj)
    assert sol._box_used and int(sol._last["box_info"][2].item()) == 0
    so = orc.BicycleOracle(0.1)
    At, Bt, ct = orc.exact_TV(so, sol.x_trj, sol.u_trj)
    xlo = np.array([-np.inf] * 4 + [-steer])
    ulo = np.array([-ubnd if ubnd else -np.inf, -np.inf])
    xo, uo, iters = orc.local_descent_box(so, At, Bt, ct, params.Q, params.Qd, params.R, params.x0, params.xd_trj,
                                          xlo, -xlo, ulo, -ulo, rho=10.0, max_iter=20000, eps=1e-10)
    np.testing.assert_allclose(u_new, uo, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x_new, xo, rtol=1e-5, atol=1e-6)
    assert np.abs(x_new[:, 4]).max() > steer - 1e-3              # the bound is active
    if ubnd is not None:
        assert np.abs(u_new[:, 0]).max() == pytest.approx(ubnd, abs=1e-9)


def test_box_descent_without_active_bounds_equals_riccati(amd):
    """Genuine but never-active bounds (pendulum swing-up, |x| stays far below the +-100 box).  Every FINITE
    entry of xbound / ubound is a genuine bound -- no magnitude threshold: the host class runs the Riccati
    descent, checks every tail's unconstrained plan against the box and keeps the result when all lie inside
    (then it IS the solution of the bounded QPs); the ADMM kernel on the same problem lands on the same
    descent; and a bound the plans do cross switches the host class to the bounded kernel."""
    from irs_mpc_amd import device as dev
    T = 40
    params = pend_params(amd, T)
    sol_u = amd.IrsLqrExact(amd.PendulumDynamics(0.05), params)
    xu, uu = sol_u.local_descent(sol_u.x_trj, sol_u.u_trj)
    assert not sol_u._box_used
    params.xbound = [-np.full(2, 100.0), np.full(2, 100.0)]
    params.ubound = np.array([[-100.0], [100.0]])
    sol_b = amd.IrsLqrExact(amd.PendulumDynamics(0.05), params)
    assert sol_b._box_bounds() is not None                       # 100 is a bound, not "none"
    xb, ub = sol_b.local_descent(sol_b.x_trj, sol_b.u_trj)
    assert not sol_b._box_used                                   # ... that no tail plan touches
    np.testing.assert_array_equal(ub, uu)
    # the bounded kernel itself on this problem == the Riccati descent
    L = sol_b._last
    o = sol_b._dm.tvlqr_box_descent(L["At"], L["Bt"], L["ct"], sol_b._Q, sol_b._Qd, sol_b._R, sol_b._xd,
                                    dev.to_dev(sol_b.x_trj[0]), *sol_b._box_bounds(), alpha_R=0.5, eps=1e-9)
    assert int(o["info"][2].item()) == 0
    np.testing.assert_allclose(o["u_new"].cpu().numpy(), uu, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o["x_new"].cpu().numpy(), xu, rtol=1e-5, atol=1e-5)
    # a torque limit the swing-up does cross: the plans leave the box, the bounded kernel takes over
    lim = 0.5 * np.abs(uu).max()
    params.ubound = np.array([[-lim], [lim]])
    sol_c = amd.IrsLqrExact(amd.PendulumDynamics(0.05), params)
    xc, uc = sol_c.local_descent(sol_c.x_trj, sol_c.u_trj)
    assert sol_c._box_used and int(sol_c._last["box_info"][2].item()) == 0
    assert np.abs(uc).max() <= lim + 1e-6 and np.abs(uc - uu).max() > 1e-3


def test_bicycle_exact_csv_end_to_end(amd, golden_dir):
    """examples/bicycle/bicycle_exact.py (T=100, steer bound active) against
    analysis/bicycle_easy_exact.csv.  The reference's curve carries OSQP's default 1e-3
    accuracy on each of 100 tail QPs per descent; this solver converges to 1e-8, so the
    curves agree to ~1 % on the first descent and settle at the same cost level (the
    reference's own tail wanders between 663.7 and 671.2)."""
    gold = np.loadtxt(os.path.join(golden_dir, "bicycle_easy_exact.csv"))
    params = bike_params(amd, 100)
    params.xbound = [-np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4]), np.array([1e4, 1e4, 1e4, 1e4, np.pi / 4])]
    params.ubound = np.array([-np.array([1e4, 1e4]), np.array([1e4, 1e4])])
    sol = amd.IrsLqrExact(amd.BicycleDynamics(0.1), params)
    sol.verbose = False
    sol.iterate(10)
    assert sol.cost_lst[0] == pytest.approx(gold[0], rel=1e-12)
    assert abs(sol.cost_lst[1] - gold[1]) / gold[1] < 0.012
    assert abs(sol.cost_lst[-1] - gold[-1]) / gold[-1] < 0.03
    assert sol.cost_lst[-1] < 0.25 * sol.cost_lst[0]


# ---------------------------------------------------------------- CEM baseline (irs_lqr/cem.py)
def cem_params(amd, T, B, n_elite):
    p = amd.CemParameters()
    p.Q, p.Qd, p.R = np.diag([1., 1.]), np.diag([20., 20.]), np.diag([1.])
    p.x0 = np.array([0., 0.])
    p.xd_trj = np.tile(np.array([np.pi, 0.]), (T + 1, 1))
    p.u_trj_initial = np.tile(np.array([0.1]), (T, 1))
    p.initial_std = np.array([1.0])
    p.batch_size, p.n_elite = B, n_elite
    return p


def test_cem_local_descent_vs_reference_fixture(amd, golden_dir):
    """Identical seed as the reference run that produced the fixture (cem.py:151-184)."""
    f = load(golden_dir, "pendulum_cem_T30_B50")
    cem = amd.CrossEntropyMethod(amd.PendulumDynamics(float(f["h"])), cem_params(amd, 30, 50, int(f["n_elite"])))
    np.random.seed(int(f["seed"]))
    x_new, u_new = cem.local_descent(cem.x_trj, cem.u_trj)
    np.testing.assert_allclose(u_new, f["u_new"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(cem.std_trj, f["std_new"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(x_new, f["x_new"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name,T,B,n_elite", [("pendulum", 30, 5000, 50), ("quadrotor", 20, 3000, 7),
                                              ("pendulum", 5, 3, 3), ("pendulum", 80, 50000, 2500)])
def test_cem_step_vs_oracle(amd, name, T, B, n_elite):
    from irs_mpc_amd import device as dev
    sys_d, sys_o = systems(amd, name)
    n, m = sys_o.dim_x, sys_o.dim_u
    params = (pend_params if name == "pendulum" else quad_params)(amd, T)
    rng = np.random.default_rng(B)
    cand = params.u_trj_initial + (0.5 if name == "pendulum" else 0.05) * rng.normal(size=(B, T, m))
    dm = sys_d.dm()
    Q, R, xd, x0 = (dev.to_dev(a) for a in (params.Q, params.R, params.xd_trj, params.x0))
    cd = dev.to_dev(cand)
    costs = dm.cem_rollout_costs(cd, x0, Q, R, xd)
    idx, u_new, std_new = dm.cem_refit(cd, costs, n_elite)
    check = min(B, 400)        # the oracle's Python rollouts are slow: verify a subset of costs ...
    sel = rng.choice(B, check, replace=False)
    co = np.array([orc.evaluate_cost(orc.rollout(sys_o, params.x0, cand[b]), cand[b], params.xd_trj, params.Q,
                                     params.R) for b in sel])
    np.testing.assert_allclose(costs.cpu().numpy()[sel], co, rtol=1e-11)
    # ... and the selection + refit against NumPy on the device's own costs
    ch = costs.cpu().numpy()
    best = np.argpartition(ch, n_elite - 1)[:n_elite] if n_elite < B else np.arange(B)
    assert sorted(idx.cpu().numpy().tolist()) == sorted(best.tolist())
    np.testing.assert_allclose(u_new.cpu().numpy(), cand[best].mean(axis=0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(std_new.cpu().numpy(), cand[best].std(axis=0), rtol=1e-9, atol=1e-12)


def test_cem_select_ties_and_nan(amd):
    """Equal costs at the threshold: lowest indices win; NaN costs are never elite."""
    from irs_mpc_amd import device as dev
    dm = amd.PendulumDynamics(0.05).dm()
    costs = np.array([5., 1., 3., 3., np.nan, 3., 0.5, 3., 9., -2.])
    cand = np.arange(10, dtype=float).reshape(10, 1, 1)
    idx, u_new, std_new = dm.cem_refit(dev.to_dev(cand), dev.to_dev(costs), 5)
    assert sorted(idx.cpu().numpy().tolist()) == [1, 2, 3, 6, 9]
    np.testing.assert_allclose(u_new.cpu().numpy().ravel(), [np.mean([1, 2, 3, 6, 9])])
    idx, _, _ = dm.cem_refit(dev.to_dev(cand), dev.to_dev(costs), 9)
    assert 4 not in idx.cpu().numpy().tolist()


def test_cem_iterate_reduces_cost(amd):
    cem = amd.CrossEntropyMethod(amd.PendulumDynamics(0.05), cem_params(amd, 30, 2000, 20))
    cem.verbose = False
    np.random.seed(0)
    cem.iterate(5)
    assert len(cem.cost_lst) == 7
    assert all(b < a for a, b in zip(cem.cost_lst, cem.cost_lst[1:]))      # monotone descent here
    assert cem.cost_lst[-1] < 0.92 * cem.cost_lst[0]


# ---------------------------------------------------------------- example scripts
@pytest.mark.parametrize("argv", [["pendulum", "zero_order", "--iters", "2", "--T", "60", "--quiet"],
                                  ["quadrotor", "first_order", "--iters", "1", "--T", "30", "--N", "200", "--quiet",
                                   "--device-rng"],
                                  ["three_cart", "zero_order", "--iters", "1", "--T", "30", "--quiet"],
                                  ["pendulum", "cem", "--iters", "2", "--T", "40", "--quiet"],
                                  ["bicycle", "exact", "--iters", "1", "--T", "40", "--quiet"]])
def test_example_runner(amd, argv, monkeypatch, capsys):
    import examples.run as run
    monkeypatch.setattr("sys.argv", ["run.py"] + argv)
    run.main()
    out = capsys.readouterr().out
    hist = [float(v) for v in out.split("cost history:")[1].split()]
    assert len(hist) == int(argv[3]) + 2 and all(np.isfinite(hist)) and hist[1] < hist[0]


# ---------------------------------------------------------------- planar quasi-dynamic contact (a7, unpinned)
HAND = orc.PlanarHandOracle
HAND_IDX = np.array([1, 4, 2, 5])        # indices_u_into_x in the reference's state order
HAND_Q = HAND.pack([1e-3, 1e-3, 10.0], [1e-3, 1e-3], [1e-3, 1e-3])          # run_planar_hand.py:113-121
HAND_GOAL = HAND.pack([0.3, -0.1, 0.5], [0, 0], [0, 0])                     # :123-125


def _hand_setup(amd, T, settle=4):
    sys_d, sys_o = amd.PlanarHandDynamics(0.1), orc.PlanarHandOracle(0.1)
    x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])   # run_planar_hand.py:31-44
    for _ in range(settle):
        x0 = sys_o.dynamics(x0, np.array([-np.pi / 4, -np.pi / 4, np.pi / 4, np.pi / 4]))
    u_trj = np.tile(x0[HAND_IDX], (T, 1)) + 0.02 * np.sin(np.arange(T))[:, None] * np.array([1, -1, -1, 1])
    return sys_d, sys_o, x0, u_trj


def test_planar_hand_dynamics_vs_oracle(amd):
    """The device functor's QP assembly + PGS sweeps reproduce the NumPy restatement (f64)."""
    sys_d, sys_o, x0, _ = _hand_setup(amd, 1)
    rng = np.random.default_rng(11)
    X = x0 + 0.03 * rng.normal(size=(512, 7))
    U = x0[HAND_IDX] + 0.1 * rng.normal(size=(512, 4))
    got = sys_d.dynamics_batch(X, U)
    np.testing.assert_allclose(got, sys_o.dynamics_batch(X, U), rtol=0, atol=1e-10)
    np.testing.assert_allclose(sys_d.dynamics(X[0], U[0]), sys_o.dynamics(X[0], U[0]), rtol=0, atol=1e-10)
    # the step's active-set derivative (the simulator's Dq_nextDq | Dq_nextDqa_cmd), f64 on the device
    U2 = x0[HAND_IDX] + 0.3 * rng.normal(size=(512, 4))
    Jd, Jo = sys_d.jacobian_xu_batch(X, U2), sys_o.jacobian_xu_batch(X, U2)
    np.testing.assert_allclose(Jd, Jo, rtol=0, atol=1e-8)
    assert np.abs(Jo[:, HAND.PERM[:3], 7:]).max() > 0.1            # contacts are active in the batch
    np.testing.assert_allclose(sys_d.jacobian_xu(X[0], U2[0]), Jo[0], rtol=0, atol=1e-8)
    np.testing.assert_allclose(sys_d.calc_AB_exact(X[1], U2[1]), Jo[1], rtol=0, atol=1e-8)


def test_planar_hand_zero_order_B_decoupled_vs_oracle(amd):
    """calc_B_zero_order + decouple_AB_matrices (quasistatic_dynamics.py:242-266,
    irs_lqr_quasistatic.py:275-284) on the contact functor."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    T, N = 6, 2000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    rng = np.random.default_rng(12)
    du = (rng.normal(size=(T, N, 4)) * 0.1).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    At, Bt, ct = o["At"], o["Bt"], o["ct"]
    assert int(o["info"].abs().sum().item()) == 0
    Ao, Bo, co = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    np.testing.assert_allclose(At.cpu().numpy(), Ao, rtol=0, atol=0)
    # f32 sample path (exact dual solve per sample) vs the f64 oracle
    np.testing.assert_allclose(Bt.cpu().numpy(), Bo, **FP32_TOL)
    np.testing.assert_allclose(ct.cpu().numpy(), co, **FP32_TOL)
    # the object rows of B see the contacts: pushing the fingers in moves the disc
    assert np.abs(Bo[:, HAND.PERM[:3], :]).max() > 0.05


def test_planar_hand_first_order_decoupled_vs_oracle(amd):
    """gradient_mode "first_order" (calc_AB_first_order, quasistatic_dynamics.py:193-208, then
    decouple_AB_matrices): every sample's contact step is differentiated through its active constraints
    INSIDE the f32 sample pass and the n x m blocks are averaged.  Fused launch == accumulate over 3
    shards + finalize == the f64 oracle on the same draws.  Tolerance: an f32 lane and the f64 oracle can
    classify a borderline row (lam_i W_ii within rounding of the 1e-7 threshold, or a pivot at the drop
    threshold) differently; one such sample moves the mean by O(1)/N."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER
    T, N = 6, 3000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    rng = np.random.default_rng(13)
    du = (rng.normal(size=(T, N, 4)) * 0.1).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    o = dm.smooth(SMOOTH_FIRST_ORDER, xd, ud, None, dev.to_dev(du, dev.F32))
    assert int(o["info"].abs().sum().item()) == 0
    Ao, Bo, co = orc.first_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    At, Bt, ct = o["At"].cpu().numpy(), o["Bt"].cpu().numpy(), o["ct"].cpu().numpy()
    np.testing.assert_allclose(At, Ao, rtol=0, atol=0)
    np.testing.assert_allclose(Bt, Bo, **FP32_TOL)
    np.testing.assert_allclose(ct, co, **FP32_TOL)
    assert np.abs(Bo[:, HAND.PERM[:3], :]).max() > 0.05
    # the estimator agrees with the zero-order one to Monte-Carlo accuracy (same smoothed dynamics)
    Az, Bz, cz = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    assert np.abs(Bz - Bo).max() < 0.15
    # sharded: sums over 3 uneven shards, then the stand-alone solve
    sums = None
    for lo, hi in ((0, 1000), (1000, 1700), (1700, N)):
        part = dm.smooth_accumulate(SMOOTH_FIRST_ORDER, xd, ud, None, dev.to_dev(np.ascontiguousarray(du[:, lo:hi]), dev.F32))
        sums = part.clone() if sums is None else sums + part
    A2, B2, c2, info = dm.smooth_finalize(SMOOTH_FIRST_ORDER, N, xd, ud, sums)
    np.testing.assert_allclose(B2.cpu().numpy(), Bt, rtol=0, atol=1e-6)
    np.testing.assert_allclose(c2.cpu().numpy(), ct, rtol=0, atol=1e-6)
    # device-drawn perturbations: same estimator, other draws
    o2 = dm.smooth_rng(SMOOTH_FIRST_ORDER, xd, ud, N, None, np.full(4, 0.1), 7, 1)
    assert np.abs(o2["Bt"].cpu().numpy() - Bt).max() < 0.1


@pytest.mark.parametrize("system,std", [("planar_hand", 0.3), ("planar_hand", 0.05), ("box_pivoting", 0.05)])
def test_first_order_contact_per_sample_classification(amd, system, std):
    """First-order smoothing of a contact model, sample by sample (irs_contact_samples_f32 = the f32 lanes of
    the sample pass; quasistatic_dynamics.py:193-208): against the f64 oracle on the SAME f32-rounded inputs,
      * the samples whose derivative block differs by more than 1e-3 -- an f32 lane and the f64 oracle on
        different faces of the piecewise-linear step, i.e. a different active set -- are < 1 % of the draw,
      * all other samples agree to f32 accuracy one by one, and their mean to the stated tolerance
        (rtol 1e-4, atol 2e-5) -- so the sample pass's error is the flipped share times O(1), nothing else,
      * every large deviation comes with a differing active-set mask (masks may differ WITHOUT consequence:
        when contact rows are dependent -- a box resting on four corners -- the multipliers are not unique,
        and the projector the derivative uses does not depend on which rows carry them)."""
    from irs_mpc_amd import device as dev
    N = 20000
    if system == "planar_hand":
        sys_d, sys_o = amd.PlanarHandDynamics(0.1), orc.PlanarHandOracle(0.1)
        x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
        x = orc.rollout(sys_o, x0, np.tile(x0[HAND_IDX], (25, 1)))[-1]       # the settled grasp
    else:
        sys_d, sys_o = amd.BoxPivotingDynamics(0.1), orc.BoxPivotOracle(0.1)
        x = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
    idx = sys_o.indices_u_into_x
    u = x[idx].copy()
    n, m = sys_o.dim_x, sys_o.dim_u
    du = (std * np.random.default_rng(5).normal(size=(N, m))).astype(np.float32)
    Xn, Bs, mask = sys_d.dm().contact_samples_f32(dev.to_dev(x), dev.to_dev(u), dev.to_dev(du, dev.F32))
    Xn, Bs, mask = Xn.cpu().numpy().astype(float), Bs.cpu().numpy().astype(float), mask.cpu().numpy()
    X = np.tile(x.astype(np.float32).astype(float), (N, 1))
    U = (u.astype(np.float32)[None] + du).astype(float)
    Bo = sys_o.jacobian_xu_batch(X, U)[:, :, n:]
    eB = np.abs(Bs - Bo).reshape(N, -1).max(1)
    flipped = eB > 1e-3
    assert flipped.mean() < 0.01, flipped.mean()
    assert eB[~flipped].max() < 5e-4, eB[~flipped].max()
    np.testing.assert_allclose(Bs[~flipped].mean(0), Bo[~flipped].mean(0), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(Xn[~flipped], sys_o.dynamics_batch(X, U)[~flipped], rtol=0, atol=2e-5)
    assert not (flipped & (mask == sys_o.active_mask_batch(X, U))).any()
    # and the sample pass is the mean of exactly these lanes
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER
    o = sys_d.dm().smooth(SMOOTH_FIRST_ORDER, dev.to_dev(np.stack([x, x])), dev.to_dev(u[None]), None,
                          dev.to_dev(du[None], dev.F32))
    free = np.setdiff1d(np.arange(n), idx)
    np.testing.assert_allclose(o["Bt"].cpu().numpy()[0][free], Bs.mean(0)[free], rtol=0, atol=2e-6)


@pytest.mark.parametrize("N", [50, 64, 130, 700, 5000, 60000])
def test_parked_samples_are_finished_exactly_once(amd, N):
    """The sample pass of the planar hand parks the samples its first attempt does not settle in a per-wave ring
    and finishes them 64 at a time (smooth.hip, DEFER).  Whatever the split -- a partial wave, several waves, a
    ring that fills and is flushed inside the loop (N = 60000: thousands of samples per wave at 8-40 % parked) --
    the launch must equal the per-sample lanes (irs_contact_samples_f32, the undeferred full method): first-order
    = their mean, zero-order-B = the least squares over their steps."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
    sys_d, sys_o = amd.PlanarHandDynamics(0.1), orc.PlanarHandOracle(0.1)
    x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    x = orc.rollout(sys_o, x0, np.tile(x0[HAND_IDX], (25, 1)))[-1]           # the settled grasp: loaded contacts
    idx = sys_o.indices_u_into_x
    u, n, m = x[idx].copy(), sys_o.dim_x, sys_o.dim_u
    free = np.setdiff1d(np.arange(n), idx)
    du = (0.3 * np.random.default_rng(11).normal(size=(N, m))).astype(np.float32)
    Xn, Bs, _ = sys_d.dm().contact_samples_f32(dev.to_dev(x), dev.to_dev(u), dev.to_dev(du, dev.F32))
    Xn, Bs = Xn.cpu().numpy().astype(float), Bs.cpu().numpy().astype(float)
    xs = dev.to_dev(np.stack([x, x]))
    o1 = sys_d.dm().smooth(SMOOTH_FIRST_ORDER, xs, dev.to_dev(u[None]), None, dev.to_dev(du[None], dev.F32))
    np.testing.assert_allclose(o1["Bt"].cpu().numpy()[0][free], Bs.mean(0)[free], rtol=0, atol=2e-6)
    o2 = sys_d.dm().smooth(SMOOTH_ZERO_ORDER_B, xs, dev.to_dev(u[None]), None, dev.to_dev(du[None], dev.F32))
    xf = x.astype(np.float32).astype(float)
    Z = du.astype(float)
    D = Xn - xf[None]
    # the kernel's estimator: B = (Z'Z)^-1 (Z'D - sum(z) (f(x,u) - x)'), f(x,u) in f64
    f0 = sys_o.dynamics(x, u) - xf
    B2 = np.linalg.solve(Z.T @ Z, Z.T @ D - np.outer(Z.sum(0), f0)).T
    np.testing.assert_allclose(o2["Bt"].cpu().numpy()[0][free], B2[free], rtol=0, atol=5e-6)


@pytest.mark.parametrize("system", ["planar_hand", "box_pivoting"])
def test_contact_sample_pass_device_rng_equals_supplied_samples(amd, system):
    """Mode G of the contact kernels (perturbations drawn inside the launch, Philox keyed by the GLOBAL sample index)
    against the same kernels fed the same draws from the host (oracle.device_gaussian_samples restates the
    generator): the RNG instantiations, the block-to-wave dealing and the parked-sample ring must not change which
    sample is which.  What the iLQR loop of bench.py runs.  Zero-order-B is continuous in the sample: equal to
    rounding.  First-order is piecewise constant in the sample, and the planar hand's grasp is statically
    indeterminate (8 contact rows on 7 dofs: the multipliers of the step QP are not unique, only its primal solution
    is): which rows end up carrying the load is decided at rounding level, two instantiations of the same f32 code
    (different fma contraction) decide it differently for ~0.4 % of the samples, and the derivative THROUGH the
    active set differs between such sets.  Measured: 3.7e-3 on B at N = 5000 -- the Monte-Carlo error of the
    estimate itself is ~3e-3.  Box pivoting has no such freedom at this state and agrees to a few samples."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
    T, N, seed, it = 3, 5000, 11, 2
    if system == "planar_hand":
        sys_d, sys_o = amd.PlanarHandDynamics(0.1), orc.PlanarHandOracle(0.1)
        x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
        x0 = orc.rollout(sys_o, x0, np.tile(x0[HAND_IDX], (25, 1)))[-1]          # the settled grasp
    else:
        sys_d, sys_o = amd.BoxPivotingDynamics(0.1), orc.BoxPivotOracle(0.1)
        x0 = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
    idx = sys_o.indices_u_into_x
    n, m = sys_o.dim_x, sys_o.dim_u
    u_trj = np.tile(x0[idx], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    su = np.full(m, 0.1)
    _, du = orc.device_gaussian_samples(T, N, n, m, np.zeros(n), su, seed, it, dtype=np.float32)
    dm = sys_d.dm()
    for mode, tol in ((SMOOTH_ZERO_ORDER_B, 2e-6), (SMOOTH_FIRST_ORDER, 2e-2 if system == "planar_hand" else 5.0 / N)):
        a = dm.smooth_rng(mode, xd, ud, N, None, su, seed, it)
        b = dm.smooth(mode, xd, ud, None, dev.to_dev(du, dev.F32))
        for k in ("At", "Bt", "ct"):
            np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=0, atol=tol, err_msg="%s mode %d" % (k, mode))


def test_planar_hand_exact_contact_solver_vs_oracle(amd):
    """contact_solver="exact" (IRS_MODEL_PLANAR_HAND_EXACT): the device's dual active-set solve of the step QP
    == the oracle's (`pgs_iters = 0`) in f64 (dynamics, active-set Jacobian), and through the f32 sample
    pass in both estimators, on the heavily loaded samples where 50 sweeps are off by up to 5e-2."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
    T, N = 6, 2000
    sys_o = orc.PlanarHandOracle(0.1, pgs_iters=0)
    sys_d = amd.PlanarHandDynamics(0.1, contact_solver="exact")
    pgs_d = amd.PlanarHandDynamics(0.1, contact_solver="pgs")
    x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
    u_trj = np.tile(x0[HAND_IDX], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    np.testing.assert_allclose(sys_d.dm().rollout_cost(dev.to_dev(x0), dev.to_dev(u_trj), dev.to_dev(np.eye(7)),
                                                       dev.to_dev(np.eye(4)), dev.to_dev(np.zeros((T + 1, 7))))[0]
                               .cpu().numpy(), x_trj, rtol=0, atol=1e-9)
    rng = np.random.default_rng(31)
    X = np.tile(x_trj[-1], (512, 1)) + 0.01 * rng.normal(size=(512, 7))
    U = u_trj[-1] + 0.3 * rng.normal(size=(512, 4))
    want = sys_o.dynamics_batch(X, U)
    np.testing.assert_allclose(sys_d.dynamics_batch(X, U), want, rtol=0, atol=1e-8)
    assert np.abs(pgs_d.dynamics_batch(X, U) - want).max() > 1e-4          # the sweeps are not there yet
    np.testing.assert_allclose(sys_d.jacobian_xu_batch(X, U), sys_o.jacobian_xu_batch(X, U), rtol=0, atol=1e-6)
    du = (0.3 * rng.normal(size=(T, N, 4))).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    assert int(o["info"].abs().sum().item()) == 0
    Ao, Bo, co = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    np.testing.assert_allclose(o["At"].cpu().numpy(), Ao, rtol=0, atol=0)
    np.testing.assert_allclose(o["Bt"].cpu().numpy(), Bo, **FP32_TOL)      # f32 lanes vs the f64 oracle
    np.testing.assert_allclose(o["ct"].cpu().numpy(), co, **FP32_TOL)
    o1 = dm.smooth(SMOOTH_FIRST_ORDER, xd, ud, None, dev.to_dev(du, dev.F32))
    _, B1, c1 = orc.first_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    np.testing.assert_allclose(o1["Bt"].cpu().numpy(), B1, **FP32_TOL)     # borderline rows: O(1)/N each
    np.testing.assert_allclose(o1["ct"].cpu().numpy(), c1, **FP32_TOL)
    with pytest.raises(ValueError):
        amd.PlanarHandDynamics(0.1, contact_solver="nope")


def test_planar_hand_exact_solver_random_states(amd):
    """The device's exact dual active-set solve on 4000 random states and commands -- separated, touching,
    deeply penetrating, up to 7 rows active -- == the oracle's (which is KKT-certified on the same kind of
    states, tests/test_oracle_golden.py); the active-set Jacobian too."""
    rng = np.random.default_rng(321)
    N = 4000
    obj = np.stack([rng.uniform(-0.3, 0.3, N), rng.uniform(0.1, 0.7, N), rng.uniform(-1, 1, N)], 1)
    left = np.stack([rng.uniform(-2.2, -0.2, N), rng.uniform(-1.5, 0.5, N)], 1)
    right = np.stack([rng.uniform(0.2, 2.2, N), rng.uniform(-0.5, 1.5, N)], 1)
    X = np.zeros((N, 7))
    X[:, HAND.PERM] = np.hstack([obj, left, right])
    U = X[:, HAND_IDX] + rng.normal(0, 0.3, (N, 4))
    sys_o = orc.PlanarHandOracle(0.1, pgs_iters=0)
    sys_d = amd.PlanarHandDynamics(0.1, contact_solver="exact")
    want = sys_o.dynamics_batch(X, U)
    got = sys_d.dynamics_batch(X, U)
    assert np.isfinite(got).all()
    step = np.abs(want - X).max(1, keepdims=True).clip(1e-3)
    assert (np.abs(got - want) <= 1e-7 * step + 1e-9).all()
    Jd, Jo = sys_d.jacobian_xu_batch(X[:500], U[:500]), sys_o.jacobian_xu_batch(X[:500], U[:500])
    assert np.abs(Jd - Jo).max() < 1e-5


def test_box_pivot_exact_solver_random_states(amd):
    """The 12-row box-pivoting step QP solved exactly on the device (IRS_MODEL_BOX_PIVOT_EXACT) == the
    oracle's exact solve on 3000 random states incl. penetration, and through the f32 sample pass."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    rng = np.random.default_rng(77)
    N = 3000
    BOX = orc.BoxPivotOracle
    box = np.stack([rng.uniform(-0.5, 0.5, N), rng.uniform(0.45, 0.9, N), rng.uniform(-1, 1, N)], 1)
    hand = np.stack([rng.uniform(-1.2, 1.2, N), rng.uniform(0.05, 1.5, N)], 1)
    X = np.zeros((N, 5))
    X[:, BOX.PERM] = np.hstack([box, hand])
    sys_o = BOX(0.1, pgs_iters=0)
    sys_d = amd.BoxPivotingDynamics(0.1, contact_solver="exact")
    U = X[:, sys_o.indices_u_into_x] + rng.normal(0, 0.2, (N, 2))
    want, got = sys_o.dynamics_batch(X, U), sys_d.dynamics_batch(X, U)
    step = np.abs(want - X).max(1, keepdims=True).clip(1e-3)
    assert np.isfinite(got).all() and (np.abs(got - want) <= 1e-7 * step + 1e-9).all()
    T, Ns = 4, 1500
    x0 = BOX.pack([0.0, 0.5, 0.0], [-0.6, 0.3])
    u_trj = np.tile(x0[sys_o.indices_u_into_x], (T, 1)) + np.linspace(0, 1, T)[:, None] * np.array([0.1, 0.0])
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (0.05 * rng.normal(size=(T, Ns, 2))).astype(np.float32)
    o = sys_d.dm().smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    Ao, Bo, co = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    np.testing.assert_allclose(o["Bt"].cpu().numpy(), Bo, **FP32_TOL)     # Kp = 5e4: f32 steps of a stiff hand
    np.testing.assert_allclose(o["ct"].cpu().numpy(), co, **FP32_TOL)
    # box pushing (2 rows): the exact functor exists too and is the default
    assert amd.BoxPushingDynamics(0.1).contact_solver == "exact"
    assert amd.BoxPushingDynamics(0.1).device_model != amd.BoxPushingDynamics(0.1, contact_solver="pgs").device_model


def test_capture_step_replays_the_two_launch_smoothing_step(amd):
    """irs_mpc_amd.distributed.capture_step (what bench.py times with several GPUs): the accumulate and
    solve launches of the sharded smoothing step captured into one HIP graph; replays reproduce the eagerly
    issued step bit for bit, also after the inputs changed in place."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    from irs_mpc_amd.distributed import capture_step
    T, N = 5, 700
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(orc.rollout(sys_o, x0, u_trj)), dev.to_dev(u_trj)
    g = torch.Generator(device="cuda").manual_seed(5)
    du = 0.1 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
    plan = dev.SmoothPlan(dm, SMOOTH_ZERO_ORDER_B, xd, ud, du=du, fuse=False, n_total=N)
    out = {}

    def step():
        plan.run()
        out["o"] = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, plan.sums, out=out.get("o"), workspace=plan.ws)

    step()
    torch.cuda.synchronize()
    ref = [t.clone() for t in out["o"][:3]]
    replay = capture_step(step)
    for t in out["o"][:3]:
        t.zero_()
    replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, out["o"][:3]))
    du.mul_(0.5)                                    # same buffers, new content: the graph reads them afresh
    replay()
    torch.cuda.synchronize()
    got = [t.clone() for t in out["o"][:3]]
    step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, out["o"][:3]))
    assert not torch.equal(got[1], ref[1])


def test_collective_step_inside_the_library(amd):
    """The multi-GPU smoothing step issued by the library itself (csrc/collective.hip): accumulate -> RCCL
    all-reduce -> solve on one stream, eagerly and as ONE replayed HIP graph, with a communicator the library
    owns (irs_comm_*; a 1-rank communicator on this one-GPU box: the all-reduce is the identity, the launch
    sequence, the capture and the replay are the real ones).  Bit for bit equal to the two-stage path."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    from irs_mpc_amd.distributed import CollectiveStep, DirectComm
    T, N = 6, 3000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (0.1 * np.random.default_rng(3).normal(size=(T, N, 4))).astype(np.float32)
    dm = sys_d.dm()
    xd, ud, dud = dev.to_dev(x_trj), dev.to_dev(u_trj), dev.to_dev(du, dev.F32)
    sums = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, xd, ud, None, dud).clone()
    ws = dm._workspace(SMOOTH_ZERO_ORDER_B, T, N, xd.device)
    A0, B0, c0, i0 = [t.clone() for t in dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, sums, workspace=ws)]
    comm = DirectComm()
    assert comm.world == 1 and comm.handle.value
    plan = dev.SmoothPlan(dm, SMOOTH_ZERO_ORDER_B, xd, ud, dx=None, du=dud, fuse=True, n_total=N)
    step = CollectiveStep(plan, comm)
    for label in ("eager", "graph"):
        for k in ("At", "Bt", "ct"):
            plan.out[k].zero_()
        if label == "graph":
            step.capture()
        o = step.run()
        torch.cuda.synchronize()
        assert torch.equal(o["Bt"], B0) and torch.equal(o["ct"], c0) and torch.equal(o["At"], A0), label
        assert torch.equal(plan.sums, sums), label
        assert int(o["info"].abs().sum().item()) == 0
    # the bare all-reduce entry point: identity on one rank
    s2 = sums.clone()
    comm.all_reduce_sums(s2)
    torch.cuda.synchronize()
    assert torch.equal(s2, sums)
    step.destroy()
    comm.destroy()


@pytest.mark.timeout(600)
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py's N > 1 code path -- sharded samples, the all-reduce of the (T,P) sums inside every step, the
    max-over-ranks timing, rank 0's single JSON line -- exercised with TWO ranks on this one GPU (gloo
    collectives: RCCL refuses two ranks on one device), launched exactly as the driver launches N > 1."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "3", "--rehearse-one-gpu", "--T", "12", "--N", "1500"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["N_total"] == 3000
    assert out["value"] > 0 and out["ilqr_iters_per_s"] > 0
    assert "all-reduce" in out["config"]["step"]


def test_quasistatic_active_set_descent_reports_a_bad_hessian(amd):
    """The matrix-core active-set kernel (compiled with -fno-honor-nans) still reports, through info[0], a control
    Hessian that is not positive definite -- an indefinite cost -- and a non-finite linearisation."""
    from irs_mpc_amd import device as dev
    T = 8
    sys_d, sys_o, x0, u_trj, x_trj, _, (At, Bt, ct), (Q, Qd, R, xd) = _hand_problem(amd, T, 400, 21)
    idx = sys_o.indices_u_into_x
    ub = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    rows = orc.quasistatic_bounds(x_trj, idx, None, ub, None)
    rows_d = [dev.to_dev(r) if np.isfinite(r).any() else None for r in rows]
    dm = sys_d.dm()
    good = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], *rows_d, solver=0,
                                      max_iter=2000, eps=1e-10)
    assert good["info"].cpu().numpy()[0] == 0
    bad = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, -50.0 * Q, -50.0 * Qd, 1e-6 * R, xd, x0)],
                                     *rows_d, solver=0, max_iter=50, eps=1e-10)
    assert bad["info"].cpu().numpy()[0] != 0
    Bn = Bt.copy()
    Bn[T // 2, 0, 0] = np.nan
    nan = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bn, ct, Q, Qd, R, xd, x0)], *rows_d, solver=0,
                                     max_iter=50, eps=1e-10)
    i3 = nan["info"].cpu().numpy()
    assert i3[0] != 0 or i3[2] != 0 or not np.isfinite(float(nan["cost"].item())), i3


def test_peer_exchange_two_ranks_on_one_gpu():
    """The all-reduce WITHOUT a collective library (csrc/collective.hip, irs_peer_*): two processes on this one GPU map
    each other's exchange regions by IPC handle and run 300 exchange launches back to back (slot reuse) -- every rank
    gets the rank-ordered total, bit for bit the closed form, and no wait times out.  (What this cannot show is the
    memory model across two L2s: that needs a multi-GPU node; include/irs_hip.h says so.)"""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IRS_PEER_TIMEOUT_MS="1500")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "tests", "helpers", "peer_worker.py"),
           "--missing-peer"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "PEER_EXCHANGE_OK world=2" in r.stdout
    # ... and a peer that never arrives ends the wait after IRS_PEER_TIMEOUT_MS with poisoned statistics (no hang)
    assert "PEER_TIMEOUT_OK" in r.stdout


def test_bench_two_ranks_peer_exchange_on_one_gpu():
    """bench.py --collective peer: the whole N > 1 step (accumulate -> peer exchange -> solve, captured into a HIP
    graph inside the library; the iLQR loop's all-reduce too) with two ranks on this one GPU."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IRS_PEER_TIMEOUT_MS="5000")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "3", "--rehearse-one-gpu", "--collective", "peer",
           "--T", "12", "--N", "1500"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["N_total"] == 3000 and out["value"] > 0
    assert out["config"]["collective"].startswith("peer") and "one HIP graph" in out["config"]["step"]


def test_peer_exchange_one_rank_step_equals_the_two_stage_path(amd):
    """One rank: accumulate -> exchange launch -> solve (eager, and replayed from the library's HIP graph) == the
    two-stage path, bit for bit (as test_collective_step_inside_the_library for the RCCL step); the counters count."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd.distributed import CollectiveStep, PeerExchange
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    T, N = 6, 3000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (0.1 * np.random.default_rng(3).normal(size=(T, N, 4))).astype(np.float32)
    dm = sys_d.dm()
    xd, ud, dud = dev.to_dev(x_trj), dev.to_dev(u_trj), dev.to_dev(du, dev.F32)
    sums = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, xd, ud, None, dud).clone()
    ws = dm._workspace(SMOOTH_ZERO_ORDER_B, T, N, xd.device)
    A0, B0, c0, i0 = [t.clone() for t in dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, sums, workspace=ws)]
    plan = dev.SmoothPlan(dm, SMOOTH_ZERO_ORDER_B, xd, ud, dx=None, du=dud, fuse=True, n_total=N)
    px = PeerExchange(plan.sums.numel())
    assert px.world == 1 and px.handle.value
    step = CollectiveStep(plan, px)
    for label in ("eager", "graph"):
        for k in ("At", "Bt", "ct"):
            plan.out[k].zero_()
        if label == "graph":
            step.capture()
        o = step.run()
        torch.cuda.synchronize()
        assert torch.equal(o["Bt"], B0) and torch.equal(o["ct"], c0) and torch.equal(o["At"], A0), label
        assert torch.equal(plan.sums, sums), label
        assert int(o["info"].abs().sum().item()) == 0
    s2 = sums.clone()
    px.all_reduce_sums(s2)
    torch.cuda.synchronize()
    assert torch.equal(s2, sums)
    launches, timeouts = px.status()
    assert timeouts == 0 and launches == 5, (launches, timeouts)      # eager, 2 warm-ups of the capture, replay, bare
    step.destroy()
    px.destroy()


def test_planar_hand_descent_runs(amd):
    """smooth -> Riccati -> closed-loop rollout through the contact functor in f64 == oracle."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    T, N = 10, 1000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    rng = np.random.default_rng(13)
    du = (rng.normal(size=(T, N, 4)) * 0.1).astype(np.float32)
    dm = sys_d.dm()
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    At, Bt, ct = o["At"], o["Bt"], o["ct"]
    Q = np.diag(HAND_Q)
    R = 5.0 * np.eye(4)
    xd = np.tile(x0 + HAND.pack([0.0, 0.0, 0.3], [0, 0], [0, 0]), (T + 1, 1))
    out = dm.tvlqr_descent(At, Bt, ct, dev.to_dev(Q), dev.to_dev(100 * Q), dev.to_dev(R), dev.to_dev(xd),
                           dev.to_dev(x0))
    K, k, x_new, cost = out["K"], out["k"], out["x_new"], out["cost"]
    assert int(out["info"].item()) == 0
    A_, B_, c_ = At.cpu().numpy(), Bt.cpu().numpy(), ct.cpu().numpy()
    Ko, ko = orc.tvlqr_riccati(A_, B_, c_, Q, 100 * Q, R, xd)
    np.testing.assert_allclose(K.cpu().numpy(), Ko, rtol=1e-7, atol=1e-9)
    xo, uo = orc.closed_loop_rollout(sys_o, K.cpu().numpy(), k.cpu().numpy(), x0)
    np.testing.assert_allclose(x_new.cpu().numpy(), xo, rtol=0, atol=1e-8)
    np.testing.assert_allclose(float(cost.item()), orc.evaluate_cost(xo, uo, xd, Q, R), rtol=1e-9)


def _hand_problem(amd, T, N, seed):
    sys_d, sys_o, x0, _ = _hand_setup(amd, T)
    u_trj = np.tile(x0[HAND_IDX], (T, 1))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (np.random.default_rng(seed).normal(size=(T, N, 4)) * 0.1).astype(np.float32)
    At, Bt, ct = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    Q, Qd, R = np.diag(HAND_Q), np.diag(100 * HAND_Q), 5.0 * np.eye(4)
    xd = np.tile(x0 + HAND_GOAL, (T + 1, 1))
    return sys_d, sys_o, x0, u_trj, x_trj, du, (At, Bt, ct), (Q, Qd, R, xd)


@pytest.mark.parametrize("bounds", ["abs", "rel", "abs+rel+x"])
def test_quasistatic_box_descent_vs_oracle(amd, bounds, as_solver):
    """IrsLqrQuasistatic.local_descent after get_TV_matrices (irs_lqr_quasistatic.py:286-345): du cost,
    per-time trust-region bounds, T re-solved tail QPs, true (contact) dynamics in the loop."""
    from irs_mpc_amd import device as dev
    T = 8
    sys_d, sys_o, x0, u_trj, x_trj, _, (At, Bt, ct), (Q, Qd, R, xd) = _hand_problem(amd, T, 400, 21)
    idx = sys_o.indices_u_into_x
    ub = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]) if "abs" in bounds else None     # run_planar_hand.py:138
    rb = np.array([-np.ones(4) * 0.03, np.ones(4) * 0.03]) if "rel" in bounds else None
    xb = np.array([-np.ones(7) * 0.04, np.ones(7) * 0.04]) if "x" in bounds else None
    rows = orc.quasistatic_bounds(x_trj, idx, xb, ub, rb)
    xo, uo, iters = orc.local_descent_quasistatic(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, *rows, rho=100.0,
                                                  max_iter=40000, eps=1e-10, relax=1.6)
    assert max(iters) < 40000
    dm = sys_d.dm()
    rows_d = [dev.to_dev(r) if np.isfinite(r).any() else None for r in rows]
    o = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], *rows_d,
                                   solver=1, rho=100.0, relax=1.6, max_iter=40000, eps=1e-10)
    info = o["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    if bounds in ("abs", "rel"):
        # the exact active-set solver on the same QPs: device == its oracle twin == the ADMM answer
        o2 = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], *rows_d,
                                        solver=as_solver, max_iter=2000, eps=1e-10)
        i2 = o2["info"].cpu().numpy()
        assert i2[0] == 0 and i2[2] == 0, i2
        lo, hi = (rows[2], rows[3]) if bounds == "abs" else (rows[4], rows[5])
        xa, ua, _ = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, bounds)
        np.testing.assert_allclose(o2["u_new"].cpu().numpy(), ua, rtol=0, atol=1e-9)
        np.testing.assert_allclose(o2["x_new"].cpu().numpy(), xa, rtol=0, atol=1e-9)
        np.testing.assert_allclose(ua, uo, rtol=0, atol=2e-7)
        np.testing.assert_allclose(float(o2["cost"].item()), orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx),
                                   rtol=1e-10)
    np.testing.assert_allclose(o["u_new"].cpu().numpy(), uo, rtol=0, atol=2e-7)
    np.testing.assert_allclose(o["x_new"].cpu().numpy(), xo, rtol=0, atol=2e-7)
    np.testing.assert_allclose(float(o["cost"].item()), orc.eval_cost_quasistatic(xo, uo, xd, Q, Qd, R, idx),
                               rtol=1e-7)
    # the bounds bind (otherwise this would not test the QP)
    un = o["u_new"].cpu().numpy()
    if "abs" in bounds:
        assert np.isclose(np.abs(un - x_trj[:-1, idx]).max(), 0.05, atol=1e-7)
    else:
        # every tail's first du is measured from the REALISED actuated position (tv_lqr.py:99-100)
        d = un - o["x_new"].cpu().numpy()[:-1, idx]
        assert np.isclose(np.abs(d).max(), 0.03, atol=1e-7)


def test_irs_lqr_quasistatic_host_twin(amd):
    """IrsLqrQuasistatic end to end (irs_lqr_quasistatic.py:44-390): identical np.random seeds give the
    oracle's first descent; the cost bookkeeping matches eval_cost; iterate() lowers the cost."""
    T, N = 8, 300
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.IrsLqrQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]),
              "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict = q_dict
    p.Qd_dict = {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)          # run_planar_hand.py:142-143
    p.std_u_initial = np.ones(4) * 0.1
    p.num_samples = N
    p.publish_every_iteration = False
    p.qp_rho, p.qp_max_iter, p.qp_eps = 100.0, 40000, 1e-10
    sol = amd.IrsLqrQuasistatic(sys_d, p)
    sol.verbose = False
    idx = sys_o.indices_u_into_x
    np.testing.assert_allclose(sum(sol.eval_cost(sol.x_trj, sol.u_trj)),
                               orc.eval_cost_quasistatic(sol.x_trj, sol.u_trj, xd, Q, Qd, R, idx), rtol=1e-12)
    # first descent from identical seeds
    np.random.seed(7)
    xn, un = sol.local_descent(sol.x_trj, sol.u_trj)
    np.random.seed(7)
    du = np.stack([np.random.normal(0, p.std_u_initial, size=[N, 4]) for _ in range(T)]).astype(np.float32)
    At, Bt, ct = orc.zero_order_B_decoupled(sys_o, sol.x_trj, sol.u_trj, du.astype(np.float64))
    rows = orc.quasistatic_bounds(sol.x_trj, idx, None, p.u_bounds_abs, None)
    xo, uo, _ = orc.local_descent_quasistatic(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, *rows, rho=100.0,
                                              max_iter=40000, eps=1e-10, relax=1.6)
    # B is fitted from f32 one-step evaluations on the device: the QP data differ at the 1e-4 level
    np.testing.assert_allclose(un, uo, rtol=0, atol=5e-4)
    np.testing.assert_allclose(xn, xo, rtol=0, atol=5e-4)
    c0 = sol.cost
    sol.iterate(3)
    assert len(sol.cost_all_list) == 5 and sol.cost_best < c0
    assert sol.x_trj_best.shape == (T + 1, 7) and sol.current_iter == 4


@pytest.mark.parametrize("kind", ["abs", "rel", "none"])
def test_quasistatic_active_set_full_horizon(amd, kind):
    """BASELINE's planar_hand horizon (T=50): the active-set descent == its oracle twin; the "rel"
    case needs the primal safeguard on some tails (the primal-dual iteration cycles there)."""
    from irs_mpc_amd import device as dev
    T = 50
    sys_d, sys_o, x0, u_trj, x_trj, _, (At, Bt, ct), (Q, Qd, R, xd) = _hand_problem(amd, T, 200, 33)
    idx = sys_o.indices_u_into_x
    ub = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]) if kind == "abs" else None
    rb = np.array([-np.ones(4) * 0.03, np.ones(4) * 0.03]) if kind == "rel" else None
    rows = orc.quasistatic_bounds(x_trj, idx, None, ub, rb)
    lo, hi = (rows[4], rows[5]) if kind == "rel" else (rows[2], rows[3])
    xa, ua, stats = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi,
                                                     "rel" if kind == "rel" else "abs")
    assert all(st[1] >= 0 for st in stats)
    dm = sys_d.dm()
    assert dm.quasistatic_descent_supported(T, 2)
    rows_d = [dev.to_dev(r) if np.isfinite(r).any() else None for r in rows]
    o = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], *rows_d,
                                   solver=0, max_iter=2000, eps=1e-10)
    info = o["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    np.testing.assert_allclose(o["u_new"].cpu().numpy(), ua, rtol=0, atol=1e-8)
    np.testing.assert_allclose(o["x_new"].cpu().numpy(), xa, rtol=0, atol=1e-8)
    np.testing.assert_allclose(float(o["cost"].item()), orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx),
                               rtol=1e-9)
    if kind == "none":
        # no bounds: the tail re-solves collapse to the du-cost Riccati policy of the augmented LQR
        Ab, Bb, cb, Qb, Qdb, xdb = orc.quasistatic_augment(At, Bt, ct, Q, Qd, xd)
        K, k = orc.tvlqr_riccati(Ab, Bb, cb, Qb, Qdb, R, xdb, alpha_R=1.0)
        x = x0.copy()
        for t in range(T):
            z = np.concatenate([x, x[idx]])
            u = z[7:] + K[t] @ z + k[t]          # z = [x; u_prev]
            np.testing.assert_allclose(ua[t], u, rtol=0, atol=1e-9)
            x = sys_o.dynamics(x, u)


def test_cem_quasistatic_vs_oracle(amd):
    """CrossEntropyMethodQuasistatic (irs_lqr/cem_quasistatic.py:39-258) on the planar hand: identical
    np.random seeds -> the oracle's candidates, costs, elite refit and trajectory."""
    T, B, n_elite = 10, 24, 6
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.CemQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]),
              "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict, p.Qd_dict = q_dict, {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.xd_trj, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.n_elite, p.batch_size, p.initial_std = n_elite, B, 0.05 * np.ones(4)
    p.publish_every_iteration = False
    sol = amd.CrossEntropyMethodQuasistatic(sys_d, p)
    sol.verbose = False
    np.random.seed(3)
    xn, un = sol.local_descent(sol.x_trj, sol.u_trj)
    np.random.seed(3)
    xo, uo, so, costs, _ = orc.cem_quasistatic_local_descent(sys_o, x0, u_trj, np.tile(p.initial_std, (T, 1)), xd,
                                                             Q, Qd, R, n_elite, B)
    np.testing.assert_allclose(sol.cost_array.cpu().numpy(), costs, rtol=1e-10)
    np.testing.assert_allclose(un, uo, rtol=0, atol=1e-12)
    np.testing.assert_allclose(sol.std_trj, so, rtol=0, atol=1e-12)
    np.testing.assert_allclose(xn, xo, rtol=0, atol=1e-9)
    c0 = sol.cost
    sol.iterate(2)
    assert len(sol.cost_all_list) == 4 and sol.cost_best <= c0 and sol.current_iter == 3
    # the analytic models are not position controlled
    from irs_mpc_amd import device as dev
    with pytest.raises(Exception):
        amd.PendulumDynamics(0.05).dm().cem_rollout_costs_quasistatic(
            dev.to_dev(np.zeros((4, 3, 1))), dev.to_dev(np.zeros(2)), dev.to_dev(np.eye(2)), dev.to_dev(np.eye(2)),
            dev.to_dev(np.eye(1)), dev.to_dev(np.zeros((4, 2))))


@pytest.mark.parametrize("argv", [["planar_hand", "irs_lqr", "--iters", "3", "--T", "20", "--N", "500"],
                                  ["planar_hand", "irs_lqr", "--iters", "2", "--T", "12", "--N", "300", "--bounds", "rel",
                                   "--device-rng"],
                                  ["planar_hand", "irs_lqr", "--iters", "2", "--T", "12", "--N", "300", "--gradient-mode",
                                   "zero_order_B"],
                                  ["planar_hand", "irs_lqr", "--iters", "2", "--T", "12", "--N", "300", "--gradient-mode",
                                   "exact"],
                                  ["planar_hand_spin", "irs_lqr", "--iters", "3", "--T", "20", "--N", "500"],
                                  ["planar_hand", "cem", "--iters", "3", "--T", "12", "--N", "60"],
                                  ["box_pivoting", "irs_lqr", "--iters", "3", "--T", "40", "--N", "500"],
                                  ["box_pivoting", "cem", "--iters", "3", "--T", "40", "--N", "80"],
                                  ["box_pushing", "irs_lqr", "--iters", "3", "--T", "40", "--N", "500"]])
def test_quasistatic_example_runner(amd, argv, monkeypatch, capsys):
    """Twins of examples/planar_hand/run_planar_hand{,_cem}.py and examples/box_pivoting/
    run_box_pivoting{,_cem}.py run end to end and descend."""
    import examples.run_quasistatic as run
    monkeypatch.setattr("sys.argv", ["run_quasistatic.py", "--quiet"] + argv)
    run.main()
    out = capsys.readouterr().out
    hist = [float(v) for v in out.split("cost history:")[1].split()]
    assert len(hist) == int(argv[3]) + 2 and all(np.isfinite(hist))
    if argv[:2] != ["box_pivoting", "cem"]:     # a 3-iteration CEM with 80 candidates does not find the pivot
        assert min(hist[1:]) < hist[0]


def test_quasistatic_active_set_warm_start_across_iterations(amd, as_solver):
    """irs_quasistatic_box_descent_ws: the active set of the first tail handed from one descent to the
    next.  Two consecutive iterations of the planar-hand problem (the second linearised around the first
    one's result): warm and cold start give the same trajectory (the QP's solution does not depend on
    the start), device == oracle twin for both, and the returned set is what the oracle's first tail
    converged to."""
    from irs_mpc_amd import device as dev
    T = 30
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    idx = sys_o.indices_u_into_x
    Q, Qd, R = np.diag(HAND_Q), np.diag(100 * HAND_Q), 5.0 * np.eye(4)
    xd = np.tile(x0 + HAND.pack([0.3, -0.1, 0.5], [0, 0], [0, 0]), (T + 1, 1))
    ub = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    rng = np.random.default_rng(21)
    dm = sys_d.dm()
    act_d = dev.to_dev(np.zeros((T, 4)))
    act_o = np.zeros((T, 4))
    x_trj = orc.rollout(sys_o, x0, u_trj)
    for it in (1, 2):
        du = (0.3 / it ** 0.8) * rng.normal(size=(T, 400, 4))
        At, Bt, ct = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du)
        rows = orc.quasistatic_bounds(x_trj, idx, None, ub, None)
        args = [dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)]
        cold = dm.quasistatic_box_descent(*args, u_lo=dev.to_dev(rows[2]), u_hi=dev.to_dev(rows[3]), solver=as_solver,
                                          max_iter=2000, eps=1e-10)
        u_cold = cold["u_new"].cpu().numpy().copy()
        warm = dm.quasistatic_box_descent(*args, u_lo=dev.to_dev(rows[2]), u_hi=dev.to_dev(rows[3]), solver=as_solver,
                                          max_iter=2000, eps=1e-10, act=act_d)
        info = warm["info"].cpu().numpy()
        assert info[0] == 0 and info[2] == 0, info
        np.testing.assert_allclose(warm["u_new"].cpu().numpy(), u_cold, rtol=0, atol=1e-8)
        xo, uo, _ = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, rows[2], rows[3], "abs",
                                                     act_io=act_o)
        np.testing.assert_allclose(warm["u_new"].cpu().numpy(), uo, rtol=0, atol=1e-8)
        np.testing.assert_array_equal(act_d.cpu().numpy(), act_o)
        assert np.abs(act_o).sum() > 10                  # the trust region binds
        x_trj, u_trj = xo, uo
    # garbage in the warm start (and a pin at an infinite bound) is harmless
    junk = dev.to_dev(rng.integers(-1, 2, size=(T, 4)).astype(np.float64) * 3.0)
    again = dm.quasistatic_box_descent(*args, u_lo=dev.to_dev(rows[2]), u_hi=dev.to_dev(rows[3]), solver=as_solver,
                                       max_iter=2000, eps=1e-10, act=junk)
    np.testing.assert_allclose(again["u_new"].cpu().numpy(), u_cold, rtol=0, atol=1e-8)
    free = dm.quasistatic_box_descent(*args, solver=as_solver, max_iter=2000, eps=1e-10, act=junk.clone())
    none = dm.quasistatic_box_descent(*args, solver=as_solver, max_iter=2000, eps=1e-10)
    np.testing.assert_allclose(free["u_new"].cpu().numpy(), none["u_new"].cpu().numpy(), rtol=0, atol=1e-9)


@pytest.mark.parametrize("T,kind", [(64, "abs"), (120, "abs"), (120, "rel")])
def test_quasistatic_descent_long_horizon(amd, T, kind):
    """Horizons beyond the LDS-resident size (planar hand: 52 steps on lanes, 55 on tiles): the matrix-core
    solver keeps its per-step records in an HBM workspace instead -- the reference has no horizon limit
    (irs_lqr_quasistatic.py:325-345).  Device == the oracle twin; the host twin picks it by itself."""
    from irs_mpc_amd import device as dev
    sys_d, sys_o, x0, u_trj, x_trj, _, (At, Bt, ct), (Q, Qd, R, xd) = _hand_problem(amd, T, 150, 40 + T)
    dm = sys_d.dm()
    assert not dm.quasistatic_descent_supported(T, 2) and dm.quasistatic_descent_supported(T, 3)
    assert dm.lib.irs_quasistatic_descent_workspace_bytes(dm.model_id, T, 3) > 0
    idx = sys_o.indices_u_into_x
    if kind == "abs":
        rows = orc.quasistatic_bounds(x_trj, idx, None, np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05]), None)
        lo, hi = rows[2], rows[3]
        b = dict(u_lo=dev.to_dev(lo), u_hi=dev.to_dev(hi))
    else:
        lo, hi = np.full((T, 4), -0.03), np.full((T, 4), 0.03)
        b = dict(du_lo=dev.to_dev(lo), du_hi=dev.to_dev(hi))
    xa, ua, _ = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, kind)
    o = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], solver=0,
                                   max_iter=2000, eps=1e-10, **b)
    info = o["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    np.testing.assert_allclose(o["u_new"].cpu().numpy(), ua, rtol=0, atol=1e-8)
    np.testing.assert_allclose(o["x_new"].cpu().numpy(), xa, rtol=0, atol=1e-7)
    np.testing.assert_allclose(float(o["cost"].item()), orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx), rtol=1e-8)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_quasistatic_active_set_random_problems(amd, seed, as_solver):
    """The active-set descent on randomised planar-hand problems (nominal trajectory, goal, bound
    widths, cost weights): device == oracle twin, both kinds, including tails that need the primal
    safeguard."""
    from irs_mpc_amd import device as dev
    rng = np.random.default_rng(100 + seed)
    T = int(rng.integers(12, 30))
    sys_d, sys_o, x0, _ = _hand_setup(amd, T)
    idx = sys_o.indices_u_into_x
    u_trj = np.tile(x0[HAND_IDX], (T, 1)) + 0.03 * rng.normal(size=(T, 4)).cumsum(axis=0) / np.sqrt(T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = 0.1 * rng.normal(size=(T, 300, 4))
    At, Bt, ct = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du)
    q = HAND_Q * rng.uniform(0.3, 3.0, size=7)
    Q, Qd, R = np.diag(q), np.diag(rng.uniform(10, 200) * q), np.diag(rng.uniform(0.5, 10, size=4))
    xd = np.tile(x0 + HAND.pack(np.concatenate([rng.uniform(-0.3, 0.3, 2), rng.uniform(-0.6, 0.6, 1)]), [0, 0], [0, 0]),
                 (T + 1, 1))
    w = rng.uniform(0.01, 0.08)
    dm = sys_d.dm()
    for kind in ("abs", "rel"):
        ub = np.array([-np.ones(4) * w, np.ones(4) * w]) if kind == "abs" else None
        rb = np.array([-np.ones(4) * w, np.ones(4) * w]) if kind == "rel" else None
        rows = orc.quasistatic_bounds(x_trj, idx, None, ub, rb)
        lo, hi = (rows[2], rows[3]) if kind == "abs" else (rows[4], rows[5])
        xa, ua, stats = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, lo, hi, kind)
        assert all(st[1] >= 0 for st in stats)
        kw = dict(u_lo=dev.to_dev(lo), u_hi=dev.to_dev(hi)) if kind == "abs" else \
            dict(du_lo=dev.to_dev(lo), du_hi=dev.to_dev(hi))
        o = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)], solver=as_solver,
                                       max_iter=2000, eps=1e-10, **kw)
        info = o["info"].cpu().numpy()
        assert info[0] == 0 and info[2] == 0, (kind, info)
        np.testing.assert_allclose(o["u_new"].cpu().numpy(), ua, rtol=0, atol=1e-8, err_msg=kind)
        np.testing.assert_allclose(o["x_new"].cpu().numpy(), xa, rtol=0, atol=1e-8, err_msg=kind)


# ---------------------------------------------------------------- box pivoting (BASELINE configs[4] model, unpinned)
def _box_setup(amd, T):
    sys_d, sys_o = amd.BoxPivotingDynamics(0.1), orc.BoxPivotOracle(0.1)
    x0 = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.5, 0.5])            # run_box_pivoting.py:31-43
    x0 = sys_o.dynamics(x0, np.array([-0.5, 0.5]))                        # resolve the initial overlap
    u_trj = np.stack([np.array([-0.5 + 0.4 * (t + 1) / T, 0.5]) for t in range(T)])   # FirstOrderHold push
    return sys_d, sys_o, x0, u_trj


def test_box_pivot_dynamics_vs_oracle(amd):
    sys_d, sys_o, x0, _ = _box_setup(amd, 1)
    assert list(sys_d.get_u_indices_into_x()) == list(sys_o.indices_u_into_x) == [0, 2]
    rng = np.random.default_rng(5)
    # box poses on and above the ground, hand inside, on and outside the box outline
    X = np.stack([orc.BoxPivotOracle.pack([rng.uniform(-0.3, 0.3), rng.uniform(0.5, 0.8), rng.uniform(-0.6, 0.6)],
                                          [rng.uniform(-1.0, 1.0), rng.uniform(0.05, 1.2)]) for _ in range(512)])
    U = X[:, [0, 2]] + 0.05 * rng.normal(size=(512, 2))
    np.testing.assert_allclose(sys_d.dynamics_batch(X, U), sys_o.dynamics_batch(X, U), rtol=0, atol=1e-9)
    # the hand centre exactly on a face (the reference's initial condition) takes the face normal
    xb = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.5, 0.5])
    np.testing.assert_allclose(sys_d.dynamics(xb, xb[[0, 2]]), sys_o.dynamics(xb, xb[[0, 2]]), rtol=0, atol=1e-9)


def test_box_pivot_quasistatic_iteration_vs_oracle(amd):
    """One IrsLqrQuasistatic descent on the box (run_box_pivoting.py:95-131: Q/R dicts, u_bounds_rel =
    +-0.15 h): zero-order-B smoothing + rate-limited tail QPs + contact rollout, device == oracle."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    T, N = 24, 1500
    sys_d, sys_o, x0, u_trj = _box_setup(amd, T)
    idx = sys_o.indices_u_into_x
    x_trj = orc.rollout(sys_o, x0, u_trj)
    du = (0.1 * np.random.default_rng(8).normal(size=(T, N, 2))).astype(np.float32)
    dm = sys_d.dm()
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    assert int(o["info"].abs().sum().item()) == 0
    At, Bt, ct = orc.zero_order_B_decoupled(sys_o, x_trj, u_trj, du.astype(np.float64))
    np.testing.assert_allclose(o["At"].cpu().numpy(), At, rtol=0, atol=0)
    # f32 samples through a stiff (kp = 5e4) contact QP
    np.testing.assert_allclose(o["Bt"].cpu().numpy(), Bt, **FP32_TOL)
    Q = np.diag(orc.BoxPivotOracle.pack([5, 5, 50], [0, 0]))
    Qd, R = Q.copy(), 1e3 * np.eye(2)
    xd = np.tile(orc.BoxPivotOracle.pack([1.0, 1.0, -np.pi / 2], [-0.5, 0.5]), (T + 1, 1))
    rows = orc.quasistatic_bounds(x_trj, idx, None, None, np.array([-np.ones(2) * 0.015, np.ones(2) * 0.015]))
    xa, ua, stats = orc.local_descent_quasistatic_as(sys_o, At, Bt, ct, Q, Qd, R, x0, xd, rows[4], rows[5], "rel")
    assert all(st[1] >= 0 for st in stats)
    out = dm.quasistatic_box_descent(*[dev.to_dev(a) for a in (At, Bt, ct, Q, Qd, R, xd, x0)],
                                     du_lo=dev.to_dev(rows[4]), du_hi=dev.to_dev(rows[5]), solver=0, eps=1e-10)
    info = out["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    np.testing.assert_allclose(out["u_new"].cpu().numpy(), ua, rtol=0, atol=1e-7)
    np.testing.assert_allclose(out["x_new"].cpu().numpy(), xa, rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(out["cost"].item()), orc.eval_cost_quasistatic(xa, ua, xd, Q, Qd, R, idx), rtol=1e-7)
    # the horizon of the reference's script (T = 120) fits the LDS-resident solver for this model
    assert dm.quasistatic_descent_supported(120, 2)


@pytest.mark.parametrize("system,solver", [("planar_hand", "pgs"), ("planar_hand", "exact"),
                                           ("box_pivoting", "pgs"), ("box_pivoting", "exact"), ("box_pushing", "pgs"),
                                           ("box_pushing", "exact")])
@pytest.mark.parametrize("T", [7, 16, 23])
def test_quasistatic_descent_outputs_are_self_consistent(amd, system, solver, T, as_solver):
    """Whatever the model, the contact solver and the parity of the horizon: the trajectory the active-set
    descent returns is a rollout of the device dynamics under the controls it returns, and the cost it
    returns is eval_cost (irs_lqr_quasistatic.py:153-194) of exactly that trajectory -- values that live in
    registers across the out-of-line contact step of every tail."""
    from irs_mpc_amd import device as dev
    rng = np.random.default_rng(T)
    if system == "planar_hand":
        sys_d = amd.PlanarHandDynamics(0.1, contact_solver=solver)
        x0 = HAND.pack([0.0, 0.35, 0.0], [-np.pi / 4, -np.pi / 4], [np.pi / 4, np.pi / 4])
        q = HAND_Q
        goal = HAND.pack([0.3, -0.1, 0.5], [0, 0], [0, 0])
        w, kind = 0.05, "abs"
    else:
        cls = amd.BoxPivotingDynamics if system == "box_pivoting" else amd.BoxPushingDynamics
        sys_d = cls(0.1, contact_solver=solver)
        x0 = orc.BoxPivotOracle.pack([0.0, 0.5, 0.0], [-0.62, 0.3] if system == "box_pivoting" else [0.0, -0.2])
        q = orc.BoxPivotOracle.pack([5, 5, 50], [0, 0])
        goal = orc.BoxPivotOracle.pack([0.5, 0.0 if system == "box_pivoting" else 0.5, -0.4], [0, 0])
        w, kind = 0.04, "rel"
    n, m = sys_d.dim_x, sys_d.dim_u
    idx = sys_d.get_u_indices_into_x()
    dm = sys_d.dm()
    u_trj = np.tile(x0[idx], (T, 1)) + 0.02 * rng.normal(size=(T, m)).cumsum(0)
    Q, Qd, R = np.diag(q), np.diag(10 * q), 5.0 * np.eye(m)
    xd = np.tile(x0 + goal, (T + 1, 1))
    x_trj = dm.rollout_cost(dev.to_dev(x0), dev.to_dev(u_trj), dev.to_dev(Q), dev.to_dev(R), dev.to_dev(xd))[0]
    du = (0.1 * rng.normal(size=(T, 600, m))).astype(np.float32)
    o = dm.smooth(2, x_trj, dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    nom = x_trj[:-1].index_select(1, torch.as_tensor(idx, device=x_trj.device))
    bounds = dict(u_lo=(nom - w).contiguous(), u_hi=(nom + w).contiguous()) if kind == "abs" else \
        dict(du_lo=dev.to_dev(np.full((T, m), -w)), du_hi=dev.to_dev(np.full((T, m), w)))
    out = dm.quasistatic_box_descent(o["At"], o["Bt"], o["ct"], dev.to_dev(Q), dev.to_dev(Qd), dev.to_dev(R),
                                     dev.to_dev(xd), dev.to_dev(x0), solver=as_solver, max_iter=2000, eps=1e-10, **bounds)
    info = out["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    xn, un = out["x_new"].cpu().numpy(), out["u_new"].cpu().numpy()
    x_roll = dm.rollout_cost(dev.to_dev(x0), dev.to_dev(un), dev.to_dev(Q), dev.to_dev(R), dev.to_dev(xd))[0].cpu().numpy()
    np.testing.assert_allclose(xn, x_roll, rtol=0, atol=1e-9)
    np.testing.assert_allclose(float(out["cost"].item()), orc.eval_cost_quasistatic(xn, un, xd, Q, Qd, R, idx), rtol=1e-10)


def test_contact_model_sums_layout_vs_oracle(amd):
    """Contact models ship [Gram | z (f - xb)' | sum z] (include/irs_hip.h): the two-stage path
    (accumulate -> finalize), the fused launch and the oracle's restatement of the layout agree, in
    ZERO_ORDER_AB (x and u noise, MFMA Gram path) as well as ZERO_ORDER_B."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_AB, SMOOTH_ZERO_ORDER_B
    T, N = 5, 3000
    sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    x_trj = orc.rollout(sys_o, x0, u_trj)
    rng = np.random.default_rng(17)
    dx = (rng.normal(size=(T, N, 7)) * 0.01).astype(np.float32)
    du = (rng.normal(size=(T, N, 4)) * 0.1).astype(np.float32)
    dm = sys_d.dm()
    xd, ud = dev.to_dev(x_trj), dev.to_dev(u_trj)
    so = orc.zero_order_sums(sys_o, x_trj, u_trj, dx.astype(np.float64), du.astype(np.float64), sum_z=True)
    assert dm.sums_len(SMOOTH_ZERO_ORDER_AB) == so.shape[1] == 66 + 77 + 11
    sums = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_AB, xd, ud, dev.to_dev(dx, dev.F32), dev.to_dev(du, dev.F32))
    scale = np.abs(so).max(axis=0) + 1e-9
    assert np.max(np.abs(sums.cpu().numpy() - so) / scale) < 2e-3          # f32 contact steps + f32 partial sums
    At, Bt, ct, info = dm.smooth_finalize(SMOOTH_ZERO_ORDER_AB, N, xd, ud, sums)
    Ao, Bo, co = orc.zero_order_from_sums(sys_o, x_trj, u_trj, sums.cpu().numpy())
    assert int(info.abs().sum().item()) == 0
    np.testing.assert_allclose(At.cpu().numpy(), Ao, rtol=0, atol=1e-7)
    np.testing.assert_allclose(Bt.cpu().numpy(), Bo, rtol=0, atol=1e-7)
    np.testing.assert_allclose(ct.cpu().numpy(), co, rtol=0, atol=1e-7)
    fused = dm.smooth(SMOOTH_ZERO_ORDER_AB, xd, ud, dev.to_dev(dx, dev.F32), dev.to_dev(du, dev.F32))
    np.testing.assert_allclose(fused["Bt"].cpu().numpy(), Bt.cpu().numpy(), rtol=0, atol=5e-5)
    # ZERO_ORDER_B: two-stage == fused
    s2 = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    assert s2.shape[1] == 10 + 28 + 4
    A2, B2, c2, _ = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, s2)
    # ... and with the nominal steps the accumulate launch left in its workspace (irs_smooth_finalize_ws)
    A3, B3, c3, i3 = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, xd, ud, s2,
                                        workspace=dm._workspace(SMOOTH_ZERO_ORDER_B, T, N, xd.device))
    assert int(i3.abs().sum().item()) == 0
    # (the f64 nominal step in the workspace comes from the sample pass's cooperative solve, csrc/smooth_ug.hip; the
    # stand-alone solve evaluates the model's step itself: same KKT point, another order of f64 operations)
    assert torch.equal(A3, A2)
    np.testing.assert_allclose(B3.cpu().numpy(), B2.cpu().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(c3.cpu().numpy(), c2.cpu().numpy(), rtol=0, atol=1e-12)
    # a single-workgroup accumulate (small N) leaves them too
    du_s = dev.to_dev(du[:, :200].copy(), dev.F32)
    s4 = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, xd, ud, None, du_s)
    r4 = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, 200, xd, ud, s4)
    r5 = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, 200, xd, ud, s4,
                            workspace=dm._workspace(SMOOTH_ZERO_ORDER_B, T, 200, xd.device))
    np.testing.assert_allclose(r4[1].cpu().numpy(), r5[1].cpu().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(r4[2].cpu().numpy(), r5[2].cpu().numpy(), rtol=0, atol=1e-12)
    f2 = dm.smooth(SMOOTH_ZERO_ORDER_B, xd, ud, None, dev.to_dev(du, dev.F32))
    np.testing.assert_allclose(f2["Bt"].cpu().numpy(), B2.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(f2["ct"].cpu().numpy(), c2.cpu().numpy(), rtol=0, atol=2e-5)


def test_irs_lqr_quasistatic_zero_order_AB_mode(amd):
    """gradient_mode "zero_order_AB" (quasistatic_dynamics.py:268-300: x and u noise, damped least
    squares) + decouple_AB through the host twin == the oracle on identical np.random seeds."""
    T, N = 6, 1500
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.IrsLqrQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]),
              "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict, p.Qd_dict = q_dict, {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)
    p.std_u_initial, p.num_samples = np.ones(4) * 0.1, N
    p.gradient_mode = "zero_order_AB"
    p.publish_every_iteration = False
    sol = amd.IrsLqrQuasistatic(sys_d, p)
    sol.verbose = False
    np.random.seed(11)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    np.random.seed(11)
    dx, du = [], []
    for _ in range(T):
        dx.append(np.random.normal(0, 1e-3, size=[N, 7]))
        du.append(np.random.normal(0, p.std_u_initial, size=[N, 4]))
    dx, du = np.stack(dx).astype(np.float32).astype(np.float64), np.stack(du).astype(np.float32).astype(np.float64)
    Ao, Bo, co = orc.zero_order_AB_damped_decoupled(sys_o, sol.x_trj, sol.u_trj, dx, du)
    np.testing.assert_allclose(At, Ao, rtol=0, atol=0)
    np.testing.assert_allclose(Bt, Bo, rtol=0, atol=1e-4)          # f32 contact steps
    np.testing.assert_allclose(ct, co, rtol=0, atol=1e-4)
    c0 = sol.cost
    sol.iterate(2)
    assert sol.cost_best < c0
    p.gradient_mode = "nope"
    with pytest.raises(RuntimeError):
        amd.IrsLqrQuasistatic(sys_d, p)


@pytest.mark.parametrize("mode", ["zero_order_B", "zero_order_AB", "first_order"])
def test_quasistatic_gradient_modes_without_decoupling(amd, mode):
    """decouple_AB = False (no example of the reference uses it): the full (A, B) of calc_B_zero_order
    (A = the step's derivative at the nominal point), calc_AB_zero_order (damped joint fit) and
    calc_AB_first_order (mean of the full sampled Jacobians) == the oracle on the reference's draws."""
    T, N = 6, 1200
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.IrsLqrQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]),
              "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict, p.Qd_dict = q_dict, {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)
    p.std_u_initial, p.num_samples = np.ones(4) * 0.1, N
    p.publish_every_iteration = False
    p.gradient_mode, p.decouple_AB = mode, False
    sol = amd.IrsLqrQuasistatic(sys_d, p)
    sol.verbose = False
    np.random.seed(9)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    np.random.seed(9)
    if mode == "zero_order_AB":
        dx, du = [], []
        for _ in range(T):
            dx.append(np.random.normal(0, 1e-3, size=[N, 7]))
            du.append(np.random.normal(0, p.std_u_initial, size=[N, 4]))
        dx = np.stack(dx).astype(np.float32).astype(np.float64)
        du = np.stack(du).astype(np.float32).astype(np.float64)
        Ao, Bo, co = orc.zero_order_AB_damped_decoupled(sys_o, sol.x_trj, sol.u_trj, dx, du, decouple=False)
        tolA, tolB = 2e-2, 5e-4           # A is fitted from 1e-3 state noise through f32 steps: ill-conditioned by design
    else:
        du = np.stack([np.random.normal(0, p.std_u_initial, size=[N, 4]) for _ in range(T)])
        if mode == "zero_order_B":
            du = du.astype(np.float32).astype(np.float64)
            Ao, Bo, co = orc.zero_order_B_decoupled(sys_o, sol.x_trj, sol.u_trj, du, decouple=False)
            tolA, tolB = 1e-8, 5e-5
        else:
            Ao, Bo, co = orc.first_order_B_decoupled(sys_o, sol.x_trj, sol.u_trj, du, decouple=False)
            tolA, tolB = 1e-8, 1e-8      # f64 lanes on the f64 draws
    np.testing.assert_allclose(At, Ao, rtol=0, atol=tolA)
    np.testing.assert_allclose(Bt, Bo, rtol=0, atol=tolB)
    np.testing.assert_allclose(ct, co, rtol=0, atol=max(tolA, tolB))
    assert np.abs(At - np.eye(7)).max() > 1e-3                  # not the decoupled structure
    c0 = sol.cost
    sol.iterate(1)
    assert np.isfinite(sol.cost_best) and sol.cost_best <= c0 * 1.5


@pytest.mark.parametrize("mode", ["first_order", "exact"])
def test_quasistatic_simulator_gradient_modes(amd, mode):
    """IrsLqrQuasistatic with the gradient modes that read the simulator's derivatives --
    examples/planar_hand/planar_hand_setup.py:28 runs gradient_mode = "first_order" -- against the oracle:
    get_TV_matrices on the reference's draws (np.random.seed), then descents that lower the cost."""
    T, N = 10, 1500
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.IrsLqrQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]),
              "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict, p.Qd_dict = q_dict, {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)
    p.std_u_initial, p.num_samples = np.ones(4) * 0.1, N
    p.publish_every_iteration = False
    p.gradient_mode = mode
    sol = amd.IrsLqrQuasistatic(sys_d, p)
    sol.verbose = False
    np.random.seed(5)
    At, Bt, ct = sol.get_TV_matrices(sol.x_trj, sol.u_trj)
    if mode == "first_order":
        np.random.seed(5)
        du = np.stack([np.random.normal(0, p.std_u_initial, size=[N, 4]) for _ in range(T)])
        du = du.astype(np.float32).astype(np.float64)
        Ao, Bo, co = orc.first_order_B_decoupled(sys_o, sol.x_trj, sol.u_trj, du)
        tol = 5e-5           # f32 lanes vs the f64 oracle; no flipped samples with the exact step QP
    else:
        Ao, Bo, co = orc.exact_contact_TV(sys_o, sol.x_trj, sol.u_trj, decouple=True)
        tol = 1e-8
    np.testing.assert_allclose(At, Ao, rtol=0, atol=0)
    np.testing.assert_allclose(Bt, Bo, rtol=0, atol=tol)
    np.testing.assert_allclose(ct, co, rtol=0, atol=tol)
    c0 = sol.cost
    sol.iterate(3)
    assert sol.cost_best < c0
    if mode == "exact":
        p.decouple_AB = False
        full = amd.IrsLqrQuasistatic(sys_d, p)
        full.verbose = False
        A2, B2, c2 = full.get_TV_matrices(full.x_trj, full.u_trj)
        Af, Bf, cf = orc.exact_contact_TV(sys_o, full.x_trj, full.u_trj, decouple=False)
        np.testing.assert_allclose(A2, Af, rtol=0, atol=1e-8)
        np.testing.assert_allclose(B2, Bf, rtol=0, atol=1e-8)
        np.testing.assert_allclose(c2, cf, rtol=0, atol=1e-8)


@pytest.mark.parametrize("mode_name", ["zero_order_B", "first_order"])
def test_planar_hand_full_size_properties(amd, mode_name):
    """BASELINE configs[3] at its per-GPU size (planar_hand, T=50, N=10^5), through properties that do
    not need the oracle at that size: 8 logical shards add up to the unsharded statistics; the solve
    of the summed shards equals the fused single launch; launches are bit-reproducible; the decoupled
    structure is exact; the device-RNG stream does not depend on the split; and a 1 % subsample of the
    estimate agrees with the NumPy oracle on the same samples.  Both sample-pass modes: the least-squares
    fit of B (zero_order_B) and the mean of the per-sample active-set derivative (first_order)."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B as _ZB
    SMOOTH_ZERO_ORDER_B = SMOOTH_FIRST_ORDER if mode_name == "first_order" else _ZB     # the mode under test
    from irs_mpc_amd.distributed import shard_range
    T, N = 50, 100000
    sys_d, sys_o, x0, _ = _hand_setup(amd, T)
    u_np = np.tile(x0[HAND_IDX], (T, 1))
    x_np = orc.rollout(sys_o, x0, u_np)
    x_trj, u_trj = dev.to_dev(x_np), dev.to_dev(u_np)
    g = torch.Generator(device="cuda").manual_seed(3)
    du = 0.3 * torch.randn((T, N, 4), generator=g, device="cuda", dtype=torch.float32)
    dm = sys_d.dm()
    full = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du).clone()
    acc = torch.zeros_like(full)
    for r in range(8):
        lo, hi = shard_range(N, r, 8)
        acc += dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du[:, lo:hi].contiguous())
    scale = full.abs().max(dim=0).values + 1e-12
    assert float(((acc - full).abs() / scale).max()) < 2e-5
    A1, B1, c1, i1 = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, x_trj, u_trj, acc)
    fused = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
    assert int(i1.abs().sum().item()) == 0 and int(fused["info"].abs().sum().item()) == 0
    np.testing.assert_allclose(B1.cpu().numpy(), fused["Bt"].cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(c1.cpu().numpy(), fused["ct"].cpu().numpy(), rtol=0, atol=2e-5)
    B_first = fused["Bt"].clone()
    again = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
    assert torch.equal(again["Bt"], B_first) and torch.equal(again["sums"], fused["sums"])
    # decouple_AB structure, exactly (irs_lqr_quasistatic.py:275-284)
    A_exp = np.eye(7)
    A_exp[:, HAND_IDX] = 0.0
    assert np.array_equal(fused["At"].cpu().numpy(), np.tile(A_exp, (T, 1, 1)))
    assert np.array_equal(fused["Bt"].cpu().numpy()[:, HAND_IDX, :], np.tile(np.eye(4), (T, 1, 1)))
    # device RNG: a pure function of the global sample index
    std_u = 0.3 * np.ones(4)
    s_all = dm.smooth_accumulate_rng(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, N, None, std_u, 99, 2).clone()
    s_two = dm.smooth_accumulate_rng(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, 60000, None, std_u, 99, 2).clone()
    s_two += dm.smooth_accumulate_rng(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, 40000, None, std_u, 99, 2, sample_offset=60000)
    assert float(((s_two - s_all).abs() / (s_all.abs().max(dim=0).values + 1e-12)).max()) < 2e-5
    # a 1 % subsample against the oracle (three time steps)
    sub = du[:, :1000].contiguous()
    o_sub = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, sub)
    for t in (0, 17, 49):
        us = u_np[t] + sub[t].cpu().numpy().astype(np.float64)
        if mode_name == "first_order":
            Bo = sys_o.jacobian_xu_batch(np.tile(x_np[t], (1000, 1)), us)[:, :, 7:].mean(0)
            tol = 5e-5          # f32 lanes vs the f64 oracle (exact step QP: no flipped samples)
        else:
            fn = sys_o.dynamics_batch(np.tile(x_np[t], (1000, 1)), us)
            Bo = orc.zero_order_B_fit(sub[t].cpu().numpy().astype(np.float64), fn - sys_o.dynamics(x_np[t], u_np[t]))
            tol = 5e-5
        Bo[HAND_IDX, :] = np.eye(4)
        np.testing.assert_allclose(o_sub["Bt"][t].cpu().numpy(), Bo, rtol=0, atol=tol)


def test_box_pivoting_full_size_properties(amd):
    """BASELINE configs[4] at its FULL size on one GPU (box_pivoting, T=80, N=5*10^4: examples/box_pivoting/
    run_box_pivoting.py:20-131 with the config's sample count): 8 logical shards -- the 8 GPUs of the config --
    add up to the unsharded statistics; the solve of the summed shards equals the fused launch; launches are
    bit-reproducible; the decoupled structure is exact; a 1 % subsample agrees with the NumPy oracle; and one
    descent with the script's rate limit (u_bounds_rel = +-0.15 h, :119-120) at that size converges and is
    self-consistent (its trajectory is the rollout of its controls, its cost their eval_cost)."""
    from examples.run_quasistatic import box_problem
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    from irs_mpc_amd.distributed import shard_range
    T, N = 80, 50000
    sys_d, x0, u_np, Q_dict, Qd_dict, R_dict, xd = box_problem(T)
    sys_o = orc.BoxPivotOracle(0.1)
    idx = sys_o.indices_u_into_x
    x_np = orc.rollout(sys_o, x0, u_np)
    dm = sys_d.dm()
    x_trj, u_trj = dev.to_dev(x_np), dev.to_dev(u_np)
    np.testing.assert_allclose(dm.rollout_cost(dev.to_dev(x0), u_trj, dev.to_dev(np.eye(5)), dev.to_dev(np.eye(2)),
                                               dev.to_dev(xd))[0].cpu().numpy(), x_np, rtol=0, atol=1e-8)
    g = torch.Generator(device="cuda").manual_seed(5)
    std = 0.1 ** 0.5                                               # :122-126 at iteration 1
    du = std * torch.randn((T, N, 2), generator=g, device="cuda", dtype=torch.float32)
    full = dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du).clone()
    acc = torch.zeros_like(full)
    for r in range(8):
        lo, hi = shard_range(N, r, 8)
        acc += dm.smooth_accumulate(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du[:, lo:hi].contiguous())
    scale = full.abs().max(dim=0).values + 1e-12
    assert float(((acc - full).abs() / scale).max()) < 2e-5
    A1, B1, c1, i1 = dm.smooth_finalize(SMOOTH_ZERO_ORDER_B, N, x_trj, u_trj, acc)
    fused = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
    assert int(i1.abs().sum().item()) == 0 and int(fused["info"].abs().sum().item()) == 0
    np.testing.assert_allclose(B1.cpu().numpy(), fused["Bt"].cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(c1.cpu().numpy(), fused["ct"].cpu().numpy(), rtol=0, atol=2e-5)
    again = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, du)
    assert torch.equal(again["Bt"], fused["Bt"]) and torch.equal(again["sums"], fused["sums"])
    A_exp = np.eye(5)
    A_exp[:, idx] = 0.0
    assert np.array_equal(fused["At"].cpu().numpy(), np.tile(A_exp, (T, 1, 1)))
    assert np.array_equal(fused["Bt"].cpu().numpy()[:, idx, :], np.tile(np.eye(2), (T, 1, 1)))
    # a 1 % subsample against the oracle (three time steps: before, at and after the hand meets the box)
    sub = du[:, :500].contiguous()
    o_sub = dm.smooth(SMOOTH_ZERO_ORDER_B, x_trj, u_trj, None, sub)
    for t in (0, 40, 79):
        us = u_np[t] + sub[t].cpu().numpy().astype(np.float64)
        fn = sys_o.dynamics_batch(np.tile(x_np[t], (500, 1)), us)
        Bo = orc.zero_order_B_fit(sub[t].cpu().numpy().astype(np.float64), fn - sys_o.dynamics(x_np[t], u_np[t]))
        Bo[idx, :] = np.eye(2)
        np.testing.assert_allclose(o_sub["Bt"][t].cpu().numpy(), Bo, **FP32_TOL)
    # one descent of the script at this size
    Q, Qd, R = (dev.to_dev(sys_d.get_Q_from_Q_dict(Q_dict)), dev.to_dev(sys_d.get_Q_from_Q_dict(Qd_dict)),
                dev.to_dev(sys_d.get_R_from_R_dict(R_dict)))
    w = 0.15 * 0.1
    out = dm.quasistatic_box_descent(fused["At"], fused["Bt"], fused["ct"], Q, Qd, R, dev.to_dev(xd), dev.to_dev(x0),
                                     du_lo=dev.to_dev(np.full((T, 2), -w)), du_hi=dev.to_dev(np.full((T, 2), w)),
                                     solver=0, max_iter=2000, eps=1e-9)
    info = out["info"].cpu().numpy()
    assert info[0] == 0 and info[2] == 0, info
    xn, un = out["x_new"].cpu().numpy(), out["u_new"].cpu().numpy()
    # the rate limit binds each tail's FIRST du, measured from the realised actuated position (tv_lqr.py:99-100)
    assert np.abs(un - xn[:-1][:, idx]).max() <= w + 1e-8 and np.abs(un - xn[:-1][:, idx]).max() > w - 1e-6
    x_roll = dm.rollout_cost(dev.to_dev(x0), dev.to_dev(un), Q, R, dev.to_dev(xd))[0].cpu().numpy()
    np.testing.assert_allclose(xn, x_roll, rtol=0, atol=1e-9)
    np.testing.assert_allclose(float(out["cost"].item()),
                               orc.eval_cost_quasistatic(xn, un, xd, Q.cpu().numpy(), Qd.cpu().numpy(), R.cpu().numpy(), idx),
                               rtol=1e-10)


def test_device_contact_qp_reproduces_reference_box_on_box(amd):
    """The device contact-QP code shared by every contact functor (csrc/contact_models.hpp:
    irs_contact_qp_step), on the reference's own 1-D example and against the closed form printed there
    (examples/box_pushing/analysis/box_on_box.py:11-20): m = 1, k = 100, h = 0.1, pusher at 0, box at 1."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    m, k, h = 1.0, 100.0, 0.1
    w1 = m / (m + h ** 2.0 * k)                 # box_on_box.py:16
    w2 = h ** 2.0 * k / (m + h ** 2.0 * k)      # box_on_box.py:17
    sys_d = amd.BoxOnBoxDynamics(h, m, k)
    u = np.linspace(-2.0, 2.0, 401)[:, None]
    X = np.tile(np.array([0.0, 1.0]), (len(u), 1))
    want = np.where(u > 1, np.hstack([w1 + w2 * u, w1 + w2 * u]), np.hstack([u, np.ones_like(u)]))   # :18 / :20
    np.testing.assert_allclose(sys_d.dynamics_batch(X, u), want, rtol=0, atol=1e-13)
    # ... and through the f32 sample pass: deep in contact the smoothed dx_u/du is w2, far from it 0
    dm = sys_d.dm()
    x_trj = np.tile(np.array([0.0, 1.0]), (3, 1))
    u_trj = np.array([[1.8], [-1.0]])
    du = (0.05 * np.random.default_rng(0).normal(size=(2, 4000, 1))).astype(np.float32)
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    B = o["Bt"].cpu().numpy()
    np.testing.assert_allclose(B[0, :, 0], [1.0, w2], rtol=0, atol=2e-6)       # pusher row decoupled, box row = w2
    np.testing.assert_allclose(B[1, :, 0], [1.0, 0.0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(o["ct"].cpu().numpy()[0], [w1 + w2 * 1.8 - 1.8, w1 + w2 * 1.8 - 1.0 - w2 * 1.8],
                               rtol=0, atol=2e-6)


# ---------------------------------------------------------------- box pushing: PINNED by the reference's simulator data
def _box_pushing_data(golden_dir):
    xu = np.load(os.path.join(golden_dir, "box_pushing_xu_quasistatic.npy"))
    J = np.load(os.path.join(golden_dir, "box_pushing_dxdu_quasistatic.npy"))
    return xu[:, :5], xu[:, 5:], J


def test_device_contact_step_reproduces_simulator_trajectory(amd, golden_dir):
    """The device contact functor against REAL simulator output: the 80-step push shipped as
    examples/box_pushing/analysis/xu_quasistatic.npy (row t = [step(x_{t-1}, u_t), u_t]) -- every
    transition in one batch, and the whole trajectory as a device rollout."""
    from irs_mpc_amd import device as dev
    x, u, _ = _box_pushing_data(golden_dir)
    sys_d = amd.BoxPushingDynamics(0.1)
    np.testing.assert_allclose(sys_d.dynamics_batch(x[:-1], u[1:]), x[1:], rtol=0, atol=3e-8)
    dm = sys_d.dm()
    x_trj, _ = dm.rollout_cost(dev.to_dev(x[0]), dev.to_dev(u[1:]), dev.to_dev(np.eye(5)), dev.to_dev(np.eye(2)),
                               dev.to_dev(np.zeros((80, 5))))
    np.testing.assert_allclose(x_trj.cpu().numpy(), x, rtol=0, atol=2e-7)


def test_device_active_set_jacobian_matches_simulator(amd, golden_dir):
    """`jacobian_xu_batch` of the pinned functor (f64 lanes: contact step + masked LDL' of the active
    rows) against all 80 of the simulator's [Dq_next/Dq | Dq_next/Dq_a_cmd] -- see
    test_box_pushing_active_set_jacobian_matches_simulator for the one onset row -- and the f32 FIRST_ORDER
    sample pass (N = 4096 u-perturbations of std 1e-4: the active set of every sample is the nominal one)
    against the same data."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER
    x, u, J = _box_pushing_data(golden_dir)
    sys_d = amd.BoxPushingDynamics(0.1)
    G = sys_d.jacobian_xu_batch(x, u)
    err = np.abs(G - J).reshape(len(x), -1).max(1)
    onset = int(np.argmax(err))
    assert err[onset] < 2e-4 and np.delete(err, onset).max() < 5e-7
    pts = [5, 12, 30, 40, 60, 78]
    rng = np.random.default_rng(3)
    du = (rng.normal(size=(len(pts), 4096, 2)) * 1e-4).astype(np.float32)
    dm = sys_d.dm()
    o = dm.smooth(SMOOTH_FIRST_ORDER, dev.to_dev(x[pts]), dev.to_dev(u[pts]), None, dev.to_dev(du, dev.F32))
    Bt = o["Bt"].cpu().numpy()
    for k, t in enumerate(pts):
        np.testing.assert_allclose(Bt[k][[1, 3, 4]], J[t][[1, 3, 4], 5:], rtol=0, atol=2e-5)   # unactuated rows
        np.testing.assert_allclose(Bt[k][[0, 2]], np.eye(2), rtol=0, atol=0)                  # decoupled


def test_device_smoothing_matches_simulator_input_jacobian(amd, golden_dir):
    """Zero-order-B smoothing (N = 20000 f32 contact steps per point, small std) of the pinned functor
    against the simulator's own Dq_next/Dq_a_cmd (dxdu_quasistatic.npy[:, :, 5:]) at points of the
    recorded push away from the contact onset: free flight and sticking contact (the hand drags and
    turns the box through the friction rows).  Box rows only: decouple_AB overwrites the hand rows."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_ZERO_ORDER_B
    x, u, J = _box_pushing_data(golden_dir)
    pts = [5, 12, 30, 40, 60, 78]
    sys_d = amd.BoxPushingDynamics(0.1)
    dm = sys_d.dm()
    # the smoothing call linearises step(x_trj[t], u_trj[t]): the recorded pairs are (x_{t-1}, u_t)
    x_trj = np.vstack([x[[t - 1 for t in pts]], x[pts[-1]][None]])
    u_trj = u[pts]
    du = (1e-3 * np.random.default_rng(4).normal(size=(len(pts), 20000, 2))).astype(np.float32)
    o = dm.smooth(SMOOTH_ZERO_ORDER_B, dev.to_dev(x_trj), dev.to_dev(u_trj), None, dev.to_dev(du, dev.F32))
    assert int(o["info"].abs().sum().item()) == 0
    B = o["Bt"].cpu().numpy()
    box_rows = [1, 3, 4]
    for k, t in enumerate(pts):
        np.testing.assert_allclose(B[k][box_rows], J[t][box_rows, 5:], rtol=0, atol=3e-3, err_msg="t=%d" % t)
    assert abs(B[3][4, 0] - 1.58) < 5e-3 and abs(B[3][3, 1] - 0.5) < 1e-3      # t = 40: dragging and pushing


# ---------------------------------------------------------------- the metric's contact model, pinned (round 3)
def test_planar_hand_spin_initial_cost_matches_reference_on_device(amd, golden_dir):
    """PIN of the device's planar-hand functor by the reference's own result files: the first entry of
    examples/planar_hand/analysis/planar_hand_spin_{exact,zero_order_B,first_order}.csv (249.6305470294...) is the
    cost of run_planar_hand_spin.py's initial rollout -- 30 steps of the external simulator on the grasp under
    gravity (tests/test_oracle_golden.py::test_planar_hand_spin_initial_cost_matches_reference has the oracle's
    side and the sensitivity to mass and friction).  Here the same number comes out of the HIP path:
    `PlanarHandDynamics` f64 rollout (irs_rollout_cost) + `IrsLqrQuasistatic.eval_cost`, through the problem
    set-up of examples/run_quasistatic.py (the twin of the reference's script)."""
    from examples.run_quasistatic import spin_problem
    gold = np.loadtxt(os.path.join(golden_dir, "planar_hand_spin_exact.csv"))[0]
    for k in ("zero_order_B", "first_order"):
        assert abs(np.loadtxt(os.path.join(golden_dir, "planar_hand_spin_%s.csv" % k))[0] - gold) < 1e-11
    T = 30
    q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, x_trj_d = spin_problem(T, 0.1)
    p = amd.IrsLqrQuasistaticParameters()
    p.Q_dict, p.Qd_dict, p.R_dict = Q_dict, Qd_dict, R_dict
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, x_trj_d, u_traj_0, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.1, np.ones(4) * 0.1])         # run_planar_hand_spin.py:142-143
    p.sampling = lambda u_initial, it: u_initial / (it ** 0.5)
    p.std_u_initial, p.num_samples = np.ones(4) * 0.1, 100
    p.publish_every_iteration = False
    sol = amd.IrsLqrQuasistatic(q_dynamics, p)
    sol.verbose = False
    np.testing.assert_allclose(sol.cost, gold, rtol=1e-8)
    assert abs(sol.cost_all_list[0] - gold) / gold < 5e-10                   # measured on the oracle: 2.1e-10
    # the device rollout IS the oracle's, step by step
    o, xo = orc.PlanarHandOracle(0.1), None
    xo = orc.rollout(o, x0, u_traj_0)
    np.testing.assert_allclose(sol.x_trj, xo, rtol=0, atol=1e-10)
    # wrong masses miss, on the device too
    for mass in (0.5, 2.0):
        qd2 = amd.PlanarHandDynamics(0.1, mass=mass)
        s2 = amd.IrsLqrQuasistatic(qd2, p)
        assert abs(s2.cost - gold) / gold > 1e-4
    # the sweep solver (opt-in) lands on the same rollout
    s3 = amd.IrsLqrQuasistatic(amd.PlanarHandDynamics(0.1, contact_solver="pgs"), p)
    np.testing.assert_allclose(s3.cost, gold, rtol=1e-7)
    # and the optimiser runs on it (first_order, the set-up file's mode) and lowers the cost
    p.gradient_mode, p.num_samples = "first_order", 2000
    np.random.seed(0)
    s4 = amd.IrsLqrQuasistatic(q_dynamics, p)
    s4.verbose = False
    s4.iterate(8)
    assert s4.cost_best < 0.75 * gold


# ---------------------------------------------------------------- a7's public estimator methods (round 3)
def _contact_nominals(amd, system, k):
    if system == "planar_hand":
        sys_d, sys_o, x0, u_trj = _hand_setup(amd, k)
    else:
        sys_d, sys_o, x0, u_trj = _box_setup(amd, k)
    x = orc.rollout(sys_o, x0, u_trj)[:k]
    return sys_d, sys_o, x, u_trj


@pytest.mark.parametrize("system", ["planar_hand", "box_pivoting"])
def test_calc_AB_estimator_methods_vs_oracle(amd, system):
    """`QuasistaticDynamics.calc_AB_first_order / calc_B_zero_order / calc_AB_zero_order / calc_AB_batch`
    (irs_lqr/quasistatic_dynamics.py:193-300) on the device twins: same names, arguments, global-generator draws
    and `[A | B]` return layout; each against the oracle's restatement of the same estimator on the same draws
    (k nominal points along a trajectory in contact -> ONE sample-pass launch per call)."""
    k, N = 5, 1500
    sys_d, sys_o, x, u = _contact_nominals(amd, system, k)
    n, m = sys_o.dim_x, sys_o.dim_u
    std_u = 0.1 * np.ones(m)
    xp = np.vstack([x, x[-1:]])

    def draws(mode):
        np.random.seed(11)
        if mode == "zero_order_AB":
            dx, du = [], []
            for _ in range(k):
                dx.append(np.random.normal(0, 1e-3, size=[N, n]))
                du.append(np.random.normal(0, std_u, size=[N, m]))
            return np.stack(dx), np.stack(du)
        return None, np.stack([np.random.normal(0, std_u, size=[N, m]) for _ in range(k)])

    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    for mode, tolA, tolB in (("first_order", 1e-8, 1e-8), ("zero_order_B", 1e-8, 5e-5), ("zero_order_AB", 3e-2, 5e-4),
                             ("exact", 1e-8, 1e-8)):
        np.random.seed(11)
        got = sys_d.calc_AB_batch(x, u, N, std_u, mode)
        assert got.shape == (k, n, n + m)
        dx, du = draws(mode)
        if mode == "first_order":
            Ao, Bo, _ = orc.first_order_B_decoupled(sys_o, xp, u, du, decouple=False)
        elif mode == "zero_order_B":
            Ao, Bo, _ = orc.zero_order_B_decoupled(sys_o, xp, u, f32(du), decouple=False)
        elif mode == "zero_order_AB":
            Ao, Bo, _ = orc.zero_order_AB_damped_decoupled(sys_o, xp, u, f32(dx), f32(du), decouple=False)
        else:
            Ao, Bo, _ = orc.exact_contact_TV(sys_o, xp, u, decouple=False)
        np.testing.assert_allclose(got[:, :, :n], Ao, rtol=0, atol=tolA, err_msg=mode)
        np.testing.assert_allclose(got[:, :, n:], Bo, rtol=0, atol=tolB, err_msg=mode)
    # the single-point methods are the batch of one point on the same draws
    np.random.seed(11)
    AB = sys_d.calc_AB_first_order(x[0], u[0], N, std_u)
    Ao, Bo, _ = orc.first_order_B_decoupled(sys_o, xp[:2], u[:1], draws("first_order")[1][:1], decouple=False)
    np.testing.assert_allclose(AB, np.hstack([Ao[0], Bo[0]]), rtol=0, atol=1e-8)
    np.random.seed(11)
    AB = sys_d.calc_B_zero_order(x[0], u[0], N, std_u)
    Ao, Bo, _ = orc.zero_order_B_decoupled(sys_o, xp[:2], u[:1], f32(draws("zero_order_B")[1][:1]), decouple=False)
    np.testing.assert_allclose(AB, np.hstack([Ao[0], Bo[0]]), rtol=0, atol=5e-5)
    np.random.seed(11)
    AB = sys_d.calc_AB_zero_order(x[0], u[0], N, std_u)
    dx, du = draws("zero_order_AB")
    Ao, Bo, _ = orc.zero_order_AB_damped_decoupled(sys_o, xp[:2], u[:1], f32(dx[:1]), f32(du[:1]), decouple=False)
    np.testing.assert_allclose(AB[:, n:], Bo[0], rtol=0, atol=5e-4)
    np.testing.assert_allclose(AB[:, :n], Ao[0], rtol=0, atol=3e-2)
    # a scalar std_u broadcasts like np.random.normal's `scale`; an unknown mode is the reference's RuntimeError
    np.random.seed(11)
    a1 = sys_d.calc_B_zero_order(x[0], u[0], 200, 0.1)
    np.random.seed(11)
    np.testing.assert_array_equal(a1, sys_d.calc_B_zero_order(x[0], u[0], 200, 0.1 * np.ones(m)))
    with pytest.raises(RuntimeError, match="is not supported"):
        sys_d.calc_AB_batch(x, u, 10, std_u, "second_order")


def test_calc_AB_estimator_methods_vs_simulator_jacobians(amd, golden_dir):
    """The same methods on the PINNED model against the simulator's own derivatives
    (examples/box_pushing/analysis/dxdu_quasistatic.npy): `calc_AB_first_order` with a vanishing std is the
    simulator's `[Dq_nextDq | Dq_nextDqa_cmd]`; `calc_B_zero_order` returns exactly `Dq_nextDq` as its A block
    and the smoothed input Jacobian as B."""
    x, u, J = _box_pushing_data(golden_dir)
    sys_d = amd.BoxPushingDynamics(0.1)
    pts = [5, 12, 30, 40, 60, 78]
    np.random.seed(2)
    AB1 = sys_d.calc_AB_batch(x[pts], u[pts], 64, 1e-7, "first_order")
    np.testing.assert_allclose(AB1, J[pts], rtol=0, atol=5e-7)
    # the recorded pairs of the zero-order data are (x_{t-1}, u_t) (test_device_smoothing_matches_simulator_input_jacobian)
    xs = x[[t - 1 for t in pts]]
    np.random.seed(2)
    AB0 = sys_d.calc_AB_batch(xs, u[pts], 20000, 1e-3, "zero_order_B")
    np.testing.assert_allclose(AB0[:, :, :5], sys_d.jacobian_xu_batch(xs, u[pts])[:, :, :5], rtol=0, atol=1e-12)
    for kk, t in enumerate(pts):
        np.testing.assert_allclose(AB0[kk][[1, 3, 4], 5:], J[t][[1, 3, 4], 5:], rtol=0, atol=3e-3)


# ---------------------------------------------------------------- non-finite samples are reported (round 3)
@pytest.mark.parametrize("system,mode_name,N", [("pendulum", "ZERO_ORDER_AB", 500), ("pendulum", "ZERO_ORDER_AB", 40000),
                                                 ("quadrotor", "FIRST_ORDER", 2000), ("quadrotor", "ZERO_ORDER_AB", 2000),
                                                 ("planar_hand", "ZERO_ORDER_B", 3000), ("planar_hand", "FIRST_ORDER", 3000),
                                                 ("box_pivoting", "ZERO_ORDER_B", 3000)])
@pytest.mark.parametrize("poison", [float("nan"), float("inf"), -float("inf")])
def test_nonfinite_sample_is_reported(amd, system, mode_name, N, poison):
    """A NaN / Inf perturbation (an upstream bug, or an f32 sample that diverged) must not come back as a
    finite-looking (A, B, c): the solve's `info` is non-zero for exactly the poisoned time step (csrc/smooth.hip:
    the statistics are tested on their BIT PATTERN, because that translation unit is built with
    -ffinite-math-only), and the host classes turn it into the reference's ValueError."""
    from irs_mpc_amd import _lib, device as dev
    MODE = getattr(_lib, "SMOOTH_" + mode_name)
    T = 6
    if system in ("pendulum", "quadrotor"):
        sys_d, sys_o = systems(amd, system)
        p = pend_params(amd, T) if system == "pendulum" else quad_params(amd, T)
        x0, u_trj = p.x0, p.u_trj_initial
    elif system == "planar_hand":
        sys_d, sys_o, x0, u_trj = _hand_setup(amd, T)
    else:
        sys_d, sys_o, x0, u_trj = _box_setup(amd, T)
    n, m = sys_o.dim_x, sys_o.dim_u
    x_trj = orc.rollout(sys_o, np.asarray(x0, float), u_trj)
    rng = np.random.default_rng(0)
    du = (0.1 * rng.normal(size=(T, N, m))).astype(np.float32)
    dx = (0.1 * rng.normal(size=(T, N, n))).astype(np.float32) if mode_name != "ZERO_ORDER_B" and system in ("pendulum", "quadrotor") else None
    t_bad, i_bad = 3, N // 2 + 7
    du[t_bad, i_bad, m - 1] = poison
    dm = sys_d.dm()
    o = dm.smooth(MODE, dev.to_dev(x_trj), dev.to_dev(u_trj), None if dx is None else dev.to_dev(dx, dev.F32),
                  dev.to_dev(du, dev.F32))
    info = o["info"].cpu().numpy()
    assert info[t_bad] != 0, info
    assert (np.delete(info, t_bad) == 0).all(), info
    for k in ("At", "Bt", "ct"):
        assert np.isfinite(np.delete(o[k].cpu().numpy(), t_bad, axis=0)).all()
    # two-stage path (what several GPUs run): the same verdict from the all-reduced sums
    sums = dm.smooth_accumulate(MODE, dev.to_dev(x_trj), dev.to_dev(u_trj), None if dx is None else dev.to_dev(dx, dev.F32),
                                dev.to_dev(du, dev.F32))
    *_, info2 = dm.smooth_finalize(MODE, N, dev.to_dev(x_trj), dev.to_dev(u_trj), sums)
    assert info2.cpu().numpy()[t_bad] != 0


def test_nonfinite_sample_raises_in_the_host_classes(amd):
    T, N = 8, 300
    calls = {"n": 0}

    def sampling(xbar, ubar, it):
        calls["n"] += 1
        dx = np.random.normal(0.0, 1.0, size=(N, 2))
        du = np.random.normal(0.0, 1.0, size=(N, 1))
        if calls["n"] == 3:
            du[5, 0] = np.nan
        return dx, du

    sol = amd.IrsLqrZeroOrder(amd.PendulumDynamics(0.05), pend_params(amd, T), sampling)
    sol.verbose = False
    with pytest.raises(ValueError):
        sol.iterate(1)
    # quasistatic twin: std_u = inf makes every draw non-finite
    sys_d, sys_o, x0, u_trj, _, _, _, (Q, Qd, R, xd) = _hand_problem(amd, T, 4, 0)
    p = amd.IrsLqrQuasistaticParameters()
    q_dict = {"sphere": np.array([1e-3, 1e-3, 10.0]), "arm_left": np.array([1e-3, 1e-3]), "arm_right": np.array([1e-3, 1e-3])}
    p.Q_dict, p.Qd_dict = q_dict, {k: 100 * v for k, v in q_dict.items()}
    p.R_dict = {"arm_left": 5 * np.ones(2), "arm_right": 5 * np.ones(2)}
    p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, xd, u_trj, T
    p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
    p.sampling = lambda u_initial, it: u_initial * np.inf
    p.std_u_initial, p.num_samples = np.ones(4) * 0.1, N
    p.publish_every_iteration = False
    for mode in ("zero_order_B", "first_order"):
        p.gradient_mode = mode
        qs = amd.IrsLqrQuasistatic(sys_d, p)
        qs.verbose = False
        with pytest.raises(ValueError):
            qs.get_TV_matrices(qs.x_trj, qs.u_trj)


# ---------------------------------------------------------------- f4: the reference's script text through the shim
_SHIM_SCRIPT = '''
import numpy as np
import time

import matplotlib.pyplot as plt
from matplotlib import cm

from {system}_dynamics import {cls}
from irs_lqr.all import IrsLqrParameters, IrsLqrExact

{name} = {cls}({h})
timesteps = {T}
params = IrsLqrParameters()
{block}
solver = IrsLqrExact({name}, params)
time_now = time.time()
solver.iterate({iters})
print("Final cost: " + str(solver.cost))
print("Elapsed time: " + str(time.time() - time_now))
plt.figure()
plt.plot(solver.cost_lst)
plt.show()
'''

_PENDULUM_BLOCK = '''params.Q = np.diag([1., 1.])
params.Qd = np.diag([20., 20.])
params.R = np.diag([1])
params.x0 = np.array([0, 0])
params.xd_trj = np.tile(np.array([np.pi, 0]), (timesteps+1,1))
params.xbound = [-np.array([1e4, 1e4]), np.array([1e4, 1e4])]
params.ubound = np.array([-np.array([1e4]), np.array([1e4])])
params.u_trj_initial = np.tile(np.array([0.1]), (timesteps,1))'''

_BICYCLE_BLOCK = '''params.Q = np.diag([5, 5, 3, 0.1, 0.1])
params.Qd = np.diag([50, 50, 30, 1, 1])
params.R = np.diag([1, 0.1])
params.x0 = np.array([0, 0, 0, 0, 0])
params.xd_trj = np.tile(np.array([3.0, 1.0, np.pi/2, 0, 0]), (timesteps+1,1))
params.xbound = [-np.array([1e4, 1e4, 1e4, 1e4, np.pi/4]), np.array([1e4, 1e4, 1e4, 1e4, np.pi/4])]
params.ubound = np.array([-np.array([1e4, 1e4]), np.array([1e4, 1e4])])
params.u_trj_initial = np.tile(np.array([0.1, 0.0]), (timesteps,1))'''


@pytest.mark.parametrize("system,cls,h,T,iters,block,csv,rtol", [
    ("pendulum", "PendulumDynamics", 0.05, 200, 7, _PENDULUM_BLOCK, "pendulum_exact.csv", 2e-9),
    ("bicycle", "BicycleDynamics", 0.1, 100, 2, _BICYCLE_BLOCK, "bicycle_easy_exact.csv", None)],
    ids=["pendulum_exact", "bicycle_exact"])
def test_reference_script_text_runs_through_the_import_shim(amd, golden_dir, tmp_path, system, cls, h, T, iters, block,
                                                           csv, rtol):
    """f4: a script in the exact SHAPE of the reference's examples (same two import lines --
    `from pendulum_dynamics import PendulumDynamics`, `from irs_lqr.all import IrsLqrParameters, IrsLqrExact` --
    same parameter block, `solver.iterate(k)`, matplotlib at the end; the text is this repo's own twin of
    examples/pendulum/pendulum_exact.py / examples/bicycle/bicycle_exact.py, the reference's files are not copied)
    runs UNMODIFIED through examples/compat/run_script.py and reproduces the reference's own result file; the
    regenerated curve is what profiles/curves/ holds."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("run_script", os.path.join(ROOT, "examples", "compat", "run_script.py"))
    rs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rs)
    script = tmp_path / ("%s_exact.py" % system)
    script.write_text(_SHIM_SCRIPT.format(system=system, cls=cls, name=system, h=h, T=T, iters=iters, block=block))
    cwd = os.getcwd()
    try:
        g = rs.run(str(script), seed=0, workdir=str(tmp_path))
    finally:
        os.chdir(cwd)
    costs = np.array(rs.cost_history(g))
    gold = np.loadtxt(os.path.join(golden_dir, csv))
    assert type(g["solver"]).__module__ == "irs_mpc_amd.irs_lqr" and len(costs) == iters + 2
    if rtol is not None:
        np.testing.assert_allclose(costs, gold[:len(costs)], rtol=rtol)
    else:
        # bicycle: the steer bound is active; the reference's curve carries OSQP's 1e-3 accuracy per tail QP
        np.testing.assert_allclose(costs[0], gold[0], rtol=1e-9)
        np.testing.assert_allclose(costs[1], gold[1], rtol=2e-2)


# ---------------------------------------------------------------- the fused iterate at the boundary (round 3)
@pytest.mark.parametrize("name,cls,N,iters", [("pendulum", "IrsLqrZeroOrder", 2000, 5), ("quadrotor", "IrsLqrFirstOrder", 400, 3),
                                              ("pendulum", "IrsLqrExact", 0, 6), ("bicycle", "IrsLqrExact", 0, 3),
                                              ("bicycle", "IrsLqrZeroOrder", 3000, 2)])
def test_fused_iterate_equals_the_host_loop(amd, name, cls, N, iters, capsys):
    """IrsLqr.iterate through ONE library call (irs_iterate, csrc/iterate.hip: every descent enqueued back to back, the
    histories read back once) == the host loop (a verbose run: one read-back per iteration) -- same kernels, same
    Philox counters: bit for bit, for the sampled estimators, the exact one, and with an ACTIVE box bound (bicycle:
    the steer limit binds, the plan test raises its flag on the device and the bounded descent runs behind it)."""
    from examples.problems import PROBLEMS
    T = {"pendulum": 40, "quadrotor": 30, "bicycle": 40}[name]

    def make(verbose):
        sysd, params, sm, _, _ = PROBLEMS[name](T)
        if cls == "IrsLqrExact":
            sol = amd.IrsLqrExact(sysd, params)
        else:
            smp = amd.GaussianSmoothing(sm["std_x"], sm["std_u"], N, power=sm["power"], seed=5)
            sol = getattr(amd, cls)(sysd, params, smp)
        sol.verbose = verbose
        return sol

    a, b = make(False), make(True)
    timing = {}
    ra = a.iterate(iters, timing=timing)
    rb = b.iterate(iters)
    capsys.readouterr()
    assert len(a.cost_lst) == len(b.cost_lst) == iters + 2 and a.iter == b.iter
    # (with an active bound the fused path takes the cost the bounded-descent kernel accumulates, the host loop a
    # separate irs_evaluate_cost launch: the same sum in another order)
    np.testing.assert_allclose(np.array(a.cost_lst), np.array(b.cost_lst), rtol=1e-13 if name == "bicycle" else 0, atol=0)
    for xa, xb in zip(a.x_trj_lst, b.x_trj_lst):
        np.testing.assert_array_equal(xa, xb)
    for ua, ub in zip(a.u_trj_lst, b.u_trj_lst):
        np.testing.assert_array_equal(ua, ub)
    np.testing.assert_array_equal(ra[0], rb[0])
    assert abs(ra[2] - rb[2]) <= 1e-13 * abs(rb[2]) and a.cost_lst[-1] < a.cost_lst[0]
    # irs_timing (SURVEY 8(b) irs_get_timing): per-phase device time of the loop
    assert timing["descents"] == iters + 1 and timing["descent_ms"] > 0 and timing["linearise_ms"] > 0
    assert timing["sample_steps"] == (iters + 1) * T * N
    if name == "bicycle":
        assert getattr(b, "_box_used", False)              # the bound was active in the host loop's last descent
    # a second call continues where the first stopped, like the reference's loop
    a.iterate(iters + 1)
    b.iterate(iters + 1)
    capsys.readouterr()
    np.testing.assert_allclose(np.array(a.cost_lst), np.array(b.cost_lst), rtol=1e-13 if name == "bicycle" else 0, atol=0)


def test_plan_within_bounds_kernel_vs_host_statement(amd):
    """irs_tvlqr_plan_within_bounds (one thread per tail, all tails advancing together) against the host's
    `_tail_plans_within_bounds` on the bicycle problem, with the steer limit loosened step by step until no plan
    touches it."""
    from irs_mpc_amd import _lib, device as dev
    from examples.problems import bicycle
    T = 60
    sysd, params, _, _, _ = bicycle(T)
    seen = set()
    for steer in (0.1, np.pi / 4, 1.5, 1e4):
        params.xbound = [-np.array([1e4, 1e4, 1e4, 1e4, steer]), np.array([1e4, 1e4, 1e4, 1e4, steer])]
        sol = amd.IrsLqrExact(sysd, params)
        sol.verbose = False
        x, u = dev.to_dev(sol.x_trj), dev.to_dev(sol.u_trj)
        At, Bt, ct = sol._get_TV_matrices_dev(x, u)
        o = sol._dm.tvlqr_descent(At, Bt, ct, sol._Q, sol._Qd, sol._R, sol._xd, x[0].contiguous(), alpha_R=0.5)
        box = sol._box_bounds()
        want = sol._tail_plans_within_bounds(At, Bt, ct, o["K"], o["k"], o["x_new"])
        flag = torch.full((1,), -7, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().irs_tvlqr_plan_within_bounds(5, 2, T, At.data_ptr(), Bt.data_ptr(), ct.data_ptr(),
                                                           o["K"].data_ptr(), o["k"].data_ptr(), o["x_new"].data_ptr(),
                                                           *(b.data_ptr() for b in box), flag.data_ptr(), dev._stream()),
                   "irs_tvlqr_plan_within_bounds")
        assert int(flag.item()) == (0 if want else 1), steer
        seen.add(bool(want))
    assert seen == {True, False}
    # T = 200 (more tails than one pass of the workgroup holds: 64 in flight), n = 2: the pendulum's horizon in the script
    from examples.problems import pendulum
    T = 200
    sysd, params, _, _, _ = pendulum(T)
    seen = set()
    for speed in (0.5, 3.0, 1e4):
        params.xbound = [-np.array([1e4, speed]), np.array([1e4, speed])]
        sol = amd.IrsLqrExact(sysd, params)
        sol.verbose = False
        x, u = dev.to_dev(sol.x_trj), dev.to_dev(sol.u_trj)
        At, Bt, ct = sol._get_TV_matrices_dev(x, u)
        o = sol._dm.tvlqr_descent(At, Bt, ct, sol._Q, sol._Qd, sol._R, sol._xd, x[0].contiguous(), alpha_R=0.5)
        box = sol._box_bounds()
        want = sol._tail_plans_within_bounds(At, Bt, ct, o["K"], o["k"], o["x_new"])
        flag = torch.full((1,), -7, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().irs_tvlqr_plan_within_bounds(2, 1, T, At.data_ptr(), Bt.data_ptr(), ct.data_ptr(),
                                                           o["K"].data_ptr(), o["k"].data_ptr(), o["x_new"].data_ptr(),
                                                           *(b.data_ptr() for b in box), flag.data_ptr(), dev._stream()),
                   "irs_tvlqr_plan_within_bounds")
        assert int(flag.item()) == (0 if want else 1), speed
        seen.add(bool(want))
    assert seen == {True, False}
    # n = 12 (16 lanes per tail, several tails per wave, T not a multiple of the tails in flight): the quadrotor with its
    # script's bounds and the pitch limit varied
    from examples.problems import quadrotor
    T = 37
    sysd, params, _, _, _ = quadrotor(T)
    seen = set()
    for pitch in (1e-3, 0.02, np.pi / 2, 1e5):
        big = np.array([1e5, 1e5, 1e5, 2.0 * np.pi, pitch, 2.0 * np.pi, 1e5, 1e5, 1e5, 1e5, 1e5, 1e5])
        params.xbound = [-big, big]
        sol = amd.IrsLqrExact(sysd, params)
        sol.verbose = False
        x, u = dev.to_dev(sol.x_trj), dev.to_dev(sol.u_trj)
        At, Bt, ct = sol._get_TV_matrices_dev(x, u)
        o = sol._dm.tvlqr_descent(At, Bt, ct, sol._Q, sol._Qd, sol._R, sol._xd, x[0].contiguous(), alpha_R=0.5)
        box = sol._box_bounds()
        want = sol._tail_plans_within_bounds(At, Bt, ct, o["K"], o["k"], o["x_new"])
        flag = torch.full((1,), -7, dtype=torch.int32, device=x.device)
        _lib.check(_lib.load().irs_tvlqr_plan_within_bounds(12, 4, T, At.data_ptr(), Bt.data_ptr(), ct.data_ptr(),
                                                           o["K"].data_ptr(), o["k"].data_ptr(), o["x_new"].data_ptr(),
                                                           *(b.data_ptr() for b in box), flag.data_ptr(), dev._stream()),
                   "irs_tvlqr_plan_within_bounds")
        assert int(flag.item()) == (0 if want else 1), pitch
        seen.add(bool(want))
    assert seen == {True, False}


def test_quasistatic_iterate_without_host_synchronisation(amd, capsys):
    """IrsLqrQuasistatic.iterate, quiet: all descents enqueued, one read-back, the reference's bookkeeping replayed
    afterwards == the verbose run (one read-back per iteration), bit for bit, on device-drawn samples."""
    from examples.run_quasistatic import problem
    T, N = 12, 1500

    def make(verbose):
        q_dynamics, x0, u_traj_0, Q_dict, Qd_dict, R_dict, x_trj_d = problem(T, 0.1)
        p = amd.IrsLqrQuasistaticParameters()
        p.Q_dict, p.Qd_dict, p.R_dict = Q_dict, Qd_dict, R_dict
        p.x0, p.x_trj_d, p.u_trj_0, p.T = x0, x_trj_d, u_traj_0, T
        p.u_bounds_abs = np.array([-np.ones(4) * 0.05, np.ones(4) * 0.05])
        p.sampling = lambda u_initial, it: u_initial / (it ** 0.8)
        p.std_u_initial, p.num_samples = np.ones(4) * 0.3, N
        p.publish_every_iteration = False
        p.device_rng_seed = 3
        sol = amd.IrsLqrQuasistatic(q_dynamics, p)
        sol.verbose = verbose
        return sol

    a, b = make(False), make(True)
    ra, rb = a.iterate(4), b.iterate(4)
    capsys.readouterr()
    assert len(a.cost_all_list) == len(b.cost_all_list) == 6 and a.current_iter == b.current_iter == 5
    np.testing.assert_array_equal(np.array(a.cost_all_list), np.array(b.cost_all_list))
    for k in ("Qu", "Qu_final", "Qa", "Qa_final", "R"):
        np.testing.assert_array_equal(getattr(a, "cost_%s_list" % k), getattr(b, "cost_%s_list" % k))
    for xa, xb in zip(a.x_trj_list, b.x_trj_list):
        np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(ra[0], rb[0])
    np.testing.assert_array_equal(a.x_trj_best, b.x_trj_best)
    assert a.cost_best == b.cost_best < a.cost_all_list[0]


# ---------------------------------------------------------------- config 4's CEM baseline at full size (round 3)
def test_cem_quasistatic_box_pivoting_full_size(amd):
    """BASELINE configs[4]: CrossEntropyMethodQuasistatic on box_pivoting at T = 80 with the batch the comparison
    runs at over 8 GPUs (batch_size = N = 50 000; run_box_pivoting_cem.py:100-135): what bench.py times as
    `cem_same_budget`.  cem_rollout_quasistatic_kernel<BoxPivotExactModel> prices every candidate with the quasistatic
    eval_cost (cem_quasistatic.py:124-165: du cost from x_0[idx], terminal Qd) over an 80-step contact rollout
    -- a subset of the 50 000 costs against the oracle's rollouts; the elite selection and the refit against NumPy
    on the device's own costs (cem_quasistatic.py:181-198)."""
    from examples.run_quasistatic import box_problem
    from irs_mpc_amd import device as dev
    T, B, n_elite = 80, 50000, 2500                       # elite fraction 5 % (run_box_pivoting_cem.py:118-119)
    sd, x0, u0, Q_dict, Qd_dict, R_dict, xd = box_problem(T)
    so = orc.BoxPivotOracle(0.1)
    Q, Qd, R = sd.get_Q_from_Q_dict(Q_dict), sd.get_Q_from_Q_dict(Qd_dict), sd.get_R_from_R_dict(R_dict)
    rng = np.random.default_rng(80)
    cand = u0[None] + 0.2 * rng.normal(size=(B, T, 2))    # initial_std 0.2 (:120)
    dm = sd.dm()
    cd = dev.to_dev(cand)
    costs = dm.cem_rollout_costs_quasistatic(cd, dev.to_dev(x0), dev.to_dev(Q), dev.to_dev(Qd), dev.to_dev(R), dev.to_dev(xd))
    ch = costs.cpu().numpy()
    assert np.isfinite(ch).all()
    sel = rng.choice(B, 48, replace=False)
    idx_u = so.indices_u_into_x
    co = np.array([orc.eval_cost_quasistatic(orc.rollout(so, x0, cand[b]), cand[b], xd, Q, Qd, R, idx_u) for b in sel])
    np.testing.assert_allclose(ch[sel], co, rtol=1e-9)
    idx, u_new, std_new = dm.cem_refit(cd, costs, n_elite)
    best = np.argpartition(ch, n_elite - 1)[:n_elite]
    assert sorted(idx.cpu().numpy().tolist()) == sorted(best.tolist())
    np.testing.assert_allclose(u_new.cpu().numpy(), cand[best].mean(axis=0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(std_new.cpu().numpy(), cand[best].std(axis=0), rtol=1e-9, atol=1e-12)
    # the per-GPU share of the comparison (6 250 candidates, 312 elites): the same kernels through the host class
    p = amd.CemQuasistaticParameters()
    p.Q_dict, p.Qd_dict, p.R_dict = Q_dict, Qd_dict, R_dict
    p.x0, p.xd_trj, p.u_trj_0, p.T = x0, xd, u0, T
    p.n_elite, p.batch_size, p.initial_std = 312, 6250, 0.2 * np.ones(2)
    p.publish_every_iteration = False
    sol = amd.CrossEntropyMethodQuasistatic(sd, p)
    sol.verbose = False
    np.random.seed(1)
    sol.iterate(2)
    # (no descent asserted: on this problem CEM at this budget hovers around its initial cost -- 10 203 -> ~10 230 --
    # which is what the comparison of bench.py / the paper shows; iRS-LQR reaches ~6 000 at the same budget)
    assert len(sol.cost_all_list) == 4 and np.isfinite(sol.cost_all_list).all() and sol.current_iter == 3


# ---------------------------------------------------------------- uniform-geometry kernel on unusual geometry (round 3)
def test_uniform_geometry_kernel_on_random_states(amd):
    """csrc/smooth_ug.hip (the table-driven pass of the exact 8-row model) against the general kernel (IRS_UG=0) and the
    oracle on nominal points that are NOT a settled grasp: separated (no row active for any sample), touching, deeply
    penetrating (up to 7 rows active), with small and large command noise, one time step per point; plus the degenerate
    sizes N = 1 and T = 1.  Same statistics layout, same (A, B, c) to the f32 tolerance; bit-reproducible."""
    from irs_mpc_amd import device as dev
    from irs_mpc_amd._lib import SMOOTH_FIRST_ORDER, SMOOTH_ZERO_ORDER_B
    rng = np.random.default_rng(99)
    k, N = 24, 1200
    obj = np.stack([rng.uniform(-0.3, 0.3, k), rng.uniform(0.1, 0.7, k), rng.uniform(-1, 1, k)], 1)
    left = np.stack([rng.uniform(-2.2, -0.2, k), rng.uniform(-1.5, 0.5, k)], 1)
    right = np.stack([rng.uniform(0.2, 2.2, k), rng.uniform(-0.5, 1.5, k)], 1)
    X = np.zeros((k, 7))
    X[:, HAND.PERM] = np.hstack([obj, left, right])
    X[0, HAND.PERM] = [0.0, 2.0, 0.0, -2.5, 0.0, 2.5, 0.0]            # far away: no contact whatever the command
    U = X[:, HAND_IDX] + rng.normal(0, 0.1, (k, 4))
    sys_o = orc.PlanarHandOracle(0.1)
    dm = amd.PlanarHandDynamics(0.1).dm()
    xd, ud = dev.to_dev(X), dev.to_dev(U)
    for std in (0.02, 0.5):
        du = (std * rng.normal(size=(k, N, 4))).astype(np.float32)
        dud = dev.to_dev(du, dev.F32)
        for mode in (SMOOTH_ZERO_ORDER_B, SMOOTH_FIRST_ORDER):
            outs = {}
            for ug in ("1", "0"):
                os.environ["IRS_UG"] = ug
                try:
                    o = dm.smooth(mode, xd, ud, None, dud)
                    outs[ug] = {kk: o[kk].cpu().numpy() for kk in ("At", "Bt", "ct", "info", "sums")}
                finally:
                    os.environ.pop("IRS_UG", None)
            assert int(np.abs(outs["1"]["info"]).sum()) == 0
            np.testing.assert_array_equal(outs["1"]["At"], outs["0"]["At"])
            if mode == SMOOTH_ZERO_ORDER_B:
                np.testing.assert_allclose(outs["1"]["Bt"], outs["0"]["Bt"], **FP32_TOL)
                np.testing.assert_allclose(outs["1"]["ct"], outs["0"]["ct"], **FP32_TOL)
                xp = np.vstack([X, X[-1:]])
                _, Bo, co = orc.zero_order_B_decoupled(sys_o, xp, U, du.astype(np.float64))
                np.testing.assert_allclose(outs["1"]["Bt"], Bo, **FP32_TOL)
                np.testing.assert_allclose(outs["1"]["ct"], co, **FP32_TOL)
            else:
                # piecewise constant in the sample: the two f32 routes may put a borderline sample on different faces
                assert np.abs(outs["1"]["Bt"] - outs["0"]["Bt"]).max() < 20.0 / N
            o2 = dm.smooth(mode, xd, ud, None, dud)
            assert np.array_equal(o2["Bt"].cpu().numpy(), outs["1"]["Bt"]) and np.array_equal(o2["ct"].cpu().numpy(), outs["1"]["ct"])
    assert np.abs(outs["1"]["Bt"][0][HAND.PERM[:3]]).max() == 0.0       # the far-away point: the disc does not see the command
    # degenerate sizes
    for T1, N1 in ((1, 1), (1, 70), (3, 1)):
        du1 = (0.1 * rng.normal(size=(T1, N1, 4))).astype(np.float32)
        o = dm.smooth(SMOOTH_FIRST_ORDER, dev.to_dev(X[:T1]), dev.to_dev(U[:T1]), None, dev.to_dev(du1, dev.F32))
        xp = np.vstack([X[:T1], X[T1 - 1:T1]])
        _, B1, c1 = orc.first_order_B_decoupled(sys_o, xp, U[:T1], du1.astype(np.float64))
        assert int(o["info"].abs().sum().item()) == 0
        np.testing.assert_allclose(o["Bt"].cpu().numpy(), B1, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(o["ct"].cpu().numpy(), c1, rtol=1e-4, atol=2e-5)
