"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/irs_hip.h declares; argument validation returns error
codes without touching a GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from irs_mpc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "irs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(irs_[a-z0-9_]+)\s*\(", src)))


def test_every_header_symbol_is_exported_and_bound(lib):
    from irs_mpc_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), "libirs_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_model_registry(lib):
    n, m, k = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.irs_model_info(0, n, m, k) == 0 and (n.value, m.value, k.value) == (2, 1, 1)
    assert lib.irs_model_info(1, n, m, k) == 0 and (n.value, m.value, k.value) == (12, 4, 9)
    assert lib.irs_model_info(99, n, m, k) == -3
    assert b"unknown model" in lib.irs_last_error()
    # P = d(d+1)/2 + d n  |  n d  |  m(m+1)/2 + m n
    assert [lib.irs_sums_len(0, mode) for mode in range(3)] == [12, 6, 3]
    assert [lib.irs_sums_len(1, mode) for mode in range(3)] == [136 + 192, 192, 10 + 48]
    # contact models append sum(z): planar hand d = 11, m = 4
    assert [lib.irs_sums_len(4, mode) for mode in (0, 2)] == [66 + 77 + 11, 10 + 28 + 4]
    assert lib.irs_sums_len(0, 7) == -1
    assert lib.irs_smooth_workspace_bytes(0, 0, 30, 10000) > 0


def test_argument_validation_without_gpu(lib):
    from irs_mpc_amd._lib import dbl_array
    p = dbl_array([0.05])
    assert lib.irs_tvlqr_riccati(0, 1, 5, None, None, None, None, None, None, 0.5, None, None, None, None, None) == -1
    assert lib.irs_tvlqr_riccati(40, 1, 5, None, None, None, None, None, None, 0.5, None, None, None, None, None) == -1
    assert lib.irs_smooth_accumulate(0, p, 1, 0, 30, 0, None, None, None, None, None, None, 0, None) == -1
    assert lib.irs_dynamics_batch(0, p, 1, None, None, 0, None, None) == -1
    assert lib.irs_smooth_finalize(0, p, 1, 9, 30, 10, None, None, None, None, None, None, None, None) == -1
    assert lib.irs_last_error() != b""
    # the quasistatic entry points: null pointers, an unknown solver, half a bound pair, a workspace
    # that cannot hold the nominal steps, an analytic model where a position-controlled one is needed
    ph = dbl_array([0.1, 10.0, 1.0, 0.25, 0.5, 50.0, 25.0, 0.3, 0.2, 0.05, 0.1, 50.0])
    one = 8     # any non-null address: validation happens before anything is dereferenced
    args = [one] * 8
    assert lib.irs_quasistatic_box_descent(4, ph, 12, 10, None, *args[:7], None, None, None, None, None, None,
                                           0, 10.0, 1.6, 100, 1e-8, one, one, None, one, None) == -1
    assert lib.irs_quasistatic_box_descent(4, ph, 12, 10, *args, None, None, None, None, None, None,
                                           7, 10.0, 1.6, 100, 1e-8, one, one, None, one, None) == -1
    assert lib.irs_quasistatic_box_descent(4, ph, 12, 10, *args, None, None, one, None, None, None,
                                           0, 10.0, 1.6, 100, 1e-8, one, one, None, one, None) == -1
    assert lib.irs_quasistatic_box_descent(4, ph, 12, 10, *args, one, one, one, one, None, None,
                                           2, 10.0, 1.6, 100, 1e-8, one, one, None, one, None) == -3   # active set: one box
    assert lib.irs_quasistatic_box_lds_bytes(0, 50, 2) == 0                 # the pendulum is not position controlled
    assert 0 < lib.irs_quasistatic_box_lds_bytes(4, 50, 2) <= 160 * 1024 - 512
    assert lib.irs_quasistatic_box_lds_bytes(4, 80, 2) > 160 * 1024 - 512   # planar hand: T <= 52 on chip
    assert 0 < lib.irs_quasistatic_box_lds_bytes(5, 120, 2) <= 160 * 1024 - 512   # box pivoting: its script's horizon fits
    # the matrix-core active-set solver (3): on chip at the benchmark horizon, a workspace beyond it
    assert 0 < lib.irs_quasistatic_box_lds_bytes(4, 50, 3) <= 160 * 1024 - 512
    assert lib.irs_quasistatic_descent_workspace_bytes(4, 50, 3) == 0
    assert lib.irs_quasistatic_descent_workspace_bytes(4, 120, 0) >= 120 * (192 + 144) * 8
    assert lib.irs_quasistatic_descent_workspace_bytes(5, 120, 3) == 0            # box pivoting, T = 120: on chip
    assert lib.irs_quasistatic_box_lds_bytes(0, 50, 3) == 0 and lib.irs_quasistatic_descent_workspace_bytes(0, 50, 3) == 0
    assert lib.irs_smooth_finalize_ws(4, ph, 12, 2, 50, 100, one, one, one, one, one, one, one, one, 64, None) == -1
    assert lib.irs_cem_rollout_costs_quasistatic(4, ph, 12, 10, 0, one, one, one, one, one, one, one, None) == -1
    assert lib.irs_model_info(11, None, None, None) != 0                     # ids 0..10 are registered
    # the in-library collective step: argument validation happens before RCCL or the GPU are touched
    assert lib.irs_comm_destroy(None) == 0 and lib.irs_step_graph_destroy(None) == 0
    assert lib.irs_comm_create(None, 1, 0, None) == -1
    assert lib.irs_allreduce_sums(None, None, 0, None) == -1
    assert lib.irs_smooth_step_collective(None, None, None) == -1
    assert lib.irs_least_squares(0, 1, 10, one, one, one, one, one, None) == -1
    assert lib.irs_tvlqr_box_solve(4, ph, 12, 10, *args[:6], 0.5, one, one, 0, None, None, None, None, one, one,
                                   10.0, 1.6, 100, 1e-8, one, one, one, None) == -1   # du bounds need the position form


def test_product_does_not_import_oracle():
    """The shipped package must never route through the oracle."""
    pkg = os.path.join(ROOT, "irs_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from irs_mpc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from irs_mpc_amd import PendulumDynamics
    with pytest.raises(RuntimeError, match="needs an AMD GPU"):
        PendulumDynamics(0.05).dynamics_batch(np.zeros((1, 2)), np.zeros((1, 1)))


def test_compat_shim_resolves_the_reference_scripts_imports():
    """f4: examples/compat lets the TEXT of the reference's analytic example scripts run with only sys.path
    changed -- every `from irs_lqr... import ...` / `from <system>_dynamics import ...` line of those scripts must
    resolve to a device-backed twin.  The names are listed here (read off examples/{pendulum,quadrotor,bicycle,
    three_cart}/*.py of the reference); when the reference tree is present (this container, not the GPU box) its
    scripts are scanned as well so that the list cannot go stale."""
    import importlib
    import os
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = os.path.join(root, "examples", "compat")
    wanted = {("irs_lqr.all", n) for n in ("IrsLqrParameters", "IrsLqrZeroOrder", "IrsLqrFirstOrder", "IrsLqrExact",
                                            "CemParameters", "CrossEntropyMethod")}
    wanted |= {("irs_lqr.irs_lqr", "IrsLqr"), ("pendulum_dynamics", "PendulumDynamics"),
               ("quadrotor_dynamics", "QuadrotorDynamics"), ("bicycle_dynamics", "BicycleDynamics"),
               ("three_cart_dynamics", "ThreeCartDynamics"),
               ("irs_lqr.irs_lqr_quasistatic", "IrsLqrQuasistatic"), ("irs_lqr.irs_lqr_quasistatic", "IrsLqrQuasistaticParameters"),
               ("irs_lqr.tv_lqr", "solve_tvlqr"), ("irs_lqr.tv_lqr", "get_solver"),
               ("irs_lqr.dynamical_system", "DynamicalSystem")}
    ref = "/root/reference/examples"
    if os.path.isdir(ref):
        pat = re.compile(r"^from\s+(irs_lqr(?:\.\w+)*|\w+_dynamics)\s+import\s+(.+)$")
        for sysname in ("pendulum", "quadrotor", "bicycle", "three_cart"):
            for fn in sorted(os.listdir(os.path.join(ref, sysname))):
                if not fn.endswith(".py") or fn.endswith("_dynamics.py") or "animation" in fn or "drake" in fn \
                        or "simulation" in fn or fn == "pendulum_nn.py":
                    continue
                for line in open(os.path.join(ref, sysname, fn)):
                    mm = pat.match(line.strip())
                    if mm:
                        wanted |= {(mm.group(1), nm.strip()) for nm in mm.group(2).split(",")}
    sys.path.insert(0, shim)
    try:
        for mod in [m for m in sys.modules if m == "irs_lqr" or m.startswith("irs_lqr.")]:
            del sys.modules[mod]
        import irs_mpc_amd
        for mod, name in sorted(wanted):
            m = importlib.import_module(mod)
            assert hasattr(m, name), (mod, name)
            assert getattr(m, name).__module__.startswith("irs_mpc_amd"), (mod, name)
        assert importlib.import_module("irs_lqr.all").IrsLqrZeroOrder is irs_mpc_amd.IrsLqrZeroOrder
    finally:
        sys.path.remove(shim)
        for mod in [m for m in sys.modules if m == "irs_lqr" or m.startswith("irs_lqr.") or m.endswith("_dynamics")]:
            del sys.modules[mod]
