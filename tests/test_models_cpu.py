"""Host-side checks of the device functors (csrc/models.hpp is __host__ __device__: no GPU needed)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_quadrotor_hand_derived_jacobian_matches_dual_numbers(tmp_path):
    """QuadrotorModel::step_jac + expand_jac (the compact, hand-derived Jacobian the first-order sample pass sums) ==
    the forward-mode dual-number Jacobian of the same step (examples/quadrotor/quadrotor_dynamics.py:40-77,
    jacobian_xu :136-138), f64, 2000 random states: 1e-11 relative; the step values are identical."""
    exe = str(tmp_path / "qjc")
    src = os.path.join(ROOT, "tests", "helpers", "quadrotor_jac_check.hip")
    r = subprocess.run([HIPCC, "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "irs_mpc_amd", "csrc"),
                        "-o", exe, src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "max rel |J_dual - J_hand|" in r.stdout
