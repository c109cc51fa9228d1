"""The CPU leg of bench.py (no GPU): the bounded oracle sample on one core and on a pool of worker
processes -- the analogue of the reference's ZMQ workers (irs_lqr_quasistatic.py:245-263)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_single_core_and_pool():
    import bench
    w = bench.Workload("planar_hand", T=6, mode=None, host_only=True)
    out = bench.cpu_baseline(w, 300, seconds=0.5, pool_cores=2)
    assert out["kind"] == "port" and out["cores"] == 1 and out["value"] > 0
    assert "oracle.zero_order_B_decoupled" in out["sample"]
    pool = out["pool"]
    assert "error" not in pool, pool
    assert pool["cores"] == 2 and pool["value"] > 0
    # the first-order twin and the pendulum workload use their own oracle functions
    w1 = bench.Workload("planar_hand", T=4, mode="first_order", host_only=True)
    part, Ns, what = bench._cpu_problem(w1, 50)
    part(0, 4)
    assert what == "oracle.first_order_B_decoupled" and Ns == 50
    w2 = bench.Workload("pendulum", T=5, host_only=True)
    part, Ns, what = bench._cpu_problem(w2, 64)
    part(1, 4)
    assert what == "oracle.zero_order_TV" and Ns == 64
