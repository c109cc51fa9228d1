"""The CPU leg of bench.py (no GPU): the bounded oracle sample on one core and on a pool of worker
processes -- the analogue of the reference's ZMQ workers (irs_lqr_quasistatic.py:245-263) -- and the
roofline bookkeeping of the JSON line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_single_core_and_pool(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "usable_cpus", lambda: 2)
    w = bench.Workload("planar_hand", T=6, mode=None, host_only=True)
    out = bench.cpu_baseline(w, 300, budget_s=2.0, reps=5)
    assert out["kind"] == "port" and out["cores"] == 1 and out["value"] > 0
    assert "oracle.zero_order_B_decoupled" in out["sample"] and "median of 5" in out["sample"]
    pool = out["pool"]
    assert "error" not in pool, pool
    assert pool["cores"] == 2 and pool["value"] > 0 and "median of 5" in pool["sample"]
    # the first-order twin and the pendulum workload use their own oracle functions
    w1 = bench.Workload("planar_hand", T=4, mode="first_order", host_only=True)
    part, what = bench._cpu_problem(w1, 50)
    part(0, 4)
    assert what == "oracle.first_order_B_decoupled"
    w2 = bench.Workload("pendulum", T=5, host_only=True)
    part, what = bench._cpu_problem(w2, 64)
    part(1, 4)
    assert what == "oracle.zero_order_TV"


def test_flop_model_counts_per_sample_work_only():
    """roofline.frac must follow from the printed formula: the hoisted contact geometry is not counted, the
    exact solver's data-dependent steps are not counted (a lower bound, said so in the formula)."""
    import bench
    f_pgs, s_pgs = bench.contact_flops_per_sample("pgs", 50, False)
    assert f_pgs == 64 + 50 * 152 + 470 + 126 + 76 and "1500" not in s_pgs
    f_ex, s_ex = bench.contact_flops_per_sample("exact", 50, False)
    assert f_ex == 64 + 32 * 152 + 316 + 128 + 126 + 76 and "lower bound" in s_ex
    f1, _ = bench.contact_flops_per_sample("exact", 50, True)
    assert f1 - f_ex == 1208 - 76


def test_adopted_profile_is_committed_and_stamped():
    """bench.py quotes roofline.traffic / valu_issue_slots from profiles/pmc_latest.json: the file must name the
    round profile it was adopted from, that profile's summaries must be under profiles/ (tracked), and the metric's
    kernel must carry the counters the formulas use, with its average duration from the single-workload pass."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    meta = d["_meta"]
    assert meta["commit"] and meta["tag"]
    for suffix in ("_kernel_stats.csv", "_pmc_summary.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", meta["tag"] + suffix)), suffix
    e = d["planar_hand_exact_zeroB_T50_N10000"]
    assert e["SQ_INSTS_VALU_raw_avg"] > 5e6 and e["FETCH_SIZE_raw_avg"] > 0 and e["WRITE_SIZE_raw_avg"] > 0
    assert "single-workload" in e["kernel_avg_source"] and 2e4 < e["kernel_avg_ns"] < 1e5
    # share of the chip's VALU capacity: a wave64 instruction executes in 2 cycles on the SIMD-32 (bench.ISSUE_CYCLES)
    slots = e["SQ_INSTS_VALU_raw_avg"] * 2 / 1024 / 2.4 / e["kernel_avg_ns"]
    assert 0.1 < slots < 1.0, slots
    # round 3: the LDS side of the uniform-geometry kernel and the matrix-core counters of the descent are kept too
    assert e["SQ_INSTS_LDS_raw_avg"] > 0 and "SQ_LDS_BANK_CONFLICT_raw_avg" in e
    dsc = d["planar_hand_exact_ctrlbox_descent_T50"]
    assert dsc["SQ_INSTS_MFMA_raw_avg"] > 1e3 and dsc["SQ_VALU_MFMA_BUSY_CYCLES_raw_avg"] > 0


def test_flop_model_of_the_uniform_geometry_kernel():
    """smooth_ug_kernel (csrc/smooth_ug.hip): what every sample does -- r, 6 sweeps, 4 table-row iterations, the
    statistics; the parked samples' dual active-set steps are not counted (a lower bound, said so)."""
    import bench
    f, s = bench.contact_flops_per_sample("ug", 50, False)
    assert f == 64 + 6 * 160 + 4 * 136 + 88 and "lower bound" in s
    f1, _ = bench.contact_flops_per_sample("ug", 50, True)
    assert f1 == f - 88 + 8
    assert bench.ISSUE_CYCLES == 2 and bench.ISSUE_CYCLES_ONE_WAVE == 4
