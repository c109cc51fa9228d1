"""Condenses the rocprofv3 passes of tools/profile_round.sh into small files for profiles/:
per-kernel call counts / average durations, and per-kernel averages of every PMC counter."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]


SIZES = {"PlanarHandExact": ("planar_hand_exact", 50, 10000), "PlanarHand": ("planar_hand", 50, 10000),
         "BoxPivotExact": ("box_pivoting_exact", 80, 6250), "BoxPivot": ("box_pivoting", 80, 6250),
         "Quadrotor": ("quadrotor", 50, 10000), "Pendulum": ("pendulum", 30, 10000)}


def short(name):
    """Key of the benchmarked kernels: <workload>[_exact]_<mode>[_rng][_unfused]_T.._N.. (bench.py's roofline()
    looks up the supplied-samples, fused variants; the device-RNG variants are the ones of the iLQR loop).
    T and N are those of the profiled command (tools/profile_round.sh), not read from the trace."""
    import re
    m = re.search(r"smooth(?:_ug)?_kernel<(?:\(anonymous namespace\)::)?(\w+)[,;] (\d)[,;] (true|false)[,;] (true|false)", name)
    if m:
        model, mode, rng, fuse = m.group(1), int(m.group(2)), m.group(3) == "true", m.group(4) == "true"
        tag = {0: "zero", 1: "first", 2: "zeroB"}[mode] + ("_rng" if rng else "") + ("" if fuse else "_unfused")
        for k in ("PlanarHandExact", "PlanarHand", "BoxPivotExact", "BoxPivot", "Quadrotor", "Pendulum"):
            if k in model:
                w, T, N = SIZES[k]
                if k == "PlanarHandExact" and mode in (1, 2) and "smooth_ug_kernel" not in name:
                    w += "_general"         # the general contact kernel (IRS_UG=0 sub-report); the default is smooth_ug_kernel
                return "%s_%s_T%d_N%d" % (w, tag, T, N)
        return None
    if "ctrlbox" in name and ("descent" in name or "mfma_kernel" in name):
        for k in ("PlanarHandExact", "PlanarHand", "BoxPivotExact", "BoxPivot"):
            if k in name:
                return "%s_ctrlbox_descent_T%d" % (SIZES[k][0], SIZES[k][1])
        return None
    if "descent_kernel" in name:
        return "quadrotor_descent_T50" if "Quadrotor" in name else "pendulum_descent_T30"
    return None


res = collections.defaultdict(dict)
prev = os.path.join(out_dir, "pmc_summary.json")
if os.path.exists(prev):                    # a later pass of the same round adds to what is there
    for k, v in json.load(open(prev)).items():
        res[k].update(v)
lines = []
f = glob.glob("%s/stats/*/*kernel_stats.csv" % out_dir)
if f:
    for r in csv.DictReader(open(f[0])):
        lines.append("%s,%s,%s,%s,%s" % (r["Name"].replace(",", ";")[:140], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
        k = short(r["Name"])
        if k:
            res[k]["kernel_avg_ns"] = float(r["AverageNs"])
            res[k]["kernel_calls"] = int(r["Calls"])
            if "smooth_kernel" in r["Name"] or "smooth_ug_kernel" in r["Name"]:
                res[k]["smooth_kernel_avg_ns"] = float(r["AverageNs"])
for d in sorted(glob.glob("%s/pmc_*" % out_dir)):
    if not os.path.isdir(d):
        continue
    ctr = os.path.basename(d)[4:]
    vals = collections.defaultdict(list)
    for fn in glob.glob("%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") == ctr:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for name, v in vals.items():
        k = short(name)
        if k:
            res[k][ctr + "_raw_avg"] = sum(v) / len(v)
            res[k][ctr + "_n"] = len(v)
# one workload per process (--no-secondary): the average launch duration of ITS kernel.  (In the default run the
# metric's kernel also serves the N = 1e3 and 1e5 points of the sweep: that average mixes three sizes.)
for f in sorted(glob.glob("%s/stats_*/*/*kernel_stats.csv" % out_dir)):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k and ("smooth_kernel" in r["Name"] or "smooth_ug_kernel" in r["Name"]) and "_rng" not in k:
            res[k]["kernel_avg_ns"] = float(r["AverageNs"])
            res[k]["smooth_kernel_avg_ns"] = float(r["AverageNs"])
            res[k]["kernel_calls"] = int(r["Calls"])
            res[k]["kernel_avg_source"] = "single-workload stats pass (--no-secondary)"
json.dump(res, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
if lines:
    open(os.path.join(out_dir, "kernel_stats.csv"), "w").write("kernel,calls,avg_ns,min_ns,max_ns\n" + "\n".join(lines) + "\n")
print(json.dumps(res, indent=1))
