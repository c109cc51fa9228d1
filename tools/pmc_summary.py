"""Condenses the rocprofv3 passes of tools/profile_round.sh into small files for profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out_dir, tag = sys.argv[1], sys.argv[2]
res = {}
lines = []
for N in (10000, 1000000):
    key = "pendulum_zero_T30_N%d" % N
    res[key] = {}
    # kernel stats
    f = glob.glob("%s/stats_N%d/*/*kernel_stats.csv" % (out_dir, N))
    if f:
        for r in csv.DictReader(open(f[0])):
            lines.append("N=%d,%s,%s,%s,%s,%s" % (N, r["Name"].replace(",", ";")[:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
            if "smooth_kernel" in r["Name"]:
                res[key]["smooth_kernel_avg_ns"] = float(r["AverageNs"])
                res[key]["smooth_kernel_calls"] = int(r["Calls"])
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = glob.glob("%s/%s_N%d/*/*counter_collection.csv" % (out_dir, sub, N))
        if not f:
            continue
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == ctr:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in vals.items():
            if "smooth_kernel" in k:
                res[key][ctr + "_raw_avg"] = sum(v) / len(v)
                res[key][ctr + "_n"] = len(v)
json.dump(res, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
open(os.path.join(out_dir, "kernel_stats.csv"), "w").write("workload,kernel,calls,avg_ns,min_ns,max_ns\n" + "\n".join(lines) + "\n")
print(json.dumps(res, indent=1))
