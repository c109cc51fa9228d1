"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) duration stats."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    k = (name[:60], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"])
    d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-60s %8s %5s %5s %5s %6s %9s %9s %9s" % ("kernel", "gridX", "gridY", "wg", "vgpr", "calls", "avg_us", "med_us", "min_us"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-60s %8s %5s %5s %5s %6d %9.2f %9.2f %9.2f" % (k + (len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3)))
