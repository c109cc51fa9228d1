#!/bin/bash
# MFMA evidence for configs[2] (quadrotor): kernel stats + matrix-core counters.
TAG=${1:-r01}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof2_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/bench_config2.py > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/tools/bench_config2.py > $OUT/pmc.log 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
res = {"kernel_stats": [], "pmc": {}}
f = glob.glob(out + "/stats/*/*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f[0])):
        if "irs" in r["Name"] or "kernel" in r["Name"]:
            res["kernel_stats"].append({"kernel": r["Name"][:110], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])})
f = glob.glob(out + "/pmc/*/*counter_collection.csv")
if f:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "descent_kernel" in k or "smooth_kernel" in k:
            d[k[:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        res["pmc"][k] = {c: sum(x) / len(x) for c, x in v.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
tail -1 $OUT/stats.log
