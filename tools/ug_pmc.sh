#!/bin/bash
# PMC passes (one counter per run) over tests/tools/ug_time.py for the uniform-geometry kernel: tools/ug_pmc.sh TAG N
TAG=${1:-ug}
N=${2:-100000}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA; do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $R/tests/tools/ug_time.py --zero $N > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
cd $R
python3 - <<PY
import csv, glob, collections
res = collections.OrderedDict()
for d in sorted(glob.glob("$OUT/*/")):
    ctr = d.rstrip("/").split("/")[-1]
    vals = []
    for fn in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if r.get("Counter_Name") == ctr and "smooth_ug_kernel" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if vals:
        res[ctr] = sum(vals) / len(vals)
open("$OUT/summary.txt", "w").write("\n".join("%s %.4g" % kv for kv in res.items()) + "\n")
print(open("$OUT/summary.txt").read())
PY
rm -rf $OUT/*/
