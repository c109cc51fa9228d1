#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_quad; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $R/tools/tune_smooth.py quadrotor zero 50 100000 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/tools/tune_smooth.py quadrotor zero 50 100000 > $OUT/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for sub in ("p1", "p2"):
    fs = glob.glob("gpurun_out/pmc_quad/%s/*/*counter_collection.csv" % sub)
    if not fs: print(sub, "no csv"); continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "smooth_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items(): print(sub, k, "avg %.4g" % (sum(v) / len(v)), "n", len(v))
PY
tail -3 $OUT/p1.log | cut -c1-200
