#!/bin/bash
# Regenerates the cost curves of the reference's example scripts on the GPU (examples/run.py, examples/run_quasistatic.py:
# the twins of the scripts; examples/compat/run_script.py runs the scripts' own text) into gpurun_out/curves/, and
# prints how each compares with the reference's result file where tests/golden/ holds one.
# Copy the directory to profiles/curves/ afterwards (tracked).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/curves
mkdir -p $OUT
cd $R
for spec in "pendulum exact" "pendulum zero_order" "pendulum first_order" "quadrotor exact" "bicycle exact" "bicycle_hard exact"; do
  set -- $spec
  timeout -k 10 300 python examples/run.py $1 $2 --quiet --csv $OUT/$1_$2.csv > $OUT/$1_$2.log 2>&1 || echo "FAILED $spec"
done
for mode in exact first_order zero_order_B; do
  timeout -k 10 300 python examples/run_quasistatic.py planar_hand_spin irs_lqr --T 30 --N 1000 --iters 20 --gradient-mode $mode --quiet \
      --csv $OUT/planar_hand_spin_$mode.csv > $OUT/planar_hand_spin_$mode.log 2>&1 || echo "FAILED spin $mode"
done
timeout -k 10 300 python examples/run_quasistatic.py box_pushing irs_lqr --T 50 --N 1000 --iters 20 --gradient-mode exact --quiet \
    --csv $OUT/box_pushing_exact.csv > $OUT/box_pushing_exact.log 2>&1 || echo "FAILED box_pushing"
python - <<PY
import numpy as np, os
out, gold = "$OUT", "$R/tests/golden"
rows = []
for name, ref in (("pendulum_exact", "pendulum_exact"), ("quadrotor_exact", "quadrotor_exact"), ("bicycle_exact", "bicycle_easy_exact"),
                  ("bicycle_hard_exact", "bicycle_hard_exact"), ("pendulum_zero_order", "pendulum_zero_order"),
                  ("pendulum_first_order", "pendulum_first_order"), ("planar_hand_spin_exact", "planar_hand_spin_exact"),
                  ("planar_hand_spin_first_order", "planar_hand_spin_first_order"),
                  ("planar_hand_spin_zero_order_B", "planar_hand_spin_zero_order_B"), ("box_pushing_exact", "box_pushing_exact")):
    p, g = os.path.join(out, name + ".csv"), os.path.join(gold, ref + ".csv")
    if not os.path.exists(p):
        rows.append("| %s | missing | | | |" % name)
        continue
    a = np.atleast_1d(np.loadtxt(p))
    b = np.atleast_1d(np.loadtxt(g)) if os.path.exists(g) else None
    if b is None:
        rows.append("| %s | %d entries | - | %.6g | %.6g |" % (name, len(a), a[0], a[-1]))
        continue
    k = min(len(a), len(b))
    rel = np.abs(a[:k] - b[:k]) / np.abs(b[:k])
    rows.append("| %s | %d / %d entries | first %.1e, max %.1e | %.6g vs %.6g | %.6g vs %.6g |" % (
        name, len(a), len(b), rel[0], rel.max(), a[0], b[0], a[k - 1], b[k - 1]))
txt = "| curve | length (ours / reference) | relative difference | first entry (ours vs reference) | last common entry |\n|---|---|---|---|---|\n" + "\n".join(rows)
open(os.path.join(out, "summary.md"), "w").write(txt + "\n")
print(txt)
PY
rm -f $OUT/*.log
