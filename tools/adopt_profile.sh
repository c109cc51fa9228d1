#!/bin/bash
# Adopt the summaries of a tools/profile_round.sh run into profiles/ (tracked), stamped with the commit they
# were taken at: bench.py's roofline.traffic / valu_issue_slots read profiles/pmc_latest.json and quote the stamp.
#   tools/adopt_profile.sh r02a
set -e
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/prof_$TAG
cp $SRC/kernel_stats.csv $R/profiles/${TAG}_kernel_stats.csv
python3 - "$SRC/pmc_summary.json" "$R/profiles/${TAG}_pmc_summary.json" "$R/profiles/pmc_latest.json" "$TAG" "$(git -C $R rev-parse --short HEAD)" <<'PY'
import json, sys
src, dst, latest, tag, commit = sys.argv[1:6]
d = json.load(open(src))
d["_meta"] = {"tag": tag, "commit": commit, "how": "tools/profile_round.sh: separate rocprofv3 --pmc passes of bench.py"}
for p in (dst, latest):
    json.dump(d, open(p, "w"), indent=1)
print("adopted", tag, "at", commit, "->", dst)
PY
