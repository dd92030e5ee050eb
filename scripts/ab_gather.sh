#!/bin/bash
# A/B the gather kernel's rocprofv3 average duration on ONE box: [BATCH=32 BATCHES=12] scripts/ab_gather.sh <outdir> "label|lib-or--" ...
OUT=$1; shift
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do
  for v in "$@"; do
    IFS='|' read -r label lib <<< "$v"
    if [ "$lib" != "-" ]; then export EKF_LIB_PATH=$REPO/$lib; else unset EKF_LIB_PATH; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$label.$r -- python3 $REPO/scripts/time_flush.py --batch ${BATCH:-32} --batches ${BATCHES:-12} --label $label > /dev/null 2>&1
    f=$(find $OUT/$label.$r -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$label" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_gather" in r["Name"]:
        print("%-12s k_gather calls %s avg %.0f ns min %s max %s" % (sys.argv[2], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  done
done
