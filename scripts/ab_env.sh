#!/bin/bash
# GPU box: interleaved A/B of one EKF_* environment knob on the one-pair pass (scripts/time_flush.py).
# The knobs exist only in the tuning build: make -C ekf_slam_amd/csrc tuning (libekfslam_tuning.so, -DEKF_TUNING).
# Usage: scripts/ab_env.sh <tag> <VAR> "<values>" "<landmark list>" [batch]
set -e -o pipefail
TAG=$1; VAR=$2; VALS=$3; LMS=${4:-10000}; B=${5:-1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for round in 1 2 3; do
  for L in $LMS; do
    for V in $VALS; do
      env EKF_LIB_PATH=$REPO/ekf_slam_amd/libekfslam_tuning.so $VAR=$V timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $L --batch $B --batches $([ $B = 1 ] && echo 256 || echo 12) --label "$VAR=$V" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
    done
  done
done
python3 - <<PY
import json, collections
rows = collections.defaultdict(list)
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    rows[(r["landmarks"], r["batch"], r["label"])].append((r["flush_ms"], r["steps_per_s"]))
for k in sorted(rows):
    print(k, " ".join("%.4f/%d" % v for v in rows[k]))
PY
