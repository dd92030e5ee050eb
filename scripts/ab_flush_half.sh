#!/bin/bash
# GPU box, tuning build: the F64 MFMA flush with 64-column work items (five wavefronts per SIMD) beside the 128-column ones:
# EKF_FLUSH_HALF_MAX = the largest pair count that takes the 64-column items (0: never, 64: always; production: 12).
# Usage: scripts/ab_flush_half.sh <tag> [landmarks] ["pairs list"]
set -e -o pipefail
TAG=$1; LM=${2:-10000}; PAIRS=${3:-"2 8 20 32"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export EKF_LIB_PATH=$REPO/ekf_slam_amd/libekfslam_tuning.so
for round in 1 2; do for B in $PAIRS; do for H in 0 12 64; do
  EKF_FLUSH_HALF_MAX=$H timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $LM --batch $B --batches 12 --label "half_max$H" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
done; done; done
python3 - <<PY
import json
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    print(r["batch"], r["label"], r["kernel"], r["flush_ms"], r["frac"], r["steps_per_s"], r["digest"][0])
PY
