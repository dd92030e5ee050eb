#!/bin/bash
# GPU box, tuning build: the F64 MFMA flush with 128-row work items (eight wavefronts per workgroup, G staged once per 128 rows) beside
# the production 64-row ones, and ablations of both shapes (EKF_FLUSH_ABL=1: no matrix work; 2: no operand staging either = the item
# shape as a plain copy).  profiles/round4_tuning.md 58.   Usage: scripts/ab_flush_waves.sh <tag> [landmarks] ["pairs list"]
set -e -o pipefail
TAG=$1; LM=${2:-10000}; PAIRS=${3:-"16 20 24 32"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export EKF_LIB_PATH=$REPO/ekf_slam_amd/libekfslam_tuning.so
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $LM --batch $B --batches 12 --label "$label" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
}
for round in 1 2; do for B in $PAIRS; do
  run prod X=0
  run w8c4 EKF_FLUSH_WAVES=8 EKF_FLUSH_CHUNK=4
  run w8c8 EKF_FLUSH_WAVES=8 EKF_FLUSH_CHUNK=8
  if [ $round = 1 ]; then
    run w4abl1 EKF_FLUSH_WAVES=4 EKF_FLUSH_CHUNK=4 EKF_FLUSH_ABL=1
    run w4abl2 EKF_FLUSH_WAVES=4 EKF_FLUSH_CHUNK=4 EKF_FLUSH_ABL=2
    run w8abl1 EKF_FLUSH_WAVES=8 EKF_FLUSH_CHUNK=4 EKF_FLUSH_ABL=1
    run w8abl2 EKF_FLUSH_WAVES=8 EKF_FLUSH_CHUNK=4 EKF_FLUSH_ABL=2
  fi
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    print(r["batch"], r["label"], r["kernel"], r["flush_ms"], r["frac"], r["steps_per_s"], r["digest"][0])
PY
