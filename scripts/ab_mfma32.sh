#!/bin/bash
# GPU box: sweep of the f32-arithmetic pass (k_flush_mfma32, cfg.pass_arith = EKF_ARITH_F32) over row groups per wavefront, operand
# chunk size and waves per SIMD (COMBOS="rg,chunk,wpe ..."; rg 0 = the production choice), tuning build only.  Usage: scripts/ab_mfma32.sh <tag> [landmarks] ["pairs list"]
set -e -o pipefail
TAG=$1; LM=${2:-40000}; PAIRS=${3:-"1 12 32 64"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export EKF_LIB_PATH=$REPO/ekf_slam_amd/libekfslam_tuning.so
for B in $PAIRS; do
  NB=$([ $B = 1 ] && echo 32 || echo 6)
  timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $LM --batch $B --batches $NB --storage f32 --label "f64-arith" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
  for V in ${COMBOS:-0,4,4 1,4,4 1,4,6 1,8,4 2,4,3 2,4,4}; do
    IFS=, read RG CH WPE <<< "$V"
    EKF_MFMA32_RG=$RG EKF_MFMA32_CHUNK=$CH EKF_MFMA32_WPE=$WPE timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $LM --batch $B --batches $NB --storage f32_mixed --label "rg$RG ch$CH w$WPE" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
  done
done
python3 - <<PY
import json
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    print(r["batch"], r["label"], r["kernel"], r["flush_ms"], r["frac"], r["steps_per_s"])
PY
