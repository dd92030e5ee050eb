#!/bin/bash
# GPU box, tuning build: the F64 MFMA flush at fewer wavefronts per SIMD (a narrower window of tile addresses in flight; the pure-copy probe
# scripts/probes/tile_stream_shapes.hip streams 64 x 64 items faster at 2-4 wavefronts per SIMD than at 5-8).  profiles/round4_tuning.md 58.
# Usage: scripts/ab_flush_wpe.sh <tag> [landmarks] ["pairs list"]
set -e -o pipefail
TAG=$1; LM=${2:-10000}; PAIRS=${3:-"2 8 12 16 20"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export EKF_LIB_PATH=$REPO/ekf_slam_amd/libekfslam_tuning.so
run() { local label=$1; shift
  env "$@" timeout -k 10 120 python3 $REPO/scripts/time_flush.py --landmarks $LM --batch $B --batches 12 --label "$label" 2>/dev/null | grep '^{' >> $OUT/ab.jsonl
}
for round in 1 2; do for B in $PAIRS; do
  run prod X=0
  for W in 2 3 4 5; do run "h64wpe$W" EKF_FLUSH_HALF_MAX=64 EKF_FLUSH_HALF_WPE=$W; done
  run c128wpe3 EKF_FLUSH_HALF_MAX=0 EKF_FLUSH_WAVES=3 EKF_FLUSH_CHUNK=4
  run c128wpe4 EKF_FLUSH_HALF_MAX=0
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.jsonl"):
    r = json.loads(l)
    print(r["batch"], r["label"], r["kernel"], r["flush_ms"], r["frac"], r["steps_per_s"], r["digest"][0])
PY
