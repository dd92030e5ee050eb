#!/bin/bash
# GPU box, tuning build: the F32-arithmetic pass -- the production choice (rg 0) against variants: "early rg chunk wpe".
export EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_tuning.so
for round in 1 2; do for B in ${PAIRS:-12 32 64}; do
 for V in ${VARIANTS:-"0,0,4,4" "1,2,4,3" "1,2,8,3" "1,2,8,2"}; do IFS=, read E RG CH W <<< "$V"
  EKF_MFMA32_EARLY=$E EKF_MFMA32_RG=$RG EKF_MFMA32_CHUNK=$CH EKF_MFMA32_WPE=$W python scripts/time_flush.py --landmarks 40000 --batch $B --batches 6 --storage f32_mixed 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print(\"$V |\", r[\"batch\"], r[\"kernel\"], r[\"flush_ms\"], r[\"frac\"], r[\"steps_per_s\"], r[\"digest\"][0])"
 done; done; done
