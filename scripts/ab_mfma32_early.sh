export EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_tuning.so
for B in 1 12 32 64; do NB=$([ $B = 1 ] && echo 24 || echo 6)
 for V in "0 0 4 4" "0 2 4 4" "0 2 4 3" "1 2 4 3" "1 2 4 2" "1 2 4 4"; do set -- $V
  EKF_MFMA32_EARLY=$1 EKF_MFMA32_RG=$2 EKF_MFMA32_CHUNK=$3 EKF_MFMA32_WPE=$4 python scripts/time_flush.py --landmarks 40000 --batch $B --batches $NB --storage f32_mixed 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print(\"$V |\", r[\"batch\"], r[\"kernel\"], r[\"flush_ms\"], r[\"frac\"], r[\"steps_per_s\"], r[\"digest\"][0])"
 done; done
