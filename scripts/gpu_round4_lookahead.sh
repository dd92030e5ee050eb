#!/bin/bash
set -o pipefail
O=gpurun_out/round4_lookahead.log
: > $O
( for mode in 0 1; do echo "== EKF_PN_MODE=$mode"; EKF_PN_MODE=$mode EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_tuning.so timeout -k 10 600 python bench.py --force-sharded --steps 320 --warmup 64 --no-cpu-baseline --no-other-configs --batch2 0; done ) 2>&1 | tee -a $O
