#!/bin/bash
# A/B on one box, interleaved: the 64-pair F32-arithmetic pass at 40 000 landmarks inside the library (scripts/bench_config5.py), strip
# form (flush32_pipe.h) against the per-item form (flush32_mfma.h); tuning build (EKF_PASS_STRIP is a constant 1 in the product library).
set -o pipefail
O=gpurun_out/ab_strip.log
: > $O
export EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_tuning.so
for r in 1 2 3; do
  for v in 1 0; do
    echo "== EKF_PASS_STRIP=$v round $r" | tee -a $O
    EKF_PASS_STRIP=$v timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --batch 64 --storage f32_mixed 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('   %.1f update-steps/s, pass %.4f ms (%s, %d pairs, %d launches)' % (d['value'], r['avg_launch_ms'], r['kernel'], r['pairs_per_launch'], r['launches']))
" | tee -a $O
  done
done
