#!/bin/bash
# usage: bench_async.sh "16 32" "0 32 64"   -> async-flush bench for each batch and number of CUs kept free for the gather chain
for r in ${2:-32}; do
for b in ${1:-32}; do
EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_tuning.so EKF_ASYNC_RESERVE_CUS=$r python bench.py --async-flush --batch $b --steps $((b*10)) --warmup $((b*2)) --no-cpu-baseline --no-other-configs 2>/tmp/a.err > /tmp/a.json || { tail -3 /tmp/a.err; exit 1; }
python - $b $r <<'PY'
import json,sys
d=json.load(open('/tmp/a.json')); print("async batch", sys.argv[1], "reserve", sys.argv[2], round(d["value"]), round(d["roofline"]["avg_launch_ms"],4), d["config"]["state_digest"][0], flush=True)
PY
done
done
