#!/bin/bash
# GPU box: rocprofv3 kernel stats of scripts/time_sharded_step.py (unsharded / sharded hinted / sharded unhinted legs)
set -e -o pipefail
TAG=${1:-shs}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/time_sharded_step.py "$@" > $OUT/under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 $REPO/scripts/analyze_trace.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) 30 > $OUT/timeline.txt
rm -rf $OUT/stats
head -14 $OUT/timeline.txt
