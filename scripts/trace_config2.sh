#!/bin/bash
# GPU box: rocprofv3 kernel trace of configs[1] in one association mode; prints per-kernel averages and the gaps between
# consecutive kernels of the last scans (scripts/analyze_trace.py).  Usage: scripts/trace_config2.sh <tag> [bench_config2 args]
set -e -o pipefail
TAG=${1:-c2t}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/bench_config2.py "$@" > $OUT/under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 $REPO/scripts/analyze_trace.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) | tee $OUT/timeline.txt
rm -rf $OUT/stats
