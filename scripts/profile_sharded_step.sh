#!/bin/bash
# GPU box: the sharded form of the as-written step on one GPU (scripts/time_sharded_step.py), plain and under rocprofv3 kernel stats.
# Usage: scripts/profile_sharded_step.sh <tag>
set -e -o pipefail
TAG=${1:-shstep}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/scripts/time_sharded_step.py > $OUT/plain.json 2> $OUT/plain.err
cat $OUT/plain.json
python3 $REPO/scripts/time_sharded_step.py --landmarks 2000 >> $OUT/plain.json 2>> $OUT/plain.err
tail -1 $OUT/plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/time_sharded_step.py --steps 256 > $OUT/under_rocprof.json 2> $OUT/rocprof.err
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp $F $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv
