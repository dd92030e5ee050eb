#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + PMC passes for HBM traffic.
# Usage: scripts/profile_round.sh <tag>   -> writes gpurun_out/<tag>/...
set -e -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --steps 160 --warmup 64 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 64 --warmup 32 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 64 --warmup 32 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
find $OUT -name '*.csv' | head -20
