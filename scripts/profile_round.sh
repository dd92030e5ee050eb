#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + PMC passes for HBM traffic (and, with MFMA=1, the
# matrix-pipe counters of the batched flush).  Usage: scripts/profile_round.sh <tag>   -> writes gpurun_out/<tag>/...
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), never a wrapper; counters are collected in their own passes.
set -e -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $REPO/scripts/show_bench.py $OUT/bench.json
ARGS="--steps 160 --warmup 64 --deferred-steps 640 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err
ARGS="--steps 64 --warmup 32 --deferred-steps 256 --no-cpu-baseline --no-other-configs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
if [ "${MFMA:-0}" = "1" ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_mfma.err
fi
find $OUT -name '*.csv' | head -20
