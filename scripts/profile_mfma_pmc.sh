#!/bin/bash
# Run on the GPU box (through gpurun): matrix-core and LDS counters of the batched flush.  Usage: scripts/profile_mfma_pmc.sh <tag>
set -e -o pipefail
TAG=${1:-mfma_pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/p2.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py --steps 128 --warmup 64 --no-cpu-baseline --no-other-configs > /dev/null 2> $OUT/p3.err || true
find $OUT -name '*counter_collection.csv'
