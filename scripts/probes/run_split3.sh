#!/bin/bash
# GPU box: the split-arithmetic pass (flush32_split.h) -- accuracy of its update against an F64 sum beside the fmaf chain's, then timing
# beside the F32-arithmetic kernels.  Build first (see flush32_bench.hip).
set -o pipefail
B=./scripts/probes/flush32_bench
O=gpurun_out/split3_probe.log
: > $O
run() { echo "== $*" | tee -a $O; timeout -k 10 "${TMO:-120}" env "$@" 2>&1 | grep -v "^check" | tee -a $O; rc=${PIPESTATUS[0]}; echo "rc=$rc" | tee -a $O; [ $rc -eq 0 ] || exit $rc; }
run ZERO_TILES=1 ACC=1 $B 600 64 0 0
run ACC=1 $B 600 64 0 0
run ZERO_TILES=1 ACC=1 $B 4000 50 0 0 40
run ZERO_TILES=1 ACC=1 $B 4000 33 0 0 0 1
run ZERO_TILES=1 ACC=1 $B 2500 64 0 0 0 0 64
TMO=300 run $B 40000 64 5 0
TMO=300 run $B 50000 64 3 0
