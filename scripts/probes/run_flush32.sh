#!/bin/bash
# GPU box: correctness of the pipelined F32 pass (bit-equal to a scalar fmaf reference) at several shapes, then timing.
set -o pipefail
B=./scripts/probes/flush32_bench
O=gpurun_out/flush32_probe.log
: > $O
run() { echo "== $*" | tee -a $O; timeout -k 10 "${TMO:-60}" $B "$@" 2>&1 | grep -v ": 0 of" | grep -v "pipe32" | tee -a $O; rc=${PIPESTATUS[0]}; echo "rc=$rc" | tee -a $O; [ $rc -eq 0 ] || exit $rc; }
run 300 64 0 1
run 600 29 0 1
run 4000 64 2 1
run 4000 37 0 1
run 4000 50 0 1 40 1
TMO=300 run 40000 64 5 1
TMO=300 run 40000 32 5 0
STAMP=2 timeout -k 10 100 $B 40000 64 0 0 2>&1 | tee -a $O
