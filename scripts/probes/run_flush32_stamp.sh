#!/bin/bash
set -o pipefail
O=gpurun_out/flush32_kreg.log
: > $O
(timeout -k 10 100 ./scripts/probes/flush32_bench 600 64 0 1 | grep -E "WRONG|strip"; timeout -k 10 100 ./scripts/probes/flush32_bench 4000 57 0 1 40 1 | grep -E "WRONG|strip"; echo "== counted wait"; timeout -k 10 200 ./scripts/probes/flush32_bench 40000 64 3 1 | grep -E "strip|mfma32|WRONG"; echo "== vmcnt(0) at the item's end"; timeout -k 10 200 ./scripts/probes/flush32_bench_waitall 40000 64 3 0 | grep -E "strip"; echo "== counted wait again"; timeout -k 10 200 ./scripts/probes/flush32_bench 40000 64 3 0 | grep -E "strip") 2>&1 | tee -a $O
