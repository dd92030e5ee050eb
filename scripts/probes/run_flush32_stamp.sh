#!/bin/bash
set -o pipefail
O=gpurun_out/flush32_ab.log
: > $O
for r in 1 2; do for b in flush32_bench flush32_benchdekf_tile_plain flush32_benchdekf_loader_prio0; do echo "== $b" | tee -a $O; timeout -k 10 200 ./scripts/probes/$b 40000 64 3 0 | grep -E "strip|mfma32|WRONG" | tee -a $O; done; done
