#!/bin/bash
set -o pipefail
O=gpurun_out/flush32_gap.log
: > $O
(echo "== back to back"; timeout -k 10 200 ./scripts/probes/flush32_bench 40576 64 3 0 | grep -E "strip"; for w in 0 500 2000 6000; do echo "== 8 ms idle, MFMA prewarm $w us"; PREWARM_US=$w GAP_MS=8 timeout -k 10 200 ./scripts/probes/flush32_bench 40576 64 3 0 | grep -E "strip"; done; for w in 1024 8192; do echo "== 8 ms idle, stream $w MB first"; PREWARM_MB=$w GAP_MS=8 timeout -k 10 200 ./scripts/probes/flush32_bench 40576 64 3 0 | grep -E "strip"; done ) 2>&1 | tee -a $O
