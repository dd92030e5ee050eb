#!/bin/bash
set -o pipefail
O=gpurun_out/flush32_prio.log
: > $O
(for r in 1 2; do for v in 0 1 2 3; do echo "== prio variant $v (round $r)"; timeout -k 10 200 ./scripts/probes/flush32_bench_p$v 40000 64 3 $((r==1)) | grep -E "strip|WRONG"; done; done; for v in 2 3; do STAMP=2 timeout -k 10 100 ./scripts/probes/flush32_bench_p$v 40000 64 0 0 | grep -E "wave  [04]"; done) 2>&1 | tee -a $O
