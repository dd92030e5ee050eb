#!/bin/bash
# GPU box: the split-arithmetic pass in the library -- configs[4] against the factored oracle and at its real size and length, then the
# configs[4] workload timed in both arithmetics in one call (boxes differ by a few per cent: only figures of one call are compared).
set -o pipefail
O=gpurun_out/round4_split3.log
: > $O
( timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py -q -s -k "factored_oracle or real_size_and_length" \
  && for st in f32_mixed f32_split; do
       timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --batch 64 --storage $st \
       && timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 9936 --batch 64 --storage $st || exit 1
     done ) 2>&1 | tee -a $O
