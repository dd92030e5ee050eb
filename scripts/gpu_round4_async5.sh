#!/bin/bash
# GPU box: configs[4] (40 000 landmarks, batch 64, split arithmetic) with the pass beside the next batch's corrections (cfg.async_flush)
# against the synchronous schedule, same call.  profiles/round4_tuning.md 59.
set -o pipefail
O=gpurun_out/round4_async5.log
: > $O
( timeout -k 10 200 python -m pytest tests/test_f32_split_gpu.py tests/test_f32_mixed_gpu.py -x -q -m gpu \
  && for A in "" "--async-flush" "" "--async-flush"; do timeout -k 10 200 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --warmup 128 --batch 64 --storage f32_split $A || exit 1; done \
  && for A in "" "--async-flush"; do timeout -k 10 200 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --warmup 128 --batch 64 --storage f32_mixed $A || exit 1; done ) 2>&1 | tee -a $O
# the whole 40 000 -> 50 000 workload (9 936 timed update-steps), same four engines
if [ "${FULL:-0}" = "1" ]; then
( for S in f32_mixed f32_split; do for A in "" "--async-flush"; do timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 9936 --warmup 64 --batch 64 --storage $S $A || exit 1; done; done ) 2>&1 | tee -a $O
fi
