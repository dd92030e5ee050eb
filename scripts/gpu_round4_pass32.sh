#!/bin/bash
# GPU box: the F32-arithmetic pass after the strip kernel went in -- probe (bit-equality with the fmaf reference), the float-tile tests,
# configs[4] at its real size, the configs[4] bench.
set -o pipefail
O=gpurun_out/round4_pass32.log
: > $O
( timeout -k 10 120 ./scripts/probes/flush32_bench 600 64 0 1 && timeout -k 10 120 ./scripts/probes/flush32_bench 4000 57 0 1 40 1 \
  && timeout -k 10 300 python -m pytest tests/test_f32_mixed_gpu.py tests/test_f32_storage_gpu.py tests/test_f32_sharded_gpu.py tests/test_f32_drift_gpu.py -x -q -m gpu \
  && timeout -k 10 600 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "config5" \
  && timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --batch 64 --storage f32_mixed \
  && timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --batch 32 --storage f32_mixed ) 2>&1 | tee -a $O
