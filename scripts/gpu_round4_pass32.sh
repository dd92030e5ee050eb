#!/bin/bash
# GPU box: the F32-arithmetic pass with the strip kernel -- the f32_mixed parity tests, configs[4] at its real size (incl. against the
# factored oracle), the configs[4] bench at batch 64.
set -o pipefail
O=gpurun_out/round4_pass32c.log
: > $O
( timeout -k 10 900 python -m pytest tests/test_f32_mixed_gpu.py -x -q -m gpu \
  && timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "config5 or factored" \
  && timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 1024 --batch 64 --storage f32_mixed \
  && timeout -k 10 300 python scripts/bench_config5.py --landmarks 40000 --steps 9936 --batch 64 --storage f32_mixed ) 2>&1 | tee -a $O
