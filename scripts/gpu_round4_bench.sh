#!/bin/bash
# GPU box: the bench line after the robustness changes (repeats / median, two deferred batches, both roofs for configs[4]) and the
# per-step breakdown of the sharded path on one GPU.
set -o pipefail
O=gpurun_out/round4_bench.log
: > $O
( ./scripts/probes/run_flush32_stamp.sh \
  && timeout -k 10 900 python -m pytest tests/test_bench_gpu.py -x -q -m gpu \
  && timeout -k 10 600 python bench.py --force-sharded --landmarks 1000 --steps 1280 --warmup 128 --no-cpu-baseline --no-other-configs \
  && timeout -k 10 600 python bench.py --force-sharded --steps 640 --warmup 64 --no-cpu-baseline --no-other-configs ) 2>&1 | tee -a $O
