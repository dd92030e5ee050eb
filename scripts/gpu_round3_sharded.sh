#!/bin/bash
# GPU box: sharded-path checks of round 3 (hinted extraction, in-place all-gather) + the sharded step's fixed cost on one GPU
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py tests/test_f32_sharded_gpu.py tests/test_full_size_gpu.py::test_ten_thousand_landmarks_eight_shards -x -q -m gpu > gpurun_out/r3_sharded_tests.log 2>&1 || { tail -40 gpurun_out/r3_sharded_tests.log; exit 1; }
tail -3 gpurun_out/r3_sharded_tests.log
for L in 10000 3536; do
  timeout -k 10 300 python scripts/time_sharded_step.py --landmarks $L --steps 512 2>gpurun_out/r3_sh.err | tee -a gpurun_out/r3_sharded_step.jsonl
done
