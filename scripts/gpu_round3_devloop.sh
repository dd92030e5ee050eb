#!/bin/bash
# GPU box: first check of the device-resident measure loop (round 3): KATs, config-2 parity in all four modes, throughput.
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kat_gpu.py tests/test_config2_uc_gpu.py tests/test_edge_cases_gpu.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r3_devloop_tests.log 2>&1 || { tail -40 gpurun_out/r3_devloop_tests.log; exit 1; }
tail -3 gpurun_out/r3_devloop_tests.log
for mode in "" "--async-flush" ${MODES:-"--verified" "--waited" "--host-decision"}; do
  for b in ${BATCHES:-8 16 32}; do
    timeout -k 10 300 python scripts/bench_config2.py --batch $b $mode 2>gpurun_out/r3_c2.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('%-10s batch %2d: %8.0f update-steps/s  %6.2f us/obs' % ('$mode' or 'devloop', $b, d['value'], 1e6 / d['value']))" | tee -a gpurun_out/r3_devloop_bench.log
  done
done
