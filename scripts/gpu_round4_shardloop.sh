#!/bin/bash
# GPU box: the device-resident measure loop on shards (ShardGroup worlds 2-8 through the exchange hook, a 1-rank RCCL communicator),
# the whole sharded test file, and configs[1] through the sharded path on one GPU beside the unsharded one.
set -o pipefail
O=gpurun_out/round4_shardloop.log
: > $O
( timeout -k 10 600 python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k "device_resident" \
  && timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py tests/test_config2_uc_gpu.py tests/test_abi_symbols.py -x -q -m gpu \
  && timeout -k 10 300 python scripts/bench_config2.py --batch 8 \
  && timeout -k 10 300 python scripts/bench_config2.py --batch 8 --force-sharded ) 2>&1 | tee -a $O
