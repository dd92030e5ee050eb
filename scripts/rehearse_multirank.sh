#!/bin/bash
# Rehearse bench.py's multi-rank flow on ONE GPU (gloo backend, exchange staged through the host): every world size must
# end with the same state digest as the single-process run.  Usage: scripts/rehearse_multirank.sh [landmarks]
L=${1:-2000}
python bench.py --landmarks $L --steps 96 --warmup 32 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/reh_1.json
for n in 2 4; do
  EKF_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520+n)) bench.py --gpus $n --landmarks $L --steps 96 --warmup 32 2>gpurun_out/reh_$n.err | grep "^{" > gpurun_out/reh_$n.json
done
python - <<'PY'
import json
for n in (1, 2, 4):
    b = json.loads(open("gpurun_out/reh_%d.json" % n).readline())
    print("world", n, "value", round(b["value"]), "digest", b["config"]["state_digest"], "transport", b["config"]["transport"],
          "lookahead" in b and round(b["lookahead"]["value"]), round(b["immediate"]["value"]))
PY
