#!/bin/bash
# Rehearse bench.py's multi-rank flow on ONE GPU (gloo backend, exchange staged through the host, every rank on device 0):
# `python bench.py --gpus N` spawns its own ranks; every world size must end with the same state digest as the
# single-process run.  Usage: scripts/rehearse_multirank.sh [landmarks]
set -e -o pipefail
L=${1:-2000}
mkdir -p gpurun_out
python bench.py --landmarks $L --steps 96 --warmup 32 --deferred-steps 128 --no-cpu-baseline 2>gpurun_out/reh_1.err > gpurun_out/reh_1.json
for n in 2 4; do
  EKF_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --landmarks $L --steps 96 --warmup 32 --deferred-steps 128 2>gpurun_out/reh_$n.err > gpurun_out/reh_$n.json
done
python - <<'PY'
import json
import numpy as np
ref = None
ref_def = None
for n in (1, 2, 4):
    b = json.loads(open("gpurun_out/reh_%d.json" % n).readline())
    assert b["n_gpus"] == n
    dg = np.array(b["config"]["state_digest"])
    print("world", n, "value", round(b["value"]), "digest", dg, "transport", b["config"]["transport"],
          "deferred", round(b["deferred"]["value"]), "lookahead", "deferred_lookahead" in b and round(b["deferred_lookahead"]["value"]))
    dd = np.array(b["deferred"]["state_digest"])
    if ref is None:
        ref, ref_def = dg, dd
    assert np.allclose(dg, ref, rtol=1e-10), "state digest differs from the single-process run"
    assert np.allclose(dd, ref_def, rtol=1e-10), "deferred leg's digest differs from the single-process run"
print("rehearsal ok")
PY
