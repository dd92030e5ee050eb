#!/usr/bin/env python3
"""Time the pass over P (downdate / flush kernel, HIP events on the engine's stream) and the whole update-step at a given batch.
One line per call; used by scripts/ab_flush.sh to A/B builds (EKF_LIB_PATH) and launch-time knobs (EKF_* environment).

    python scripts/time_flush.py [--landmarks 10000] [--batch 32] [--batches 12] [--storage f64] [--label text]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--storage", default="f64")
    ap.add_argument("--label", default="")
    ap.add_argument("--async-flush", action="store_true")
    ap.add_argument("--pass-direction", type=int, default=0, help="cfg.pass_direction: 0 auto, 1 forwards, 2 alternate")
    ap.add_argument("--no-kernel-timing", action="store_true", help="no HIP events around the kernels: the throughput column is then undisturbed")
    a = ap.parse_args()
    import bench
    from ekf_slam_amd import Engine, _lib as L
    N = a.landmarks
    w, x, s, d, U = bench.make_state(N, 20260104)
    nsteps = a.batch * a.batches
    steps = bench.make_steps(w, N, a.batch * 2 + nsteps, [.01, 5.0])
    e = Engine(capacity=N, tile=a.tile, batch=a.batch, storage=a.storage, async_flush=a.async_flush, pass_direction=a.pass_direction)
    e.load_lowrank_state(x, s, d, U)
    warm, timed = e.marshal_steps(steps[:a.batch * 2]), e.marshal_steps(steps[a.batch * 2:])
    for i in range(warm["m"]):
        e.step_raw(warm, i)
    e.flush(); e.sync()
    if not a.no_kernel_timing:
        e.timing_enable(L.EKF_KERNEL_DOWNDATE, True, launches=nsteps + 8)
        e.timing_enable(L.EKF_KERNEL_GATHER, True, launches=nsteps + 8)
    e.sync()
    t0 = time.perf_counter()
    for i in range(timed["m"]):
        e.step_raw(timed, i)
    e.flush(); e.sync()
    dt = time.perf_counter() - t0
    nl, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE)
    ng, gms = e.timing_read(L.EKF_KERNEL_GATHER)
    name, pairs = e.downdate_kernel_name()
    n = 3 + 2 * N
    b_alg = (8 if a.storage == "f64" else 4) * n * (n + 1)
    avg = ms / max(nl, 1)
    print(json.dumps({"label": a.label, "landmarks": N, "batch": a.batch, "kernel": name, "pairs": pairs, "launches": nl,
                      "flush_ms": round(avg, 4), "frac": round(b_alg / (avg * 1e-3) / 8e12, 4) if avg > 0 else None,
                      "gather_us": round(gms / max(ng, 1) * 1e3, 2), "steps_per_s": round(nsteps / dt),
                      "finite": bool(np.isfinite(e.get_x()).all()), "digest": [float(v) for v in e.digest()]}), flush=True)
    e.close()


if __name__ == "__main__":
    main()
