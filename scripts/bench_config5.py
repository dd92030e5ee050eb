#!/usr/bin/env python3
"""BASELINE.json configs[4] shape on ONE GPU: F32 tile storage / F64 solve, streaming landmark append.

Starts from N0 bulk-loaded landmarks (P = D + U U'), capacity N0 + steps; every step = predict + append of one new
landmark + one correction on a cycling landmark (EKF_SLAM.m:40-51, :67-98, :124-145).  Prints one JSON line.

    python scripts/bench_config5.py [--landmarks 40000] [--steps 512] [--batch 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=40000)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--storage", default="f32")
    args = ap.parse_args()
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd.world import World
    N0, total = args.landmarks, args.steps + args.warmup
    cap = N0 + total
    w = World(cap, 20260101 + 5)
    rng = np.random.default_rng(77)
    n0 = 3 + 2 * N0
    x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks[:N0].reshape(-1)])
    d = rng.uniform(0.01, 0.1, n0)
    U = rng.normal(0.0, 0.01, (n0, 8))
    s = np.arange(1, N0 + 1.0)
    e = Engine(mode="known", capacity=cap, storage=args.storage, batch=args.batch)
    t0 = time.perf_counter()
    e.load_lowrank_state(x, s, d, U)
    e.sync()
    t_load = time.perf_counter() - t0
    Rc = [.01, 5.0]
    steps = []
    for t in range(total):
        u = w.step()
        k = (t * 37) % N0
        (_, r, b), = w.observe([k])
        steps.append((u, np.array([r, b]), np.diag([r * Rc[0], b * Rc[1]]), k, w.landmarks[N0 + t]))

    def run(chunk):
        for (u, z, R, k, pos) in chunk:
            e.predict(u)
            e.append(u, R, pos, e_N[0] + 1)
            e_N[0] += 1
            e.correct(z, R, k)
        e.flush()

    e_N = [N0]
    run(steps[:args.warmup])
    e.sync()
    e.timing_enable(L.EKF_KERNEL_DOWNDATE, True, launches=args.steps + 8)
    n_start = 3 + 2 * e_N[0]
    t0 = time.perf_counter()
    run(steps[args.warmup:])
    e.sync()
    dt = time.perf_counter() - t0
    launches, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE)
    kernel, kpairs = e.downdate_kernel_name()
    n_end = 3 + 2 * e_N[0]
    w_bytes = 4 if args.storage == "f32" else 8
    n_mid = (n_start + n_end) / 2
    b_alg = w_bytes * n_mid * (n_mid + 1)
    avg_ms = ms / max(launches, 1)
    finite = bool(np.isfinite(e.get_x()).all())
    out = {"metric": "EKF update-steps/sec at N landmarks; HBM GB/s on (I−KH)P vs roofline",
           "value": args.steps / dt, "unit": "update-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "dtype": "f64 solve / %s tiles" % args.storage, "data": "synthetic",
           "config": {"workload": "configs[4] shape on 1 GPU: %d -> %d landmarks, %s tile storage, F64 solve, step = predict + "
                                  "append + 1 correction (streaming landmark append)" % (N0 + args.warmup, e_N[0], args.storage),
                      "deferred_batch": args.batch, "tile": int(e.cfg.tile), "device_GB": e.device_bytes() / 1e9,
                      "bulk_load_s": t_load, "state_finite": finite},
           "roofline": {"bound": "hbm", "achieved": b_alg / (avg_ms * 1e-3) / 1e9, "peak": bench.HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": b_alg / (avg_ms * 1e-3) / bench.HBM_PEAK, "traffic": None, "kernel": kernel, "pairs_per_launch": kpairs,
                        "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": b_alg,
                        "update_steps_per_launch": args.steps / max(launches, 1)}}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
