#!/usr/bin/env python3
"""BASELINE.json configs[4] shape on ONE GPU: F32 tile storage / F64 solve, streaming landmark append.

Starts from N0 bulk-loaded landmarks (P = D + U U'), capacity N0 + steps; every step = predict + append of one new
landmark + one correction on a cycling landmark (EKF_SLAM.m:40-51, :67-98, :124-145).  Prints one JSON line.

    python scripts/bench_config5.py [--landmarks 40000] [--steps 512] [--batch 12] [--storage f32|f32_mixed|f32_split|f64]
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


def load_pmc(N0, batch, storage):
    """HBM bytes per full-batch launch of the pass from the newest committed PMC summary of this workload
    (profiles/round*_config5_pmc.json: separate rocprofv3 --pmc passes of this command, profiles/README.md), else None."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_config5_pmc.json")),
                   key=lambda f: int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
    for path in reversed(files):
        try:
            with open(path) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        for leg in rec.get("legs", []):
            if leg.get("landmarks") == N0 and leg.get("batch") == batch and leg.get("storage") == storage:
                return {"file": os.path.relpath(path, ROOT), "git_blob": bench.git_blob_hash(path), "kernel": leg.get("kernel"),
                        "pairs_per_launch": leg.get("pairs_per_launch"), "landmarks_at_launch": leg.get("landmarks_at_launch"),
                        "hbm_bytes_per_launch": leg.get("hbm_bytes_per_launch"), "matrix_pipe_busy": leg.get("matrix_pipe_busy"),
                        "l2_hit_rate": leg.get("l2_hit_rate")}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", type=int, default=40000)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--batch", type=int, default=12, help="12 is the usable point of the F64-arithmetic pass on float tiles; 32-64 with --storage f32_mixed")
    ap.add_argument("--async-flush", action="store_true", help="cfg.async_flush: the pass on a second, CU-masked stream beside the next batch's corrections (twice the tile memory)")
    ap.add_argument("--storage", default="f32", help="f32: float tiles, F64 arithmetic; f32_mixed: float tiles, the pass in F32 arithmetic (cfg.pass_arith); f32_split: the same in split arithmetic (every float operand cut into three bfloat16 pieces, bf16 matrix pipe); f64")
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
    e = Engine(mode="known", capacity=cap, storage=args.storage, batch=args.batch, async_flush=args.async_flush)
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

    # the steps marshalled once (Engine.marshal_steps) and driven through the C ABI's entry points directly, as bench.py does: the timed
    # loop measures the library, not numpy conversions of the Python front-end (~20 us per call, three calls per step)
    import ctypes
    marsh = e.marshal_steps([(u, z, R, k) for (u, z, R, k, _) in steps])
    POS = np.ascontiguousarray([p for (_, _, _, _, p) in steps], dtype=np.float64)
    f_pred, f_corr = e._raw[0], e._raw[1]
    f_app = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double)(("ekf_append", e.lib))

    def run(i0, i1):
        for i in range(i0, i1):
            rc = f_pred(e.h, marsh["u"] + 16 * i) or f_app(e.h, marsh["u"] + 16 * i, marsh["r"] + 32 * i, POS.ctypes.data + 16 * i, float(e_N[0] + 1))
            e_N[0] += 1
            rc = rc or f_corr(e.h, marsh["z"] + 16 * i, marsh["r"] + 32 * i, marsh["k"][i])
            if rc:
                e._check(rc)
        e.flush()

    e_N = [N0]
    run(0, args.warmup)
    e.sync()
    kernel_full, kpairs_full = e.downdate_kernel_name()       # the warm-up ended with a pass: a FULL batch when warmup % batch == 0
    e.timing_enable(L.EKF_KERNEL_DOWNDATE, True, launches=args.steps + 8)
    n_start = 3 + 2 * e_N[0]
    t0 = time.perf_counter()
    run(args.warmup, total)
    e.sync()
    dt = time.perf_counter() - t0
    launches, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE)
    kernel, kpairs = e.downdate_kernel_name()
    n_end = 3 + 2 * e_N[0]
    w_bytes = 4 if args.storage.startswith("f32") else 8
    n_mid = (n_start + n_end) / 2
    b_alg = w_bytes * n_mid * (n_mid + 1)
    avg_ms = ms / max(launches, 1)
    finite = bool(np.isfinite(e.get_x()).all())
    # Which roof binds the pass?  HBM: every unique entry of P read and written once (b_alg).  Matrix pipe: one launch applies
    # `pairs` rank-2 terms to every stored entry -- 2 x 2 pairs FLOPs per entry of the lower triangle, n (n + 1) / 2 entries (the
    # kernels compute whole diagonal tiles: counted as the triangle, the algorithmic figure).  `bound` is the roof with the larger
    # achieved / peak: the one this launch shape sits closer to.
    pairs_avg = args.steps / max(launches, 1)
    flops = 4.0 * pairs_avg * n_mid * (n_mid + 1) / 2
    f32_pass = args.storage == "f32_mixed"
    split_pass = args.storage == "f32_split"
    # split arithmetic: six bf16 partial products per product -- the matrix pipe EXECUTES 6 x the algorithmic flops, priced against the dense
    # bf16 peak; `algorithmic_flops_per_launch` stays the algorithmic figure
    exec_factor = 6.0 if split_pass else 1.0
    pipe_peak = bench.BF16_MATRIX_PEAK if split_pass else bench.F32_MATRIX_PEAK if f32_pass else bench.F64_MATRIX_PEAK
    hbm_frac = b_alg / (avg_ms * 1e-3) / bench.HBM_PEAK
    pipe_frac = exec_factor * flops / (avg_ms * 1e-3) / pipe_peak
    bound = "mfma" if pipe_frac > hbm_frac else "hbm"
    pmc = load_pmc(N0, args.batch, args.storage)
    if pmc is not None and pmc.get("kernel") and pmc["kernel"].split("<")[0] != kernel_full.split("<")[0]:
        pmc = None                                            # measured on another kernel: does not describe these launches
    roof = {"bound": bound,
            "achieved": exec_factor * flops / (avg_ms * 1e-3) / 1e12 if bound == "mfma" else b_alg / (avg_ms * 1e-3) / 1e9,
            "peak": pipe_peak / 1e12 if bound == "mfma" else bench.HBM_PEAK / 1e9,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
            "frac": pipe_frac if bound == "mfma" else hbm_frac,
            "roofs": {"hbm": {"achieved": b_alg / (avg_ms * 1e-3) / 1e9, "peak": bench.HBM_PEAK / 1e9, "unit": "GB/s", "frac": hbm_frac},
                      "mfma": {"achieved": exec_factor * flops / (avg_ms * 1e-3) / 1e12, "peak": pipe_peak / 1e12, "unit": "TFLOP/s", "frac": pipe_frac,
                               "executed_over_algorithmic_flops": exec_factor,
                               "pipe": "v_mfma_f32_16x16x32_bf16" if split_pass else "v_mfma_f32_16x16x4_f32" if f32_pass else "v_mfma_f64_16x16x4_f64"}},
            # HBM bytes per FULL-batch launch from PMC counters (their own rocprofv3 passes): from the committed summary of this
            # launch shape, null when none matches
            "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "from_committed_profile": pmc,
            "kernel": kernel_full, "pairs_per_launch": kpairs_full, "last_launch": {"kernel": kernel, "pairs": kpairs},
            "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": b_alg,
            "algorithmic_flops_per_launch": flops, "update_steps_per_launch": pairs_avg}
    out = {"metric": "EKF update-steps/sec at N landmarks; HBM GB/s on (I\u2212KH)P vs roofline",
           "value": args.steps / dt, "unit": "update-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "dtype": "f64 solve / %s" % {"f32": "f32 tiles, f64 pass arithmetic", "f32_mixed": "f32 tiles, f32 pass arithmetic (matrix pipe)", "f32_split": "f32 tiles, f32-equivalent split pass arithmetic (3 x bf16 pieces per operand, six exact partial products, f32 accumulation: bf16 matrix pipe)", "f64": "f64 tiles"}[args.storage], "data": "synthetic",
           "config": {"workload": "configs[4] shape on 1 GPU: %d -> %d landmarks, %s tile storage, F64 solve, step = predict + "
                                  "append + 1 correction (streaming landmark append)" % (N0 + args.warmup, e_N[0], args.storage),
                      "deferred_batch": args.batch, "async_flush": bool(args.async_flush), "tile": int(e.cfg.tile), "device_GB": e.device_bytes() / 1e9,
                      "bulk_load_s": t_load, "state_finite": finite},
           "roofline": roof}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
