#!/usr/bin/env python3
"""BASELINE.json configs[1]: 1 000 landmarks, unknown correspondence (EKF_SLAM_UC.m + Correspondence.m), F64, 1 GPU.

SURVEY.md 8d config 2: seed 20260102; warm-up sweep appends all 1 000 landmarks, then timed SLAM iterations
(1 predict + measure() over the m = 8 nearest landmarks).  Reports SLAM iterations/s and update-steps/s, checks the
final state against the CPU oracle on the same inputs.  Prints one JSON line.

    python scripts/bench_config2.py [--steps 200] [--batch 8]
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
    ap.add_argument("--landmarks", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-conditioning", action="store_true", help="skip the untimed device-conditioning replay on a throw-away engine")
    ap.add_argument("--async-flush", action="store_true", help="cfg.async_flush: the pass over P on a second stream")
    ap.add_argument("--check", action="store_true", help="replay the same inputs through the CPU oracle and compare")
    ap.add_argument("--kernel-timing", action="store_true",
                    help="bracket k_associate / k_gather with HIP events (adds ~2 us per launch: use for the per-kernel figures, not for "
                         "the throughput line)")
    ap.add_argument("--host-decision", action="store_true",
                    help="cfg.device_assoc = 0: take the association decision from the host mirror of s (legitimate: the reference's "
                         "live likelihood is signature-only, Correspondence.m:75); default here is the device kernels per observation")
    ap.add_argument("--waited", action="store_true",
                    help="cfg.device_assoc = 1: k_associate per observation, the host waits for every decision (round 2's primary mode); "
                         "default here is cfg.device_assoc = 3, the device-resident loop")
    ap.add_argument("--verified", action="store_true",
                    help="cfg.device_assoc = 2: k_associate runs for every observation, the host dispatches on its mirror's decision "
                         "without waiting and verifies every device decision before measure() returns")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the SHARDED code path on this one GPU (cfg.force_sharded, the library's own 1-rank RCCL communicator): per "
                         "correction k_rowpanel<kDev> -> ncclAllGather -> k_gather<sharded, kDev>; what a shard's measure() loop costs")
    args = ap.parse_args()
    from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
    from ekf_slam_amd.world import SyntheticLandmark, make_run
    N = args.landmarks
    _, run = make_run(N, 20260102, 2 + args.steps, policy="nearest", m=args.m)
    shard_kw = {"force_sharded": 1} if args.force_sharded else {}

    def attach(engine):
        if args.force_sharded:
            import ctypes
            from ekf_slam_amd import _lib as L_
            raw = ctypes.create_string_buffer(L_.EKF_COMM_ID_BYTES)
            if L_.lib().ekf_comm_unique_id(raw) != 0:
                sys.exit("bench_config2.py: --force-sharded needs librccl (ekf_comm_unique_id failed)")
            engine.comm_init(raw.raw)

    e = EKF_SLAM_UC(capacity=N, tile=args.tile, batch=args.batch, async_flush=args.async_flush, device_assoc=(0 if args.host_decision else 2 if args.verified else 1 if args.waited else 3), **shard_kw)
    attach(e._e)
    lm = Landmark('SYNTHETIC')
    t0 = time.perf_counter()
    for u, scan in run[:2]:                      # warm-up sweep: appends every landmark
        e.predict(u); e.measure(scan, u, lm)
    e.sync()
    t_sweep = time.perf_counter() - t0
    assert e._e.N == N
    # pre-resolve the landmark source so that the timed loop measures the engine, not the Python front-end stand-in
    feeds = []
    x_for_loc = e.x
    for u, scan in run[2:]:
        obs = lm.getLandmark(scan, x_for_loc)
        idx, loc = lm.landmarkObj.table()
        feeds.append((u, obs, idx.copy(), loc.copy()))
    eng = e._e
    from ekf_slam_amd import _lib as L
    # Device conditioning, outside the timed region (as bench.py does): the first sustained burst of launches in a process sees a one-off
    # 35-70 ms stall (scripts/probe_queue.py) -- several times this benchmark's whole timed region.  Burn it on a throw-away engine of
    # the same configuration that replays the first scans from the same state.
    if not args.no_conditioning:
        warm = EKF_SLAM_UC(capacity=N, tile=args.tile, batch=args.batch, async_flush=args.async_flush, device_assoc=int(eng.cfg.device_assoc), **shard_kw)
        attach(warm._e)
        warm.x, warm.s, warm.P = e.x, e.s, e.P
        for rep in range(3):
            for u, obs, idx, loc in feeds[:64]:
                warm._e.predict(u)
                warm._e.measure(obs, u, idx, loc)
        warm._e.sync()
        warm._e.close()
    if args.kernel_timing:
        eng.timing_enable(L.EKF_KERNEL_ASSOCIATE, True, launches=args.steps * args.m + 8)
        eng.timing_enable(L.EKF_KERNEL_GATHER, True, launches=args.steps * args.m + 8)
    eng.sync()
    t0 = time.perf_counter()
    for u, obs, idx, loc in feeds:
        eng.predict(u)
        eng.measure(obs, u, idx, loc)
    eng.flush(); eng.sync()
    dt = time.perf_counter() - t0
    n_as, ms_as = eng.timing_read(L.EKF_KERNEL_ASSOCIATE) if args.kernel_timing else (0, 0.0)
    n_ga, ms_ga = eng.timing_read(L.EKF_KERNEL_GATHER) if args.kernel_timing else (0, 0.0)
    x_end = eng.get_x()
    # k_associate: one lane per landmark reads its 3x2 strip block, its 2x2 diagonal block, x_j and s_k (+ the shared 3x3 robot
    # block): 13 doubles = 104 B per landmark and observation -- ~100 KB per launch at 1 k landmarks, i.e. a
    # latency-bound launch; the HBM roofline is quoted only to show how far from bandwidth-bound it is
    assoc = None
    if n_as:
        b_assoc = 13 * 8 * N
        t_as = ms_as / n_as * 1e-3
        assoc = {"kernel": "k_associate", "launches": n_as, "avg_launch_us": t_as * 1e6,
                 "algorithmic_bytes_per_launch": b_assoc, "achieved_GBps": b_assoc / t_as / 1e9,
                 "roofline": {"bound": "latency", "hbm_frac": b_assoc / t_as / 8e12}}
    out = {"metric": "EKF update-steps/sec at N landmarks; HBM GB/s on (I−KH)P vs roofline",
           "value": args.steps * args.m / dt, "unit": "update-steps/s", "slam_iterations_per_s": args.steps / dt,
           "n_gpus": 1, "steps": args.steps, "ms_per_iteration": dt / args.steps * 1e3, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "configs[1]: %d landmarks, unknown correspondence (EKF_SLAM_UC.m + Correspondence.m), F64; "
                                  "iteration = predict + measure() over the %d nearest landmarks" % (N, args.m),
                      "deferred_batch": args.batch, "async_flush": args.async_flush, "tile": args.tile, "warmup_sweep_s": t_sweep,
                      "force_sharded": bool(args.force_sharded),
                      "device_association": ("host mirror" if args.host_decision else "device, verified after dispatch" if args.verified
                                             else "device, waited for" if args.waited else
                                             "device-resident loop: decision produced and consumed on the device, no host wait"),
                      "associate": assoc, "gather_avg_us": (ms_ga / n_ga * 1e3) if n_ga else None,
                      "state_finite": bool(np.isfinite(x_end).all())}}
    if args.check:
        from oracle.ekf_structured import StructuredEKF
        ref = StructuredEKF(N, "uc")
        lr = SyntheticLandmark()
        for u, scan in run[:2]:
            ref.predict(u); ref.measure(scan, u, lr)
        tc = time.perf_counter()
        for u, obs, idx, loc in feeds:           # same resolved observations as the GPU run
            ref.predict(u)
            for ii in range(obs.shape[0]):
                z = obs[ii]
                R = ref._R(z)
                new, k = ref.associate(z, R)
                assert not new
                ref.correct(z, R, k)
        tc = time.perf_counter() - tc
        from oracle.ekf_structured import available_cores
        out["cpu_baseline"] = {"value": args.steps * args.m / tc, "unit": "update-steps/s", "cores": available_cores(),
                               "kind": "port", "sample": "the same %d iterations through oracle/ekf_structured.c "
                               "(associate + correct per observation, full n x n P)" % args.steps}
        P = eng.get_P()
        out["parity"] = {"rel_err_x": float(np.abs(x_end - ref.x).max() / np.abs(ref.x).max()),
                         "rel_err_P": float(np.abs(P - ref.P).max() / np.abs(ref.P).max())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
