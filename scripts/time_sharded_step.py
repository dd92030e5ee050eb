#!/usr/bin/env python3
"""What the SHARDED form of the as-written update-step costs beyond its pass over P, measured on ONE GPU: the handle is forced
onto the sharded code path (cfg.force_sharded = 1: k_rowpanel -> all-gather -> k_gather<sharded> -> downdate) with a 1-rank RCCL
communicator (the all-gather is then a device copy by RCCL's kernel: its launch and kernel cost are in, its xGMI hops are
not), and timed beside the unsharded handle on the same steps.  Run under `rocprofv3 --kernel-trace --stats` for the
per-kernel split.

    python scripts/time_sharded_step.py [--landmarks 10000] [--steps 512]
"""
import argparse
import ctypes
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
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--async-flush", action="store_true", help="also time the sharded handle with cfg.async_flush")
    a = ap.parse_args()
    import bench
    from ekf_slam_amd import Engine, _lib as L
    N = a.landmarks
    w, x, s, d, U = bench.make_state(N, 20260104)
    steps = bench.make_steps(w, N, 64 + a.steps, [.01, 5.0])
    out = {"landmarks": N, "steps": a.steps, "batch": a.batch}
    digests = []
    # "_nohint": the sharded leg without ekf_hint_next (round 2's flow: k_rowpanel launch per step; the all-gather is in place either way)
    legs = [("unsharded", "0", False), ("sharded_1rank_rccl", "1", False), ("sharded_1rank_rccl_nohint", "1", False)]
    if a.async_flush:
        legs += [("unsharded_async", "0", True), ("sharded_1rank_rccl_async", "1", True)]
    for name, forced, asy in legs:
        e = Engine(capacity=N, batch=a.batch, async_flush=asy, force_sharded=int(forced))
        if forced == "1":
            raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
            assert L.lib().ekf_comm_unique_id(raw) == 0
            e.comm_init(raw.raw)
            if name.endswith("_nohint"):
                e._hints = False
        e.load_lowrank_state(x, s, d, U)
        warm, timed = e.marshal_steps(steps[:64]), e.marshal_steps(steps[64:])
        for i in range(warm["m"]):
            e.step_raw(warm, i)
        e.flush(); e.sync()
        t0 = time.perf_counter()
        for i in range(timed["m"]):
            e.step_raw(timed, i)
        e.flush(); e.sync()
        dt = time.perf_counter() - t0
        out[name] = {"ms_per_step": round(dt / a.steps * 1e3, 5), "steps_per_s": round(a.steps / dt, 1)}
        digests.append(e.digest())
        e.close()
    out["extra_us_per_step"] = round((out["sharded_1rank_rccl"]["ms_per_step"] - out["unsharded"]["ms_per_step"]) * 1e3, 2)
    out["extra_us_per_step_nohint"] = round((out["sharded_1rank_rccl_nohint"]["ms_per_step"] - out["unsharded"]["ms_per_step"]) * 1e3, 2)
    out["same_digest"] = all(bool(np.array_equal(digests[0], dg)) for dg in digests[1:])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
