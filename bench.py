#!/usr/bin/env python3
"""Headline benchmark: EKF update-steps/sec at N landmarks and HBM GB/s of the (I - K H) P downdate.

Workload (BASELINE.json configs[2] / configs[3]; SURVEY.md section 8d "Config 3/4"): N = 10 000 landmarks,
known correspondence, F64.  The state is bulk-loaded (x from the seeded world, P = D + U U' with
D = diag(U(0.01,0.1)), U = n x 8 N(0,0.01^2)); one step = 1 predict + 1 correction (EKF_SLAM.m:40-51 and
:124-145) on a cycling landmark index, with range/bearing taken from the world's true pose.  Inputs are
resident in HBM before the timed region; the only per-step host->device traffic is kernel arguments.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--landmarks 10000] [--tile 64]

For N > 1 launch one rank per GPU (torch.distributed.run); P is split over the ranks (tile (I,J) on rank
(I+J) mod N) and each step carries one all-gather of the 2 x n landmark row-panel.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md chip table


def make_state(N, seed):
    """SURVEY.md 8d config 3 state: world landmarks as the map, P = D + U U' (SPD, dense)."""
    from ekf_slam_amd.world import World
    w = World(N, seed)
    rng = np.random.default_rng(seed + 1)
    n = 3 + 2 * N
    x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks.reshape(-1)])
    d = rng.uniform(0.01, 0.1, size=n)
    U = rng.normal(0.0, 0.01, size=(n, 8))
    s = np.arange(1, N + 1, dtype=np.float64)
    return w, x, s, d, U


def make_steps(w, N, count, Rc):
    """(u, z, R, idx0) per step: noisy odometry, observation of a cycling landmark from the true pose."""
    steps = []
    for t in range(count):
        u = w.step()
        k = (t * 37) % N
        (_, r, b), = w.observe([k])
        steps.append((u, np.array([r, b]), np.diag([r * Rc[0], b * Rc[1]]), k))
    return steps


def cpu_baseline(N, x, s, d, U, steps, budget_s=20.0):
    """The structured C restatement (oracle/, kind "port") timed on this host's cores on a bounded sample."""
    from oracle.ekf_structured import StructuredEKF, available_cores
    o = StructuredEKF(N, "known")
    n = 3 + 2 * N
    P = o.raw_P()
    blk = 2048
    for r0 in range(0, n, blk):                       # build D + U U' in row blocks, in place
        r1 = min(n, r0 + blk)
        P[r0:r1, :n] = U[r0:r1] @ U.T
    P[np.arange(n), np.arange(n)] += d
    o._x[:n] = x
    o._s[:N] = s
    o.L.oekf_set_num_landmarks(o.h, N)
    done, t0 = 0, time.perf_counter()
    for (u, z, R, k) in steps:
        o.predict(u)
        o.correct(z, R, k + 1)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "update-steps/s", "cores": available_cores(), "kind": "port",
            "sample": "%d steps (1 predict + 1 correct) of the same %d-landmark workload, structured O(n^2) C "
                      "restatement with OpenMP (oracle/ekf_structured.c), full n x n P" % (done, N)}


def load_traffic(N, tile):
    """HBM bytes per downdate launch from the committed PMC summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "downdate_pmc.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        if rec.get("landmarks") == N and rec.get("tile") == tile:
            return rec.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--landmarks", type=int, default=10000)
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with one rank per GPU "
                     "(python -m torch.distributed.run --nproc-per-node %d ...)" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    from ekf_slam_amd import Engine
    from ekf_slam_amd import _lib as L

    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    N = args.landmarks
    seed = 20260101 + 3
    w, x, s, d, U = make_state(N, seed)
    e = Engine(mode="known", capacity=N, tile=args.tile, device=local_rank, rank=rank, world=world)
    Rc = [e.cfg.Rc[0], e.cfg.Rc[1]]
    steps = make_steps(w, N, args.warmup + args.steps, Rc)
    e.load_lowrank_state(x, s, d, U)
    if world > 1:
        from ekf_slam_amd.sharding import attach_communicator
        transport = attach_communicator(e, dist, torch)
    else:
        transport = "none"

    def barrier():
        e.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(chunk):
        for (u, z, R, k) in chunk:
            e.predict(u)
            e.correct(z, R, k)

    run(steps[:args.warmup])
    barrier()
    e.timing_enable(L.EKF_KERNEL_DOWNDATE, True)
    barrier()
    t0 = time.perf_counter()
    run(steps[args.warmup:])
    barrier()
    dt = time.perf_counter() - t0
    launches, kernel_ms = e.timing_read(L.EKF_KERNEL_DOWNDATE)
    e.timing_enable(L.EKF_KERNEL_DOWNDATE, False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    x_end = e.get_x()
    finite = bool(np.isfinite(x_end).all())
    n = 3 + 2 * N
    b_alg_total = 8 * n * (n + 1)                       # every unique entry read + written once
    b_alg_rank = b_alg_total / world                    # this rank's share of the launch
    avg_ms = kernel_ms / max(launches, 1)
    achieved = b_alg_rank / (avg_ms * 1e-3)
    traffic = load_traffic(N, args.tile) if world == 1 else None

    if rank == 0:
        out = {
            "metric": "EKF update-steps/sec at N landmarks; HBM GB/s on (I−KH)P vs roofline",
            "value": args.steps / dt,
            "unit": "update-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[2]: %d landmarks, known correspondence (EKF_SLAM.m), F64; step = 1 predict"
                                   " + 1 correction on a cycling landmark; P split over %d GPU(s)" % (N, world),
                       "landmarks": N, "state_dim": n, "tile": args.tile, "storage": "f64",
                       "shard": "tile (I,J) on rank (I+J) mod %d" % world, "transport": transport,
                       "state_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "kernel": "k_downdate", "launches": launches, "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": b_alg_rank},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, x, s, d, U, steps[args.warmup:])
        print(json.dumps(out), flush=True)
    e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
