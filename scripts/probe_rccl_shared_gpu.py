#!/usr/bin/env python3
"""Probe: can TWO processes on ONE GPU form a native RCCL communicator (ekf_comm_init)?  If RCCL allows it, this is a real
multi-process run of the library's own transport (unique id from rank 0, ncclCommInitRank, ncclAllGather per correction) on the
single test GPU; if it refuses ("duplicate GPU"), the error is printed and the exit code is 3.  Rendezvous over gloo.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/probe_rccl_shared_gpu.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd.sharding import attach_communicator
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    N = 300
    rng = np.random.default_rng(5)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 6))
    P = np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T
    s = np.arange(1, N + 1.0)
    e = Engine(mode="uc", capacity=N, tile=32, device=0, rank=rank, world=world)
    try:
        transport = attach_communicator(e, dist, torch, transport="rccl")
    except L.EkfError as ex:
        print("[rank %d] native RCCL communicator refused: %s" % (rank, ex), flush=True)
        sys.exit(3)
    one = Engine(mode="uc", capacity=N, tile=32)
    for eng in (e, one):
        eng.set_params(w_pos=1.0, s_cost=200.0, s_thresh=1e9)
        eng.set_state(x, P, s)
    for k in (0, 17, 150, 299, 18):
        z = [float(rng.uniform(1, 30)), float(rng.uniform(1, 359))]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        e.predict([0.1, 3.0]); one.predict([0.1, 3.0])
        e.correct(z, R, k); one.correct(z, R, k)
        assert e.associate(z + [5.0], R) == one.associate(z + [5.0], R)
    np.testing.assert_array_equal(e.get_x(), one.get_x())
    dg = torch.tensor(e.digest())
    dist.all_reduce(dg)
    if rank == 0:
        print("transport", transport, "sharded digest", dg.numpy(), "unsharded", one.digest(), flush=True)
        assert np.allclose(dg.numpy(), one.digest(), rtol=1e-12)
        print("two ranks on one GPU over native RCCL: ok", flush=True)
    dist.barrier()


if __name__ == "__main__":
    main()
