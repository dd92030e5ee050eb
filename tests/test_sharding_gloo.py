"""CPU, world_size 2 over gloo: the shard plan of the multi-GPU path (tile (I,J) on shard (I+J) mod world, the
equal-count all-gather of the landmark row-panel, downdate of owned tiles only) exercised across two real
processes.  The per-shard arithmetic is emulated in NumPy here (no GPU in this container); ownership, slab
layout and chunk routing come from the library's host-only plan functions (ekf_shard_owner /
ekf_shard_panel_source), i.e. the same code the HIP path uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

T = 16
N = 40          # 80 landmark-block rows -> 5 tile rows
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _state(seed=5):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.2, -0.1, 30.0], rng.uniform(-15, 15, 2 * N)])
    U = rng.normal(0, 0.05, (n, 5))
    P = np.diag(rng.uniform(0.01, 0.1, n)) + U @ U.T
    return x, P, np.arange(1, N + 1.0)


def _worker(rank, port, outq):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ekf_slam_amd import sharding
        from oracle.ekf_structured import StructuredEKF
        from oracle.matlab_compat import atan2d, inv2, wrapTo360

        x, P, s = _state()
        ref = StructuredEKF(N, "known")
        ref.set_state(x, P, s)
        nt = (2 * N + T - 1) // T
        Pmm = P[3:, 3:]
        tiles = {(I, J): Pmm[I * T:(I + 1) * T, J * T:(J + 1) * T].copy()
                 for I in range(nt) for J in range(I + 1) if sharding.owner(WORLD, I, J) == rank}
        # local slots never collide (a tile row may leave at most one slot unused)
        mine = sorted(tiles)
        slots = [sharding.slot(WORLD, I, J) for I, J in mine]
        assert len(set(slots)) == len(slots) and max(slots) < len(slots) + nt
        strip, prr, xs = P[0:3, 3:].copy(), P[0:3, 0:3].copy(), x.copy()

        def low(r, c):          # canonical lower-triangle entry from an OWNED tile
            if r < c:
                r, c = c, r
            return tiles[(r // T, c // T)][r % T, c % T]

        rng = np.random.default_rng(9)
        for idx0 in (0, 7, 8, 23, N - 1):
            j, Ij = 2 * idx0, (2 * idx0) // T
            cmax = (nt + WORLD - 1) // WORLD
            send = np.zeros((cmax * T, 2))
            for k in range(nt):
                o, kl = sharding.panel_source(WORLD, Ij, k)
                if o == rank:
                    for cc in range(T):
                        send[kl * T + cc] = (low(j, k * T + cc), low(j + 1, k * T + cc))
            slabs = [torch.zeros(cmax * T, 2, dtype=torch.float64) for _ in range(WORLD)]
            dist.all_gather(slabs, torch.from_numpy(send))                     # THE exchange of an update-step
            M = np.zeros((2, 2 * N))
            for k in range(nt):
                o, kl = sharding.panel_source(WORLD, Ij, k)
                M[:, k * T:(k + 1) * T] = slabs[o].numpy()[kl * T:(kl + 1) * T].T
            np.testing.assert_array_equal(M, Pmm_full_rows(tiles, j, nt, rank, M))   # routing sanity (own chunks)
            # replicated solve (what k_gather does on every shard)
            z = np.array([rng.uniform(2, 20), rng.uniform(5, 355)])
            R = np.diag([z[0] * .01, z[1] * 5.0])
            d = xs[3 + j:5 + j] - xs[0:2]
            q = d @ d
            sq = np.sqrt(q)
            zhat = np.array([sq, wrapTo360(atan2d(d[1], d[0]) - xs[2])])
            Hs = (1 / q) * np.array([[-sq * d[0], -sq * d[1], 0, sq * d[0], sq * d[1]], [d[1], -d[0], -q, -d[1], d[0]]])
            G = Hs[:, 0:3] @ strip + Hs[:, 3:5] @ M                                   # 2 x 2N
            Gr = Hs[:, 0:3] @ prr + Hs[:, 3:5] @ strip[:, j:j + 2].T                   # 2 x 3
            GS = np.hstack([Gr, G[:, j:j + 2]])
            Phi = inv2(GS @ Hs.T + R)
            K, Kr = G.T @ Phi, Gr.T @ Phi
            nu = z - zhat
            xs[3:] += K @ nu
            xs[0:3] += Kr @ nu
            strip -= Kr @ G
            prr -= Kr @ Gr
            for (I, J), t in tiles.items():                                           # downdate of OWNED tiles only
                t -= K[I * T:(I + 1) * T] @ G[:, J * T:(J + 1) * T]
            ref.correct(z, R, idx0 + 1)
        # gather every shard's tiles and compare the assembled covariance with the oracle
        everything = [None] * WORLD
        dist.all_gather_object(everything, tiles)
        full = np.zeros((2 * N, 2 * N))
        for part in everything:
            for (I, J), t in part.items():
                full[I * T:(I + 1) * T, J * T:(J + 1) * T] = t
        full = np.tril(full) + np.tril(full, -1).T
        Pref = ref.P
        err = max(np.abs(full - Pref[3:, 3:]).max(), np.abs(strip - Pref[0:3, 3:]).max(), np.abs(prr - Pref[0:3, 0:3]).max())
        errx = np.abs(xs - ref.x).max() / np.abs(ref.x).max()
        outq.put((rank, float(err / np.abs(Pref).max()), float(errx), len(tiles)))
    except BaseException as ex:  # noqa: BLE001 -- report instead of letting the parent wait for its timeout
        import traceback
        outq.put((rank, "error", traceback.format_exc(), repr(ex)))
        raise
    finally:
        dist.destroy_process_group()


def Pmm_full_rows(tiles, j, nt, rank, M):
    """M restricted to what this rank can verify (its own tiles); other entries are taken from M itself."""
    out = M.copy()
    for (I, J), t in tiles.items():
        for r in (j, j + 1):
            if I == r // T:                      # row part
                for cc in range(T):
                    c = J * T + cc
                    if c <= r:
                        out[r - j, c] = t[r % T, cc]
            if J == r // T:                      # column part
                for rr in range(T):
                    c = I * T + rr
                    if c > r:
                        out[r - j, c] = t[rr, r % T]
    return out


def test_shard_plan_two_processes_gloo(oracle_lib):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = []
    for _ in range(WORLD):
        r = q.get(timeout=120)
        if r[1] == "error":
            for p in procs:
                p.join(10)
                if p.is_alive():
                    p.terminate()
            pytest.fail("rank %d failed:\n%s" % (r[0], r[2]))
        results.append(r)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    counts = []
    for rank, errP, errx, ntiles in sorted(results):
        assert errP < 1e-12 and errx < 1e-12, (rank, errP, errx)
        counts.append(ntiles)
    assert sum(counts) == 15 and abs(counts[0] - counts[1]) <= 3      # 5 tile rows: 15 tiles; diagonal tiles sit on even shards


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_plan_properties(world):
    """Host-only plan: every tile has one owner; slots are a dense 0..count-1 range per shard; the per-row and
    total loads are balanced for any number of active tile rows (streaming append never re-balances)."""
    from ekf_slam_amd import sharding
    for nt in (1, 2, 5, 16, 37):
        per_rank = [[] for _ in range(world)]
        for I in range(nt):
            row = [0] * world
            for J in range(I + 1):
                o = sharding.owner(world, I, J)
                assert 0 <= o < world
                per_rank[o].append(sharding.slot(world, I, J))
                row[o] += 1
            assert max(row) - min(row) <= 1
        for r in range(world):
            assert sorted(per_rank[r]) == sorted(set(per_rank[r]))            # no slot collision
            assert all(0 <= sl < nt * (nt + 1) // 2 + nt for sl in per_rank[r])
        total = [len(v) for v in per_rank]
        assert max(total) - min(total) <= nt + 1      # diagonal tiles sit on even shards when world is even
        for Ij in range(nt):                                                 # panel routing is a cyclic deal
            for k in range(nt):
                o, kl = sharding.panel_source(world, Ij, k)
                assert o == (Ij + k) % world and kl == k // world
                I, J = (Ij, k) if k <= Ij else (k, Ij)
                assert o == sharding.owner(world, I, J)
