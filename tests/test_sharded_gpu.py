"""GPU: the sharded correction path (tile (I,J) on shard (I+J) mod world; one all-gather of the landmark
row-panel per update-step) against the CPU oracle and against the unsharded HIP path.

Only one GPU is available to the tests, so `world` shards of one filter live on that GPU in ONE process and the
exchange is ekf_exchange_local (transport (c) of include/ekfslam.h); the RCCL and torch.distributed transports
are exercised with world == 1 (a 1-rank communicator), which runs the same extract -> all-gather -> solve code."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _state(N, seed):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 6))
    d = rng.uniform(0.01, 0.1, size=n)
    return x, np.diag(d) + U @ U.T, np.arange(1, N + 1.0), d, U


@pytest.mark.parametrize("world,tile", [(2, 16), (3, 16), (4, 32), (8, 16), (2, 64)])
def test_shard_group_matches_oracle_and_unsharded(world, tile, oracle_lib):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N = 150
    x, P, s, _, _ = _state(N, 21)
    g = ShardGroup(world, capacity=N + 8, tile=tile)
    one = Engine(capacity=N + 8, tile=tile)
    ref = StructuredEKF(N + 8, "known")
    g.set_state(x, P, s); one.set_state(x, P, s); ref.set_state(x, P, s)
    assert rel_err(g.get_P(), P) == 0.0
    rng = np.random.default_rng(4)
    seq = [0, 3, N // 2, N - 1, 9, 77]
    for step, idx0 in enumerate(seq):
        u = [0.1, 3.0]
        g.predict(u); one.predict(u); ref.predict(u)
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        g.correct(z, R, idx0); one.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
        if step % 2 == 1:                         # grow the map while sharded (streaming append)
            pos = rng.uniform(-5, 5, 2)
            sig = g.N + 1
            g.append(u, R, pos, sig); one.append(u, R, pos, sig); ref.append(u, R, pos, sig)
            g.correct(z, R, g.N - 1); one.correct(z, R, one.N - 1); ref.correct(z, R, ref.N)
    Pg, Po = g.get_P(), one.get_P()
    assert not np.isnan(Pg).any()
    assert rel_err(g.get_x(), ref.x) < REL and rel_err(Pg, ref.P) < REL
    # same kernels, same arithmetic: sharding must not change a single bit
    np.testing.assert_array_equal(Pg, Po)
    np.testing.assert_array_equal(g.get_x(), one.get_x())
    np.testing.assert_allclose(g.digest(), one.digest(), rtol=1e-12)
    g.close(); one.close()


def test_shard_group_uc_association(oracle_lib):
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N = 90
    x, P, s, _, _ = _state(N, 23)
    g = ShardGroup(3, mode="uc", capacity=N, tile=16)
    ref = StructuredEKF(N, "uc")
    g.set_state(x, P, s); ref.set_state(x, P, s)
    R = np.diag([1.0, 50.0])
    for sig in (5.0, 90.0, 91.0):
        new_g, idx_g = g.associate([7.0, 123.0, sig], R)
        new_r, idx_r = ref.associate([7.0, 123.0, sig], R)
        assert (new_g, idx_g + 1) == (new_r, idx_r)
    g.close()


@pytest.mark.parametrize("world,tile,batch", [(2, 16, 1), (3, 16, 4), (4, 32, 1), (8, 16, 8), (2, 64, 4)])
def test_shard_group_association_with_position_cost(world, tile, batch, oracle_lib):
    """with the position cost in the likelihood (w_pos != 0, Correspondence.m:74) every shard scores the landmarks
    whose 2x2 diagonal block it holds, ONE all-gather carries the candidates (+ the position costs when asked for), and every
    shard takes the same arg-min: decision and cost vectors equal the unsharded engine's bit for bit (same kernel arithmetic,
    pending pairs applied on the fly when batch > 1) and the structured oracle's to 1e-6."""
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N = 150
    x, P, s, _, _ = _state(N, 51)
    kw = dict(mode="uc", capacity=N + 4, tile=tile, batch=batch)
    g, one, ref = ShardGroup(world, **kw), Engine(**kw), StructuredEKF(N + 4, "uc")
    for e in (g, one):
        e.set_params(w_pos=1.0, s_cost=200.0, s_thresh=1e9)
    ref.w_pos, ref.s_cost, ref.s_thresh = 1.0, 200.0, 1e9
    g.set_state(x, P, s); one.set_state(x, P, s); ref.set_state(x, P, s)
    rng = np.random.default_rng(6)
    for step in range(6):
        u = [0.1, 3.0]
        g.predict(u); one.predict(u); ref.predict(u)
        # an observation of landmark k from the current estimate (+ noise), so that the position cost picks it
        k = int(rng.integers(0, N))
        xe = one.get_x()
        dx, dy = xe[3 + 2 * k] - xe[0], xe[4 + 2 * k] - xe[1]
        z = [np.hypot(dx, dy) + rng.normal(0, .02), (np.degrees(np.arctan2(dy, dx)) - xe[2]) % 360.0, float(rng.integers(1, N))]
        R = np.diag([z[0] * .1, max(z[1], 1.0) * 5.0])
        ng, ig, pcg, scg = g.associate(z, R, want_costs=True)
        no, io, pco, sco = one.associate(z, R, want_costs=True)
        nr, ir, pcr, scr = ref.associate(z, R, want_costs=True)
        assert (ng, ig) == (no, io) == (nr, ir - 1)
        np.testing.assert_array_equal(pcg, pco)
        np.testing.assert_array_equal(scg, sco)
        assert not np.isnan(pcg).any()
        np.testing.assert_allclose(pcg, pcr, rtol=1e-6)
        assert g.associate(z, R) == (no, io)                 # without the cost vectors: 4 doubles per shard travel
        # a correction in between: with batch > 1 it stays pending and the next association patches the blocks on the fly
        g.correct(z, R, io); one.correct(z, R, io); ref.correct(z, R, io + 1)
    # nothing passes a tiny threshold: new landmark, default index N (0-based) on every shard
    for e in (g, one):
        e.set_params(s_thresh=1e-12)
    assert g.associate(z, R) == one.associate(z, R) == (True, N)
    np.testing.assert_array_equal(g.get_x(), one.get_x())
    g.close(); one.close()


def test_position_cost_on_a_lone_shard_needs_no_exchange():
    """Until round 3 a shard could only score the landmarks whose diagonal TILE it held, and ekf_associate with w_pos != 0 (or asking for
    the cost vector) needed an all-gather of candidates.  The landmarks' 2x2 diagonal blocks are now replicated (live F64 copies kept by
    every correction's gather): a lone shard, no communicator, answers like the unsharded engine -- decision and costs, bit for bit."""
    from ekf_slam_amd import Engine, EkfError
    from ekf_slam_amd.sharding import ShardGroup
    N = 40
    x, P, s, _, _ = _state(N, 52)
    g = ShardGroup(2, mode="uc", capacity=N, tile=16, batch=4)
    one = Engine(mode="uc", capacity=N, tile=16, batch=4)
    g.set_state(x, P, s); one.set_state(x, P, s)
    R = np.diag([1.0, 50.0])
    z = [7.0, 123.0, 5.0]
    g.correct([6.0, 100.0], R, 3); one.correct([6.0, 100.0], R, 3)       # a pending pair: the live blocks carry it already
    assert g.shards[1].associate(z, R) == one.associate(z, R) == (False, 4)
    for sh in g.shards:
        a, b = sh.associate(z, R, want_costs=True), one.associate(z, R, want_costs=True)
        assert a[:2] == b[:2]
        np.testing.assert_array_equal(a[2], b[2]); np.testing.assert_array_equal(a[3], b[3])
    g.set_params(w_pos=1.0); one.set_params(w_pos=1.0)
    assert g.shards[0].associate(z, R) == g.shards[1].associate(z, R) == one.associate(z, R)
    with pytest.raises(EkfError):                              # finish without begin
        g.shards[0].associate_finish()
    g.close(); one.close()


_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import ctypes
from ekf_slam_amd import Engine, _lib as L
rng = np.random.default_rng(31)
N = 120; n = 3 + 2 * N
x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, 2 * N)])
U = rng.normal(0, 0.05, (n, 6)); P = np.diag(rng.uniform(0.01, 0.1, n)) + U @ U.T
s = np.arange(1, N + 1.0)
mode = sys.argv[1]
e = Engine(capacity=N, tile=32, batch=4, force_sharded=1)     # the sharded code path with one rank (cfg.force_sharded)
ref = Engine(capacity=N, tile=32, batch=4)
e.set_state(x, P, s); ref.set_state(x, P, s)
if mode == "rccl":
    raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
    assert L.lib().ekf_comm_unique_id(raw) == 0
    e.comm_init(raw.raw)
    transport = "rccl-native"
else:
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from ekf_slam_amd.sharding import attach_communicator
    transport = attach_communicator(e, dist, torch, transport=mode)
for idx0 in (0, 50, 119, 7):
    z = [rng.uniform(1, 30), rng.uniform(1, 359)]; R = np.diag([z[0] * .01, z[1] * 5.0])
    e.predict([0.1, 3.0]); ref.predict([0.1, 3.0])
    e.correct(z, R, idx0); ref.correct(z, R, idx0)
# one exchange for the next three corrections, then a whole scan through measure() (which prefetches by itself)
e.prefetch_rows([3, 60, 61])
for idx0 in (60, 3, 61):
    z = [rng.uniform(1, 30), rng.uniform(1, 359)]; R = np.diag([z[0] * .01, z[1] * 5.0])
    e.predict([0.1, 3.0]); ref.predict([0.1, 3.0])
    e.correct(z, R, idx0); ref.correct(z, R, idx0)
np.testing.assert_array_equal(e.get_P(), ref.get_P())
np.testing.assert_array_equal(e.get_x(), ref.get_x())
# association with the position cost in the likelihood: candidates through the same transport
for eng in (e, ref):
    eng.set_params(w_pos=1.0, s_cost=200.0, s_thresh=1e9)
za, Ra = [9.0, 200.0, 17.0], np.diag([0.9, 1000.0])
ra, rb = e.associate(za, Ra, want_costs=True), ref.associate(za, Ra, want_costs=True)
assert ra[:2] == rb[:2] == e.associate(za, Ra)
np.testing.assert_array_equal(ra[2], rb[2]); np.testing.assert_array_equal(ra[3], rb[3])
if mode == "rccl":
    # a whole scan through ekf_measure: on a sharded handle with a communicator it fetches the scan's row-panels in
    # one exchange; rows 1..3 correct landmarks 1..3 (known correspondence: idx = row number), row 4 appends
    obs = np.array([[5.0, 40.0, 1.0], [6.0, 50.0, 2.0], [7.0, 60.0, 3.0], [8.0, 70.0, N + 1.0]])
    idx = np.arange(1, N + 2, dtype=np.float64); loc = rng.uniform(-20, 20, (N + 1, 2))
    e2 = Engine(capacity=N + 2, tile=32, batch=4); r2 = Engine(capacity=N + 2, tile=32, batch=4)
    e3 = Engine(capacity=N + 2, tile=32, batch=4, force_sharded=1)
    raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
    assert L.lib().ekf_comm_unique_id(raw) == 0
    e3.comm_init(raw.raw)
    for eng in (e3, r2):
        eng.set_state(x, P, s)
        eng.predict([0.1, 3.0])
        eng.measure(obs, [0.1, 3.0], idx, loc)
    assert e3.N == r2.N == N + 1
    np.testing.assert_array_equal(e3.get_x(), r2.get_x())
    np.testing.assert_array_equal(e3.get_P(), r2.get_P())
    # unknown correspondence with the position cost in the likelihood, whole scan through ekf_measure on the sharded handle:
    # every row's association exchanges its candidates (ekf_associate_begin -> ncclAllGather -> finish inside the library)
    e4 = Engine(mode="uc", capacity=N + 2, tile=32, batch=4, force_sharded=1)
    r4 = Engine(mode="uc", capacity=N + 2, tile=32, batch=4)
    raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
    assert L.lib().ekf_comm_unique_id(raw) == 0
    e4.comm_init(raw.raw)
    xs = r2.get_x()
    rows = []
    for k in (4, 70, 4):
        dx, dy = x[3 + 2 * k] - x[0], x[4 + 2 * k] - x[1]
        rows.append([np.hypot(dx, dy), (np.degrees(np.arctan2(dy, dx)) - x[2] - 3.0) %% 360.0, k + 1.0])
    rows.append([3.0, 10.0, 9999.0])                     # matches nothing within the threshold -> appended
    obs4 = np.array(rows)
    for eng in (e4, r4):
        eng.set_params(w_pos=1.0, s_cost=1e6, s_thresh=50.0)
        eng.set_state(x, P, s)
        eng.predict([0.1, 3.0])
        eng.measure(obs4, [0.1, 3.0], np.array([N + 1.0]), np.array([[1.5, -2.5]]))
    assert e4.N == r4.N and e4.N in (N, N + 1), (e4.N, r4.N)
    np.testing.assert_array_equal(e4.get_x(), r4.get_x())
    np.testing.assert_array_equal(e4.get_P(), r4.get_P())
print("TRANSPORT", transport)
"""


@pytest.mark.parametrize("mode,expect", [("rccl", "rccl-native"), ("torch", "torch.distributed")])
def test_one_rank_communicators(mode, expect):
    """Real RCCL communicator / torch.distributed (nccl) all-gather with a single rank, in a child process."""
    out = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}, mode], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "TRANSPORT " + expect in out.stdout


@pytest.mark.parametrize("world,batch,tile", [(2, 8, 16), (4, 8, 32), (8, 16, 16), (3, 5, 64)])
def test_prefetched_row_panels_need_no_per_step_exchange(world, batch, tile):
    """ekf_prefetch_rows: ONE exchange carries the base row-panels of the next `batch` corrections; each of them then
    runs without an exchange of its own (ekf_correct on a shard with no communicator would otherwise fail) and the
    result is bit-identical to the unsharded engine."""
    from ekf_slam_amd import Engine, EkfError, _lib as L
    from ekf_slam_amd.sharding import ShardGroup
    N = 140
    x, P, s, _, _ = _state(N, 29)
    g = ShardGroup(world, capacity=N + 4, tile=tile, batch=batch)
    one = Engine(capacity=N + 4, tile=tile, batch=batch)
    g.set_state(x, P, s); one.set_state(x, P, s)
    rng = np.random.default_rng(10)
    plan = [int(i) for i in rng.integers(0, N, size=3 * batch)]
    plan[3] = plan[1]                                     # the same landmark twice inside one batch
    for b0 in range(0, len(plan), batch):
        chunk = plan[b0:b0 + batch]
        g.prefetch_rows(sorted(set(chunk)))
        for idx0 in chunk:
            u = [0.1, 3.0]
            z = [rng.uniform(1, 30), rng.uniform(1, 359)]
            R = np.diag([z[0] * .01, z[1] * 5.0])
            g.predict(u); one.predict(u)
            g.correct_local(z, R, idx0); one.correct(z, R, idx0)
        assert g.shards[0].pending() == one.pending() == 0        # the batch boundary flushed
    np.testing.assert_array_equal(g.get_x(), one.get_x())
    np.testing.assert_array_equal(g.get_P(), one.get_P())
    # after the flush the prefetch is gone: a local correction has nothing to work from
    with pytest.raises(EkfError) as ei:
        g.correct_local([5.0, 50.0], np.diag([0.05, 250.0]), plan[0])
    assert ei.value.status == L.EKF_ERR_STATE
    # growing the map drops it as well; the per-step exchange still works and stays bit-identical
    g.prefetch_rows([1, 2, 3])
    pos = rng.uniform(-5, 5, 2)
    R = np.diag([0.2, 40.0])
    g.append([0.1, 3.0], R, pos, N + 1); one.append([0.1, 3.0], R, pos, N + 1)
    with pytest.raises(EkfError):
        g.correct_local([5.0, 50.0], R, 2)
    g.correct([5.0, 50.0], R, 2); one.correct([5.0, 50.0], R, 2)
    g.correct([6.0, 60.0], R, N); one.correct([6.0, 60.0], R, N)
    np.testing.assert_array_equal(g.get_P(), one.get_P())
    g.close(); one.close()


def test_state_load_drops_a_prefetch():
    """A prefetch holds BASE row-panels of the covariance it was taken from: loading another state (ekf_set_P,
    ekf_set_x, ekf_load_lowrank_state, a checkpoint) must drop it, or the next correction on a prefetched landmark would
    silently solve against the old rows.  After the load a local correction has nothing to work from (EKF_ERR_STATE); the
    exchanged correction matches the unsharded engine on the NEW state bit for bit."""
    from ekf_slam_amd import Engine, EkfError, _lib as L
    from ekf_slam_amd.sharding import ShardGroup
    N = 100
    x, P, s, d, U = _state(N, 41)
    x2, P2, s2, d2, U2 = _state(N, 42)
    z, R = [7.0, 80.0], np.diag([0.07, 400.0])
    for how in ("set_state", "set_P_only", "lowrank"):
        g = ShardGroup(2, capacity=N, tile=16, batch=4)
        one = Engine(capacity=N, tile=16, batch=4)
        g.set_state(x, P, s); one.set_state(x, P, s)
        g.prefetch_rows([5, 6])
        if how == "set_state":
            g.set_state(x2, P2, s2); one.set_state(x2, P2, s2)
        elif how == "set_P_only":
            for e in g.shards + [one]:
                Pf = np.asfortranarray(P2)
                e._check(e.lib.ekf_set_P(e.h, Pf.reshape(-1, order="F").ctypes.data_as(L._dp), Pf.shape[0]))
        else:
            g.load_lowrank_state(x2, s2, d2, U2); one.load_lowrank_state(x2, s2, d2, U2)
        with pytest.raises(EkfError) as ei:
            g.correct_local(z, R, 5)
        assert ei.value.status == L.EKF_ERR_STATE, how
        g.correct(z, R, 5); one.correct(z, R, 5)
        np.testing.assert_array_equal(g.get_x(), one.get_x())
        np.testing.assert_array_equal(g.get_P(), one.get_P())
        g.close(); one.close()


def test_measure_is_refused_up_front_on_a_shard_without_communicator():
    """ekf_measure decides append / correct row by row and a correction on a shard needs an exchange: without the
    library-owned communicator it is refused BEFORE any row changes the state (hosts that run the exchange themselves
    drive append / correct_begin / finish per row)."""
    from ekf_slam_amd import EkfError, _lib as L
    from ekf_slam_amd.sharding import ShardGroup
    N = 20
    x, P, s, _, _ = _state(N, 43)
    g = ShardGroup(2, capacity=N + 2, tile=16)
    g.set_state(x, P, s)
    x0 = g.get_x()
    obs = np.array([[5.0, 40.0, 1.0], [8.0, 70.0, N + 1.0]])
    with pytest.raises(EkfError) as ei:
        g.shards[0].measure(obs, [0.1, 3.0], np.arange(1, N + 2.0), np.zeros((N + 1, 2)))
    assert ei.value.status == L.EKF_ERR_STATE and "communicator" in str(ei.value)
    assert g.shards[0].N == N
    np.testing.assert_array_equal(g.shards[0].get_x(), x0)
    g.close()


@pytest.mark.parametrize("world,tile,storage", [(2, 128, "f64"), (3, 64, "f64"), (4, 128, "f64"), (8, 64, "f64"), (2, 128, "f32"), (1, 128, "f64")])
def test_hinted_pass_extracts_the_next_row_panel(world, tile, storage):
    """ekf_hint_next: with cfg.batch = 1 the pass over P that ends a correction also extracts the row-panel of the landmark the
    NEXT correction names (k_downdate_w<..., +rowpanel>), so that step starts with its exchange.  Right hints, wrong hints, no
    hint, a hint overtaken by an append, landmarks on the first / last rows of tiles and of the map: the state must equal the
    unsharded engine's bit for bit throughout, and the extraction must actually have been the launched instance."""
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    N = 200
    x, P, s, _, _ = _state(N, 91)
    kw = dict(capacity=N + 4, tile=tile, storage=storage, batch=1)
    g = ShardGroup(world, **kw) if world > 1 else None
    if g is None:
        import ctypes
        from ekf_slam_amd import _lib as L
        g1 = Engine(force_sharded=1, **kw)
        raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
        assert L.lib().ekf_comm_unique_id(raw) == 0
        g1.comm_init(raw.raw)                    # 1-rank RCCL communicator: the in-place all-gather of the library's own transport
    one = Engine(**kw)
    for e in ([g] if g else [g1]) + [one]:
        e.set_state(x, P, s)
    sh = g if g else g1
    first = g.shards[0] if g else g1
    rng = np.random.default_rng(5)
    T2 = tile // 2                               # landmarks per tile row
    seq = [0, 1, T2 - 1, T2, T2 + 1, N - 1, 0, 2 * T2 - 1, 2 * T2, 77, 77, N - 1, N - 2, 5] + [int(v) for v in rng.integers(0, N, 20)]
    used = 0
    for t, k in enumerate(seq):
        z = [float(rng.uniform(1, 30)), float(rng.uniform(1, 359))]
        R = np.array([[z[0] * .01, 0.001], [0.001, z[1] * 5.0]])
        nxt = seq[t + 1] if t + 1 < len(seq) else None
        mode = t % 5
        if nxt is not None and mode in (0, 1, 2):
            sh.hint_next(nxt)                    # right hint
        elif nxt is not None and mode == 3:
            sh.hint_next((nxt + 3) % N)          # wrong hint: the extraction is wasted, k_rowpanel runs as usual
        sh.predict([0.1, 2.0]); one.predict([0.1, 2.0])
        sh.correct(z, R, k); one.correct(z, R, k)
        if nxt is not None and mode in (0, 1, 2, 3):
            assert "+rowpanel" in first.downdate_kernel_name()[0], first.downdate_kernel_name()
            used += 1
        if t == 9:                               # an append between a hinted pass and the correction it was for
            pos = rng.uniform(-5, 5, 2)
            sh.append([0.1, 2.0], R, pos, N + 1); one.append([0.1, 2.0], R, pos, N + 1)
        np.testing.assert_array_equal(sh.get_x(), one.get_x(), err_msg="step %d" % t)
    assert used > 20
    Ps = sh.get_P()
    np.testing.assert_array_equal(Ps, one.get_P())
    sh.close(); one.close()


@pytest.mark.parametrize("world,tile,batch", [(2, 16, 1), (4, 16, 4), (8, 16, 8), (2, 64, 8), (3, 32, 24)])
def test_device_resident_measure_loop_on_shards(world, tile, batch):
    """EKF_SLAM_UC.measure (EKF_SLAM_UC.m:107-151) on a ShardGroup with cfg.device_assoc = 3, the default: every shard runs the
    device-resident loop -- k_associate for a scan's first row, then per correction k_rowpanel<kDev> (the row-panel of the landmark the
    DEVICE names) -> all-gather -> k_gather<sharded, kDev> (whose epilogue scores the next observation) -- one host thread per shard,
    the exchanges through the hook of transport (d).  Bit-identical to the unsharded device loop, scan after scan."""
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd.sharding import ShardGroup
    from ekf_slam_amd.slam import Landmark
    from ekf_slam_amd.world import make_run
    N, M, ITERS = 150, 6, 14
    _, run = make_run(N, 20260110 + world, 2 + ITERS, policy="nearest", m=M)
    kw = dict(mode="uc", capacity=N, tile=tile, batch=batch)
    one, g = Engine(**kw), ShardGroup(world, **kw)
    assert one.cfg.device_assoc == 3 and all(e.cfg.device_assoc == 3 for e in g.shards)
    for kid in (L.EKF_KERNEL_ASSOCIATE, L.EKF_KERNEL_ROWPANEL, L.EKF_KERNEL_GATHER):
        g.shards[-1].timing_enable(kid, True, launches=4096)
    lm1, lmg = Landmark('SYNTHETIC'), Landmark('SYNTHETIC')
    corrections = scans = 0
    for t, (u, scan) in enumerate(run):
        one.predict(u); g.predict(u)
        n_before = one.N
        for eng, lm in ((one, lm1), (g, lmg)):
            obs = lm.getLandmark(scan, eng.get_x())
            idx, loc = lm.landmarkObj.table()
            eng.measure(obs, u, idx, loc)
        corrections += len(scan) - (one.N - n_before)
        scans += 1
        assert g.N == one.N
        np.testing.assert_array_equal(g.get_x(), one.get_x())
    assert one.N == N
    one.flush(); g.flush()
    np.testing.assert_array_equal(g.get_P(), one.get_P())
    np.testing.assert_allclose(g.digest(), one.digest(), rtol=1e-12)
    # the loop ran on the shards: k_associate launches (the host-mirror path launches none when w_pos == 0), and one device-named
    # extraction + one gather per correction
    n_as, _ = g.shards[-1].timing_read(L.EKF_KERNEL_ASSOCIATE)
    n_rp, _ = g.shards[-1].timing_read(L.EKF_KERNEL_ROWPANEL)
    n_ga, _ = g.shards[-1].timing_read(L.EKF_KERNEL_GATHER)
    assert n_as >= 1 and n_rp == corrections and n_ga == corrections, (n_as, n_rp, n_ga, corrections)
    g.close(); one.close()


def test_device_resident_measure_loop_on_a_one_rank_communicator():
    """The same loop through the library's own RCCL communicator (transport (a)), one rank on this GPU (cfg.force_sharded):
    bit-identical to the unsharded handle."""
    import ctypes
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd.slam import Landmark
    from ekf_slam_amd.world import make_run
    raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
    if L.lib().ekf_comm_unique_id(raw) != 0:
        pytest.skip("librccl not loadable")
    N, M, ITERS = 300, 8, 20
    _, run = make_run(N, 20260117, 2 + ITERS, policy="nearest", m=M)
    kw = dict(mode="uc", capacity=N, tile=32, batch=8)
    one, sh = Engine(**kw), Engine(force_sharded=1, **kw)
    sh.comm_init(raw.raw)
    sh.timing_enable(L.EKF_KERNEL_EXCHANGE, True, launches=4096)
    lms = [Landmark('SYNTHETIC'), Landmark('SYNTHETIC')]
    corrections = 0
    for u, scan in run:
        n_before = one.N
        for eng, lm in zip((one, sh), lms):
            eng.predict(u)
            obs = lm.getLandmark(scan, eng.get_x())
            idx, loc = lm.landmarkObj.table()
            eng.measure(obs, u, idx, loc)
        corrections += len(scan) - (one.N - n_before)
    one.flush(); sh.flush()
    np.testing.assert_array_equal(sh.get_x(), one.get_x())
    np.testing.assert_array_equal(sh.get_P(), one.get_P())
    assert corrections >= ITERS * M and sh.timing_read(L.EKF_KERNEL_EXCHANGE)[0] == corrections      # one ncclAllGather per correction
    one.close(); sh.close()


def _lookahead_run(eng_or_group, steps, batch, announce, threaded):
    """predict + correct over `steps` in batches; the landmarks of a batch are prefetched -- by ekf_prefetch_rows at the batch's start,
    or (announce) by ekf_prefetch_next during the batch before."""
    def drive(e):
        nb = (len(steps) + batch - 1) // batch
        lists = [sorted(set(k for (_, _, _, k) in steps[b * batch:(b + 1) * batch])) for b in range(nb)]
        for b in range(nb):
            if b == 0 or not announce:
                e.prefetch_rows(lists[b])
            if announce and b + 1 < nb:
                e.prefetch_next(lists[b + 1])
            for (u, z, R, k) in steps[b * batch:(b + 1) * batch]:
                e.predict(u)
                e.correct(z, R, k)
        e.flush()
    if threaded:
        eng_or_group.run_threaded(drive)
    else:
        drive(eng_or_group)


@pytest.mark.parametrize("world,tile,batch,storage", [(2, 16, 4, "f64"), (4, 32, 8, "f64"), (3, 16, 20, "f32"), (1, 32, 8, "f64"), (1, 64, 20, "f32")])
def test_prefetch_announced_before_the_pass_is_bit_identical(world, tile, batch, storage):
    """ekf_prefetch_next: the next batch's row-panels extracted in front of the current batch's pass AS THAT PASS WILL LEAVE THEM, their
    all-gather beside the pass (world 1: the library's own 1-rank RCCL communicator and its exchange stream; worlds 2-4: one host
    thread per shard, the exchange hook).  Same bits as prefetching after the pass, and as the unsharded deferred engine."""
    import ctypes
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd.sharding import ShardGroup
    import bench
    N = 200
    w, x, s, d, U = bench.make_state(N, 20260120 + world)
    steps = bench.make_steps(w, N, 5 * batch + 3, [.01, 5.0])
    kw = dict(capacity=N, tile=tile, batch=batch, storage=storage)
    one = Engine(**kw)
    one.load_lowrank_state(x, s, d, U)
    for (u, z, R, k) in steps:
        one.predict(u); one.correct(z, R, k)
    one.flush()
    results = []
    for announce in (False, True):
        if world == 1:
            raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
            if L.lib().ekf_comm_unique_id(raw) != 0:
                pytest.skip("librccl not loadable")
            g = Engine(force_sharded=1, **kw)
            g.comm_init(raw.raw)
        else:
            g = ShardGroup(world, **kw)
        g.load_lowrank_state(x, s, d, U)
        last = g if world == 1 else g.shards[-1]
        last.timing_enable(L.EKF_KERNEL_ROWPANEL, True, launches=4096)
        _lookahead_run(g, steps, batch, announce, threaded=world > 1)
        results.append((g.get_x(), g.get_P(), last.timing_read(L.EKF_KERNEL_ROWPANEL)[0]))
        g.close()
    for (xg, Pg, _) in results:
        np.testing.assert_array_equal(xg, one.get_x())
        np.testing.assert_array_equal(Pg, one.get_P())
    # one extraction launch per batch either way (k_rowpanel_base at the batch's start / k_rowpanel_next in front of the pass): every
    # correction found its landmark prefetched
    nb = (len(steps) + batch - 1) // batch
    assert results[0][2] == nb and results[1][2] == nb, (results[0][2], results[1][2], nb)
    one.close()


def test_prefetch_next_statuses_and_withdrawal():
    """ekf_prefetch_next: a no-op on an unsharded handle; refused (EKF_ERR_STATE) without a communicator or hook, with batch 1, with
    the pass in F32 arithmetic; bad lists refused; an append before the batch completes drops the announcement -- results as without."""
    import ctypes
    from ekf_slam_amd import Engine, _lib as L
    import bench
    N, batch = 120, 4
    w, x, s, d, U = bench.make_state(N, 20260131)
    steps = bench.make_steps(w, N + 1, 3 * batch, [.01, 5.0])
    plain = Engine(capacity=N + 2, tile=16, batch=batch)
    plain.prefetch_next([1, 2])                                      # unsharded: nothing to exchange, nothing to refuse
    raw = ctypes.create_string_buffer(L.EKF_COMM_ID_BYTES)
    if L.lib().ekf_comm_unique_id(raw) != 0:
        pytest.skip("librccl not loadable")
    lone = Engine(force_sharded=1, capacity=N + 2, tile=16, batch=batch)
    lone.load_lowrank_state(x, s, d, U)
    with pytest.raises(L.EkfError) as ei:
        lone.prefetch_next([1, 2])                                   # no communicator, no hook
    assert ei.value.status == L.EKF_ERR_STATE
    lone.comm_init(raw.raw)
    with pytest.raises(L.EkfError) as ei:
        lone.prefetch_next(list(range(batch + 1)))                   # more landmarks than a batch holds
    assert ei.value.status == L.EKF_ERR_INVALID_ARG
    with pytest.raises(L.EkfError) as ei:
        lone.prefetch_next([N + 5])
    assert ei.value.status == L.EKF_ERR_INDEX
    b1 = Engine(force_sharded=1, capacity=N, tile=16, batch=1)
    with pytest.raises(L.EkfError) as ei:
        b1.prefetch_next([1])
    assert ei.value.status == L.EKF_ERR_STATE
    f32 = Engine(force_sharded=1, capacity=N, storage="f32_mixed", batch=batch)
    with pytest.raises(L.EkfError) as ei:
        f32.prefetch_next([1])
    assert ei.value.status == L.EKF_ERR_STATE
    # an append between the announcement and the batch's end: the announcement is dropped, the run equals the plain engine's
    plain.load_lowrank_state(x, s, d, U)
    R2 = np.diag([0.3, 4.0])
    for eng in (plain, lone):
        for t, (u, z, R, k) in enumerate(steps):
            if t % batch == 0 and t + batch < len(steps):
                eng.prefetch_next(sorted(set(kk for (_, _, _, kk) in steps[t + batch:t + 2 * batch] if kk < eng.N)))
            eng.predict(u)
            if t == 1:
                eng.append(u, R2, [3.0, -2.0], float(N + 1))
            eng.correct(z, R, min(k, eng.N - 1))
        eng.flush()
    np.testing.assert_array_equal(lone.get_x(), plain.get_x())
    np.testing.assert_array_equal(lone.get_P(), plain.get_P())
    for e in (plain, lone, b1, f32):
        e.close()
