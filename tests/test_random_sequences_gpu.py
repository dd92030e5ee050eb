"""GPU: seeded random operation sequences.  Each seed draws an engine configuration (tile edge, deferred batch, shard count,
asynchronous flush, association weight) and a sequence of the path's operations in random order -- predict, correct, append,
associate (with and without the cost vectors), whole scans through measure(), partial reads of P (which force a flush), a
checkpoint round trip -- and runs it
on (a) the configured engine, (b) the plain engine (one GPU, every correction rewriting P at once) and (c) the structured CPU
oracle.  (a) must equal (b) bit for bit: deferral, sharding, the second stream and the walking direction of the pass are
schedules, not arithmetic.  (b) must equal (c) to 1e-6 (F64 on both sides; measured ~1e-13)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


class _Table:
    """Landmark.m-shaped source handing the oracle the same observed rows / landmark table the GPU call gets."""
    def __init__(self, rows, index, loc):
        class _E:
            def __init__(self, i, l): self.index, self.loc = i, np.asarray(l, dtype=float)
        class _O: pass
        self.rows = np.asarray(rows, dtype=float).reshape(-1, 3)
        self.landmarkObj = _O()
        self.landmarkObj.landmark = [_E(i, l) for i, l in zip(index, loc)]

    def getLandmark(self, laser, x):
        return self.rows


def _initial(N, rng):
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-15, 15, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 5))
    return x, np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T, np.arange(1, N + 1.0)


@pytest.mark.parametrize("seed", range(48))
def test_random_sequence(seed, oracle_lib, tmp_path):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    rng = np.random.default_rng(9000 + seed)
    tile = int(rng.choice([16, 32, 64]))
    batch = int(rng.choice([1, 2, 5, 8]))
    world = int(rng.choice([1, 1, 2, 3, 4]))
    asy = bool(rng.integers(0, 2)) and batch > 1
    w_pos = float(rng.choice([0.0, 1.0]))
    N0 = int(rng.integers(3, 70))
    cap = N0 + 12
    kw = dict(mode="uc", capacity=cap, tile=tile)
    # measure(): the configured engine in a random association mode (3 = the default device-resident loop: decision produced and
    # consumed on the device), the plain engine always host-decided -- two independent routes to the same rows
    assoc = int(rng.choice([3, 3, 1, 2, 0]))
    cfgd = ShardGroup(world, batch=batch, async_flush=asy, **kw) if world > 1 else \
        Engine(batch=batch, async_flush=asy, device_assoc=assoc, **kw)
    plain = Engine(batch=1, device_assoc=0, **kw)
    ref = StructuredEKF(cap, "uc")
    x, P, s = _initial(N0, rng)
    for e in (cfgd, plain):
        e.set_params(w_pos=w_pos, s_cost=50.0, s_thresh=1e9)
        e.set_state(x, P, s)
    ref.w_pos, ref.s_cost, ref.s_thresh = w_pos, 50.0, 1e9
    ref.set_state(x, P, s)
    what = "seed %d: tile %d batch %d world %d async %s w_pos %g N0 %d assoc %d" % (seed, tile, batch, world, asy, w_pos, N0, assoc)

    def obs_of(k):
        xe = plain.get_x()
        dx, dy = xe[3 + 2 * k] - xe[0], xe[4 + 2 * k] - xe[1]
        return [float(np.hypot(dx, dy) + rng.normal(0, .05)), float((np.degrees(np.arctan2(dy, dx)) - xe[2] + rng.normal(0, 1.0)) % 360.0)]

    u = [0.1, 3.0]
    for step in range(45):
        op = rng.choice(["predict", "correct", "correct", "correct", "append", "associate", "read", "checkpoint", "measure"],
                        p=[.2, .2, .15, .1, .1, .1, .07, .03, .05])
        N = plain.N
        if op == "predict":
            u = [float(rng.uniform(0, .3)), float(rng.uniform(-8, 8))]
            for e in (cfgd, plain, ref):
                e.predict(u)
        elif op == "correct":
            k = int(rng.integers(0, N))
            z = obs_of(k)
            R = np.array([[z[0] * .01, 0.001], [0.001, max(z[1], 1.0) * 5.0]])
            cfgd.correct(z, R, k); plain.correct(z, R, k); ref.correct(z, R, k + 1)
        elif op == "append" and N < cap:
            pos = rng.uniform(-15, 15, 2)
            R = np.diag([0.2, 40.0])
            for e in (cfgd, plain, ref):
                e.append(u, R, pos, float(N + 1))
        elif op == "associate":
            k = int(rng.integers(0, N))
            z = obs_of(k) + [float(rng.integers(1, N + 1))]
            R = np.diag([z[0] * .1, max(z[1], 1.0) * 5.0])
            costs = bool(rng.integers(0, 2))
            a, b, c = cfgd.associate(z, R, want_costs=costs), plain.associate(z, R, want_costs=costs), ref.associate(z, R, want_costs=costs)
            assert a[:2] == b[:2] == (c[0], c[1] - 1), what
            if costs:
                np.testing.assert_array_equal(a[2], b[2], err_msg=what)
                np.testing.assert_array_equal(a[3], b[3], err_msg=what)
                np.testing.assert_allclose(b[2], c[2], rtol=1e-6, err_msg=what)
        elif op == "read":
            r0 = int(rng.integers(0, 3 + 2 * N - 1)); nr = int(rng.integers(1, min(9, 3 + 2 * N - r0) + 1))
            c0 = int(rng.integers(0, 3 + 2 * N - 1)); nc = int(rng.integers(1, min(9, 3 + 2 * N - c0) + 1))
            blk_p = plain.get_P_block(r0, c0, nr, nc)
            if world == 1:
                np.testing.assert_array_equal(cfgd.get_P_block(r0, c0, nr, nc), blk_p, err_msg=what)
            assert rel_err(blk_p, ref.P[r0:r0 + nr, c0:c0 + nc]) < REL or np.abs(blk_p).max() < 1e-12, what
        elif op == "measure" and world == 1 and N + 1 < cap:
            # a whole scan through measure() (EKF_SLAM_UC.m:102-152): rows of known landmarks (associated, then corrected) and
            # one row whose signature matches nothing (appended from the table: index = N + 1)
            rows = [obs_of(int(k)) + [float(k + 1)] for k in rng.integers(0, N, size=int(rng.integers(1, 4)))]
            strict = dict(s_cost=1e-3, s_thresh=1.0)      # signature must match: the far row below becomes a new landmark
            for e in (cfgd, plain):
                e.set_params(w_pos=0.0, **strict)
            ref.w_pos, ref.s_cost, ref.s_thresh = 0.0, 1e-3, 1.0
            rows.append([4.0, 77.0, 5000.0 + step])
            idx, loc = np.array([N + 1.0]), rng.uniform(-15, 15, (1, 2))
            cfgd.measure(rows, u, idx, loc); plain.measure(rows, u, idx, loc); ref.measure(None, u, _Table(rows, idx, loc))
            assert cfgd.N == plain.N == ref.N == N + 1, what
            for e in (cfgd, plain):
                e.set_params(w_pos=w_pos, s_cost=50.0, s_thresh=1e9)
            ref.w_pos, ref.s_cost, ref.s_thresh = w_pos, 50.0, 1e9
        elif op == "checkpoint" and world == 1:
            path = str(tmp_path / ("ck_%d.bin" % step))
            cfgd.checkpoint_save(path)
            cfgd.predict([9.0, 9.0])                       # wander off, then come back
            cfgd.checkpoint_load(path)
            os.remove(path)
        np.testing.assert_array_equal(cfgd.get_x(), plain.get_x(), err_msg=what + " step %d %s" % (step, op))
    Pc, Pp = cfgd.get_P(), plain.get_P()
    np.testing.assert_array_equal(Pc, Pp, err_msg=what)
    n = 3 + 2 * plain.N
    assert rel_err(plain.get_x(), ref.x[:n]) < REL and rel_err(Pp, ref.P[:n, :n]) < REL, what
    cfgd.close(); plain.close()
