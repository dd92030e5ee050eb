"""GPU: BASELINE.json's full size (10 000 landmarks, F64).  The structured oracle runs the same predict / correct
steps on a full 20 003 x 20 003 matrix (3.2 GB); parity is checked on x (all of it), on the digests of P
(trace / sum / sum of squares over the lower triangle), on the robot rows and on sampled blocks spread over first,
middle and last tile rows, plus size-independent properties: trace non-increasing across a correction, and the
deferred engine (batch 4, asynchronous flush) equal to the immediate one bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6
N = 10000


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_ten_thousand_landmarks_against_oracle(oracle_lib):
    import bench
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    w, x, s, d, U = bench.make_state(N, 20260104)
    steps = bench.make_steps(w, N, 9, [.01, 5.0])
    n = 3 + 2 * N
    ref = StructuredEKF(N, "known")
    P = ref.raw_P()
    for r0 in range(0, n, 2048):
        r1 = min(n, r0 + 2048)
        P[r0:r1, :n] = U[r0:r1] @ U.T
    P[np.arange(n), np.arange(n)] += d
    ref._x[:n] = x; ref._s[:N] = s
    ref.L.oekf_set_num_landmarks(ref.h, N)
    imm = Engine(capacity=N, batch=1)
    dfr = Engine(capacity=N, batch=4, async_flush=True)      # 9 corrections: two asynchronous flushes + one pending pair
    for e in (imm, dfr):
        e.load_lowrank_state(x, s, d, U)
    tr0 = imm.digest()[0]
    np.testing.assert_allclose(tr0, float(np.trace(P[:n, :n])), rtol=1e-12)
    traces = [tr0]
    for t, (u, z, R, k) in enumerate(steps):
        for e in (imm, dfr, ref):
            e.predict(u)
        if t == 4:
            traces.append(imm.digest()[0])        # after the predict, before the correction
        imm.correct(z, R, k); dfr.correct(z, R, k); ref.correct(z, R, k + 1)
        if t == 4:
            traces.append(imm.digest()[0])
    assert traces[2] <= traces[1] + 1e-9          # a correction never increases trace(P)
    assert dfr.pending() in (1, 5)                # 9 corrections, batch 4: one new pair (+ 4 of a flush still in flight)
    xg = imm.get_x()
    np.testing.assert_array_equal(dfr.get_x(), xg)
    assert rel_err(xg, ref._x[:n]) < REL
    dg_i, dg_d = imm.digest(), dfr.digest()       # flushes the deferred engine
    np.testing.assert_allclose(dg_d, dg_i, rtol=1e-13)
    Pv = ref._P                                   # un-copied view of the oracle's matrix
    tr = float(np.trace(Pv[:n, :n]))
    assert abs(dg_i[0] - tr) / abs(tr) < 1e-9
    # sum over the lower triangle, row blocks at a time (no 3.2 GB temporaries)
    sm = sq = 0.0
    for r0 in range(0, n, 1024):
        r1 = min(n, r0 + 1024)
        blk = Pv[r0:r1, :r1].copy()
        rows = np.arange(r0, r1)[:, None]
        cols = np.arange(r1)[None, :]
        blk[cols > rows] = 0.0
        sm += blk.sum(); sq += (blk * blk).sum()
    assert abs(dg_i[1] - sm) / abs(sm) < 1e-7 and abs(dg_i[2] - sq) / abs(sq) < 1e-7
    # robot rows and sampled blocks (first / middle / last tile rows, on and off the diagonal)
    assert rel_err(imm.get_P_block(0, 0, 3, n), Pv[0:3, :n]) < REL
    rng = np.random.default_rng(1)
    for r0, c0 in [(3, 3), (3 + 2 * 63, 3), (3 + 2 * 5000, 3 + 2 * 4999), (n - 6, 5), (n - 6, n - 6),
                   (3 + 2 * 7777, 3 + 2 * 123)] + [tuple(int(v) for v in rng.integers(3, n - 8, 2)) for _ in range(20)]:
        a = imm.get_P_block(r0, c0, 6, 6)
        assert rel_err(a, Pv[r0:r0 + 6, c0:c0 + 6]) < REL
        np.testing.assert_array_equal(dfr.get_P_block(r0, c0, 6, 6), a)
    imm.close(); dfr.close()
