"""GPU: BASELINE.json configs[4]'s code path -- F32 tile storage / F64 solve, P split over shards, streaming append -- at a
size the oracle finishes in seconds.  `world` shards of one filter live on the one test GPU in one process
(ekf_exchange_local); the kernels are the ones a multi-GPU run uses (k_rowpanel<float>, k_rowpanel_base<float>,
k_gather<float,true,*>, k_flush_mfma<float,256,*>, k_append<float>).

Checked: against the F64 structured oracle at the F32 tolerance of tests/test_f32_storage_gpu.py (1e-6 relative on x and P:
every landmark-block entry is rounded to float once per pass, x / robot block / strip / solves stay F64), and BIT FOR BIT
against the unsharded F32 engine (sharding changes where a tile lives, not one operation on it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_X, TOL_P = 1e-6, 1e-6


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _state(N, seed):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 6))
    return x, np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T, np.arange(1, N + 1.0)


@pytest.mark.parametrize("world,batch", [(2, 1), (4, 1), (8, 1), (2, 8), (4, 8), (8, 8)])
def test_f32_tiles_sharded_streaming_append(world, batch, oracle_lib):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    T = 256
    N = 638                                   # 1276 landmark-block rows: 5 tile rows; the appends below cross into the 6th at 1280
    cap = N + 8
    x, P, s = _state(N, 71)
    g = ShardGroup(world, capacity=cap, tile=T, storage="f32", batch=batch)
    one = Engine(capacity=cap, tile=T, storage="f32", batch=batch)
    ref = StructuredEKF(cap, "known")
    g.set_state(x, P, s); one.set_state(x, P, s); ref.set_state(x, P, s)
    assert rel_err(g.get_P(), P) < 1e-7                         # one float rounding of the landmark block
    rng = np.random.default_rng(17)

    def step(idx0, exchange):
        u = [0.1, 3.0]
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        g.predict(u); one.predict(u); ref.predict(u)
        (g.correct if exchange else g.correct_local)(z, R, idx0)
        one.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
        return u, R

    def grow(u, R):
        pos = rng.uniform(-5, 5, 2)
        sig = g.N + 1
        g.append(u, R, pos, sig); one.append(u, R, pos, sig); ref.append(u, R, pos, sig)

    # per-step exchange (k_rowpanel<float> patches the pending pairs), streaming appends across the tile-row edge 1280
    for k, idx0 in enumerate([0, 127, 128, N - 1, 300, 5]):
        u, R = step(idx0, True)
        if k % 2 == 1:
            grow(u, R)                        # N -> 639, 640 (rows 1276..1279: last slots of tile row 4), 641 (rows 1280-1: tile row 5)
            step(g.N - 1, True)               # correct the landmark just appended
    assert g.N == one.N == ref.N == N + 3 and 2 * g.N > 5 * T
    if batch > 1:
        # one exchange for a batch (k_rowpanel_base<float>), corrections without an exchange of their own, incl. a repeat
        g.flush(); one.flush()
        plan = [3, 640, 200, 3, 639, 77, 511, 512][:batch]
        g.prefetch_rows(sorted(set(plan)))
        for idx0 in plan:
            step(idx0, False)
    xg, Pg = g.get_x(), g.get_P()
    assert not np.isnan(Pg).any()
    ex, eP = rel_err(xg, ref.x), rel_err(Pg, ref.P)
    print("f32 x %d shards, batch %d: x %.2e P %.2e" % (world, batch, ex, eP))
    assert ex < TOL_X and eP < TOL_P
    np.testing.assert_array_equal(xg, one.get_x())
    np.testing.assert_array_equal(Pg, one.get_P())
    np.testing.assert_allclose(g.digest(), one.digest(), rtol=1e-12)
    g.close(); one.close()
