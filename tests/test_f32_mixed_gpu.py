"""GPU: "F32 mixed precision with F64 innovation solve" (BASELINE.json configs[4]) -- cfg.storage = EKF_STORE_F32 with
cfg.pass_arith = EKF_ARITH_F32: the pass over the float tiles runs on the f32 matrix pipe (k_flush_mfma32: -K and G rounded to float,
products accumulated in float), while the innovation, S, its inverse, K, x, the robot block, the strip and the landmarks' 2x2 diagonal
blocks stay F64 (EKF_SLAM.m:124-145 is F64 throughout).  The pass's update -sum K_i G_i is summed in float from zero and added to the float
tile value once: one rounding at the entry's magnitude per pass, as with the F64-arithmetic pass.

Tolerance, stated here: 1e-6 relative (max-norm) on x and P against the F64 oracle after 40 update-steps from a dense random state
(measured 0.4-6e-8 on x, 0.8-5e-9 on P, the same as the F64-arithmetic pass on the same float tiles); at configs[4]'s length the
drift bound of tests/test_f32_drift_gpu.py holds unchanged, and at its real size AND length tests/test_full_size_gpu.py.  Sharding and the association modes change WHERE a tile is updated and who
decides, not one operation on it: bit-identical to the plain engine of the same batch.  (The asynchronous pass is not, with float tiles of
either arithmetic: the corrections that run beside a pass read the float tiles of BEFORE it plus its pairs in F64 -- unrounded -- where the
synchronous engine reads the rounded result; it is held to the oracle tolerance instead.  With F64 tiles nothing is rounded and all
schedules agree bit for bit, tests/test_deferred_gpu.py.)"""
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


def _run(engines, ref, steps, seed, appends=()):
    rng = np.random.default_rng(seed)
    for step in range(steps):
        u = [0.1, 3.0]
        idx0 = int(rng.integers(0, engines[0].N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        for e in engines:
            e.predict(u); e.correct(z, R, idx0)
        if ref is not None:
            ref.predict(u); ref.correct(z, R, idx0 + 1)
        if step in appends:
            pos = rng.uniform(-5, 5, 2)
            for e in engines:
                e.append(u, R, pos, e.N + 1)
            if ref is not None:
                ref.append(u, R, pos, ref.N + 1)


def _pass_kernel(pairs):
    """which kernel a pass of `pairs` pending pairs over float tiles in F32 arithmetic runs (kernels.hip::launch_flush_mfma)"""
    return "k_flush_strip32<" if pairs > 56 else "k_flush_mfma32<256," if pairs > 2 else "k_flush_mfma<float,256,"


@pytest.mark.parametrize("batch", [1, 8, 13, 40, 57, 64])
def test_f32_arithmetic_pass_against_f64_oracle(batch, oracle_lib):
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 300                                                  # 600 landmark rows: 3 tile rows of 256, the last one ragged
    x, P, s = _state(N, 61)
    e = Engine(capacity=N + 8, storage="f32_mixed", batch=batch)
    plain = Engine(capacity=N + 8, storage="f32", batch=batch)
    ref = StructuredEKF(N + 8, "known")
    for q in (e, plain, ref):
        q.set_state(x, P, s)
    steps = 40 if batch < 57 else 2 * batch + 5              # (the strip form needs a full batch of 57-64 pairs to run at all)
    _run([e, plain], ref, steps, 14, appends=(11, 29))
    if batch >= 57:
        assert e.downdate_kernel_name() == ("k_flush_strip32<8>", batch), e.downdate_kernel_name()
    e.flush()
    name, pairs = e.downdate_kernel_name()                   # the run's last pass: steps % batch pairs (one or two pairs take the F64-arithmetic kernel)
    assert name.startswith(_pass_kernel(pairs)), (name, pairs)
    assert plain.downdate_kernel_name()[0].startswith("k_flush_mfma<float,256,")
    ex, eP = rel_err(e.get_x(), ref.x), rel_err(e.get_P(), ref.P)
    px, pP = rel_err(plain.get_x(), ref.x), rel_err(plain.get_P(), ref.P)
    print("f32 arithmetic, batch %d: x %.2e P %.2e   (F64 arithmetic on the same tiles: x %.2e P %.2e)" % (batch, ex, eP, px, pP))
    assert ex < TOL_X and eP < TOL_P
    # the robot block and the landmarks' own 2x2 blocks never see a float: they agree with the oracle far below float eps
    Pg = e.get_P()
    assert rel_err(Pg[:3, :3], ref.P[:3, :3]) < 5e-7
    d = np.arange(3, 3 + 2 * e.N)
    assert np.abs(Pg[d, d] - ref.P[d, d]).max() / np.abs(ref.P[d, d]).max() < 5e-7


def test_f32_arithmetic_needs_float_tiles_of_edge_256():
    from ekf_slam_amd import Engine, _lib as L
    for kw in (dict(storage="f64", pass_arith=L.EKF_ARITH_F32), dict(storage="f32", tile=128, pass_arith=L.EKF_ARITH_F32),
               dict(storage="f32", pass_arith=7)):
        with pytest.raises(L.EkfError):
            Engine(capacity=64, **kw)
    assert Engine(capacity=64, storage="f32_mixed").cfg.pass_arith == L.EKF_ARITH_F32
    with pytest.raises(TypeError):
        Engine(capacity=64, storage="f16")


@pytest.mark.parametrize("world,batch", [(2, 1), (4, 8), (8, 20), (2, 64), (3, 60)])
def test_f32_arithmetic_sharded_equals_the_plain_engine_bitwise(world, batch, oracle_lib):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N = 638                                                  # the appends cross the tile-row edge at 1280 rows
    cap = N + 8
    x, P, s = _state(N, 71)
    g = ShardGroup(world, capacity=cap, storage="f32_mixed", batch=batch)
    one = Engine(capacity=cap, storage="f32_mixed", batch=batch)
    asy = Engine(capacity=cap, storage="f32_mixed", batch=batch, async_flush=True)
    ref = StructuredEKF(cap, "known")
    for q in (g, one, asy, ref):
        q.set_state(x, P, s)
    _run([g, one, asy], ref, 2 * batch + 7, 17, appends=(3, 4, 5))
    xg, Pg = g.get_x(), g.get_P()
    assert g.N == one.N == N + 3 and 2 * g.N > 5 * 256
    assert rel_err(xg, ref.x) < TOL_X and rel_err(Pg, ref.P) < TOL_P
    np.testing.assert_array_equal(xg, one.get_x())
    np.testing.assert_array_equal(Pg, one.get_P())
    assert rel_err(asy.get_x(), ref.x) < TOL_X and rel_err(asy.get_P(), ref.P) < TOL_P


def test_f32_arithmetic_unknown_correspondence_device_loop(oracle_lib):
    """EKF_SLAM_UC.measure on the mixed-precision engine: the device-resident loop's decisions (association from float tiles + F64
    pending pairs + the F64 diagonal blocks) equal the host-decided mode's, so the two states are equal bit for bit."""
    from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
    from ekf_slam_amd.world import SyntheticLandmark, make_run
    from oracle.ekf_structured import StructuredEKF
    _, run = make_run(150, 7, 14, policy="nearest", m=6)
    dev = EKF_SLAM_UC(capacity=160, batch=8, storage="f32_mixed")
    host = EKF_SLAM_UC(capacity=160, batch=8, storage="f32_mixed", device_assoc=0)
    ref = StructuredEKF(160, "uc")
    ld, lh, lr = Landmark('SYNTHETIC'), Landmark('SYNTHETIC'), SyntheticLandmark()
    for u, scan in run:
        for e, l in ((dev, ld), (host, lh), (ref, lr)):
            e.predict(u); e.measure(scan, u, l)
    assert dev._e.N == host._e.N == ref.N == 150
    np.testing.assert_array_equal(dev.x, host.x)
    np.testing.assert_array_equal(dev.P, host.P)
    assert rel_err(dev.x, ref.x) < TOL_X and rel_err(dev.P, ref.P) < TOL_P


@pytest.mark.parametrize("how", ["set_state", "lowrank", "checkpoint"])
def test_a_reused_handle_leaves_no_stale_float_pairs_beyond_a_smaller_map(how, tmp_path):
    """The float copies of the pending pairs (DevState::Gp32 / Kp32) must read as zero beyond the active columns, like the F64 pairs: a
    handle that ran at N1 landmarks and is then loaded with a smaller state N0 < N1, streams appends across a 256-column tile edge with
    several pairs pending -- its pass reads whole tile-wide slices of K and G, i.e. slots' columns beyond the map as it was when the slot
    was written.  Equal bit for bit to a fresh handle given the same state and steps."""
    from ekf_slam_amd import Engine
    N1, N0, batch = 700, 380, 6                              # 760 rows: the appends cross the tile edge at 768 rows
    cap = N1 + 8
    used = Engine(capacity=cap, storage="f32_mixed", batch=batch)
    fresh = Engine(capacity=cap, storage="f32_mixed", batch=batch)
    x1, P1, s1 = _state(N1, 5)
    used.set_state(x1, P1, s1)
    _run([used], None, 2 * batch + 3, 3)                     # every ring slot written at 1 400 columns, three pairs left pending
    rng = np.random.default_rng(8)
    n0 = 3 + 2 * N0
    x0 = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N0)])
    d0, U0 = rng.uniform(0.01, 0.1, size=n0), rng.normal(0, 0.05, size=(n0, 6))
    s0 = np.arange(1, N0 + 1.0)
    if how == "set_state":
        P0 = np.diag(d0) + U0 @ U0.T
        used.set_state(x0, P0, s0); fresh.set_state(x0, P0, s0)
    elif how == "lowrank":
        used.load_lowrank_state(x0, s0, d0, U0); fresh.load_lowrank_state(x0, s0, d0, U0)
    else:
        fresh.load_lowrank_state(x0, s0, d0, U0)
        fresh.checkpoint_save(tmp_path / "n0.ckpt")
        used.checkpoint_load(tmp_path / "n0.ckpt")
    _run([used, fresh], None, 3 * batch + 1, 21, appends=(1, 2, 3, 4, 5, 7, 8, 9))     # 388 landmarks = 776 rows > 768
    assert used.N == fresh.N == N0 + 8 and 2 * used.N > 768
    used.flush(); fresh.flush()
    assert used.downdate_kernel_name()[0].startswith("k_flush_")
    np.testing.assert_array_equal(used.get_x(), fresh.get_x())
    np.testing.assert_array_equal(used.get_P(), fresh.get_P())
