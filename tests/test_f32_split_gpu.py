"""GPU: cfg.pass_arith = EKF_ARITH_SPLIT3 ("f32_split") -- configs[4]'s mixed precision with the pass over the float tiles on the BF16
matrix pipe: every float operand (the float copies of the pending pairs the F32-arithmetic pass reads) is cut EXACTLY into three
bfloat16 pieces, a product is the sum of six partial products (each exact in float; what is dropped is at most 2^-23 of the product, 0.09 x 2^-24
in the root mean square: tests/test_split3_arith_cpu.py),
summed in float from zero, added to the float tile value once (ekf_slam_amd/csrc/flush32_split.h; EKF_SLAM.m:145 x m).

What is asserted, and against what:
  * the F64 oracle (oracle/ekf_structured, the restatement of EKF_SLAM.m) at the tolerance of the F32-arithmetic pass, 1e-6 relative on x and
    P -- the SAME bar as tests/test_f32_mixed_gpu.py, stated there;
  * the F32-arithmetic engine ("f32_mixed", a k-ordered fmaf chain): NOT bit-identical (another summation order, exact partial
    products), but inside one float rounding of the row's largest entry per pass;
  * closer to the F64-arithmetic pass on the same float tiles ("f32": the update summed in double, rounded once) than the fmaf chain is, or
    as close: the split sum's error is not larger than the fmaf chain's (entry by entry against an F64 sum at 40 000 landmarks:
    scripts/probes/flush32_bench.hip ACC=1, profiles/round4_tuning.md 53: max 3.6 against 4.7, mean 0.21 against 0.28 float ulps of sum |k g|);
  * sharding changes where a tile is updated, not one operation on it: bit-identical to the plain engine.
Passes of up to 27 pairs run the F32-arithmetic kernels (faster there: the pass is HBM-bound either way)."""
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


@pytest.mark.parametrize("batch", [28, 33, 40, 50, 64])
def test_split_arithmetic_pass_against_f64_oracle(batch, oracle_lib):
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 300                                                  # 600 landmark rows: 3 tile rows of 256, the last one ragged
    x, P, s = _state(N, 61)
    e = Engine(capacity=N + 8, storage="f32_split", batch=batch)
    mixed = Engine(capacity=N + 8, storage="f32_mixed", batch=batch)
    plain = Engine(capacity=N + 8, storage="f32", batch=batch)
    ref = StructuredEKF(N + 8, "known")
    for q in (e, mixed, plain, ref):
        q.set_state(x, P, s)
    steps = 2 * batch + 5
    _run([e, mixed, plain], ref, steps, 14, appends=(11, 29))
    assert e.downdate_kernel_name() == ("k_flush_split3<2>", batch), e.downdate_kernel_name()
    e.flush(); mixed.flush(); plain.flush()
    name, pairs = e.downdate_kernel_name()                   # the run's last pass: 5 pairs -- the F32-arithmetic kernel
    assert pairs == 5 and name.startswith("k_flush_mfma32<256,"), (name, pairs)
    ex, eP = rel_err(e.get_x(), ref.x), rel_err(e.get_P(), ref.P)
    mx, mP = rel_err(mixed.get_x(), ref.x), rel_err(mixed.get_P(), ref.P)
    print("split arithmetic, batch %d: x %.2e P %.2e   (fmaf chain: x %.2e P %.2e)" % (batch, ex, eP, mx, mP))
    assert ex < TOL_X and eP < TOL_P
    Pe, Pm, Pp = e.get_P(), mixed.get_P(), plain.get_P()
    # the robot block and the landmarks' own 2x2 blocks never see a float: far below float eps, as in every storage mode
    assert rel_err(Pe[:3, :3], ref.P[:3, :3]) < 5e-7
    d = np.arange(3, 3 + 2 * e.N)
    assert np.abs(Pe[d, d] - ref.P[d, d]).max() / np.abs(ref.P[d, d]).max() < 5e-7
    # not the fmaf chain's bits, but its neighbourhood: both are float sums of the same 2m products added to the same float tile
    assert not np.array_equal(Pe, Pm)
    scale = np.abs(Pp).max()
    assert np.abs(Pe - Pm).max() / scale < 3e-7
    # and no further (root mean square over the landmark block) from the F64-arithmetic pass on the same float tiles -- the update summed in
    # double from the F64 pairs, rounded once -- than the fmaf chain is.  Both read the same float copies of the pairs, so both carry the same
    # operand rounding; what differs is the summation (maxima: single entries, printed, both far below a float ulp of the largest entry).
    de, dm = np.abs(Pe - Pp).max() / scale, np.abs(Pm - Pp).max() / scale
    re, rm = np.sqrt(np.mean((Pe - Pp) ** 2)) / scale, np.sqrt(np.mean((Pm - Pp) ** 2)) / scale
    print("   distance to the F64-arithmetic pass on the same tiles: split max %.2e rms %.2e, fmaf chain max %.2e rms %.2e" % (de, re, dm, rm))
    assert de < 6e-8 and dm < 6e-8
    assert re <= 1.25 * rm


def test_split_arithmetic_needs_float_tiles_of_edge_256():
    from ekf_slam_amd import Engine, _lib as L
    for kw in (dict(storage="f64", pass_arith=L.EKF_ARITH_SPLIT3), dict(storage="f32", tile=128, pass_arith=L.EKF_ARITH_SPLIT3)):
        with pytest.raises(L.EkfError):
            Engine(capacity=64, **kw)
    assert Engine(capacity=64, storage="f32_split").cfg.pass_arith == L.EKF_ARITH_SPLIT3


@pytest.mark.parametrize("world,batch", [(2, 64), (3, 40), (8, 48)])
def test_split_arithmetic_sharded_equals_the_plain_engine_bitwise(world, batch, oracle_lib):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N = 638                                                  # the appends cross the tile-row edge at 1280 rows
    cap = N + 8
    x, P, s = _state(N, 71)
    g = ShardGroup(world, capacity=cap, storage="f32_split", batch=batch)
    one = Engine(capacity=cap, storage="f32_split", batch=batch)
    asy = Engine(capacity=cap, storage="f32_split", batch=batch, async_flush=True)
    ref = StructuredEKF(cap, "known")
    for q in (g, one, asy, ref):
        q.set_state(x, P, s)
    _run([g, one, asy], ref, 2 * batch + 7, 17, appends=(3, 4, 5))
    assert one.downdate_kernel_name()[0] in ("k_flush_split3<2>", "k_flush_mfma32<256,4,2,3,early>")
    xg, Pg = g.get_x(), g.get_P()
    assert g.N == one.N == N + 3 and 2 * g.N > 5 * 256
    assert rel_err(xg, ref.x) < TOL_X and rel_err(Pg, ref.P) < TOL_P
    np.testing.assert_array_equal(xg, one.get_x())
    np.testing.assert_array_equal(Pg, one.get_P())
    # the asynchronous engine's ring start moves (pstart != 0): k_split_pairs resolves the ring; held to the oracle like every float-tile async run
    assert rel_err(asy.get_x(), ref.x) < TOL_X and rel_err(asy.get_P(), ref.P) < TOL_P


def test_split_arithmetic_shrink_and_reload_leave_no_stale_planes(oracle_lib):
    """The bf16 planes are cut afresh from the float copies in front of every pass, over exactly the active tile rows: an engine that ran at a
    larger size, was reloaded smaller and grew again across a tile edge equals a fresh engine bit for bit."""
    from ekf_slam_amd import Engine
    N1, N0, batch = 700, 250, 40
    x1, P1, s1 = _state(N1, 5)
    x0, P0, s0 = _state(N0, 6)
    used = Engine(capacity=N1 + 16, storage="f32_split", batch=batch)
    used.set_state(x1, P1, s1)
    _run([used], None, batch + 3, 3)
    fresh = Engine(capacity=N1 + 16, storage="f32_split", batch=batch)
    for q in (used, fresh):
        q.set_state(x0, P0, s0)
    _run([used, fresh], None, 2 * batch + 9, 9, appends=tuple(range(2, 12)))     # 250 -> 260 landmarks: rows 500 -> 520 cross the edge at 512
    assert used.N == fresh.N == N0 + 10
    np.testing.assert_array_equal(used.get_x(), fresh.get_x())
    np.testing.assert_array_equal(used.get_P(), fresh.get_P())


def test_split_arithmetic_unknown_correspondence_device_loop(oracle_lib):
    """EKF_SLAM_UC.measure (EKF_SLAM_UC.m:107-151) on the split-arithmetic engine at batch 40: the device-resident loop's decisions equal the
    host-decided mode's (the association reads x, s, the strip and the F64 diagonal blocks: nothing the pass's arithmetic touches), so the
    two states are equal bit for bit -- and both are on the oracle."""
    from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
    from ekf_slam_amd.world import SyntheticLandmark, make_run
    from oracle.ekf_structured import StructuredEKF
    _, run = make_run(150, 7, 16, policy="nearest", m=6)
    dev = EKF_SLAM_UC(capacity=160, batch=40, storage="f32_split")
    host = EKF_SLAM_UC(capacity=160, batch=40, storage="f32_split", device_assoc=0)
    ref = StructuredEKF(160, "uc")
    ld, lh, lr = Landmark('SYNTHETIC'), Landmark('SYNTHETIC'), SyntheticLandmark()
    split_passes = 0
    for u, scan in run:
        for e, l in ((dev, ld), (host, lh), (ref, lr)):
            e.predict(u); e.measure(scan, u, l)
        split_passes += dev._e.downdate_kernel_name() == ("k_flush_split3<2>", 40)
    assert split_passes > 0                                  # passes of 40 pairs ran in split arithmetic
    assert dev._e.N == host._e.N == ref.N == 150
    np.testing.assert_array_equal(dev.x, host.x)
    np.testing.assert_array_equal(dev.P, host.P)
    assert rel_err(dev.x, ref.x) < TOL_X and rel_err(dev.P, ref.P) < TOL_P


@pytest.mark.parametrize("storage,batch", [("f32_split", 40), ("f32_split", 64), ("f32_mixed", 64)])
def test_async_pass_with_an_append_every_step_on_float_tiles(storage, batch, oracle_lib):
    """configs[4]'s step on the asynchronous engine with float tiles (tile edge 256): every step appends a landmark beside the pass in flight --
    the new rows go to the store the pass reads and are copied to the other one when the pass retires (k_copy_tile_rows<float>), across the
    tile-row edge at 1 280 rows.  Not the synchronous engine's bits (the corrections beside a pass read its pairs unrounded): both are held to
    the F64 oracle at the float-tile tolerance, and to each other."""
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 560
    steps = 3 * batch + 9                                     # 560 -> 560 + steps landmarks: rows 1 120 -> > 1 500
    x, P, s = _state(N, 83)
    syn = Engine(capacity=N + steps, storage=storage, batch=batch)
    asy = Engine(capacity=N + steps, storage=storage, batch=batch, async_flush=True)
    ref = StructuredEKF(N + steps, "known")
    for q in (syn, asy, ref):
        q.set_state(x, P, s)
    _run([syn, asy], ref, steps, 23, appends=tuple(range(steps)))
    assert syn.N == asy.N == ref.N == N + steps and 2 * N < 1280 < 2 * asy.N
    xs, xa, Ps, Pa = syn.get_x(), asy.get_x(), syn.get_P(), asy.get_P()
    # (one landmark appended per step with R up to diag(0.3, 1 800): after ~200 steps the float tiles are 1e-6 of max |P| from the oracle
    # on BOTH engines -- the tolerance here is 3e-6, and the two engines must agree with each other as closely as with the oracle)
    errs = dict(xa=rel_err(xa, ref.x), Pa=rel_err(Pa, ref.P), xs=rel_err(xs, ref.x), Ps=rel_err(Ps, ref.P), x_as=rel_err(xa, xs), P_as=rel_err(Pa, Ps))
    assert max(errs["xa"], errs["xs"], errs["x_as"]) < 3 * TOL_X and max(errs["Pa"], errs["Ps"], errs["P_as"]) < 3 * TOL_P, errs
    assert errs["Pa"] < 2 * errs["Ps"] + 1e-7, errs          # the asynchronous engine is not the less accurate one
    # the appended landmarks' rows made it into the store that is current now: no zero rows where the oracle has entries
    new_rows = slice(3 + 2 * N, 3 + 2 * asy.N)
    assert np.abs(Pa[new_rows, :3 + 2 * N]).max() > 0 and rel_err(Pa[new_rows], ref.P[new_rows]) < 1e-5
