"""GPU: deferred rank-2m downdate (cfg.batch > 1).  Pending (K,G) pairs are patched into the rows later
corrections / associations read and applied to P in one pass; the results must be BIT-IDENTICAL to the
immediate path (batch = 1), whatever is interleaved (predicts, appends, associations, sharding), and within
1e-6 of the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _state(N, seed):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 6))
    return x, np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T, np.arange(1, N + 1.0)


@pytest.mark.parametrize("batch,tile", [(1, 16), (1, 128), (2, 16), (5, 16), (8, 64), (16, 32), (64, 64)])
def test_deferred_equals_immediate_bitwise(batch, tile, oracle_lib):
    # batch 1 exercises the asynchronous engine alone: every update-step's pass over P runs on the second stream beside the next
    # step's gather, which patches its rows with the pair still in flight
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 140
    x, P, s = _state(N, 41)
    imm = Engine(capacity=N + 30, tile=tile, batch=1)
    dfr = Engine(capacity=N + 30, tile=tile, batch=batch)
    asy = Engine(capacity=N + 30, tile=tile, batch=batch, async_flush=True)     # flush on a second stream, two tile stores
    ref = StructuredEKF(N + 30, "known")
    for e in (imm, dfr, asy, ref):
        e.set_state(x, P, s)
    rng = np.random.default_rng(8)
    for step in range(37):
        u = [0.1, 3.0]
        for e in (imm, dfr, asy, ref):
            e.predict(u)
        for _ in range(int(rng.integers(1, 4))):
            idx0 = int(rng.integers(0, imm.N))
            z = [rng.uniform(1, 30), rng.uniform(1, 359)]
            R = np.diag([z[0] * .01, z[1] * 5.0])
            imm.correct(z, R, idx0); dfr.correct(z, R, idx0); asy.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
        if step % 5 == 2:                  # grow the map while pairs are pending
            pos, sig = rng.uniform(-5, 5, 2), imm.N + 1
            R = np.diag([0.2, 40.0])
            for e in (imm, dfr, asy, ref):
                e.append(u, R, pos, sig)
            z = [rng.uniform(1, 30), rng.uniform(1, 359)]
            imm.correct(z, R, imm.N - 1); dfr.correct(z, R, dfr.N - 1); asy.correct(z, R, asy.N - 1); ref.correct(z, R, ref.N)
        np.testing.assert_array_equal(dfr.get_x(), imm.get_x())      # x is always current, no flush involved
        np.testing.assert_array_equal(asy.get_x(), imm.get_x())
    assert batch <= 2 or dfr.pending() > 0 or batch > 37
    Pd, Pi = dfr.get_P(), imm.get_P()                                 # get_P flushes
    assert dfr.pending() == 0
    np.testing.assert_array_equal(Pd, Pi)
    np.testing.assert_array_equal(asy.get_P(), Pi)
    assert asy.pending() == 0
    assert rel_err(Pd, ref.P) < REL and rel_err(dfr.get_x(), ref.x) < REL
    np.testing.assert_array_equal(dfr.digest(), imm.digest())
    np.testing.assert_array_equal(imm.digest(), imm.digest())         # and repeatable


def test_deferred_uc_slam_run_matches_golden():
    from golden_util import load, rel_err as rerr, replay_slam
    from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
    g = load("slam120_uc_nearest.npz")
    e = EKF_SLAM_UC(capacity=128, tile=32, batch=16, Rc=list(g["Rc"]))
    poses = replay_slam(e, Landmark('SYNTHETIC'), g)
    assert rerr(poses, g["poses"]) < REL and rerr(e.x, g["x"]) < REL and rerr(e.P, g["P"]) < REL


def test_deferred_association_costs_see_pending_pairs(oracle_lib):
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 300
    x, P, s = _state(N, 43)
    e = Engine(mode="uc", capacity=N, tile=64, batch=8)
    ref = StructuredEKF(N, "uc")
    e.set_state(x, P, s); ref.set_state(x, P, s)
    rng = np.random.default_rng(2)
    for idx0 in (0, 150, 299, 31, 32):
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .1, z[1] * 5.0])
        e.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
    assert e.pending() == 5
    R = np.diag([1.0, 50.0])
    new_g, idx_g, pc_g, sc_g = e.associate([7.0, 123.0, 151.0], R, want_costs=True)
    new_r, idx_r, pc_r, sc_r = ref.associate([7.0, 123.0, 151.0], R, want_costs=True)
    assert (new_g, idx_g + 1) == (new_r, idx_r) == (False, 151)
    assert rel_err(pc_g, pc_r) < REL
    assert e.pending() == 5


@pytest.mark.parametrize("world,batch,asy", [(2, 4, False), (3, 7, False), (4, 16, False), (2, 4, True), (3, 5, True), (2, 1, True), (4, 1, True)])
def test_deferred_sharded_bitwise(world, batch, asy):
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    N = 120
    x, P, s = _state(N, 47)
    g = ShardGroup(world, capacity=N + 8, tile=16, batch=batch, async_flush=asy)
    one = Engine(capacity=N + 8, tile=16, batch=1)
    g.set_state(x, P, s); one.set_state(x, P, s)
    rng = np.random.default_rng(6)
    for step in range(23):
        u = [0.1, 3.0]
        g.predict(u); one.predict(u)
        idx0 = int(rng.integers(0, one.N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        g.correct(z, R, idx0); one.correct(z, R, idx0)
        if step == 11:
            pos = rng.uniform(-5, 5, 2)
            g.append(u, R, pos, one.N + 1); one.append(u, R, pos, one.N + 1)
    np.testing.assert_array_equal(g.get_x(), one.get_x())
    np.testing.assert_array_equal(g.get_P(), one.get_P())
    g.close()


def test_fused_predict_equals_standalone_predict_bitwise(oracle_lib):
    """ekf_predict is lazy: the next correction folds it into its gather kernel.  Forcing the standalone k_predict
    (any read of x materialises it) must not change a single bit."""
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 130
    x, P, s = _state(N, 53)
    fused = Engine(capacity=N, tile=64, batch=4)
    split = Engine(capacity=N, tile=64, batch=4)
    ref = StructuredEKF(N, "known")
    for e in (fused, split, ref):
        e.set_state(x, P, s)
    rng = np.random.default_rng(12)
    for step in range(11):
        u = [0.1 + 0.01 * step, 3.0 - step]
        fused.predict(u); split.predict(u); ref.predict(u)
        split.get_x()                                   # materialises the predict on its own
        if step == 5:
            fused.predict(u); split.predict(u); ref.predict(u)      # two predicts in a row
        idx0 = int(rng.integers(0, N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        fused.correct(z, R, idx0); split.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
    np.testing.assert_array_equal(fused.get_Q3(), split.get_Q3())
    np.testing.assert_array_equal(fused.get_x(), split.get_x())
    np.testing.assert_array_equal(fused.get_P(), split.get_P())
    assert rel_err(fused.get_P(), ref.P) < REL and rel_err(fused.get_x(), ref.x) < REL


@pytest.mark.parametrize("N,tile,batch,storage", [(0, 16, 1, "f64"), (5, 16, 1, "f64"), (130, 64, 4, "f64"), (1100, 0, 2, "f64"), (300, 0, 8, "f32")])
def test_predict_folded_into_append_equals_standalone_predict_bitwise(N, tile, batch, storage, oracle_lib):
    """predict -> append (the streaming-append step of BASELINE configs[4], and a new landmark on a scan's first row): the append launch
    carries the recorded predict out itself (k_append<.., kPredict>).  Forcing the standalone predict kernel first (any read of x
    materialises it; k_predict_mfma at >= 1024 landmarks) must not change a single bit; both follow the oracle."""
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    cap = N + 12
    fused = Engine(capacity=cap, tile=tile, batch=batch, storage=storage)
    split = Engine(capacity=cap, tile=tile, batch=batch, storage=storage)
    ref = StructuredEKF(cap, "known")
    if N:
        x, P, s = _state(N, 57)
        for e in (fused, split, ref):
            e.set_state(x, P, s)
    rng = np.random.default_rng(21)
    for step in range(9):
        u = [0.1 + 0.02 * step, 7.0 * step - 13.0]
        fused.predict(u); split.predict(u); ref.predict(u)
        split.get_x()                                   # materialises the predict on its own
        if step == 4:
            fused.predict(u); split.predict(u); ref.predict(u)      # two predicts in a row: the first one is materialised by the second
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        pos = rng.uniform(-5, 5, 2)
        fused.append(u, R, pos, fused.N + 1); split.append(u, R, pos, split.N + 1); ref.append(u, R, pos, ref.N + 1)
        if step % 2:
            idx0 = int(rng.integers(0, fused.N))
            fused.correct(z, R, idx0); split.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
    assert fused.N == split.N == ref.N == N + 9
    np.testing.assert_array_equal(fused.get_Q3(), split.get_Q3())
    np.testing.assert_array_equal(fused.get_x(), split.get_x())
    np.testing.assert_array_equal(fused.get_P(), split.get_P())
    np.testing.assert_array_equal(fused.get_s(), split.get_s())
    tol = REL if storage == "f64" else 1e-6
    assert rel_err(fused.get_P(), ref.P) < tol and rel_err(fused.get_x(), ref.x) < tol


def test_mfma_predict_panel_matches_valu_and_oracle(oracle_lib):
    """At >= 1024 landmarks the standalone predict runs its 3x3 * 3x2N panel product on v_mfma_f64_16x16x4_f64; the
    predict folded into a correction uses plain FMAs.  The f64 MFMA is a k-ordered chain of correctly rounded FMAs
    (scripts/probes/mfma_f64_order.*), so with the identity / F(1:2,3) operands it performs exactly fma(fa, s2, s0):
    the two paths agree bit for bit."""
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 1100                                        # 2200 strip columns: MFMA path, ragged last 16-column slice
    x, P, s = _state(N, 59)
    fused = Engine(capacity=N, batch=2)
    split = Engine(capacity=N, batch=2)
    ref = StructuredEKF(N, "known")
    for e in (fused, split, ref):
        e.set_state(x, P, s)
    rng = np.random.default_rng(3)
    for step in range(5):
        u = [0.3 + 0.1 * step, 17.0 * step - 20.0]
        fused.predict(u); split.predict(u); ref.predict(u)
        split.get_x()                               # materialises the predict: k_predict_mfma
        idx0 = int(rng.integers(0, N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        fused.correct(z, R, idx0); split.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
    fused.predict([0.2, 5.0]); split.predict([0.2, 5.0]); ref.predict([0.2, 5.0])
    Pf, Ps = fused.get_P(), split.get_P()
    np.testing.assert_array_equal(fused.get_x(), split.get_x())
    np.testing.assert_array_equal(Pf, Ps)
    assert rel_err(Pf, ref.P) < REL and rel_err(fused.get_x(), ref.x) < REL


@pytest.mark.parametrize("batch", [2, 3, 5, 8, 9, 13, 27, 32, 33, 64])
def test_mfma_flush_equals_immediate_bitwise(batch):
    """Production tiles (T = 128, F64): two or more pending pairs are applied by k_flush_mfma on the matrix cores.  Odd pair
    counts (padded k-step), chunk boundaries (4 / 8 pairs per LDS chunk, both instances) and partial final batches must all
    reproduce the one-pair VALU downdate bit for bit, signed zeros included."""
    from ekf_slam_amd import Engine
    N = 300                                         # 600 landmark rows: 5 tile rows, ragged last tile
    x, P, s = _state(N, 61)
    imm = Engine(capacity=N, tile=128, batch=1)
    dfr = Engine(capacity=N, tile=128, batch=batch)
    imm.set_state(x, P, s); dfr.set_state(x, P, s)
    rng = np.random.default_rng(batch)
    steps = 2 * batch + 3                           # two full flushes and a partial one
    for step in range(steps):
        imm.predict([0.1, 2.0]); dfr.predict([0.1, 2.0])
        idx0 = int(rng.integers(0, N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        imm.correct(z, R, idx0); dfr.correct(z, R, idx0)
    assert dfr.pending() == steps % batch
    Pd, Pi = dfr.get_P(), imm.get_P()
    np.testing.assert_array_equal(Pd, Pi)
    assert Pd.tobytes() == Pi.tobytes()             # signed zeros too
    np.testing.assert_array_equal(dfr.get_x(), imm.get_x())
    np.testing.assert_array_equal(dfr.digest(), imm.digest())      # fixed reduction order: equal states, equal digests


def _flush_run(storage, tile, batch, **cfg):
    """2*batch + 3 corrections on a 300-landmark map; returns P (the final get_P flushes the partial batch)."""
    from ekf_slam_amd import Engine
    N = 300
    rng = np.random.default_rng(71)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-20, 20, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 6))
    P = np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T
    e = Engine(capacity=N, tile=tile, storage=storage, batch=batch, **cfg)
    e.set_state(x, P, np.arange(1, N + 1.0))
    for step in range(2 * batch + 3):
        e.predict([0.1, 2.0])
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        e.correct(z, np.diag([z[0] * .01, z[1] * 5.0]), int(rng.integers(0, N)))
    name = e.downdate_kernel_name()[0]
    Pout = e.get_P()
    e.close()
    return Pout, name


@pytest.mark.parametrize("storage,tile,valu_tile,batch", [("f32", 256, 128, 5), ("f32", 256, 128, 12), ("f32", 256, 128, 33),
                                                          ("f64", 128, 64, 33), ("f64", 128, 64, 7)])
def test_mfma_flush_equals_valu_flush_bitwise(storage, tile, valu_tile, batch):
    """The matrix-core flush (f64 tiles T = 128, f32 tiles T = 256) against the VALU flush k_downdate_w with several pairs, which
    the same engine runs at the next smaller tile edge -- the tile edge changes the kernel, not the arithmetic (per element: the
    pending pairs in slot order, two FMAs each; one rounding to float per flush with F32 tiles).  With F32 tiles a deferred run is
    not bit-equal to an immediate one (rounded once per flush, not once per correction), so this is the bitwise check of the
    F32 instance."""
    P_mfma, k_mfma = _flush_run(storage, tile, batch)
    P_valu, k_valu = _flush_run(storage, valu_tile, batch)
    assert k_mfma.startswith("k_flush_mfma") and k_valu.startswith("k_downdate_w"), (k_mfma, k_valu)
    assert np.isfinite(P_mfma).all()
    assert P_mfma.tobytes() == P_valu.tobytes()


@pytest.mark.parametrize("storage,tile,batch", [("f64", 16, 1), ("f64", 128, 1), ("f64", 128, 6), ("f32", 256, 3), ("f64", 64, 4)])
def test_pass_direction_does_not_change_a_bit(storage, tile, batch):
    """Above 256 MiB of tiles every other pass over P walks its work list backwards (the Infinity Cache then serves what the
    previous pass wrote last).  Every element is updated independently, so the order must not matter: forced on
    for a small map (cfg.pass_direction = 2), every pass kernel -- generic, one-pair streaming, VALU and MFMA flushes -- against
    always-forwards (cfg.pass_direction = 1), bit for bit.  (At full size the 10 k-landmark tests run with the alternation on by
    itself.)"""
    P_alt, _ = _flush_run(storage, tile, batch, pass_direction=2)
    P_fwd, _ = _flush_run(storage, tile, batch, pass_direction=1)
    assert np.isfinite(P_alt).all()
    assert P_alt.tobytes() == P_fwd.tobytes()


def test_marshalled_steps_equal_plain_calls():
    """Engine.marshal_steps / step_raw (pre-marshalled inputs, integer addresses only) is the same sequence of ABI calls as
    predict() + correct() with per-call conversion."""
    from ekf_slam_amd import Engine
    N = 150
    x, P, s = _state(N, 67)
    a = Engine(capacity=N, tile=64, batch=4)
    b = Engine(capacity=N, tile=64, batch=4)
    a.set_state(x, P, s); b.set_state(x, P, s)
    rng = np.random.default_rng(4)
    steps = []
    for _ in range(19):
        z = np.array([rng.uniform(1, 30), rng.uniform(1, 359)])
        steps.append(([0.1, float(rng.uniform(-3, 3))], z, np.array([[z[0] * .01, 0.002], [0.002, z[1] * 5.0]]), int(rng.integers(0, N))))
    for (u, z, R, k) in steps:
        a.predict(u); a.correct(z, R, k)
    run = b.marshal_steps(steps)
    for i in range(run["m"]):
        b.step_raw(run, i)
    np.testing.assert_array_equal(a.get_x(), b.get_x())
    np.testing.assert_array_equal(a.get_P(), b.get_P())


@pytest.mark.parametrize("tile", [16, 128])
def test_small_map_fused_downdate_equals_the_two_launch_form_bitwise(tile, oracle_lib):
    """Up to 24 landmarks an immediate-mode correction applies its rank-2 downdate inside the gather kernel (one launch per
    update-step); beyond, and in deferred mode, k_downdate / the flush does it.  Same arithmetic, so a batch-1 engine growing
    from 18 to 28 landmarks (fused, then two launches) must equal a batch-3 engine (never fused) bit for bit, and the oracle
    to 1e-6."""
    from ekf_slam_amd import Engine
    from oracle.ekf_structured import StructuredEKF
    N = 18
    x, P, s = _state(N, 47)
    imm = Engine(capacity=40, tile=tile, batch=1)
    dfr = Engine(capacity=40, tile=tile, batch=3)
    ref = StructuredEKF(40, "known")
    for e in (imm, dfr, ref):
        e.set_state(x, P, s)
    rng = np.random.default_rng(9)
    fused_seen = split_seen = False
    for step in range(40):
        u = [0.1, 3.0]
        for e in (imm, dfr, ref):
            e.predict(u)
        idx0 = int(rng.integers(0, imm.N))
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        imm.correct(z, R, idx0); dfr.correct(z, R, idx0); ref.correct(z, R, idx0 + 1)
        name, _ = imm.downdate_kernel_name()
        fused_seen |= "fused" in name
        split_seen |= "fused" not in name
        if step % 4 == 1 and imm.N < 28:
            pos, sig = rng.uniform(-5, 5, 2), imm.N + 1
            for e in (imm, dfr, ref):
                e.append(u, R, pos, sig)
    assert fused_seen and split_seen and imm.N == 28
    np.testing.assert_array_equal(imm.get_x(), dfr.get_x())
    np.testing.assert_array_equal(imm.get_P(), dfr.get_P())
    assert rel_err(imm.get_P(), ref.P) < REL and rel_err(imm.get_x(), ref.x) < REL


@pytest.mark.parametrize("storage,tile", [("f64", 64), ("f32", 128)])
def test_diag_blocks_are_live_without_a_pass(storage, tile):
    """What plot() reads (EKF_SLAM.m:180,205: P(1:2,1:2) and every landmark's 2x2 diagonal block) comes from the live copies every
    correction updates at once: ekf_get_P_diag_blocks answers with pending pairs still pending (no pass over P), and -- with F64
    tiles -- returns exactly what the flushed matrix holds."""
    from ekf_slam_amd import Engine
    N = 90
    x, P, s = _state(N, 21)
    e = Engine(capacity=N + 2, tile=tile, storage=storage, batch=16)
    e.set_state(x, P, s)
    rng = np.random.default_rng(8)
    for step in range(11):
        e.predict([0.1, 2.0])
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        e.correct(z, np.diag([z[0] * .01, z[1] * 5.0]), int(rng.integers(0, N)))
        if step == 5:
            e.append([0.1, 2.0], np.diag([.2, 40.0]), [3.0, -4.0], N + 1)
    assert e.pending() == 11
    blocks = e.get_P_diag_blocks()
    assert e.pending() == 11                                   # ... and they still are
    Pf = e.get_P()                                             # this one flushes
    assert e.pending() == 0
    want = [Pf[0:2, 0:2]] + [Pf[3 + 2 * k:5 + 2 * k, 3 + 2 * k:5 + 2 * k] for k in range(e.N)]
    for got, w in zip(blocks, want):
        np.testing.assert_array_equal(got, w)
    if storage == "f64":                                       # F64 tiles: the live copies ARE the tile entries the pass writes, bit for bit
        ref = Engine(capacity=N + 2, tile=tile, storage=storage, batch=1)
        ref.set_state(x, P, s)
        rng = np.random.default_rng(8)
        for step in range(11):
            ref.predict([0.1, 2.0])
            z = [rng.uniform(1, 30), rng.uniform(1, 359)]
            ref.correct(z, np.diag([z[0] * .01, z[1] * 5.0]), int(rng.integers(0, N)))
            if step == 5:
                ref.append([0.1, 2.0], np.diag([.2, 40.0]), [3.0, -4.0], N + 1)
        np.testing.assert_array_equal(ref.get_P(), Pf)


@pytest.mark.parametrize("batch,tile,world", [(4, 16, 1), (8, 16, 1), (16, 32, 1), (3, 16, 1), (8, 16, 3)])
def test_async_pass_with_an_append_every_step_equals_immediate_bitwise(batch, tile, world, oracle_lib):
    """configs[4]'s step (predict + append + correct, EKF_SLAM.m:67-98 then :124-145) on the asynchronous engine: landmarks appended WHILE a
    pass is in flight go to the store the pass reads and are copied to the store it writes when it retires (k_copy_tile_rows) -- the append
    does not wait for the pass.  Rows cross several tile edges per batch (tile 16: 8 landmarks per tile row); F64 tiles: bit-identical."""
    from ekf_slam_amd import Engine
    from ekf_slam_amd.sharding import ShardGroup
    from oracle.ekf_structured import StructuredEKF
    N, steps = 21, 70
    x, P, s = _state(N, 43)
    imm = Engine(capacity=N + steps, tile=tile, batch=1)
    asy = (Engine(capacity=N + steps, tile=tile, batch=batch, async_flush=True) if world == 1 else
           ShardGroup(world, capacity=N + steps, tile=tile, batch=batch, async_flush=True))
    ref = StructuredEKF(N + steps, "known")
    for e in (imm, asy, ref):
        e.set_state(x, P, s)
    rng = np.random.default_rng(9)
    for step in range(steps):
        u = [0.1, 3.0]
        pos, sig = rng.uniform(-5, 5, 2), imm.N + 1
        Ra = np.diag([0.2, 40.0])
        idx0 = int(rng.integers(0, imm.N + 1))                          # now and then the landmark appended in this very step
        z = [rng.uniform(1, 30), rng.uniform(1, 359)]
        R = np.diag([z[0] * .01, z[1] * 5.0])
        for e in (imm, asy, ref):
            e.predict(u)
            e.append(u, Ra, pos, sig)
            e.correct(z, R, idx0 + (1 if e is ref else 0))
        if step % 9 == 4:
            np.testing.assert_array_equal(asy.get_x(), imm.get_x())
        if step == 40:                                                   # a read in the middle: retires the pass in flight, copies the rows
            np.testing.assert_array_equal(asy.get_P(), imm.get_P())
    assert asy.N == imm.N == N + steps
    Pi = imm.get_P()
    np.testing.assert_array_equal(asy.get_x(), imm.get_x())
    np.testing.assert_array_equal(asy.get_P(), Pi)
    assert rel_err(Pi, ref.P) < REL and rel_err(imm.get_x(), ref.x) < REL
