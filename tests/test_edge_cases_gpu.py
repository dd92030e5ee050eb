"""GPU: edge cases of the reference's path, each against the literal-dense oracle on the same inputs (1e-6) -- the cases a reader
of EKF_SLAM.m / Correspondence.m would try to break it with: a bearing of exactly 0 (R(2,2) = 0, EKF_SLAM.m:108), an empty scan,
the same landmark twice in one scan, a full (non-diagonal) R that sends the 2x2 inverse through its pivoting branch
(EKF_SLAM.m:143), a landmark exactly at the robot (q = 0: the reference divides by zero -- non-finite state, not a crash), the
heading wrapped at exactly 360 (wrapTo360(360) = 360), capacity reached exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-6


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _pair(N, seed, tile=16, batch=1, mode="known"):
    from ekf_slam_amd import Engine
    from oracle import ekf_dense as D
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.4, -0.3, 25.0], rng.uniform(-8, 8, 2 * N)])
    U = rng.normal(0, 0.05, (n, 5))
    P = np.diag(rng.uniform(0.01, 0.1, n)) + U @ U.T
    s = list(range(1, N + 1))
    e = Engine(mode=mode, capacity=N + 4, tile=tile, batch=batch)
    e.set_state(x, P, np.array(s, dtype=float))
    d = (D.EKF_SLAM if mode == "known" else D.EKF_SLAM_UC)()
    d.x, d.P, d.s = x.copy(), P.copy(), list(s)
    return e, d


class _Table:
    """Landmark.m-shaped source that hands the oracle the same observed rows / table the GPU gets."""
    def __init__(self, rows, index, loc):
        class _E:
            def __init__(self, i, l): self.index, self.loc = i, np.asarray(l, dtype=float)
        class _O: pass
        self.rows = np.asarray(rows, dtype=float).reshape(-1, 3)
        self.landmarkObj = _O()
        self.landmarkObj.landmark = [_E(i, l) for i, l in zip(index, loc)]

    def getLandmark(self, laser, x):
        return self.rows


@pytest.mark.parametrize("batch", [1, 4])
def test_zero_bearing_gives_singular_R_and_still_matches(batch):
    e, d = _pair(6, 1, batch=batch)
    rows = [[5.0, 0.0, 1.0], [3.0, 0.0, 2.0], [7.5, 120.0, 3.0]]          # R = diag(r * .01, 0) for the first two rows
    idx, loc = np.arange(1, 7.0), np.zeros((6, 2))
    u = [0.1, 3.0]
    e.predict(u); d.predict(u)
    e.measure(rows, u, idx, loc); d.measure(None, u, _Table(rows, idx, loc))
    assert np.isfinite(d.x).all()
    assert rel_err(e.get_x(), d.x) < REL and rel_err(e.get_P(), d.P) < REL


def test_empty_scan_is_a_no_op_and_duplicates_are_applied_in_order():
    e, d = _pair(5, 2, batch=3)
    u = [0.2, -4.0]
    e.predict(u); d.predict(u)
    x0 = e.get_x()
    e.measure(np.zeros((0, 3)), u, np.arange(1, 6.0), np.zeros((5, 2)))
    np.testing.assert_array_equal(e.get_x(), x0)
    # known correspondence corrects landmark ii = the ROW NUMBER (EKF_SLAM.m:123): rows 1..3 with repeated third columns
    rows = [[4.0, 10.0, 2.0], [4.1, 11.0, 2.0], [6.0, 200.0, 1.0]]
    idx, loc = np.arange(1, 6.0), np.zeros((5, 2))
    e.measure(rows, u, idx, loc); d.measure(None, u, _Table(rows, idx, loc))
    assert rel_err(e.get_x(), d.x) < REL and rel_err(e.get_P(), d.P) < REL


def test_full_R_takes_the_pivoting_branch_of_the_2x2_inverse():
    from oracle.matlab_compat import inv2
    e, d = _pair(4, 3)
    R = np.array([[1e-4, 9.0], [9.0, 2e-4]])                              # |phi(2,1)| > |phi(1,1)|: rows swap in the LU
    z = [6.0, 75.0]
    e.predict([0.1, 2.0]); d.predict([0.1, 2.0])
    from oracle.ekf_dense import _innovation_terms
    _, H = _innovation_terms(d.x, 2)
    phi = H @ d.P @ H.T + R
    assert abs(phi[1, 0]) > abs(phi[0, 0])                                # the case is what it claims to be
    np.testing.assert_allclose(inv2(phi), np.linalg.inv(phi), rtol=1e-12)
    e.correct(z, R, 1); d._correct(z, R, 2)
    assert rel_err(e.get_x(), d.x) < REL and rel_err(e.get_P(), d.P) < REL


def test_landmark_at_the_robot_divides_by_zero_like_the_reference(oracle_lib):
    """q = 0 (EKF_SLAM.m:127,137): MATLAB's 1/q_k is Inf and the update turns x and P into NaN / Inf without an error.  Checked
    against the structured C restatement (IEEE division; the NumPy one raises on a Python float division)."""
    from oracle.ekf_structured import StructuredEKF
    e, d = _pair(3, 4)
    x = d.x.copy(); x[5:7] = x[0:2]                                       # landmark 2 exactly at the robot
    ref = StructuredEKF(8, "known")
    e.set_state(x, d.P, np.arange(1, 4.0)); ref.set_state(x, d.P, np.arange(1, 4.0))
    z, R = [1.0, 10.0], np.diag([0.01, 50.0])
    e.correct(z, R, 1); ref.correct(z, R, 2)
    xe, xr = e.get_x(), ref.x
    assert not np.isfinite(xr).any() and not np.isfinite(xe).any()        # every entry non-finite on both sides, no crash
    assert not np.isfinite(e.get_P()).any() and not np.isfinite(ref.P).any()


def test_heading_landing_exactly_on_360_stays_360():
    e, d = _pair(2, 5)
    for eng in (e, d):
        xs = (eng.get_x() if hasattr(eng, "get_x") else eng.x).copy()
    x = d.x.copy(); x[2] = 350.0
    e.set_state(x, d.P, np.arange(1, 3.0)); d.x = x.copy()
    e.predict([0.5, 10.0]); d.predict([0.5, 10.0])
    assert d.x[2] == 360.0 and e.get_x()[2] == 360.0
    z, R = [5.0, 33.0], np.diag([0.05, 165.0])
    e.correct(z, R, 0); d._correct(z, R, 1)
    assert rel_err(e.get_x(), d.x) < REL and rel_err(e.get_P(), d.P) < REL


def test_capacity_reached_exactly_then_refused():
    from ekf_slam_amd import Engine, EkfError, _lib as L
    from oracle import ekf_dense as D
    e, d = Engine(capacity=3, tile=16), D.EKF_SLAM()
    R = np.diag([0.02, 10.0])
    for k in range(3):
        e.append([0.1, 5.0], R, [1.0 + k, 2.0 - k], k + 1); d.append([0.1, 5.0], R, [1.0 + k, 2.0 - k], k + 1)
    assert e.N == 3 and rel_err(e.get_P(), d.P) < REL
    with pytest.raises(EkfError) as ei:
        e.append([0.1, 5.0], R, [9.0, 9.0], 4)
    assert ei.value.status == L.EKF_ERR_CAPACITY and e.N == 3
    e.correct([2.0, 40.0], R, 2); d._correct([2.0, 40.0], R, 3)            # still usable at capacity
    assert rel_err(e.get_x(), d.x) < REL and rel_err(e.get_P(), d.P) < REL


@pytest.mark.parametrize("device_assoc", [2, 3])
def test_signature_index_of_the_host_mirror_agrees_with_the_device_on_ties_and_threshold_edges(device_assoc):
    """From 256 landmarks on, the host's prediction of an observation's association (the mirror of s) evaluates only the landmarks
    whose signature lies within the threshold of z(3), found in a sorted index; the device evaluates all N.  Duplicate signatures
    (the lowest index must win, Correspondence.m:81), signatures exactly on / just beyond the threshold, observations matching
    nothing (appended, which also extends the index): every device decision is checked against the host's (modes 2 and 3 verify)."""
    from ekf_slam_amd import Engine
    rng = np.random.default_rng(123)
    N = 400
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-30, 30, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 4))
    P = np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T
    s = rng.integers(1, 60, size=N).astype(float) + rng.choice([0.0, 0.05, 0.1, -0.05], size=N)       # many duplicates and near-duplicates
    e = Engine(mode="uc", capacity=N + 40, tile=64, batch=4, device_assoc=device_assoc, s_cost=1.0, s_thresh=0.01)   # |d| <= 0.1 passes
    ref = Engine(mode="uc", capacity=N + 40, tile=64, batch=4, device_assoc=1, s_cost=1.0, s_thresh=0.01)           # waited: the device decides
    for g in (e, ref):
        g.set_state(x, P, s)
    u = [0.1, 2.0]
    appended = 0
    for scan in range(12):
        rows = []
        for _ in range(6):
            base = float(rng.integers(0, 62))
            z3 = base + float(rng.choice([0.0, 0.05, 0.1, 0.1000001, -0.1, -0.0999999, 0.15, 0.5]))
            rows.append([float(rng.uniform(2, 30)), float(rng.uniform(1, 359)), z3])
        Ncur = e.N
        idx = np.arange(Ncur + 1, Ncur + 8, dtype=float)                      # table entries for every index an append may ask for
        loc = rng.uniform(-30, 30, (7, 2))
        for g in (e, ref):
            g.predict(u)
            g.measure(rows, u, idx, loc)
        e.sync()                                                              # mode 3 reports a mismatch here at the latest
        appended += e.N - Ncur
        assert e.N == ref.N
    assert appended > 0 and e.N > N
    np.testing.assert_array_equal(e.get_s(), ref.get_s())
    np.testing.assert_array_equal(e.get_x(), ref.get_x())
    np.testing.assert_array_equal(e.get_P(), ref.get_P())


@pytest.mark.parametrize("kind", ["correction_contradicted", "append_contradicted"])
def test_device_loop_mismatch_is_reported_once_at_the_next_synchronising_call(kind):
    """include/ekfslam.h, cfg.device_assoc = 3: the host queues a scan's launches from its mirror's prediction of every association and
    waits for nothing; what the device decided is compared afterwards.  The two cannot differ unless the device's signatures changed behind
    the library's back -- which ekf_diag_poke_device_signature does on purpose.  What the caller sees: ekf_measure returns EKF_OK; the
    first synchronising call returns EKF_ERR_STATE exactly once, naming both decisions; every launch stayed inside the state (finite, N as the
    host predicted); the handle stays usable and a state reload makes it the reference's again (bit-identical to a fresh engine)."""
    from ekf_slam_amd import Engine, _lib as L
    from ekf_slam_amd._lib import EkfError
    rng = np.random.default_rng(9)
    N = 40
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-30, 30, size=2 * N)])
    U = rng.normal(0, 0.05, size=(n, 4))
    P = np.diag(rng.uniform(0.01, 0.1, size=n)) + U @ U.T
    s = np.arange(1, N + 1.0)
    e = Engine(mode="uc", capacity=N + 8, tile=16, batch=4, device_assoc=3)
    e.set_state(x, P, s)
    u = [0.1, 2.0]
    idx, loc = np.array([N + 1.0]), np.array([[5.0, 6.0]])
    if kind == "correction_contradicted":
        # the host's mirror matches signature 7 to landmark 7; on the device that landmark's signature is now far away: nothing passes
        assert e.lib.ekf_diag_poke_device_signature(e.h, 6, 1000.0) == L.EKF_OK
        rows = [[12.0, 30.0, 7.0]]
    else:
        # the host's mirror finds no landmark for signature 500 and queues an append; on the device landmark 3 carries that signature
        assert e.lib.ekf_diag_poke_device_signature(e.h, 2, 500.0) == L.EKF_OK
        rows = [[12.0, 30.0, 500.0]]
    assert e.lib.ekf_diag_poke_device_signature(e.h, N, 1.0) == L.EKF_ERR_INVALID_ARG      # no such landmark: refused, nothing written
    e.predict(u)
    e.measure(rows, u, idx, loc)                                              # EKF_OK: nothing is waited for
    with pytest.raises(EkfError) as ei:
        e.sync()
    assert ei.value.status == L.EKF_ERR_STATE and "device association decided" in str(ei.value) and "predicted" in str(ei.value)
    e.sync()                                                                  # reported once
    assert e.N == (N if kind == "correction_contradicted" else N + 1)         # what the host queued ran, inside the state
    assert np.isfinite(e.get_x()).all() and np.isfinite(e.get_P()).all()
    # reload: the reference's state again
    fresh = Engine(mode="uc", capacity=N + 8, tile=16, batch=4, device_assoc=3)
    for g in (e, fresh):
        g.set_state(x, P, s)
        g.predict(u)
        g.measure([[12.0, 30.0, 7.0], [3.0, 200.0, 99.0]], u, idx, loc)
        g.sync()
    assert e.N == fresh.N == N + 1
    np.testing.assert_array_equal(e.get_x(), fresh.get_x())
    np.testing.assert_array_equal(e.get_P(), fresh.get_P())
