"""CPU tests of the oracle (test infrastructure): hand-derived known-answer tests from the reference source
(SURVEY.md section 4, KAT-1..4, and KAT-5..7 for the correction body, append and association -- derivations in
tests/kat_cases.py; the reference itself holds no golden vectors: PARITY UNPINNED), KAT-8..12 at non-zero headings (predict,
correction, append, two corrections in a row, the UC new-landmark dispatch), and literal-dense == structured
agreement.  The GPU twins of KAT-5..12 are in tests/test_kat_gpu.py."""
import numpy as np
import pytest

from ekf_slam_amd.world import SyntheticLandmark, make_run
from oracle import ekf_dense as D
from oracle.ekf_structured import StructuredEKF
from oracle.matlab_compat import atan2d, cosd, inv2, sind, wrapTo360

import kat_cases as K


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def test_kat1_predict_from_ctor_state_dense():
    # EKF_SLAM.m:28-31,42-50,58-64 by hand: W=[1;0;0], Q_rr=diag(.2,0,0), F(1,3)=0, F(2,3)=1
    e = D.EKF_SLAM()
    e.predict([1, 0])
    np.testing.assert_allclose(e.x, [1, 0, 0], atol=0)
    np.testing.assert_allclose(e.P, [[0.3, 0, 0], [0, 0.2, 0.1], [0, 0.1, 0.1]], atol=1e-16)


def test_kat1_structured(oracle_lib):
    e = StructuredEKF(4, "known")
    e.predict([1, 0])
    np.testing.assert_allclose(e.x, [1, 0, 0], atol=0)
    np.testing.assert_allclose(e.P, [[0.3, 0, 0], [0, 0.2, 0.1], [0, 0.1, 0.1]], atol=1e-16)
    np.testing.assert_allclose(e.Q, np.diag([0.2, 0, 0]), atol=1e-16)


def test_kat2_degree_builtins(oracle_lib):
    assert wrapTo360(360.0) == 360.0 and wrapTo360(720.0) == 360.0 and wrapTo360(0.0) == 0.0
    assert wrapTo360(-90.0) == 270.0 and wrapTo360(450.0) == 90.0 and wrapTo360(-360.0) == 0.0
    for a, s, c in [(0, 0, 1), (90, 1, 0), (180, 0, -1), (270, -1, 0), (360, 0, 1), (-90, -1, 0), (450, 1, 0)]:
        assert sind(a) == s and cosd(a) == c
        assert oracle_lib.oekf_sind(a) == s and oracle_lib.oekf_cosd(a) == c
    for a in np.linspace(-720, 720, 97):
        assert abs(sind(a) - np.sin(np.deg2rad(a))) < 1e-14
        assert abs(cosd(a) - np.cos(np.deg2rad(a))) < 1e-14
        assert sind(a) == oracle_lib.oekf_sind(a) and cosd(a) == oracle_lib.oekf_cosd(a)
        assert wrapTo360(a) == oracle_lib.oekf_wrapTo360(a)
    assert atan2d(1, 0) == 90.0 and atan2d(0, -1) == 180.0
    A = np.array([[2.0, 1.0], [0.5, 3.0]])
    np.testing.assert_allclose(inv2(A), np.linalg.inv(A), rtol=1e-15)
    B = np.array([[0.1, 2.0], [4.0, 1.0]])      # pivoting branch
    np.testing.assert_allclose(inv2(B), np.linalg.inv(B), rtol=1e-15)


def test_kat3_first_observation_only_appends():
    # EKF_SLAM.m:110-111: with length(x) < 4 the row appends (signature 1) and performs no correction
    e = D.EKF_SLAM()
    lm = SyntheticLandmark()
    e.predict([0.1, 3.0])
    x_before, P_before = e.x.copy(), e.P.copy()
    e.measure([(0, 2.0, 30.0)], [0.1, 3.0], lm)
    assert len(e.x) == 5 and e.s == [1]
    np.testing.assert_array_equal(e.x[:3], x_before)
    np.testing.assert_array_equal(e.P[:3, :3], P_before)


def test_kat3_malformed_first_row_raises():
    e = D.EKF_SLAM()
    lm = SyntheticLandmark()
    with pytest.raises(D.LandmarkLookupError):
        e.measure([(0, 2.0, 30.0), (1, 3.0, 40.0)], [0.1, 3.0], lm)   # two indexed landmarks: find([index]) has 2 hits


def test_kat4_association_is_signature_only():
    # Correspondence.m:71-85: (false,k) iff some s(k) == z(3) (first such k), else (true, N+1)
    rng = np.random.default_rng(0)
    N = 6
    x = np.concatenate([[0, 0, 10.0], rng.uniform(-5, 5, 2 * N)])
    A = rng.normal(size=(3 + 2 * N, 3 + 2 * N))
    P = A @ A.T + np.eye(3 + 2 * N)
    s = [1, 2, 3, 4, 3, 6]
    c = D.Correspondence(1e-11, 1e9, 'EKF_SLAM_UC')
    R = np.diag([0.1, 5.0])
    assert c.estimateCorrespondence([3.0, 20.0, 3], R, x, P, s) == (False, 3)
    assert c.estimateCorrespondence([3.0, 20.0, 6], R, x, P, s) == (False, 6)
    assert c.estimateCorrespondence([3.0, 20.0, 7], R, x, P, s) == (True, N + 1)
    assert c.estimateCorrespondence([3.0, 20.0, 3.05], R, x, P, s) == (False, 3)      # |d| <= 0.1 passes the threshold
    assert c.estimateCorrespondence([3.0, 20.0, 3.2], R, x, P, s) == (True, N + 1)
    with pytest.warns(UserWarning):
        assert D.Correspondence(1e-11, 1e9, 'other').method == 'ML'                    # Correspondence.m:19-23


def test_append_free_function_guard():
    # append.m:4: only appends when numOfLandmarks < idx
    x = np.array([0.0, 0, 0, 1, 1])
    P = np.eye(5)
    x2, P2 = D.append(x, P, [0.1, 1], 1, np.eye(2), [2, 2])
    assert len(x2) == 5 and P2.shape == (5, 5)
    x3, P3 = D.append(x, P, [0.1, 1], 2, np.eye(2), [2, 2])
    assert len(x3) == 7 and P3.shape == (7, 7)
    np.testing.assert_allclose(P3, P3.T, atol=1e-15)


def test_kat5_correction_body_by_hand(oracle_lib):
    """EKF_SLAM.m:124-145 on the paper case of tests/kat_cases.py (KAT-5): exact rationals, both restatements."""
    d = D.EKF_SLAM()
    d.x, d.P, d.s = K.K5_X.copy(), K.K5_P.copy(), [1]
    d._correct(K.K5_Z, K.K5_R, 1)
    st = StructuredEKF(4, "known")
    st.set_state(K.K5_X, K.K5_P, [1.0])
    st.correct(K.K5_Z, K.K5_R, 1)
    for x, P in ((d.x, d.P), (st.x, st.P)):
        np.testing.assert_allclose(x, K.K5_X_OUT, rtol=0, atol=2e-16)
        np.testing.assert_allclose(P, K.K5_P_OUT, rtol=0, atol=2e-16)


def test_kat6_append_twice_by_hand(oracle_lib):
    """EKF_SLAM.m:67-98 (KAT-6): two appends from the KAT-1 state, incl. the old-landmark loop :94-97."""
    d = D.EKF_SLAM()
    st = StructuredEKF(4, "known")
    for e in (d, st):
        e.predict([1, 0])
        for a in K.K6_APPENDS:
            e.append(a["u"], a["R"], a["pos"], a["sig"])
        np.testing.assert_allclose(e.x, K.K6_X_OUT, rtol=0, atol=0)
        np.testing.assert_allclose(e.P, K.K6_P_OUT, rtol=0, atol=4e-15)        # 17 = 1.0 + 16 carries one rounding of .1 sums
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), K.K6_S_OUT)
    # free function append.m:1-27 gives the same first append and refuses the guarded one
    x1, P1 = D.append([1.0, 0, 0], np.array([[.3, 0, 0], [0, .2, .1], [0, .1, .1]]), K.K6_APPENDS[0]["u"], 1,
                      K.K6_APPENDS[0]["R"], K.K6_APPENDS[0]["pos"])
    np.testing.assert_allclose(P1, K.K6_P_OUT[:5, :5], rtol=0, atol=4e-15)
    x2, _ = D.append(x1, P1, [2, 180], 1, np.eye(2), [9, 9])                    # numOfLandmarks (1) < idx (1) is false
    assert len(x2) == 5


def test_kat7_association_costs_by_hand(oracle_lib):
    """Correspondence.m:49-87 (KAT-7): Mahalanobis cost per landmark, live (signature-only) and commented-out (position +
    signature) likelihoods, threshold and tie rules."""
    c = D.Correspondence(1.0, 1e9, 'EKF_SLAM_UC')
    for z, pc in ((K.K7_ZA, K.K7_PC_A), (K.K7_ZB, K.K7_PC_B)):
        assert c.estimateCorrespondence(z, K.K7_R, K.K7_X, K.K7_P, K.K7_S) == (False, 1)      # :75 live line: tie -> first
        np.testing.assert_allclose(c.last_position_cost, pc, rtol=1e-14)
        np.testing.assert_array_equal(c.last_signature_cost, [0.0, 0.0])
    for w_pos, thresh, z, want in [(0.0, 1e9, K.K7_ZA, (False, 1)), (0.0, 1e9, K.K7_ZB, (False, 1)),
                                   (1.0, 1e9, K.K7_ZA, (False, 1)), (1.0, 1e9, K.K7_ZB, (False, 2)),
                                   (1.0, 100.0, K.K7_ZA, (False, 1)), (1.0, 2.0, K.K7_ZA, (True, 3)),
                                   (1.0, 1e9, [4.2, 85.0, 6.0], (False, 2))]:
        st = StructuredEKF(4, "uc", s_cost=1.0, s_thresh=thresh, w_pos=w_pos)
        st.set_state(K.K7_X, K.K7_P, K.K7_S)
        new, idx, pc, sc = st.associate(z, K.K7_R, want_costs=True)
        assert (new, idx) == want, (w_pos, thresh, z)
        np.testing.assert_allclose(pc, K.K7_PC_A if z[0] == 2.5 else K.K7_PC_B, rtol=1e-14)
        np.testing.assert_array_equal(sc, [(z[2] - 5.0) ** 2] * 2)


def _both(mode="known", capacity=8):
    """The two restatements behind one surface: (name, engine, correct(z, R, idx1))."""
    d = (D.EKF_SLAM if mode == "known" else D.EKF_SLAM_UC)()
    st = StructuredEKF(capacity, mode)
    return [("dense", d, d._correct), ("structured", st, st.correct)]


def _load(e, x, P, s):
    if isinstance(e, StructuredEKF):
        e.set_state(x, P, s)
    else:
        e.x, e.P, e.s = np.array(x, dtype=float), np.array(P, dtype=float), list(s)


def test_kat8_predict_at_heading_90_by_hand(oracle_lib):
    """EKF_SLAM.m:40-51, :56-65 (KAT-8): W and F(1:2,3) with the PRE-motion heading 90, pose with 90 + u2, strip rows move."""
    for name, e, _ in _both():
        _load(e, K.K8_X, K.K8_P, [1.0])
        e.predict(K.K8_U)
        np.testing.assert_array_equal(e.x, K.K8_X_OUT, err_msg=name)
        np.testing.assert_allclose(e.P, K.K8_P_OUT, rtol=0, atol=3e-13, err_msg=name)      # 1620.3: one ulp is 2.3e-13
        np.testing.assert_allclose(np.asarray(e.Q)[:3, :3], K.K8_Q_OUT, rtol=0, atol=3e-13, err_msg=name)
    x_new, F = D.f(K.K8_X, K.K8_U)                                                           # the public f(x,u), :56-65
    np.testing.assert_array_equal(x_new, [-1, 2, 180, 3, 4])
    Fw = np.eye(5); Fw[0, 2] = -2.0; Fw[1, 2] = 0.0
    np.testing.assert_array_equal(F, Fw)


@pytest.mark.parametrize("case", ["a", "b"])
def test_kat9_correction_with_a_heading_by_hand(case, oracle_lib):
    """EKF_SLAM.m:124-145 (KAT-9): z_k = wrapTo360(atan2d(dy,dx) - x(3)) at headings 90 / 270, innovation NOT wrapped."""
    x0, xo, Po = (K.K9A_X, K.K9A_X_OUT, K.K9A_P_OUT) if case == "a" else (K.K9B_X, K.K9B_X_OUT, K.K9B_P_OUT)
    for name, e, correct in _both():
        _load(e, x0, K.K9_P, [1.0])
        correct(K.K9_Z, K.K9_R, 1)
        np.testing.assert_allclose(e.x, xo, rtol=0, atol=6e-14, err_msg=name)               # 270 - 4/201: one ulp is 5.7e-14
        np.testing.assert_allclose(e.P, Po, rtol=0, atol=2e-16, err_msg=name)


def test_kat10_append_at_heading_90_by_hand(oracle_lib):
    """EKF_SLAM.m:67-98 (KAT-10): jxr with the state's heading 90, jz with u2, the old-landmark loop on a non-diagonal P."""
    a = K.K10_APPEND
    for name, e, _ in _both():
        _load(e, K.K8_X, K.K8_P, [4.0])
        e.append(a["u"], a["R"], a["pos"], a["sig"])
        np.testing.assert_array_equal(e.x, K.K10_X_OUT, err_msg=name)
        np.testing.assert_allclose(e.P, K.K10_P_OUT, rtol=0, atol=4e-15, err_msg=name)
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), [4.0, 9.0])
    x1, P1 = D.append(K.K8_X, K.K8_P, a["u"], 2, a["R"], a["pos"])                          # append.m:1-27, 1 < 2
    np.testing.assert_allclose(P1, K.K10_P_OUT, rtol=0, atol=4e-15)


def test_kat11_two_corrections_in_a_row_by_hand(oracle_lib):
    """EKF_SLAM.m:107,124-145 (KAT-11): the second correction is linearised at the first one's x+ and P+."""
    for name, e, correct in _both():
        _load(e, K.K11_X, K.K11_P, [1.0, 2.0])
        correct(K.K11_Z1, K.K11_R1, 1)
        np.testing.assert_allclose(e.x, K.K11_X1, rtol=0, atol=5e-16, err_msg=name)
        np.testing.assert_allclose(e.P, K.K11_P1, rtol=0, atol=1e-16, err_msg=name)
        correct(K.K11_Z2, K.K11_R2, 2)
        np.testing.assert_allclose(e.x, K.K11_X2, rtol=0, atol=2e-14, err_msg=name)         # 102.2: one ulp is 1.4e-14
        np.testing.assert_allclose(e.P, K.K11_P2, rtol=0, atol=2e-16, err_msg=name)


def test_kat12_uc_measure_with_a_new_landmark_by_hand(oracle_lib):
    """EKF_SLAM_UC.m:102-152 (KAT-12): correction, then a row no signature matches -> append with signature N+1 and the loc of
    the table entry whose index is N+1, then a correction on the grown state."""
    for name, e, _ in _both("uc", 8):
        _load(e, K.K12_X, K.K12_P, K.K12_S)
        e.measure(None, K.K12_U, K.KatTable(K.K12_TABLE, K.K12_OBSERVED[:2]))
        np.testing.assert_allclose(e.x, K.K12_X_AFTER2, rtol=0, atol=5e-16, err_msg=name)
        np.testing.assert_allclose(e.P, K.K12_P_AFTER2, rtol=0, atol=2e-13, err_msg=name)   # 600.25: one ulp is 1.1e-13
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), K.K12_S_OUT)             # signature 3, not the observed 7
    for name, e, _ in _both("uc", 8):
        _load(e, K.K12_X, K.K12_P, K.K12_S)
        e.measure(None, K.K12_U, K.KatTable(K.K12_TABLE, K.K12_OBSERVED))
        np.testing.assert_allclose(e.x, K.K12_X_OUT, rtol=0, atol=2e-14, err_msg=name)
        np.testing.assert_allclose(e.P, K.K12_P_OUT, rtol=0, atol=2e-13, err_msg=name)
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), K.K12_S_OUT)
    # the table entry with index N+1 missing -> the reference's comma-separated-list expansion fails (EKF_SLAM_UC.m:123)
    d = D.EKF_SLAM_UC()
    _load(d, K.K12_X, K.K12_P, K.K12_S)
    with pytest.raises(D.LandmarkLookupError):
        d.measure(None, K.K12_U, K.KatTable(K.K12_TABLE[:3], K.K12_OBSERVED))


def test_kat13_association_costs_at_heading_90_by_hand(oracle_lib):
    """Correspondence.m:49-87 (KAT-13): KAT-7's scene turned by 90 degrees about the robot -- same ranges, same relative bearings, so
    the same Mahalanobis costs and decisions, provided z_k = wrapTo360(atan2d(dy,dx) - x(3)) is evaluated as written (:56)."""
    c = D.Correspondence(1.0, 1e9, 'EKF_SLAM_UC')
    for z, pc in ((K.K7_ZA, K.K7_PC_A), (K.K7_ZB, K.K7_PC_B)):
        assert c.estimateCorrespondence(z, K.K7_R, K.K13_X, K.K7_P, K.K7_S) == (False, 1)
        np.testing.assert_allclose(c.last_position_cost, pc, rtol=1e-14)
    for w_pos, thresh, z, want in [(1.0, 1e9, K.K7_ZA, (False, 1)), (1.0, 1e9, K.K7_ZB, (False, 2)), (1.0, 2.0, K.K7_ZA, (True, 3))]:
        st = StructuredEKF(4, "uc", s_cost=1.0, s_thresh=thresh, w_pos=w_pos)
        st.set_state(K.K13_X, K.K7_P, K.K7_S)
        new, idx, pc, sc = st.associate(z, K.K7_R, want_costs=True)
        assert (new, idx) == want, (w_pos, thresh, z)
        np.testing.assert_allclose(pc, K.K7_PC_A if z[0] == 2.5 else K.K7_PC_B, rtol=1e-14)


def test_kat14_known_correspondence_dispatch_quirks_by_hand(oracle_lib):
    """EKF_SLAM.m:116-123 (KAT-14): idx = ii corrects the landmark of the ROW NUMBER whatever z(3) says; z(3) > N appends with
    signature z(3) and the loc of the table entry whose index is z(3)."""
    for name, e, _ in _both("known", 8):
        _load(e, K.K14A_X, K.K14A_P, [1.0, 2.0])
        e.measure(None, [0.1, 0.0], K.KatTable(K.K12_TABLE, K.K14A_OBSERVED))
        np.testing.assert_allclose(e.x, K.K14A_X_OUT, rtol=0, atol=2e-16, err_msg=name)
        np.testing.assert_allclose(e.P, K.K14A_P_OUT, rtol=0, atol=2e-16, err_msg=name)
    for name, e, _ in _both("known", 8):
        _load(e, K.K14B_X, K.K14B_P, [1.0, 2.0])
        e.measure(None, K.K12_U, K.KatTable(K.K14B_TABLE, K.K14B_OBSERVED))
        np.testing.assert_allclose(e.x, K.K14B_X_OUT, rtol=0, atol=5e-16, err_msg=name)
        np.testing.assert_allclose(e.P, K.K14B_P_OUT, rtol=0, atol=2e-13, err_msg=name)
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), K.K14B_S_OUT)


@pytest.mark.parametrize("mode", ["known", "uc"])
def test_dense_equals_structured_20_landmarks(mode, oracle_lib):
    _, run = make_run(20, 20260101, 60, policy="all")
    d = (D.EKF_SLAM if mode == "known" else D.EKF_SLAM_UC)()
    st = StructuredEKF(32, mode)
    la, lb = SyntheticLandmark(), SyntheticLandmark()
    for u, scan in run:
        d.predict(u); st.predict(u)
        d.measure(scan, u, la); st.measure(scan, u, lb)
        assert rel_err(st.x, d.x) < 1e-11 and rel_err(st.P, d.P) < 1e-11
    assert st.N == 20 and list(st.s) == list(d.s)
    # invariants: symmetric to rounding, PSD
    assert np.abs(d.P - d.P.T).max() < 1e-12
    assert np.linalg.eigvalsh((d.P + d.P.T) / 2).min() > -1e-10


def test_dense_equals_structured_200_landmarks_uc(oracle_lib):
    _, run = make_run(200, 20260102, 6, policy="nearest", m=8)
    d, st = D.EKF_SLAM_UC(), StructuredEKF(256, "uc")
    la, lb = SyntheticLandmark(), SyntheticLandmark()
    for u, scan in run:
        d.predict(u); st.predict(u)
        d.measure(scan, u, la); st.measure(scan, u, lb)
    assert st.N == 200
    assert rel_err(st.x, d.x) < 1e-11 and rel_err(st.P, d.P) < 1e-11


def test_trace_non_increasing_across_correction(oracle_lib):
    _, run = make_run(20, 20260101, 10, policy="all")
    st = StructuredEKF(32, "known")
    lb = SyntheticLandmark()
    for u, scan in run:
        st.predict(u); st.measure(scan, u, lb)
    z = [3.0, 45.0]
    before = np.trace(st.P)
    st.correct(z, np.diag([0.03, 225.0]), 4)
    assert np.trace(st.P) <= before + 1e-12


def test_kat15_append_then_correct_the_appended_landmark_by_hand(oracle_lib):
    """EKF_SLAM.m:67-98 then :124-145 (KAT-15): the correction reads the cross-covariances the append wrote; the robot columns of
    G = H P cancel only if P(1:3,new) = Prr jxr' carries the signs the reference gives it."""
    a = K.K15_APPEND
    for name, e, correct in _both():
        _load(e, K.K15_X, K.K15_P, [])
        e.append(a["u"], a["R"], a["pos"], a["sig"])
        np.testing.assert_array_equal(e.x, K.K15_X_A, err_msg=name)
        np.testing.assert_allclose(e.P, K.K15_P_A, rtol=0, atol=2e-16, err_msg=name)
        correct(K.K15_Z, K.K15_R, 1)
        np.testing.assert_allclose(e.x, K.K15_X_OUT, rtol=0, atol=2e-15, err_msg=name)
        np.testing.assert_allclose(e.P, K.K15_P_OUT, rtol=0, atol=2e-16, err_msg=name)
        np.testing.assert_array_equal(np.asarray(e.s, dtype=float), [5.0])


def test_kat16_correction_with_a_full_phi_by_hand(oracle_lib):
    """EKF_SLAM.m:124-145 (KAT-16): phi_k is a full 2 x 2 matrix (phi12 = .25), its inverse has off-diagonals [10 -5; -5 4.5] and is taken
    on the pivoting branch (|phi21| > |phi11|); K, x+ and P+ worked out as exact rationals in tests/kat_cases.py -- all three restatements."""
    from oracle.ekf_factored import FactoredEKF
    from oracle.matlab_compat import inv2
    assert abs(K.K16_PHI[1, 0]) > abs(K.K16_PHI[0, 0])
    np.testing.assert_allclose(inv2(K.K16_PHI), [[10, -5], [-5, 4.5]], rtol=0, atol=2e-14)
    fac = FactoredEKF(4, "known")
    for name, e, correct in _both() + [("factored", fac, fac.correct)]:
        if name == "factored":
            e.set_state(K.K16_X, K.K16_P, [1.0])
        else:
            _load(e, K.K16_X, K.K16_P, [1.0])
        correct(K.K16_Z, K.K16_R, 1)
        np.testing.assert_allclose(e.x, K.K16_X_OUT, rtol=0, atol=1e-15, err_msg=name)
        np.testing.assert_allclose(e.P, K.K16_P_OUT, rtol=0, atol=2e-16, err_msg=name)


def test_kat17_association_sees_the_robot_landmark_cross_covariance_by_hand(oracle_lib):
    """Correspondence.m:66-69 (KAT-17): with a non-zero strip P(1:3, 4:7) the cross terms of H_k P H_k' change the position costs and the
    decision of the commented-out likelihood (:74, w_pos = 1) flips from landmark 1 (KAT-7's diagonal P) to landmark 2."""
    from oracle.ekf_factored import FactoredEKF
    c = D.Correspondence(1.0, 1e9, 'EKF_SLAM_UC')
    for P, pc in ((K.K17_P_DIAG, K.K17_PC_DIAG), (K.K17_P, K.K17_PC)):
        assert c.estimateCorrespondence(K.K17_Z, K.K17_R, K.K17_X, P, K.K17_S) == (False, 1)       # the live line :75: signature only
        np.testing.assert_allclose(c.last_position_cost, pc, rtol=1e-14)
    assert K.K17_PC_DIAG[0] < K.K17_PC_DIAG[1] and K.K17_PC[0] > K.K17_PC[1]
    for P, pc, want in ((K.K17_P_DIAG, K.K17_PC_DIAG, (False, 1)), (K.K17_P, K.K17_PC, (False, 2))):
        st = StructuredEKF(4, "uc", s_cost=1.0, s_thresh=1e9, w_pos=1.0)
        st.set_state(K.K17_X, P, K.K17_S)
        new, idx, pcs, sc = st.associate(K.K17_Z, K.K17_R, want_costs=True)
        assert (new, idx) == want
        np.testing.assert_allclose(pcs, pc, rtol=1e-14)
        fac = FactoredEKF(4, "uc")
        fac.set_state(K.K17_X, P, K.K17_S)
        fac.s_cost, fac.w_pos = 1.0, 1.0
        assert fac.estimateCorrespondence(K.K17_Z, K.K17_R) == want
        np.testing.assert_allclose(fac.last_position_cost, pc, rtol=1e-14)
