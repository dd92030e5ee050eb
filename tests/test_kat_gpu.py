"""GPU twins of the hand-derived known-answer tests KAT-5..12 (derivations: tests/kat_cases.py; CPU side:
tests/test_oracle_kat.py): the HIP path, through the C ABI, against values worked out on paper from EKF_SLAM.m:40-65,
:67-98, :124-145, EKF_SLAM_UC.m:102-152 and Correspondence.m:49-87 -- the one pin that does not pass through the builder's
own restatements.  KAT-8..12 sit at headings 90 / 180 / 270."""
import numpy as np
import pytest

import kat_cases as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile,batch", [(16, 1), (128, 1), (16, 4), (0, 8)])
def test_kat5_correction_body_on_the_gpu(tile, batch):
    from ekf_slam_amd import Engine
    e = Engine(capacity=4, tile=tile, batch=batch)
    e.set_state(K.K5_X, K.K5_P, [1.0])
    e.correct(K.K5_Z, K.K5_R, 0)
    np.testing.assert_allclose(e.get_x(), K.K5_X_OUT, rtol=0, atol=2e-16)
    np.testing.assert_allclose(e.get_P(), K.K5_P_OUT, rtol=0, atol=2e-16)       # get_P applies the pending pair first
    e.close()


@pytest.mark.parametrize("tile", [16, 128])
def test_kat6_append_twice_on_the_gpu(tile):
    from ekf_slam_amd import Engine
    e = Engine(capacity=4, tile=tile)
    e.predict([1, 0])
    for a in K.K6_APPENDS:
        e.append(a["u"], a["R"], a["pos"], a["sig"])
    np.testing.assert_array_equal(e.get_x(), K.K6_X_OUT)
    np.testing.assert_allclose(e.get_P(), K.K6_P_OUT, rtol=0, atol=4e-15)
    np.testing.assert_array_equal(e.get_s(), K.K6_S_OUT)
    e.close()


def test_kat6_free_function_append_on_the_gpu():
    from ekf_slam_amd.slam import append
    P0 = np.array([[.3, 0, 0], [0, .2, .1], [0, .1, .1]])
    a = K.K6_APPENDS[0]
    x1, P1 = append([1.0, 0, 0], P0, a["u"], 1, a["R"], a["pos"])
    np.testing.assert_array_equal(x1, K.K6_X_OUT[:5])
    np.testing.assert_allclose(P1, K.K6_P_OUT[:5, :5], rtol=0, atol=4e-15)
    x2, _ = append(x1, P1, [2, 180], 1, np.eye(2), [9, 9])           # append.m:4 guard: 1 < 1 is false
    assert len(x2) == 5


@pytest.mark.parametrize("batch", [1, 4])
def test_kat7_association_on_the_gpu(batch):
    from ekf_slam_amd import Engine
    for w_pos, thresh, z, want in [(0.0, 1e9, K.K7_ZA, (False, 1)), (0.0, 1e9, K.K7_ZB, (False, 1)),
                                   (1.0, 1e9, K.K7_ZA, (False, 1)), (1.0, 1e9, K.K7_ZB, (False, 2)),
                                   (1.0, 100.0, K.K7_ZA, (False, 1)), (1.0, 2.0, K.K7_ZA, (True, 3)),
                                   (1.0, 1e9, [4.2, 85.0, 6.0], (False, 2))]:
        e = Engine(mode="uc", capacity=4, tile=16, batch=batch, s_cost=1.0, s_thresh=thresh, w_pos=w_pos)
        e.set_state(K.K7_X, K.K7_P, K.K7_S)
        new, idx0, pc, sc = e.associate(z, K.K7_R, want_costs=True)
        assert (new, idx0 + 1) == want, (w_pos, thresh, z)
        np.testing.assert_allclose(pc, K.K7_PC_A if z[0] == 2.5 else K.K7_PC_B, rtol=1e-14)
        np.testing.assert_array_equal(sc, [(z[2] - 5.0) ** 2] * 2)
        e.close()


def test_kat7_correspondence_class_forwards_cost_and_threshold():
    """The reference-named class (Correspondence.m:12-25,28) with its own cost / threshold, live likelihood."""
    from ekf_slam_amd.slam import Correspondence
    c = Correspondence(1.0, 1e9, 'EKF_SLAM_UC')
    assert c.estimateCorrespondence(K.K7_ZA, K.K7_R, K.K7_X, K.K7_P, K.K7_S) == (False, 1)
    np.testing.assert_allclose(c.position_cost, K.K7_PC_A, rtol=1e-14)
    assert Correspondence(1.0, 0.5, 'EKF_SLAM_UC').estimateCorrespondence([2.5, 10.0, 6.0], K.K7_R, K.K7_X, K.K7_P, K.K7_S) \
        == (True, 3)                                   # signature cost (6-5)^2/1 = 1 > 0.5 for both -> new landmark N+1
    with pytest.warns(UserWarning):
        assert Correspondence(1.0, 1e9, 'other').method == 'ML'


# ---- KAT-8..12: non-zero headings, two corrections in a row, the UC new-landmark dispatch ----
_SHAPES = [(16, 1), (128, 1), (16, 4), (0, 8)]          # (tile, batch): small and production tiles, immediate and deferred


@pytest.mark.parametrize("tile,batch", _SHAPES)
@pytest.mark.parametrize("lazy", [False, True])
def test_kat8_predict_at_heading_90_on_the_gpu(tile, batch, lazy):
    """lazy: the predict stays recorded and is folded into the next correction's gather kernel (checked through KAT-9's
    arithmetic being untouched: x and P are read right after, which launches the standalone predict) -- both forms."""
    from ekf_slam_amd import Engine
    e = Engine(capacity=4, tile=tile, batch=batch)
    e.set_state(K.K8_X, K.K8_P, [1.0])
    e.predict(K.K8_U)
    if not lazy:
        np.testing.assert_array_equal(e.get_x(), K.K8_X_OUT)
        np.testing.assert_allclose(e.get_P(), K.K8_P_OUT, rtol=0, atol=3e-13)
        np.testing.assert_allclose(e.get_Q3(), K.K8_Q_OUT, rtol=0, atol=3e-13)
    else:
        # fold the predict into a correction's gather, then undo nothing: compare with predict-then-correct of the dense oracle's
        # arithmetic by running the same correction on an engine loaded with the hand-derived post-predict state
        z, R = [2.0, 10.0], np.diag([.02, 50.0])
        e.correct(z, R, 0)
        ref = Engine(capacity=4, tile=tile, batch=batch)
        ref.set_state(K.K8_X_OUT, K.K8_P_OUT, [1.0])
        ref.correct(z, R, 0)
        np.testing.assert_allclose(e.get_x(), ref.get_x(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(e.get_P(), ref.get_P(), rtol=0, atol=1e-12)
        ref.close()
    e.close()


def test_kat8_public_f_on_the_host_function():
    from ekf_slam_amd.slam import EKF_SLAM
    h = EKF_SLAM(capacity=4)
    x_new, F = h.f(K.K8_X, K.K8_U)                     # EKF_SLAM.m:56-65: F(1,3) = -2 sind(90), F(2,3) = 2 cosd(90)
    np.testing.assert_array_equal(x_new, [-1, 2, 180, 3, 4])
    Fw = np.eye(5); Fw[0, 2] = -2.0
    np.testing.assert_array_equal(F, Fw)


@pytest.mark.parametrize("tile,batch", _SHAPES)
@pytest.mark.parametrize("case", ["a", "b"])
def test_kat9_correction_with_a_heading_on_the_gpu(case, tile, batch):
    from ekf_slam_amd import Engine
    x0, xo, Po = (K.K9A_X, K.K9A_X_OUT, K.K9A_P_OUT) if case == "a" else (K.K9B_X, K.K9B_X_OUT, K.K9B_P_OUT)
    e = Engine(capacity=4, tile=tile, batch=batch)
    e.set_state(x0, K.K9_P, [1.0])
    e.correct(K.K9_Z, K.K9_R, 0)
    np.testing.assert_allclose(e.get_x(), xo, rtol=0, atol=6e-14)
    np.testing.assert_allclose(e.get_P(), Po, rtol=0, atol=2e-16)
    e.close()


@pytest.mark.parametrize("tile", [16, 128])
def test_kat10_append_at_heading_90_on_the_gpu(tile):
    from ekf_slam_amd import Engine
    a = K.K10_APPEND
    e = Engine(capacity=4, tile=tile)
    e.set_state(K.K8_X, K.K8_P, [4.0])
    e.append(a["u"], a["R"], a["pos"], a["sig"])
    np.testing.assert_array_equal(e.get_x(), K.K10_X_OUT)
    np.testing.assert_allclose(e.get_P(), K.K10_P_OUT, rtol=0, atol=4e-15)
    np.testing.assert_array_equal(e.get_s(), [4.0, 9.0])
    e.close()


@pytest.mark.parametrize("tile,batch", _SHAPES)
def test_kat11_two_corrections_in_a_row_on_the_gpu(tile, batch):
    """With batch > 1 the second correction reads landmark 2's rows through the first one's PENDING pair."""
    from ekf_slam_amd import Engine
    e = Engine(capacity=4, tile=tile, batch=batch)
    e.set_state(K.K11_X, K.K11_P, [1.0, 2.0])
    e.correct(K.K11_Z1, K.K11_R1, 0)
    if batch == 1:
        np.testing.assert_allclose(e.get_x(), K.K11_X1, rtol=0, atol=5e-16)
        np.testing.assert_allclose(e.get_P(), K.K11_P1, rtol=0, atol=1e-16)
    e.correct(K.K11_Z2, K.K11_R2, 1)
    np.testing.assert_allclose(e.get_x(), K.K11_X2, rtol=0, atol=2e-14)
    np.testing.assert_allclose(e.get_P(), K.K11_P2, rtol=0, atol=2e-16)
    e.close()


@pytest.mark.parametrize("tile,batch", _SHAPES)
@pytest.mark.parametrize("device_assoc", [0, 1, 2, 3])
def test_kat12_uc_measure_with_a_new_landmark_on_the_gpu(device_assoc, tile, batch):
    """EKF_SLAM_UC.measure through the reference-named class: row 1 corrects landmark 1, row 2 matches no signature and is
    appended with signature N+1 = 3 and the loc of the table entry whose index is 3, row 3 corrects landmark 2 on n = 9."""
    from ekf_slam_amd.slam import EKF_SLAM_UC
    for rows, xo, Po, tol in ((2, K.K12_X_AFTER2, K.K12_P_AFTER2, 5e-16), (3, K.K12_X_OUT, K.K12_P_OUT, 2e-14)):
        h = EKF_SLAM_UC(capacity=4, tile=tile, batch=batch, device_assoc=device_assoc)
        h.x, h.s, h.P = K.K12_X, K.K12_S, K.K12_P
        h.measure(None, K.K12_U, K.KatTable(K.K12_TABLE, K.K12_OBSERVED[:rows]))
        np.testing.assert_allclose(h.x, xo, rtol=0, atol=tol)
        np.testing.assert_allclose(h.P, Po, rtol=0, atol=2e-13)
        np.testing.assert_array_equal(h.s, K.K12_S_OUT)
    # the table entry of index N+1 missing: EKF_ERR_LOOKUP, as the reference's failed comma-list expansion (EKF_SLAM_UC.m:123)
    from ekf_slam_amd._lib import EkfError, EKF_ERR_LOOKUP
    h = EKF_SLAM_UC(capacity=4, tile=tile, batch=batch, device_assoc=device_assoc)
    h.x, h.s, h.P = K.K12_X, K.K12_S, K.K12_P
    with pytest.raises(EkfError) as ei:
        h.measure(None, K.K12_U, K.KatTable(K.K12_TABLE[:3], K.K12_OBSERVED))
    assert ei.value.status == EKF_ERR_LOOKUP


@pytest.mark.parametrize("batch", [1, 4])
def test_kat13_association_costs_at_heading_90_on_the_gpu(batch):
    """KAT-7's scene turned by 90 degrees (tests/kat_cases.py): k_associate's z_k = wrapTo360(atan2d(dy,dx) - x(3)), Correspondence.m:56."""
    from ekf_slam_amd import Engine
    for w_pos, thresh, z, want in [(0.0, 1e9, K.K7_ZA, (False, 1)), (1.0, 1e9, K.K7_ZA, (False, 1)), (1.0, 1e9, K.K7_ZB, (False, 2)),
                                   (1.0, 2.0, K.K7_ZA, (True, 3))]:
        e = Engine(mode="uc", capacity=4, tile=16, batch=batch, s_cost=1.0, s_thresh=thresh, w_pos=w_pos)
        e.set_state(K.K13_X, K.K7_P, K.K7_S)
        new, idx0, pc, sc = e.associate(z, K.K7_R, want_costs=True)
        assert (new, idx0 + 1) == want, (w_pos, thresh, z)
        np.testing.assert_allclose(pc, K.K7_PC_A if z[0] == 2.5 else K.K7_PC_B, rtol=1e-14)
        e.close()


@pytest.mark.parametrize("tile,batch", _SHAPES)
def test_kat14_known_correspondence_dispatch_quirks_on_the_gpu(tile, batch):
    """EKF_SLAM.measure through the reference-named class: idx = ii (EKF_SLAM.m:123) and the z(3) > N append (:118-120)."""
    from ekf_slam_amd.slam import EKF_SLAM
    h = EKF_SLAM(capacity=4, tile=tile, batch=batch)
    h.x, h.s, h.P = K.K14A_X, [1.0, 2.0], K.K14A_P
    h.measure(None, [0.1, 0.0], K.KatTable(K.K12_TABLE, K.K14A_OBSERVED))
    np.testing.assert_allclose(h.x, K.K14A_X_OUT, rtol=0, atol=2e-16)
    np.testing.assert_allclose(h.P, K.K14A_P_OUT, rtol=0, atol=2e-16)
    h = EKF_SLAM(capacity=4, tile=tile, batch=batch)
    h.x, h.s, h.P = K.K14B_X, [1.0, 2.0], K.K14B_P
    h.measure(None, K.K12_U, K.KatTable(K.K14B_TABLE, K.K14B_OBSERVED))
    np.testing.assert_allclose(h.x, K.K14B_X_OUT, rtol=0, atol=5e-16)
    np.testing.assert_allclose(h.P, K.K14B_P_OUT, rtol=0, atol=2e-13)
    np.testing.assert_array_equal(h.s, K.K14B_S_OUT)


@pytest.mark.parametrize("tile,batch", _SHAPES)
def test_kat15_append_then_correct_the_appended_landmark_on_the_gpu(tile, batch):
    """KAT-15 (tests/kat_cases.py): append at heading 90 onto an empty map, then the correction of that landmark; with a predict of zero
    motion in front of the append the launch that carries the predict out is the append's own (k_append<.., kPredict>)."""
    from ekf_slam_amd import Engine
    a = K.K15_APPEND
    for zero_motion_first in (False, True):
        e = Engine(capacity=4, tile=tile, batch=batch)
        e.set_state(K.K15_X, K.K15_P, [])
        if zero_motion_first:
            e.set_params(C=0.0)                              # Q = (W C) W' = 0 and F = I for u = [0 0]: the predict changes nothing
            e.predict([0.0, 0.0])
        e.append(a["u"], a["R"], a["pos"], a["sig"])
        np.testing.assert_array_equal(e.get_x(), K.K15_X_A)
        np.testing.assert_allclose(e.get_P(), K.K15_P_A, rtol=0, atol=2e-16)
        e.correct(K.K15_Z, K.K15_R, 0)
        np.testing.assert_allclose(e.get_x(), K.K15_X_OUT, rtol=0, atol=2e-15)
        np.testing.assert_allclose(e.get_P(), K.K15_P_OUT, rtol=0, atol=2e-16)
        np.testing.assert_array_equal(e.get_s(), [5.0])
        e.close()


@pytest.mark.parametrize("tile,batch", _SHAPES)
def test_kat16_correction_with_a_full_phi_on_the_gpu(tile, batch):
    """KAT-16 (tests/kat_cases.py): phi_k full, its inverse taken on the pivoting branch (|phi21| > |phi11|): k_gather's solve against the
    paper values of K = P H' phi^-1, x+ and P+ (EKF_SLAM.m:141-145)."""
    from ekf_slam_amd import Engine
    e = Engine(capacity=4, tile=tile, batch=batch)
    e.set_state(K.K16_X, K.K16_P, [1.0])
    e.correct(K.K16_Z, K.K16_R, 0)
    np.testing.assert_allclose(e.get_x(), K.K16_X_OUT, rtol=0, atol=1e-15)
    np.testing.assert_allclose(e.get_P(), K.K16_P_OUT, rtol=0, atol=2e-16)
    e.close()


@pytest.mark.parametrize("batch", [1, 4])
def test_kat17_association_with_cross_covariance_on_the_gpu(batch):
    """KAT-17 (tests/kat_cases.py): the strip P(1:3, 4:7) enters k_associate's phi_k (Correspondence.m:66) and flips the w_pos = 1 decision
    relative to the diagonal P of KAT-7; the live, signature-only likelihood (:75) is unmoved."""
    from ekf_slam_amd import Engine
    for P, pc, w_pos, want in ((K.K17_P_DIAG, K.K17_PC_DIAG, 1.0, (False, 1)), (K.K17_P, K.K17_PC, 1.0, (False, 2)), (K.K17_P, K.K17_PC, 0.0, (False, 1))):
        e = Engine(mode="uc", capacity=4, tile=16, batch=batch, s_cost=1.0, s_thresh=1e9, w_pos=w_pos)
        e.set_state(K.K17_X, P, K.K17_S)
        new, idx0, pcs, sc = e.associate(K.K17_Z, K.K17_R, want_costs=True)
        assert (new, idx0 + 1) == want, (w_pos, want)
        np.testing.assert_allclose(pcs, pc, rtol=1e-14)
        e.close()


@pytest.mark.parametrize("device_assoc", [0, 1, 2, 3])
def test_kat17_in_the_measure_loop_of_every_association_mode(device_assoc):
    """the same scene through EKF_SLAM_UC.measure (EKF_SLAM_UC.m:107-151) in the four association modes: with the reference's live likelihood
    the row corrects landmark 1 whatever the strip holds -- all modes leave the same bits"""
    from ekf_slam_amd import Engine
    ref = None
    e = Engine(mode="uc", capacity=4, tile=16, batch=1, device_assoc=device_assoc)
    e.set_state(K.K17_X, K.K17_P, K.K17_S)
    e.measure([K.K17_Z], [0.0, 0.0], [1.0, 2.0, 3.0], [[2.0, 0.0], [0.0, 4.0], [9.0, 9.0]])
    x, P = e.get_x(), e.get_P()
    assert e.N == 2 and np.isfinite(x).all()
    host = Engine(mode="uc", capacity=4, tile=16, batch=1, device_assoc=0)
    host.set_state(K.K17_X, K.K17_P, K.K17_S)
    R = np.diag([K.K17_Z[0] * .1, K.K17_Z[1] * 5.0])                   # EKF_SLAM_UC.m:110, Rc = [.1 5]
    host.correct(K.K17_Z[:2], R, 0)
    np.testing.assert_array_equal(x, host.get_x())
    np.testing.assert_array_equal(P, host.get_P())
    e.close(); host.close()
