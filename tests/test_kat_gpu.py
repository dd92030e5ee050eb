"""GPU twins of the hand-derived known-answer tests KAT-5..7 (derivations: tests/kat_cases.py; CPU side:
tests/test_oracle_kat.py): the HIP path, through the C ABI, against values worked out on paper from EKF_SLAM.m:67-98,
:124-145 and Correspondence.m:49-87 -- the one pin that does not pass through the builder's own restatements."""
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
