"""GPU: the HIP path (through the C ABI) replayed over the committed golden fixtures.  Tolerance 1e-6 relative
(BASELINE.json north_star), F64."""
import numpy as np
import pytest

from golden_util import load, rel_err, replay_append3, replay_slam

pytestmark = pytest.mark.gpu
REL = 1e-6


@pytest.mark.parametrize("name,mode,tile", [("slam20_known.npz", "known", 16), ("slam20_uc.npz", "uc", 16),
                                            ("slam20_known.npz", "known", 64), ("slam120_uc_nearest.npz", "uc", 32),
                                            ("slam120_uc_nearest.npz", "uc", 64)])
def test_hip_path_reproduces_golden_slam_runs(name, mode, tile):
    from ekf_slam_amd.slam import EKF_SLAM, EKF_SLAM_UC, Landmark
    g = load(name)
    e = (EKF_SLAM if mode == "known" else EKF_SLAM_UC)(capacity=128, tile=tile, Rc=list(g["Rc"]))
    poses = replay_slam(e, Landmark('SYNTHETIC'), g)
    assert e._e.N == int(g["counts"][-1])
    assert rel_err(poses, g["poses"]) < REL
    assert rel_err(e.x, g["x"]) < REL
    assert rel_err(e.P, g["P"]) < REL
    np.testing.assert_array_equal(e.s, g["s"])
    print("%s tile %d: x %.2e P %.2e" % (name, tile, rel_err(e.x, g["x"]), rel_err(e.P, g["P"])))


def test_hip_path_reproduces_append3():
    from ekf_slam_amd import Engine
    g = load("append3.npz")
    e = Engine(capacity=4, tile=16)
    worst = replay_append3(g, e.predict, e.append, lambda z, R, idx: e.correct(z, R, idx - 1),
                           lambda: (e.get_x(), e.get_P()))
    assert worst < REL
    print("append3 worst %.2e" % worst)


def test_class_surface_matches_reference_names():
    """The host mirror keeps the reference's public names (SLAM.m, EKF_SLAM.m, Correspondence.m, append.m)."""
    from ekf_slam_amd import slam
    from ekf_slam_amd.world import make_run
    _, run = make_run(5, 1, 6, policy="all")
    s = slam.SLAM('EKF_SLAM_UC', feed=run, capacity=8, tile=16)
    for _ in range(6):
        s.runSlam()
    assert s.slam._e.N == 5
    x = s.slam.x
    x_new, F = s.slam.f(x, [0.2, 10.0])
    assert F.shape == (13, 13) and F[0, 2] != 0.0
    assert s.slam.Q.shape == (13, 13) and np.count_nonzero(s.slam.Q[3:, :]) == 0
    pose, blocks = s.slam.plot_data()
    assert len(blocks) == 6 and blocks[1].shape == (2, 2)
    # Correspondence / append as free-standing calls on caller arrays
    c = slam.Correspondence(1e-11, 1e9, 'EKF_SLAM_UC')
    P = s.slam.P
    new, idx = c.estimateCorrespondence([2.0, 30.0, 3.0], np.diag([0.2, 150.0]), x, P, s.slam.s)
    assert (new, idx) == (False, 3)
    x2, P2 = slam.append(x, P, [0.1, 3.0], 6, np.diag([0.2, 150.0]), [1.0, 2.0])
    assert len(x2) == 15 and P2.shape == (15, 15)
    x3, _ = slam.append(x, P, [0.1, 3.0], 5, np.diag([0.2, 150.0]), [1.0, 2.0])
    assert len(x3) == 13


def test_runslam_end_to_end_with_ransac_bookkeeping():
    """SLAM('EKF_SLAM') with Landmark('RANSAC'): the reference's landmark-list bookkeeping (consensus counts, index
    assignment, single observed row) fed with synthetic wall foot-points, against the literal-dense oracle driven by
    its own copy of the same bookkeeping."""
    from ekf_slam_amd import slam
    from ekf_slam_amd.ransac_bookkeeping import RansacBookkeeping
    from oracle import ekf_dense as D

    rng = np.random.default_rng(8)
    walls = np.array([[2.0, 1.0], [-1.5, 2.5]])
    feed = []
    for t in range(45):
        u = [0.05 + 0.001 * t, 2.0]
        pts = walls[:1 if t < 20 else 2] + rng.normal(0, 0.01, (1 if t < 20 else 2, 2))
        feed.append((u, pts))
    s = slam.SLAM('EKF_SLAM', feed=feed, capacity=8, tile=16, landmark_method='RANSAC')

    class _LM:                       # Landmark.m shape around the oracle's own bookkeeping object
        def __init__(self):
            self.landmarkObj = RansacBookkeeping()

        def getLandmark(self, laser, x):
            return self.landmarkObj.getLandmark(laser, x)
    ref, lm = D.EKF_SLAM(), _LM()
    for u, pts in feed:
        s.runSlam()
        ref.predict(u); ref.measure(pts, u, lm)
    assert s.slam._e.N == (len(ref.x) - 3) // 2 >= 1
    assert rel_err(s.slam.x, ref.x) < REL and rel_err(s.slam.P, ref.P) < REL
    assert [e.index for e in s.LM.landmarkObj.landmark] == [e.index for e in lm.landmarkObj.landmark]
