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


@pytest.mark.parametrize("device_assoc", [3, 0])
def test_reassigned_association_properties_reach_the_engine(device_assoc):
    """The reference's tunables are public, assignable properties (EKF_SLAM.m:14-16; EKF_SLAM_UC.m:16 `correspondence`): a run that
    reassigns them MID-RUN -- a new Correspondence object with a signature threshold so tight that a re-observed landmark's noisy
    signature no longer matches, later edited in place back to the default -- against oracle.ekf_dense.EKF_SLAM_UC given the same
    reassignments.  With the threshold ignored (the round-2 mirror never forwarded it) the tight phase would append nothing."""
    from ekf_slam_amd import slam
    from ekf_slam_amd.world import SyntheticLandmark, make_run
    from oracle import ekf_dense as D
    _, run = make_run(12, 77, 12, policy="all")
    gpu = slam.EKF_SLAM_UC(capacity=40, tile=16, batch=4, device_assoc=device_assoc)
    ref = D.EKF_SLAM_UC()
    lg, lr = slam.Landmark('SYNTHETIC'), SyntheticLandmark()
    for t, (u, scan) in enumerate(run):
        if t == 4:
            # signatures are the integers 1..12; shift what the "sensor" reports by 0.05: |d| = 0.05 passes the default test
            # (d^2 / 1e-11 = 2.5e8 <= 1e9) and fails this one (2.5e8 > 1e8) -> every row becomes a new landmark
            gpu.correspondence = slam.Correspondence(1e-11, 1e8, 'EKF_SLAM_UC')
            ref.correspondence = D.Correspondence(1e-11, 1e8, 'EKF_SLAM_UC')
        if t == 6:
            gpu.correspondence.s_thresh = 1e9                    # edited in place, like a MATLAB property of a value object
            ref.correspondence.s_thresh = 1e9
        scan_t = [(wid, r, b) for (wid, r, b) in scan]
        gpu.predict(u); ref.predict(u)
        if 4 <= t < 6:
            # the tight phase: the table must offer entries N+1, N+2, ... for the appends (EKF_SLAM_UC.m:123) -- use a source
            # whose signatures are off by 0.05 and whose table grows with the state
            obs_g = _shifted(lg, scan_t, gpu.x, 0.05, gpu._e.N)
            obs_r = _shifted(lr, scan_t, ref.x, 0.05, len(ref.s))
            gpu.measure(None, u, obs_g); ref.measure(None, u, obs_r)
        else:
            gpu.measure(scan_t, u, lg); ref.measure(scan_t, u, lr)
        assert gpu._e.N == len(ref.s), t
    assert gpu._e.N > 12                                         # the tight phase did append
    np.testing.assert_array_equal(gpu.s, np.asarray(ref.s, dtype=float))
    assert rel_err(gpu.x, ref.x) < REL and rel_err(gpu.P, ref.P) < REL
    # the known-correspondence class carries s_cost / s_thresh as plain assignable properties (EKF_SLAM.m:14-16)
    k = slam.EKF_SLAM(capacity=4, tile=16)
    k.s_cost, k.s_thresh = 2.5, 7.0
    assert (k.s_cost, k.s_thresh) == (2.5, 7.0) and (k._e.cfg.s_cost, k._e.cfg.s_thresh) == (2.5, 7.0)


class _Shifted:
    """A landmark_list whose observed signatures are off by `shift` and whose table holds one entry per possible new index."""

    def __init__(self, rows, table):
        self._rows = np.asarray(rows, dtype=float).reshape(-1, 3)

        class _Obj:
            pass
        self.landmarkObj = _Obj()

        class _E:
            def __init__(self, index, loc):
                self.index, self.loc = index, np.asarray(loc, dtype=float)
        self.landmarkObj.landmark = [_E(i, l) for i, l in table]
        self.landmarkObj.table = lambda: (np.array([e.index for e in self.landmarkObj.landmark], dtype=float),
                                          np.array([e.loc for e in self.landmarkObj.landmark], dtype=float).reshape(-1, 2))

    def getLandmark(self, laserdata, x):
        return self._rows


def _shifted(src, scan, x, shift, N):
    rows = np.asarray(src.getLandmark(scan, x), dtype=float).reshape(-1, 3).copy()
    rows[:, 2] += shift
    table = [(N + 1 + q, (100.0 + q, -50.0 - q)) for q in range(len(rows))]      # entries for the indices the appends will ask for
    return _Shifted(rows, table)


def test_runslam_end_to_end_with_ransac_bookkeeping():
    """SLAM('EKF_SLAM') with Landmark('RANSAC') on the GPU -- the product's landmark-list bookkeeping fed with wall
    foot-points -- against the literal-dense oracle driven by a HAND-SCRIPTED trace of what that bookkeeping must output
    (tests/ransac_script.py: derived from RANSAC.m:234-334 on paper, no product class on the oracle side).  Checked at every
    step: the observed row and the landmark table of the product equal the script's; at the end: x and P."""
    import ransac_script as S
    from ekf_slam_amd import slam
    from oracle import ekf_dense as D

    feed = S.feed()
    s = slam.SLAM('EKF_SLAM', feed=feed, capacity=8, tile=16, landmark_method='RANSAC')
    ref, src = D.EKF_SLAM(), S.ScriptedSource()
    for t, (u, pts) in enumerate(feed):
        s.runSlam()
        ref.predict(u); ref.measure(pts, u, src)
        got = s.slam.observed
        want = src.rows[-1]
        assert got.shape == want.shape, t
        if len(want):
            assert got[0, 2] == want[0, 2] == S.expected_row_index(t)
            np.testing.assert_allclose(got[0, :2], want[0, :2], rtol=1e-9)
        assert [(e.index, tuple(e.loc)) for e in s.LM.landmarkObj.landmark] == \
               [(i, tuple(loc)) for i, loc in S.expected_table(t)], t
    assert s.slam._e.N == (len(ref.x) - 3) // 2 == 2
    assert rel_err(s.slam.x, ref.x) < REL and rel_err(s.slam.P, ref.P) < REL
