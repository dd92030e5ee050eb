"""CPU: hand-derived cases for the re-statement of the reference's landmark-list bookkeeping (RANSAC.m:234-373).
The reference holds no fixtures for it and cannot run here: parity unpinned, pinned by these cases only."""
import numpy as np

from ekf_slam_amd.ransac_bookkeeping import RansacBookkeeping

POSE = np.array([0.0, 0.0, 0.0])


def test_first_detection_seeds_only_the_first_potential_landmark():
    r = RansacBookkeeping()
    obs = r.getLandmark([[1.0, 0.0], [5.0, 5.0]], POSE)                  # RANSAC.m:236-241
    assert len(obs) == 0 and len(r.landmark) == 1
    e = r.landmark[0]
    assert (e.observe, e.index) == (1, 0) and e.fresh == 49              # seeded with 50, decremented the same call (:316-320)
    np.testing.assert_array_equal(e.loc, [1.0, 0.0])


def test_index_after_consensus_then_single_observed_row():
    r = RansacBookkeeping()
    for k in range(10):
        assert len(r.getLandmark([[1.0, 0.0]], POSE)) == 0               # observe 1..10: not yet > 10
    assert r.landmark[0].observe == 10 and r.landmark[0].index == 0
    obs = r.getLandmark([[1.2, 0.1]], POSE)                              # 11th sighting: observe 11 > 10 -> index 1 (:261-264)
    assert r.landmark[0].index == 1
    np.testing.assert_array_equal(r.landmark[0].loc, [1.2, 0.1])         # loc replaced by the detection (:268)
    assert obs.shape == (1, 3) and obs[0, 2] == 1.0
    np.testing.assert_allclose(obs[0, 0], np.hypot(1.2, 0.1))
    np.testing.assert_allclose(obs[0, 1], np.degrees(np.arctan2(0.1, 1.2)))
    # two indexed landmarks re-observed in one call: still ONE row (the `elseif ~find(...)` of :283 never fires)
    for k in range(12):
        r.getLandmark([[1.2, 0.1], [4.0, 4.0]], POSE)
    assert sorted(e.index for e in r.landmark) == [1, 2]
    assert r.getLandmark([[1.2, 0.1], [4.0, 4.0]], POSE).shape == (1, 3)


def test_no_break_a_detection_increments_every_entry_in_range():
    r = RansacBookkeeping()
    r.getLandmark([[0.0, 0.0]], POSE)
    r.getLandmark([[0.8, 0.0]], POSE)                                    # 0.8 away: new entry
    assert len(r.landmark) == 2
    r.getLandmark([[0.4, 0.0]], POSE)                                    # within 0.5 of BOTH: `jj = size(...)` is not a break
    assert [e.observe for e in r.landmark] == [2, 2]


def test_unindexed_entries_expire_after_freshness_timer():
    r = RansacBookkeeping()
    r.getLandmark([[1.0, 0.0]], POSE)                                    # entry A: fresh 50 -> 49 in the same call
    for k in range(5):
        r.getLandmark(None, POSE)                                        # scans without walls do no bookkeeping (:143-145)
    assert len(r.landmark) == 1 and r.landmark[0].fresh == 49
    for k in range(48):
        r.getLandmark([[9.0, 9.0]], POSE)                                # every call with walls costs A one `fresh`
    assert [tuple(e.loc) for e in r.landmark][0] == (1.0, 0.0) and r.landmark[0].fresh == 1
    r.getLandmark([[9.0, 9.0]], POSE)                                    # 49th: fresh 0 -> A is deleted (:321-324)
    assert [tuple(e.loc) for e in r.landmark] == [(9.0, 9.0)]
    assert r.landmark[0].index == 1                                      # B passed the consensus long ago and never expires


def test_update_landmark_list_touches_only_the_last_state_landmark():
    r = RansacBookkeeping()
    for k in range(12):
        r.getLandmark([[1.0, 0.0], [4.0, 4.0]] if k else [[1.0, 0.0]], POSE)
    for k in range(12):
        r.getLandmark([[1.0, 0.0], [4.0, 4.0]], POSE)
    by_index = {e.index: e for e in r.landmark}
    assert set(by_index) == {1, 2}
    x = np.array([0.0, 0.0, 0.0, 1.5, 0.5, 4.5, 3.5])                    # the filter moved both landmarks
    r.updateLandmarkList(x)                                              # `for ii = N` -> only landmark 2 (RANSAC.m:354)
    np.testing.assert_array_equal(by_index[2].loc, [4.5, 3.5])
    np.testing.assert_array_equal(by_index[1].loc, [1.0, 0.0])


def test_table_shape_matches_what_measure_takes():
    r = RansacBookkeeping()
    for k in range(12):
        r.getLandmark([[1.0, 0.0]], POSE)
    idx, loc = r.table()
    assert idx.tolist() == [1.0] and loc.shape == (1, 2)


def test_bookkeeping_plays_the_hand_scripted_scenario():
    """The scenario of tests/ransac_script.py (expected rows / tables derived on paper from RANSAC.m:234-334), with the poses
    of the literal-dense oracle: the product's bookkeeping must reproduce it step by step."""
    import ransac_script as S
    from oracle import ekf_dense as D
    ref, src, rb = D.EKF_SLAM(), S.ScriptedSource(), RansacBookkeeping()
    for t, (u, pts) in enumerate(S.feed()):
        ref.predict(u)
        got = rb.getLandmark(pts, ref.x)
        ref.measure(pts, u, src)                       # the oracle advances on the SCRIPTED source, not on `rb`
        want = src.rows[-1]
        assert got.shape == want.shape, t
        if len(want):
            assert got[0, 2] == S.expected_row_index(t)
            np.testing.assert_allclose(got, want, rtol=1e-12)
        assert [(e.index, tuple(e.loc)) for e in rb.landmark] == [(i, tuple(loc)) for i, loc in S.expected_table(t)], t
    assert (len(ref.x) - 3) // 2 == 2
