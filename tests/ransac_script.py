"""A hand-scripted landmark-front-end scenario: what the reference's bookkeeping (RANSAC.m:234-334) must output, step by
step, for a fixed sequence of noise-free wall foot-points -- derived by hand from the source, NOT by running
ekf_slam_amd.ransac_bookkeeping.  Used to drive the ORACLE side of the end-to-end tests independently of the product class.

Scenario (foot-points A = (2,1), B = (-1.5,2.5); both further than landmarkDistance = .5 apart):
  t = 0..19   potentials [A]        t = 20..T-1  potentials [B, A]   (B listed first)
Derivation, with landmarkCountConsensus = 10, freshnessTimer = 50:
  t = 0      the list is empty: seeded with A, observe 1, index 0 (RANSAC.m:236-241); no row.
  t = 1..9   A is re-detected: observe 2..10; `observe > 10` is false (:261); no row.
  t = 10     observe 11 > 10 and index 0 -> index = max(index)+1 = 1 (:262); indexed -> loc = A and the row
             [dist, ang, 1] from the pose handed in (:268-280).  From now on a row with index 1 every step.
  t = 20     B matches nothing: appended, observe 1, index 0, fresh 50 -> 49 the same call (:292-298, :316-320);
             A (second potential) matches entry 1 -> row index 1.
  t = 21..29 B: observe 2..10 (fresh 48..40: no expiry); rows stay index 1.
  t = 30     B: observe 11 -> index 2, loc = B, row [dist, ang, 2] is the FIRST re-observed row; A matches too but a second
             row is never added (:279-284).  From now on every row carries index 2.
  Filter side (EKF_SLAM.m:107-123, known correspondence): t = 10 first row on an empty map -> append (signature 1, loc of
  the one indexed entry); t = 11..29 z(3) = 1 <= N -> correct landmark ii = 1; t = 30 z(3) = 2 > N = 1 -> append loc(index==2)
  = B, signature 2; t >= 31 z(3) = 2 <= N = 2 -> corrects landmark ii = 1 (the row number, :123) with the measurement taken
  to B -- the reference's quirk D5, reproduced.
  `updateLandmarkList` (:336-373, called first) overwrites the loc of the entry whose index equals N with the filter's last
  landmark; the matching detection then puts the foot-point back (:268) as long as the two are within .5 -- true over this
  horizon (asserted by the tests through the table they compare).
"""
import numpy as np

from oracle.matlab_compat import atan2d, wrapTo360

A = np.array([2.0, 1.0])
B = np.array([-1.5, 2.5])
T = 32


def feed():
    """[(u, foot-points)] -- odometry is deterministic too."""
    return [([0.05 + 0.001 * t, 2.0], np.array([A]) if t < 20 else np.array([B, A])) for t in range(T)]


def expected_row_index(t):
    return None if t < 10 else (1 if t < 30 else 2)


def expected_table(t):
    """(index, loc) of the struct array after step t's bookkeeping, storage order."""
    tab = [(0 if t < 10 else 1, A)]
    if t >= 20:
        tab.append((0 if t < 30 else 2, B))
    return tab


class ScriptedSource:
    """Landmark.m-shaped source that plays the hand-derived trace: row from the pose handed in (RANSAC.m:275-277)."""

    class _Entry:
        def __init__(self, index, loc):
            self.index, self.loc = index, np.array(loc, dtype=np.float64)

    class _Obj:
        landmark = []

    def __init__(self):
        self.landmarkObj = self._Obj()
        self.t = 0
        self.rows = []

    def getLandmark(self, laserdata, x):
        t = self.t
        self.t += 1
        self.landmarkObj.landmark = [self._Entry(i, loc) for i, loc in expected_table(t)]
        k = expected_row_index(t)
        if k is None:
            row = np.zeros((0, 3))
        else:
            loc = A if k == 1 else B
            dist = np.sqrt((x[0] - loc[0]) ** 2 + (x[1] - loc[1]) ** 2)
            ang = wrapTo360(atan2d(loc[1] - x[1], loc[0] - x[0]) - x[2])
            row = np.array([[dist, ang, float(k)]])
        self.rows.append(row)
        return row
