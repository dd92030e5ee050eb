"""Landmark front-end step before the EKF path: a deterministic, toolbox-free host
re-statement of the *bookkeeping* half of the reference's RANSAC class, fed with wall foot-points instead of laser scans.

The reference's `RANSAC.getLandmark` (RANSAC.m:14-152) does two things: (1) extract wall lines from a ROS LaserScan
(`findPoints`, `findPotentialLine`, `getOrthogPoints`, RANSAC.m:154-232 -- needs `datasample`'s RNG, `polyfit` and the
Symbolic Toolbox; out of scope) and (2) maintain the landmark list those foot-points feed
(`getOutputLandmarkListAndObservedLandmarkList` RANSAC.m:234-334, `updateLandmarkList` RANSAC.m:336-373).  This
module restates (2) so that `SLAM.runSlam` can run end to end from recorded or synthetic foot-points, with the same
struct fields the EKF classes read (`.landmark(k).loc / .observe / .index / .fresh`, RANSAC.m:238-241) and the same
quirks:

  * the first detection ever seeds the list with the FIRST potential landmark only (RANSAC.m:236-241);
  * `jj = size(...)` inside the `for jj` loop (RANSAC.m:286) does not break a MATLAB for loop: every list entry within
    `landmarkDistance` of a potential landmark is incremented;
  * an entry gets its index (max index + 1) once `observe > landmarkCountConsensus` (RANSAC.m:261-264) and from then on
    its `loc` is overwritten by each matching detection (RANSAC.m:268-269);
  * the observed list gets its first row from the first re-observed indexed entry; the `elseif ~find(...)` of
    RANSAC.m:283 never fires (`~[]` is `[]`, and `~k` is false), so at most ONE row is ever returned;
  * un-indexed entries lose one `fresh` per call and are deleted at zero (RANSAC.m:316-327);
  * `updateLandmarkList` (called first when the list is non-empty, RANSAC.m:17-19) loops `for ii = N`, i.e. only over the
    LAST landmark of the state vector, and refreshes the `loc` of the entry whose index is N (RANSAC.m:354-364).

Parity status: unpinned (no MATLAB here, no fixtures in the reference); pinned by hand-derived cases in
tests/test_ransac_bookkeeping.py.
"""
import math

import numpy as np

_D2R = math.pi / 180.0


def _wrap360(a):
    w = math.fmod(a, 360.0)
    if w < 0.0:
        w += 360.0
    if w == 0.0 and a > 0.0:
        w = 360.0
    return w


class LandmarkEntry:
    """One element of the struct array (RANSAC.m:238-241)."""
    __slots__ = ("loc", "observe", "index", "fresh")

    def __init__(self, loc, observe, index, fresh):
        self.loc = np.array(loc, dtype=np.float64)
        self.observe, self.index, self.fresh = observe, index, fresh


class RansacBookkeeping:
    """Stands where `Landmark.landmarkObj` (a RANSAC object) stands; `laserdata` is a k x 2 array of wall foot-points in
    the world frame (what `getOrthogPoints` would have produced), or None / empty for a scan without walls."""

    landmarkCountConsensus = 10      # RANSAC.m:88
    landmarkDistance = .50           # RANSAC.m:84
    freshnessTimer = 50              # RANSAC.m:91

    def __init__(self):
        self.landmark = []           # RANSAC.m:11
        self.observed = None

    # ---- RANSAC.m:14-152 with the line extraction replaced by its output ----
    def getLandmark(self, laserdata, pose):
        if self.landmark:
            self.updateLandmarkList(pose)
        potential = None if laserdata is None else np.asarray(laserdata, dtype=np.float64).reshape(-1, 2)
        if potential is not None and len(potential):
            observed_LL = self._bookkeep(potential, np.asarray(pose, dtype=np.float64)[:3])
        else:
            observed_LL = np.zeros((0, 3))      # RANSAC.m:143-145
        self.observed = observed_LL
        return observed_LL

    # ---- RANSAC.m:234-334 ----
    def _bookkeep(self, potential, pose):
        reobserved = []
        lst = self.landmark
        if not lst:
            lst.append(LandmarkEntry(potential[0], 1, 0, self.freshnessTimer))           # :236-241
        else:
            for ii in range(len(potential)):                                              # :244
                flag = 0
                for jj in range(len(lst)):                                                # :247 (no break, see module doc)
                    e = lst[jj]
                    d = float(np.linalg.norm(potential[ii] - e.loc))                      # :250
                    if d < self.landmarkDistance:                                         # :253
                        e.observe += 1                                                    # :256
                        flag = 1
                        if e.observe > self.landmarkCountConsensus and e.index == 0:      # :261
                            e.index = max(x.index for x in lst) + 1                       # :262
                        if e.index != 0:                                                  # :267
                            e.loc = potential[ii].copy()                                  # :268
                            dist = math.sqrt((pose[0] - e.loc[0]) ** 2 + (pose[1] - e.loc[1]) ** 2)      # :275
                            ang = _wrap360(math.atan2(e.loc[1] - pose[1], e.loc[0] - pose[0]) / _D2R - pose[2])   # :276-277
                            if not reobserved:                                            # :279-280
                                reobserved.append((dist, ang, float(e.index)))
                            # :283 `elseif ~find(...)`: never true, a second row is never added
                if flag == 0:                                                             # :292-298
                    lst.append(LandmarkEntry(potential[ii], 1, 0, self.freshnessTimer))
        ii = 0                                                                            # :316-327
        while ii < len(lst):
            if lst[ii].index == 0:
                lst[ii].fresh -= 1
                if lst[ii].fresh == 0:
                    del lst[ii]
                    ii -= 1
            ii += 1
        return np.array(reobserved, dtype=np.float64).reshape(-1, 3)

    # ---- RANSAC.m:336-373 ----
    def updateLandmarkList(self, state_vector):
        n = len(state_vector)
        if n > 3:
            ii = (n - 3) // 2                         # `for ii = (length(state_vector)-3)/2`: the last landmark only
            for e in self.landmark:
                if e.index == ii:
                    e.loc = np.array([state_vector[2 * (ii - 1) + 3], state_vector[2 * (ii - 1) + 4]], dtype=np.float64)

    def table(self):
        """(index[], loc[][2]) of the struct array in storage order -- what ekf_measure takes."""
        idx = np.array([e.index for e in self.landmark], dtype=np.float64)
        loc = np.array([e.loc for e in self.landmark], dtype=np.float64).reshape(-1, 2)
        return idx, loc
