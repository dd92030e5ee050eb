"""Seeded synthetic world + landmark source standing in for the reference's RANSAC/ROS front-end.

The reference has no synthetic world (example.m is a broken script); its landmark
front-end (RANSAC.m, out of scope) needs a live ROS laser scan.  The EKF hot path only touches the
front-end's *shape*:

    observed_LL = landmark_list.getLandmark(laserData, x)         EKF_SLAM.m:102, Landmark.m:25-28
    landmark_list.landmarkObj.landmark(k).loc / .index            EKF_SLAM.m:111,120, RANSAC.m:238-241

``SyntheticLandmark`` offers exactly that shape, fed by a precomputed, filter-independent scan list, so the
same odometry + range/bearing inputs can be replayed into the HIP engine and into the CPU oracle.

World: N landmarks uniform in [-L,L]^2 with L = 5*sqrt(N/20) m; robot starts at
[0,0,0 deg] and drives a circle, u = [0.10 m, 3 deg] per step; odometry noise N(0,(0.005 m)^2),
N(0,(0.1 deg)^2); range/bearing noise N(0,(0.02 m)^2), N(0,(0.5 deg)^2).  Indices are handed out
RANSAC-style (RANSAC.m:262-264: next index = max(index)+1) in order of first sighting, and `loc` is
re-derived from the *filter's* pose on every sighting (RANSAC.m:103-105,270-277).
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


class World:
    """Ground truth + per-step odometry and scans.  A scan is a list of (world_id, range, bearing_deg)."""

    def __init__(self, n_landmarks, seed, u_nominal=(0.10, 3.0), odo_sigma=(0.005, 0.1),
                 obs_sigma=(0.02, 0.5)):
        self.N = int(n_landmarks)
        self.rng = np.random.default_rng(seed)
        self.L = 5.0 * math.sqrt(max(self.N, 1) / 20.0)
        self.landmarks = self.rng.uniform(-self.L, self.L, size=(self.N, 2))
        self.u_nominal = u_nominal
        self.odo_sigma = odo_sigma
        self.obs_sigma = obs_sigma
        self.pose = np.array([0.0, 0.0, 0.0])

    def step(self):
        """Advance the true pose by the nominal control; return the noisy odometry u = [dD, dTheta_deg]."""
        d, dth = self.u_nominal
        th = self.pose[2] + dth
        self.pose = np.array([self.pose[0] + d * math.cos(th * _D2R),
                              self.pose[1] + d * math.sin(th * _D2R), th])
        return np.array([d + self.rng.normal(0.0, self.odo_sigma[0]),
                         dth + self.rng.normal(0.0, self.odo_sigma[1])])

    def observe(self, ids):
        """Noisy range/bearing of the given world landmark ids from the true pose."""
        scan = []
        for k in ids:
            dx = self.landmarks[k, 0] - self.pose[0]
            dy = self.landmarks[k, 1] - self.pose[1]
            r = math.hypot(dx, dy) + self.rng.normal(0.0, self.obs_sigma[0])
            b = _wrap360(math.atan2(dy, dx) / _D2R - self.pose[2]) + self.rng.normal(0.0, self.obs_sigma[1])
            scan.append((int(k), max(r, 1e-3), _wrap360(b)))
        return scan

    def nearest(self, m, among=None):
        ids = np.arange(self.N) if among is None else np.asarray(among)
        d2 = ((self.landmarks[ids] - self.pose[:2]) ** 2).sum(axis=1)
        return [int(i) for i in ids[np.argsort(d2, kind="stable")[:m]]]


def make_run(n_landmarks, seed, steps, policy="all", m=8):
    """Precompute (u_t, scan_t) for `steps` SLAM iterations.

    policy "all":     step 0 sights landmark 0 only (the reference's empty-map branch, EKF_SLAM.m:110-111,
                      is only a valid MATLAB call when exactly one landmark is indexed); later steps sight
                      every landmark.
    policy "nearest": step 0 sights landmark 0, step 1 sights every landmark (warm-up sweep that appends
                      them all), later steps sight the m nearest.
    """
    w = World(n_landmarks, seed)
    run = []
    for t in range(steps):
        u = w.step()
        if t == 0:
            ids = [0]
        elif policy == "all" or t == 1:
            ids = list(range(w.N))
        else:
            ids = w.nearest(m)
        run.append((u, w.observe(ids)))
    return w, run


class _LandmarkEntry:
    """One element of RANSAC's struct array (RANSAC.m:238-241)."""
    __slots__ = ("loc", "observe", "index", "fresh")

    def __init__(self, loc, index):
        self.loc = loc
        self.observe = 1
        self.index = index
        self.fresh = 0


class _SyntheticSource:
    """Plays the role of the RANSAC object held in Landmark.landmarkObj."""

    def __init__(self):
        self.landmark = []
        self._by_world_id = {}

    def getLandmark(self, laserdata, x, cosd, sind):
        rows = []
        for (wid, r, b) in laserdata:
            loc = np.array([x[0] + r * cosd(b + x[2]), x[1] + r * sind(b + x[2])])
            e = self._by_world_id.get(wid)
            if e is None:
                nxt = max([lm.index for lm in self.landmark], default=0) + 1
                e = _LandmarkEntry(loc, nxt)
                self._by_world_id[wid] = e
                self.landmark.append(e)
            else:
                e.loc = loc
                e.observe += 1
            rows.append((r, b, float(e.index)))
        rows.sort(key=lambda t: t[2])
        return np.array(rows, dtype=np.float64).reshape(-1, 3)

    def table(self):
        """(index[], loc[][2]) arrays of the struct array, in storage order."""
        idx = np.array([lm.index for lm in self.landmark], dtype=np.float64)
        loc = np.array([lm.loc for lm in self.landmark], dtype=np.float64).reshape(-1, 2)
        return idx, loc


class SyntheticLandmark:
    """Landmark.m surface (Landmark.m:12-33) over the synthetic source instead of RANSAC."""

    def __init__(self, method="SYNTHETIC", trig=None):
        self.method = method
        self.landmarkObj = _SyntheticSource()
        if trig is None:
            trig = (lambda a: math.cos(a * _D2R), lambda a: math.sin(a * _D2R))
        self._cosd, self._sind = trig

    def getLandmark(self, laserdata, x):
        return self.landmarkObj.getLandmark(laserdata, x, self._cosd, self._sind)
