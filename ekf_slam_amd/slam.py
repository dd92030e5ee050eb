"""Host mirror of the reference's MATLAB class surface over libekfslam (no arithmetic lives here).

Same names, argument meaning and 1-based landmark indices as the reference, so a user of
``EKF_SLAM`` / ``EKF_SLAM_UC`` / ``Correspondence`` / ``Landmark`` / ``SLAM`` / ``append`` finds the same calls:

    SLAM(name)                         SLAM.m:21-40       .predict(u) :42  .measure(laserdata,u) :52  .runSlam() :70
    EKF_SLAM() / EKF_SLAM_UC()         EKF_SLAM.m:26-34 / EKF_SLAM_UC.m:27-36
      .predict(u)                      EKF_SLAM.m:40-51
      [x_new,F] = .f(x,u)              EKF_SLAM.m:56-65
      .append(u,R,landmarkPos,sig)     EKF_SLAM.m:67-98
      .measure(laserData,u,lm_list)    EKF_SLAM.m:100-151 / EKF_SLAM_UC.m:102-152
      properties x P Q s C Rc ...      EKF_SLAM.m:5-22
    Correspondence(cost,thresh,method) Correspondence.m:12-25   .estimateCorrespondence(z,R,x,P,s) :28-88
    Landmark(method)                   Landmark.m:12-33         .getLandmark(laserdata,x)
    [x,P] = append(x,P,u,idx,R,pos)    append.m:1-27

Differences forced by the host language: x is a 1-D array (the reference's 1 x n row vector); plotting
(EKF_SLAM.m:154-234, out of scope) is replaced by ``plot_data()`` which returns what plot() reads (pose and
the 2x2 diagonal covariance blocks); the ROS subscribers of SLAM.m:23-24 are replaced by a scripted
(u, scan) feed.
"""
import ctypes
import warnings

import numpy as np

from . import _lib as L
from .engine import Engine, _p, _vec
from .world import SyntheticLandmark

_DEFAULT_CAPACITY = 1024


class _EkfBase:
    _mode = "known"

    def __init__(self, capacity=_DEFAULT_CAPACITY, **engine_kw):
        self._e = Engine(mode=self._mode, capacity=capacity, **engine_kw)
        self.landmark_list = None
        self.observed = None
        self.log = None             # a TrajectoryLog records what predict / measure consume

    # ---- the reference's public properties, pulled from HBM on demand ----
    @property
    def x(self):
        return self._e.get_x()

    @x.setter
    def x(self, v):                 # assignable like the reference's (EKF_SLAM.m:6-9); x fixes the landmark count
        v = _vec(v)
        self._e._check(self._e.lib.ekf_set_x(self._e.h, _p(v), v.size))

    @property
    def P(self):
        return self._e.get_P()

    @P.setter
    def P(self, v):
        Pf = np.asfortranarray(np.asarray(v, dtype=np.float64))
        self._e._check(self._e.lib.ekf_set_P(self._e.h, _p(Pf.reshape(-1, order="F")), Pf.shape[0]))

    @property
    def s(self):
        return self._e.get_s()

    @s.setter
    def s(self, v):
        v = _vec(v)
        self._e._check(self._e.lib.ekf_set_s(self._e.h, _p(v) if v.size else None, v.size))

    @property
    def Q(self):
        """zeros(size(P)) with the 3x3 process-noise block (EKF_SLAM.m:43-44)."""
        n = self._e.n
        Q = np.zeros((n, n))
        Q[0:3, 0:3] = self._e.get_Q3()
        return Q

    @property
    def C(self):
        return self._e.cfg.C

    @C.setter
    def C(self, v):
        self._e.set_params(C=v)

    @property
    def Rc(self):
        return [self._e.cfg.Rc[0], self._e.cfg.Rc[1]]

    @Rc.setter
    def Rc(self, v):
        self._e.set_params(Rc=v)

    @property
    def s_cost(self):               # public assignable properties of the reference (EKF_SLAM.m:14-16)
        return self._e.cfg.s_cost

    @s_cost.setter
    def s_cost(self, v):
        self._e.set_params(s_cost=v)

    @property
    def s_thresh(self):
        return self._e.cfg.s_thresh

    @s_thresh.setter
    def s_thresh(self, v):
        self._e.set_params(s_thresh=v)

    # ---- methods ----
    def predict(self, u):
        self._last_u = np.asarray(u, dtype=np.float64)
        self._e.predict(u)

    def f(self, x, u):
        x = _vec(x)
        n = x.size
        x_new = np.empty(n)
        F = np.empty(n * n)
        rc = self._e.lib.ekf_motion_model(_p(x), n, _p(_vec(u, 2)), _p(x_new), _p(F))
        if rc:
            raise L.EkfError(rc, "ekf_motion_model")
        return x_new, F.reshape(n, n, order="F")

    def append(self, u, R, landmarkPos, signature):
        self._e.append(u, R, landmarkPos, signature)

    def _push_params(self):
        pass

    def measure(self, laserData, u, landmark_list):
        self._push_params()
        observed_LL = landmark_list.getLandmark(laserData, self.x)       # EKF_SLAM.m:102
        self.observed = observed_LL                                      # :103
        idx, loc = landmark_list.landmarkObj.table()
        if self.log is not None:
            self.log.record(u, observed_LL, idx, loc)
        if observed_LL is None or len(observed_LL) == 0:                 # :105
            return
        self._e.measure(observed_LL, u, idx, loc)

    def plot_data(self):
        """What plot() reads: pose and the 2x2 diagonal blocks of P (EKF_SLAM.m:180,205)."""
        return self.x, list(self._e.get_P_diag_blocks())

    def plot(self, landmark_list=None):
        """plot(h, landmark_list) (EKF_SLAM.m:154-234) minus the drawing: returns what the figure is made of -- pose,
        landmark positions, the 2x2 covariance blocks -- and forwards to the landmark source's own plot if it has one
        (EKF_SLAM.m:167)."""
        x, blocks = self.plot_data()
        src = getattr(landmark_list, "landmarkObj", None)
        if src is not None and hasattr(src, "plot"):
            src.plot(x, self.observed)
        return {"pose": x[:3], "landmarks": x[3:].reshape(-1, 2), "covariance_blocks": blocks}

    def sync(self):
        self._e.sync()


class EKF_SLAM(_EkfBase):
    """Known correspondence (EKF_SLAM.m)."""
    _mode = "known"


class EKF_SLAM_UC(_EkfBase):
    """Unknown correspondence (EKF_SLAM_UC.m); owns a Correspondence (EKF_SLAM_UC.m:16)."""
    _mode = "uc"

    def __init__(self, capacity=_DEFAULT_CAPACITY, **engine_kw):
        super().__init__(capacity, **engine_kw)
        # EKF_SLAM_UC.m:16: Correspondence(.00000000001, 1000000000, 'EKF_SLAM_UC') -- the engine's defaults (ekf_config_default);
        # s_cost / s_thresh given as engine keywords arrive here too, so that the property and the engine start out equal
        self.correspondence = Correspondence(self._e.cfg.s_cost, self._e.cfg.s_thresh, 'EKF_SLAM_UC')

    def _push_params(self):
        """measure() associates with h.correspondence's cost / threshold (EKF_SLAM_UC.m:16,119): the property is public and
        may be replaced or edited at any time, so its values go to the engine before every scan (as matlab/EKF_SLAM_UC.m's
        pushParams does)."""
        c = self.correspondence
        cfg = self._e.cfg
        if float(c.s_cost) != cfg.s_cost or float(c.s_thresh) != cfg.s_thresh:
            self._e.set_params(s_cost=float(c.s_cost), s_thresh=float(c.s_thresh))


class Correspondence:
    """Value class of Correspondence.m; the computation runs on the device through a scratch handle."""

    def __init__(self, cost, thresh, method):
        self.method = method
        self.s_cost = cost
        self.s_thresh = thresh
        if method != 'EKF_SLAM_UC':                                      # Correspondence.m:19-23
            warnings.warn('Improper method specified. Using ML as default.')
            self.method = 'ML'
        self.position_cost = None
        self.signature_cost = None

    def estimateCorrespondence(self, z, R, x, P, s, **engine_kw):
        x = _vec(x)
        N = (x.size - 3) // 2
        e = Engine(mode="uc", capacity=max(N, 1), s_cost=float(self.s_cost), s_thresh=float(self.s_thresh), **engine_kw)
        try:
            e.set_state(x, P, s)
            is_new, idx0, pc, sc = e.associate(z, R, want_costs=True)
        finally:
            e.close()
        self.position_cost, self.signature_cost = pc, sc
        return is_new, idx0 + 1                                          # 1-based like the reference


class Landmark:
    """Landmark.m surface.  'SYNTHETIC' is the seeded source of ekf_slam_amd.world; 'RANSAC' (Landmark.m:14-16) is the
    reference's landmark-list bookkeeping (ekf_slam_amd.ransac_bookkeeping) fed with wall foot-points -- the laser-scan
    line extraction itself needs ROS and MATLAB toolboxes and is out of scope."""

    def __init__(self, method):
        self.method = method
        if method == 'SYNTHETIC':
            self._src = SyntheticLandmark(method)
            self.landmarkObj = self._src.landmarkObj
        elif method == 'RANSAC':
            from .ransac_bookkeeping import RansacBookkeeping
            self.landmarkObj = RansacBookkeeping()
            self._src = self.landmarkObj
        else:
            warnings.warn('Improper landmark recognition method.')       # Landmark.m:18
            self._src = SyntheticLandmark('SYNTHETIC')
            self.landmarkObj = self._src.landmarkObj

    def getLandmark(self, laserdata, x):
        return self._src.getLandmark(laserdata, x)


def append(x, P, u, idx, R, pos, **engine_kw):
    """[x,P] = append(x,P,u,idx,R,pos)  (append.m:1-27): appends only if numOfLandmarks < idx."""
    x = _vec(x)
    N = (x.size - 3) // 2
    if not (N < idx):
        return x, np.asarray(P, dtype=np.float64)
    e = Engine(mode="known", capacity=N + 1, **engine_kw)
    try:
        e.set_state(x, P, np.zeros(N))
        e.append(u, R, pos, 0.0)
        return e.get_x(), e.get_P()
    finally:
        e.close()


class SLAM:
    """SLAM.m facade with the ROS subscribers replaced by a scripted feed of (u, scan) pairs."""

    def __init__(self, inputString, feed=None, capacity=_DEFAULT_CAPACITY, landmark_method='SYNTHETIC', **engine_kw):
        self.algorithmName = inputString
        self.feed = iter(feed) if feed is not None else None
        self.u = np.zeros(3)
        if inputString == 'EKF_SLAM':                                    # SLAM.m:26-35
            self.slam = EKF_SLAM(capacity, **engine_kw)
        elif inputString == 'EKF_SLAM_UC':
            self.slam = EKF_SLAM_UC(capacity, **engine_kw)
        else:
            self.slam = None
        self.LM = Landmark(landmark_method)       # SLAM.m:29,34 use 'RANSAC'

    def predict(self, u):
        if self.slam is not None:
            self.slam.predict(u)

    def measure(self, laserdata, u):
        if self.slam is not None:
            self.slam.measure(laserdata, u, self.LM)

    def plot(self):                               # SLAM.m:61-68
        if self.slam is not None:
            return self.slam.plot(self.LM)

    def runSlam(self):
        """One SLAM iteration: predict then measure (SLAM.m:105-116)."""
        u, scan = next(self.feed)
        self.u = np.asarray(u, dtype=np.float64)
        self.slam.predict(self.u)
        self.slam.measure(scan, self.u, self.LM)
