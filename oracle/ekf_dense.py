"""Literal-dense NumPy restatement of the reference's EKF-SLAM hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- PARITY UNPINNED (no MATLAB /
Octave in the image, no golden vectors in the reference).

Every dense MATLAB expression of the reference is executed as the same dense
NumPy expression (eye(n), zeros(n), the 5 x n selector F_k, n x n x n products),
so this file is the semantic ground truth for N <~ 1k landmarks; it is O(n^3)
per step exactly like the reference.  Angles are degrees everywhere.

Follows (file:line in /root/reference):
  EKF_SLAM.m:26-34    constructor             EKF_SLAM_UC.m:27-36
  EKF_SLAM.m:40-51    predict                 EKF_SLAM_UC.m:42-53
  EKF_SLAM.m:56-65    f (motion model)        EKF_SLAM_UC.m:58-67
  EKF_SLAM.m:67-98    append                  EKF_SLAM_UC.m:69-100, append.m:1-27
  EKF_SLAM.m:100-151  measure (known corr.)   EKF_SLAM_UC.m:102-152 (unknown corr.)
  Correspondence.m:12-25, 28-88               constructor, estimateCorrespondence
"""
import warnings

import numpy as np

from .matlab_compat import atan2d, cosd, inv2, sind, wrapTo360


class LandmarkLookupError(RuntimeError):
    """MATLAB would raise: `landmark(find(...)).loc` did not expand to exactly one argument."""


def _lookup_loc(landmark_list, key=None):
    """landmark_list.landmarkObj.landmark(find([...index] == key)).loc

    key=None reproduces EKF_SLAM.m:111 `find([landmark.index])` (all non-zero indices).
    The comma-separated-list expansion only forms a valid append() call when exactly one
    landmark matches; anything else is a MATLAB error.
    """
    lms = landmark_list.landmarkObj.landmark
    if key is None:
        hits = [lm for lm in lms if lm.index != 0]
    else:
        hits = [lm for lm in lms if lm.index == key]
    if len(hits) != 1:
        raise LandmarkLookupError("landmark lookup matched %d entries" % len(hits))
    return hits[0].loc


def f(x, u):
    """[x_new,F] = f(x,u)   EKF_SLAM.m:56-65"""
    x = np.asarray(x, dtype=np.float64)
    x_new = x.copy()
    x_new[0] = x[0] + u[0] * cosd(x[2] + u[1])
    x_new[1] = x[1] + u[0] * sind(x[2] + u[1])
    x_new[2] = x[2] + u[1]
    F = np.eye(len(x))
    F[0, 2] = -1 * u[0] * sind(x[2])
    F[1, 2] = u[0] * cosd(x[2])
    return x_new, F


def predict(x, u, P, C):
    """[x,P] = predict(x,u,P,C)   example.m:24-35 == EKF_SLAM.m:40-51; also returns Q."""
    W = np.array([[u[0] * cosd(x[2])], [u[0] * sind(x[2])], [u[1]]])
    Q = np.zeros(P.shape)
    Q[0:3, 0:3] = (W * C) @ W.T
    x, F = f(x, u)
    P = F @ P @ F.T + Q
    x[2] = wrapTo360(x[2])
    return x, P, Q


def append(x, P, u, idx, R, pos):
    """[x,P] = append(x,P,u,idx,R,pos)   append.m:1-27 (guarded by numOfLandmarks < idx)."""
    x = np.asarray(x, dtype=np.float64)
    numOfLandmarks = (len(x) - 3) // 2
    if numOfLandmarks < idx:
        x, P = _append_core(x, P, u, R, pos)
    return x, P


def _append_core(x, P, u, R, pos):
    numOfLandmarks = (len(x) - 3) // 2
    n = P.shape[0]
    x = np.concatenate([x, [pos[0], pos[1]]])
    jxr = np.array([[1.0, 0.0, -u[0] * sind(x[2])], [0.0, 1.0, u[0] * cosd(x[2])]])
    jz = np.array([[cosd(u[1]), -u[0] * sind(u[1])], [sind(u[1]), u[0] * cosd(u[1])]])
    Pn = np.zeros((n + 2, n + 2))
    Pn[:n, :n] = P
    Pn[n:n + 2, n:n + 2] = jxr @ Pn[0:3, 0:3] @ jxr.T + jz @ R @ jz.T      # C
    Pn[0:3, n:n + 2] = Pn[0:3, 0:3] @ jxr.T                               # I
    Pn[n:n + 2, 0:3] = Pn[0:3, n:n + 2].T                                 # H
    for k in range(numOfLandmarks):
        c = 3 + 2 * k
        Pn[n:n + 2, c:c + 2] = jxr @ Pn[c:c + 2, 0:3].T                   # F
        Pn[c:c + 2, n:n + 2] = Pn[n:n + 2, c:c + 2].T                     # G
    return x, Pn


def _innovation_terms(x, idx):
    """delta_k, q_k, z_k, H_k for 1-based landmark idx (EKF_SLAM.m:125-138, Correspondence.m:50-63)."""
    n = len(x)
    numOfLandmarks = (n - 3) // 2
    j = 3 + 2 * (idx - 1)
    mu_k = np.array([[x[j]], [x[j + 1]]])
    delta_k = mu_k - x[0:2].reshape(2, 1)
    q_k = float((delta_k.T @ delta_k)[0, 0])
    z_k = np.array([[np.sqrt(q_k)], [wrapTo360(atan2d(delta_k[1, 0], delta_k[0, 0]) - x[2])]])
    F_k = np.zeros((5, numOfLandmarks * 2 + 3))
    F_k[0:3, 0:3] = np.eye(3)
    F_k[3:5, j:j + 2] = np.eye(2)
    sq = np.sqrt(q_k)
    d0, d1 = delta_k[0, 0], delta_k[1, 0]
    H_k = ((1 / q_k) * np.array([[-sq * d0, -sq * d1, 0.0, sq * d0, sq * d1],
                                 [d1, -d0, -q_k, -d1, d0]])) @ F_k
    return z_k, H_k


class Correspondence:
    """Correspondence.m:1-92 (value class)."""

    def __init__(self, cost, thresh, method):
        self.method = method
        if method == 'EKF_SLAM_UC':
            self.s_cost = cost
            self.s_thresh = thresh
        else:
            warnings.warn('Improper method specified. Using ML as default.')
            self.s_cost = cost
            self.s_thresh = thresh
            self.method = 'ML'
        self.last_position_cost = None  # diagnostic only; Correspondence.m:69 computes and discards it
        self.last_signature_cost = None

    def estimateCorrespondence(self, z, R, x, P, s):
        """[newLL,index] = estimateCorrespondence(z,R,x,P,s)   Correspondence.m:28-88"""
        x = np.asarray(x, dtype=np.float64)
        newLL = True
        numOfLandmarks = (len(x) - 3) // 2
        index = numOfLandmarks + 1
        min_log_likelihood = np.inf
        log_likelihood = np.zeros(numOfLandmarks)
        pos_costs = np.zeros(numOfLandmarks)
        zc = np.array([[z[0]], [z[1]]])
        for kk in range(1, numOfLandmarks + 1):
            z_k, H_k = _innovation_terms(x, kk)
            phi_k = H_k @ P @ H_k.T + R
            nu = zc - z_k
            position_cost = float((nu.T @ inv2(phi_k) @ nu)[0, 0])      # :69 (unused by the decision)
            d = z[2] - s[kk - 1]
            signiture_cost = d * (1.0 / self.s_cost) * d                # :71
            pos_costs[kk - 1] = position_cost
            log_likelihood[kk - 1] = signiture_cost                     # :75
            if log_likelihood[kk - 1] <= self.s_thresh:
                if log_likelihood[kk - 1] < min_log_likelihood:
                    newLL = False
                    min_log_likelihood = log_likelihood[kk - 1]
                    index = kk
        self.last_position_cost = pos_costs
        self.last_signature_cost = log_likelihood
        return newLL, index


class EKF_SLAM:
    """EKF_SLAM.m (known correspondence).  x is the 1 x n row vector, P is n x n."""

    Rc_default = (.01, 5)

    def __init__(self):
        self.x = np.zeros(3)
        self.P = np.eye(3) * 0.1
        self.Q = None
        self.s = []
        self.C = 0.2
        self.Rc = list(self.Rc_default)
        self.s_cost = .00000000001
        self.s_thresh = 1000000000
        self.landmark_list = None
        self.observed = None

    def predict(self, u):
        self.x, self.P, self.Q = predict(self.x, u, self.P, self.C)

    def f(self, x, u):
        return f(x, u)

    def append(self, u, R, landmarkPos, signature):
        self.s.append(signature)
        self.x, self.P = _append_core(self.x, self.P, u, R, landmarkPos)

    def _R(self, row):
        R = np.zeros((2, 2))
        R[0, 0] = row[0] * self.Rc[0]
        R[1, 1] = row[1] * self.Rc[1]
        return R

    def _correct(self, z, R, idx):
        """EKF_SLAM.m:124-145"""
        z_k, H_k = _innovation_terms(self.x, idx)
        phi_k = H_k @ self.P @ H_k.T + R
        K = self.P @ H_k.T @ inv2(phi_k)
        zc = np.array([[z[0]], [z[1]]])
        self.x = self.x + (K @ (zc - z_k)).T[0]
        self.P = (np.eye(self.P.shape[0]) - K @ H_k) @ self.P

    def measure(self, laserData, u, landmark_list):
        observed_LL = landmark_list.getLandmark(laserData, self.x)
        self.observed = observed_LL
        if observed_LL is None or len(observed_LL) == 0:
            return
        observed_LL = np.asarray(observed_LL, dtype=np.float64).reshape(-1, 3)
        for ii in range(1, observed_LL.shape[0] + 1):
            row = observed_LL[ii - 1]
            R = self._R(row)
            if len(self.x) < 4:
                self.append(u, R, _lookup_loc(landmark_list, None), 1)
            else:
                self._measure_row(ii, row, R, u, landmark_list)

    def _measure_row(self, ii, z, R, u, landmark_list):
        numOfLandmarks = (len(self.x) - 3) // 2
        if z[2] > numOfLandmarks:
            self.append(u, R, _lookup_loc(landmark_list, z[2]), z[2])
        else:
            idx = ii                                                     # EKF_SLAM.m:123
            if idx > numOfLandmarks:
                raise IndexError("idx=ii exceeds the state (MATLAB: index exceeds matrix dimensions)")
            self._correct(z, R, idx)


class EKF_SLAM_UC(EKF_SLAM):
    """EKF_SLAM_UC.m (unknown correspondence): differs in Rc, the Correspondence object and measure."""

    Rc_default = (.1, 5)

    def __init__(self):
        super().__init__()
        self.correspondence = Correspondence(.00000000001, 1000000000, 'EKF_SLAM_UC')

    def _measure_row(self, ii, z, R, u, landmark_list):
        new_LM, idx = self.correspondence.estimateCorrespondence(z, R, self.x, self.P, self.s)
        if new_LM:
            self.append(u, R, _lookup_loc(landmark_list, idx), idx)
        else:
            self._correct(z, R, idx)
