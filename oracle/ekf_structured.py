"""ctypes front for oracle/libekf_oracle.so (the structured C restatement) with the same class surface as
oracle/ekf_dense.py.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED.

The measure() dispatch loops restate EKF_SLAM.m:100-122 / EKF_SLAM_UC.m:102-124; the arithmetic is in
ekf_structured.c.
"""
import ctypes
import os
import subprocess

import numpy as np

from .ekf_dense import LandmarkLookupError, _lookup_loc  # shared restatement of the struct-array lookup

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_d = ctypes.c_double
_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64


def available_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libekf_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libekf_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oekf_create.restype = ctypes.c_void_p
        L.oekf_create.argtypes = [_i64, _d]
        L.oekf_destroy.argtypes = [ctypes.c_void_p]
        for name in ("oekf_x", "oekf_P", "oekf_s", "oekf_Q"):
            getattr(L, name).restype = _dp
            getattr(L, name).argtypes = [ctypes.c_void_p]
        L.oekf_num_landmarks.restype = _i64
        L.oekf_num_landmarks.argtypes = [ctypes.c_void_p]
        L.oekf_ld.restype = _i64
        L.oekf_ld.argtypes = [ctypes.c_void_p]
        L.oekf_set_num_landmarks.argtypes = [ctypes.c_void_p, _i64]
        L.oekf_predict.argtypes = [ctypes.c_void_p, _dp]
        L.oekf_append.argtypes = [ctypes.c_void_p, _dp, _dp, _dp, _d]
        L.oekf_append.restype = ctypes.c_int
        L.oekf_correct.argtypes = [ctypes.c_void_p, _dp, _dp, _i64]
        L.oekf_correct.restype = ctypes.c_int
        L.oekf_associate.argtypes = [ctypes.c_void_p, _dp, _dp, _d, _d, _d,
                                     ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(_i64), _dp, _dp]
        L.oekf_associate.restype = ctypes.c_int
        for name in ("oekf_sind", "oekf_cosd", "oekf_wrapTo360"):
            getattr(L, name).restype = _d
            getattr(L, name).argtypes = [_d]
        L.oekf_threads.restype = ctypes.c_int
        L.oekf_set_threads.argtypes = [ctypes.c_int]
        L.oekf_set_threads(available_cores())
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(_dp)


def _vec(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    assert a.size == n
    return a


class StructuredEKF:
    """mode 'known' = EKF_SLAM.m, mode 'uc' = EKF_SLAM_UC.m."""

    def __init__(self, capacity, mode="known", C=0.2, Rc=None, s_cost=1e-11, s_thresh=1e9, w_pos=0.0):
        self.L = lib()
        self.mode = mode
        self.C = C
        self.Rc = list(Rc) if Rc is not None else ([.01, 5] if mode == "known" else [.1, 5])
        self.s_cost, self.s_thresh, self.w_pos = s_cost, s_thresh, w_pos
        self.h = self.L.oekf_create(int(capacity), float(C))
        if not self.h:
            raise MemoryError("oekf_create failed")
        self.ld = self.L.oekf_ld(self.h)
        self._x = np.ctypeslib.as_array(self.L.oekf_x(self.h), shape=(self.ld,))
        self._P = np.ctypeslib.as_array(self.L.oekf_P(self.h), shape=(self.ld, self.ld))
        self._s = np.ctypeslib.as_array(self.L.oekf_s(self.h), shape=(max(capacity, 1),))
        self.observed = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oekf_destroy(self.h)
            self.h = None

    # ---- state views ----
    @property
    def N(self):
        return int(self.L.oekf_num_landmarks(self.h))

    @property
    def n(self):
        return 3 + 2 * self.N

    @property
    def x(self):
        return self._x[:self.n].copy()

    @property
    def P(self):
        return self._P[:self.n, :self.n].copy()

    @property
    def s(self):
        return self._s[:self.N].copy()

    @property
    def Q(self):
        return np.ctypeslib.as_array(self.L.oekf_Q(self.h), shape=(3, 3)).copy()

    def set_state(self, x, P, s):
        N = (len(x) - 3) // 2
        n = 3 + 2 * N
        self._x[:n] = x
        self._P[:n, :n] = P
        self._s[:N] = s
        self.L.oekf_set_num_landmarks(self.h, N)

    def raw_P(self):
        """Un-copied view of the full ld x ld buffer (for in-place bulk initialisation)."""
        return self._P

    # ---- operations ----
    def predict(self, u):
        self.L.oekf_predict(self.h, _p(_vec(u, 2)))

    def append(self, u, R, landmarkPos, signature):
        rc = self.L.oekf_append(self.h, _p(_vec(u, 2)), _p(_vec(R, 4)), _p(_vec(landmarkPos, 2)), float(signature))
        if rc:
            raise MemoryError("oracle capacity exceeded")

    def correct(self, z, R, idx):
        rc = self.L.oekf_correct(self.h, _p(_vec(z[:2], 2)), _p(_vec(R, 4)), int(idx))
        if rc:
            raise IndexError("landmark index out of range")

    def associate(self, z, R, want_costs=False):
        is_new = ctypes.c_int32()
        index = _i64()
        N = self.N
        pc = np.zeros(max(N, 1))
        sc = np.zeros(max(N, 1))
        self.L.oekf_associate(self.h, _p(_vec(z, 3)), _p(_vec(R, 4)), self.s_cost, self.s_thresh, self.w_pos,
                              ctypes.byref(is_new), ctypes.byref(index), _p(pc), _p(sc))
        if want_costs:
            return bool(is_new.value), int(index.value), pc[:N], sc[:N]
        return bool(is_new.value), int(index.value)

    def _R(self, row):
        return np.array([[row[0] * self.Rc[0], 0.0], [0.0, row[1] * self.Rc[1]]])

    def measure(self, laserData, u, landmark_list):
        observed_LL = landmark_list.getLandmark(laserData, self.x)
        self.observed = observed_LL
        if observed_LL is None or len(observed_LL) == 0:
            return
        observed_LL = np.asarray(observed_LL, dtype=np.float64).reshape(-1, 3)
        for ii in range(1, observed_LL.shape[0] + 1):
            z = observed_LL[ii - 1]
            R = self._R(z)
            if self.N == 0:
                self.append(u, R, _lookup_loc(landmark_list, None), 1)
            elif self.mode == "known":
                if z[2] > self.N:
                    self.append(u, R, _lookup_loc(landmark_list, z[2]), z[2])
                else:
                    self.correct(z, R, ii)
            else:
                new_LM, idx = self.associate(z, R)
                if new_LM:
                    self.append(u, R, _lookup_loc(landmark_list, idx), idx)
                else:
                    self.correct(z, R, idx)


__all__ = ["StructuredEKF", "LandmarkLookupError", "build", "lib"]
