"""MathWorks built-in semantics the reference's hot path relies on.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: these built-ins
are third-party (MATLAB base + Mapping Toolbox, version not pinned anywhere in
the reference) and absent from the image; what is restated here is their
*documented* behaviour at the reference's call sites:

  cosd / sind   EKF_SLAM.m:42,58-59,63-64,84-88   Correspondence.m:36
  atan2d        EKF_SLAM.m:130                    Correspondence.m:56
  wrapTo360     EKF_SLAM.m:50,130                 Correspondence.m:56
  mpower(A,-1)  EKF_SLAM.m:143                    Correspondence.m:69,71

Assumed semantics:
  * sind/cosd are exact at multiples of 90 deg (argument reduced in degrees,
    then sin/cos of a remainder in [-45,45] deg) .
  * atan2d(y,x) = atan2(y,x) * 180/pi.
  * wrapTo360(a) = mod(a,360), except that positive multiples of 360 map to 360.
  * A^-1 for a 2x2 is inv(A): LU with partial pivoting, no symmetry assumed.
"""
import math

import numpy as np

_D2R = math.pi / 180.0
_R2D = 180.0 / math.pi


def _reduce90(a):
    """a (deg) -> (r, m): a = 90*n + r, r in [-45,45], m = n mod 4."""
    q = a / 90.0
    n = math.copysign(math.floor(abs(q) + 0.5), q)  # MATLAB round(): half away from zero
    r = a - n * 90.0
    m = int(np.mod(n, 4.0))
    return r, m


def sind(a):
    a = float(a)
    if not math.isfinite(a):
        return math.nan
    r, m = _reduce90(math.fmod(a, 360.0))
    if m == 0:
        return math.sin(_D2R * r)
    if m == 1:
        return math.cos(_D2R * r)
    if m == 2:
        return -math.sin(_D2R * r)
    return -math.cos(_D2R * r)


def cosd(a):
    a = float(a)
    if not math.isfinite(a):
        return math.nan
    r, m = _reduce90(math.fmod(a, 360.0))
    if m == 0:
        return math.cos(_D2R * r)
    if m == 1:
        return -math.sin(_D2R * r)
    if m == 2:
        return -math.cos(_D2R * r)
    return math.sin(_D2R * r)


def atan2d(y, x):
    return math.atan2(float(y), float(x)) * _R2D


def wrapTo360(a):
    a = float(a)
    positive = a > 0.0
    if not math.isfinite(a):
        return math.nan
    w = math.fmod(a, 360.0)  # exact remainder, sign of a
    if w < 0.0:
        w += 360.0  # mod(): result takes the sign of the divisor
    if w == 0.0 and positive:
        w = 360.0
    return w


def inv2(A):
    """inv() of a 2x2 by LU with partial pivoting (what mpower(A,-1) calls)."""
    a, b, c, d = float(A[0, 0]), float(A[0, 1]), float(A[1, 0]), float(A[1, 1])
    # pivot on the larger |first-column| entry
    if abs(c) > abs(a):
        # rows swapped: [c d; a b]
        l = a / c
        u22 = b - l * d
        # solve for inverse columns of the permuted system, then undo permutation
        # P A = L U with P = swap  ->  A^-1 = U^-1 L^-1 P
        i11, i12 = 1.0 / c, -d / (c * u22)
        i22 = 1.0 / u22
        # U^-1 = [i11 i12; 0 i22], L^-1 = [1 0; -l 1]
        m11 = i11 + i12 * (-l)
        m12 = i12
        m21 = i22 * (-l)
        m22 = i22
        # times P (swap columns)
        return np.array([[m12, m11], [m22, m21]])
    l = c / a
    u22 = d - l * b
    i11, i12 = 1.0 / a, -b / (a * u22)
    i22 = 1.0 / u22
    m11 = i11 + i12 * (-l)
    m12 = i12
    m21 = i22 * (-l)
    m22 = i22
    return np.array([[m11, m12], [m21, m22]])
