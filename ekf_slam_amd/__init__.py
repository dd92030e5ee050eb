"""ekf_slam_amd -- host side of the MI355X-native EKF-SLAM update engine.

The product is ``libekfslam.so`` (C ABI in include/ekfslam.h, gfx950 kernels in ekf_slam_amd/csrc).  This
package is the thin host mirror of the reference's MATLAB class surface over that ABI; it holds no
arithmetic of its own and has no CPU fallback.
"""
from ._lib import EkfError, build, lib  # noqa: F401
from .engine import Engine  # noqa: F401
