"""CPU: the polynomial sind / cosd / atan2d kernels of ekf_slam_amd/csrc/device_math.h (shared by the HIP kernels and the
host-side ekf_motion_model) against glibc, compiled for the host with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_math_matches_libm(tmp_path):
    exe = str(tmp_path / "host_math_check")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-I", os.path.join(ROOT, "ekf_slam_amd", "csrc"),
                    os.path.join(ROOT, "tests", "support", "host_math_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe, "2000000"], check=True, capture_output=True, text=True).stdout.split()
    ms, mc, ma, ma2, bad = float(out[0]), float(out[1]), float(out[2]), float(out[3]), int(out[4])
    assert ms < 1.0 and mc < 1.0          # ulp
    assert ma < 2.0 and ma2 < 3.0         # atan2 includes the rounding of y / x
    assert bad == 0                       # zeros, infinities, NaN, exact values at multiples of 90 / 45 degrees
