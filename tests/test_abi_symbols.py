"""CPU tests of the boundary: libekfslam.so builds for gfx950, loads without a GPU, exports every symbol
include/ekfslam.h declares, and fails loudly (no fallback) when no HIP device is present.  No compute here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import ekf_slam_amd
    ekf_slam_amd.build()
    return ekf_slam_amd.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ekfslam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ekf_[a-z_0-9A-Z]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from ekf_slam_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "libekfslam.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes binding and header disagree"


def test_abi_version_and_status_strings(L):
    assert L.ekf_abi_version() == 1
    assert L.ekf_status_string(0) == b"ok"
    assert b"capacity" in L.ekf_status_string(4)


def test_config_defaults_match_reference_properties(L):
    from ekf_slam_amd import _lib
    cfg = _lib.EkfConfig()
    assert L.ekf_config_default(ctypes.byref(cfg), _lib.EKF_MODE_KNOWN) == 0
    assert (cfg.C, cfg.Rc[0], cfg.Rc[1]) == (0.2, .01, 5.0)            # EKF_SLAM.m:12-13
    assert (cfg.s_cost, cfg.s_thresh, cfg.w_pos) == (1e-11, 1e9, 0.0)  # EKF_SLAM.m:14,16
    assert L.ekf_config_default(ctypes.byref(cfg), _lib.EKF_MODE_UC) == 0
    assert (cfg.Rc[0], cfg.Rc[1]) == (.1, 5.0)                         # EKF_SLAM_UC.m:13
    assert L.ekf_config_default(ctypes.byref(cfg), 7) == _lib.EKF_ERR_INVALID_ARG


def test_motion_model_is_host_only_and_matches_kat(L):
    # [x_new,F] = f(x,u)  EKF_SLAM.m:56-65 -- a pure function, runs without a GPU
    from ekf_slam_amd.engine import _p
    x = np.array([1.0, 2.0, 90.0, 5.0, 6.0])
    u = np.array([2.0, 90.0])
    xn, F = np.empty(5), np.empty(25)
    assert L.ekf_motion_model(_p(x), 5, _p(u), _p(xn), _p(F)) == 0
    np.testing.assert_allclose(xn, [1.0 - 2.0, 2.0, 180.0, 5.0, 6.0], atol=1e-15)
    F = F.reshape(5, 5, order="F")
    expect = np.eye(5); expect[0, 2] = -2.0; expect[1, 2] = 0.0
    np.testing.assert_allclose(F, expect, atol=1e-15)


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ekf_slam_amd import EkfError, Engine, _lib
    with pytest.raises(EkfError) as ei:
        Engine(capacity=4)
    assert ei.value.status == _lib.EKF_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    """The product path must never route through oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "ekf_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                text = open(os.path.join(dirpath, f)).read()
                where = os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+\.*oracle\b", text, flags=re.M), "%s imports the oracle" % where
                assert "libekf_oracle" not in text and "ekf_structured" not in text and "ekf_dense" not in text, where
