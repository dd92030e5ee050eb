"""GPU: checkpoint (ekf_checkpoint_save / _load) and trajectory replay reproduce a run bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("storage,tile,batch", [("f64", 16, 1), ("f64", 0, 8), ("f32", 0, 4)])
def test_checkpoint_resume_is_bit_identical(tmp_path, storage, tile, batch):
    from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
    from ekf_slam_amd.trajectory import TrajectoryLog
    from ekf_slam_amd.world import make_run
    _, run = make_run(40, 11, 24, policy="nearest", m=6)
    full = EKF_SLAM_UC(capacity=64, tile=tile, storage=storage, batch=batch)
    full.log = TrajectoryLog()
    lm = Landmark('SYNTHETIC')
    for k, (u, scan) in enumerate(run):
        full.predict(u); full.measure(scan, u, lm)
        if k == 13:
            full._e.checkpoint_save(tmp_path / "mid.ckpt")      # pending pairs are flushed into the file
    log_path = tmp_path / "run.npz"
    full.log.save(log_path)
    # resume: new handle, load the checkpoint taken after step 13, replay steps 14.. from the log
    resumed = EKF_SLAM_UC(capacity=64, tile=tile, storage=storage, batch=batch)
    resumed._e.checkpoint_load(tmp_path / "mid.ckpt")
    assert resumed._e.N == 40
    log = TrajectoryLog.load(log_path)
    log.replay(resumed._e, start=14)
    np.testing.assert_array_equal(resumed.x, full.x)
    np.testing.assert_array_equal(resumed.P, full.P)
    np.testing.assert_array_equal(resumed.s, full.s)
    # a fresh engine replaying the whole log reproduces the run too -- bit for bit with F64 tiles; with F32 tiles the
    # checkpoint's flush moved one float rounding (a flush is where the landmark block is rounded), so only closely
    fresh = EKF_SLAM_UC(capacity=64, tile=tile, storage=storage, batch=batch)
    log.replay(fresh._e)
    if storage == "f64":
        np.testing.assert_array_equal(fresh.P, full.P)
    else:
        assert np.abs(fresh.P - full.P).max() / np.abs(full.P).max() < 1e-5


def test_checkpoint_refuses_mismatched_handle(tmp_path):
    from ekf_slam_amd import Engine, EkfError, _lib as L
    e = Engine(capacity=8, tile=16)
    e.append([0.1, 1.0], np.eye(2), [1.0, 2.0], 1)
    e.checkpoint_save(tmp_path / "a.ckpt")
    with pytest.raises(EkfError) as ei:
        Engine(capacity=8, tile=32).checkpoint_load(tmp_path / "a.ckpt")
    assert ei.value.status == L.EKF_ERR_STATE
    with pytest.raises(EkfError):
        Engine(capacity=8, tile=16).checkpoint_load(tmp_path / "missing.ckpt")
    ok = Engine(capacity=8, tile=16)
    ok.checkpoint_load(tmp_path / "a.ckpt")
    np.testing.assert_array_equal(ok.get_P(), e.get_P())
