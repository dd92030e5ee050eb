"""CPU: the trajectory-log file format round-trips and replays into the CPU oracle exactly like the live run."""
import numpy as np

from ekf_slam_amd.trajectory import TrajectoryLog
from ekf_slam_amd.world import SyntheticLandmark, make_run
from oracle.ekf_structured import StructuredEKF


class _OracleAsEngine:
    """Adapter: the log replays into anything with predict(u) / measure(obs, u, idx, loc)."""

    def __init__(self, ref):
        self.ref = ref

    def predict(self, u):
        self.ref.predict(u)

    def measure(self, obs, u, idx, loc):
        class _Tab:
            pass
        src = _Tab(); src.landmarkObj = _Tab()
        src.landmarkObj.landmark = [type("E", (), {"index": i, "loc": l})() for i, l in zip(idx, loc)]
        src.getLandmark = lambda laser, x: obs
        self.ref.measure(None, u, src)


def test_log_roundtrip_and_replay(tmp_path, oracle_lib):
    _, run = make_run(12, 7, 15, policy="all")
    live = StructuredEKF(16, "uc")
    lm = SyntheticLandmark()
    log = TrajectoryLog()
    for u, scan in run:
        live.predict(u)
        obs = lm.getLandmark(scan, live.x)
        idx, loc = lm.landmarkObj.table()
        log.record(u, obs, idx, loc)
        _OracleAsEngine(live).measure(obs, u, idx, loc)
    path = tmp_path / "run.npz"
    log.save(path)
    back = TrajectoryLog.load(path)
    assert len(back) == len(log) == 15
    for k in range(15):
        np.testing.assert_array_equal(back.u[k], log.u[k])
        np.testing.assert_array_equal(back.obs[k], log.obs[k])
        np.testing.assert_array_equal(back.lm_index[k], log.lm_index[k])
        np.testing.assert_array_equal(back.lm_loc[k], log.lm_loc[k])
    again = StructuredEKF(16, "uc")
    back.replay(_OracleAsEngine(again))
    np.testing.assert_array_equal(again.x, live.x)
    np.testing.assert_array_equal(again.P, live.P)
