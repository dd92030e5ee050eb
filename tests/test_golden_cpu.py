"""CPU: the structured C oracle replayed over the committed golden fixtures (tests/golden/, produced by the
literal-dense restatement -- parity with MATLAB itself is unpinned, see oracle/__init__.py)."""
import numpy as np
import pytest

from ekf_slam_amd.world import SyntheticLandmark
from golden_util import load, rel_err, replay_append3, replay_slam
from oracle.ekf_structured import StructuredEKF


@pytest.mark.parametrize("name,mode", [("slam20_known.npz", "known"), ("slam20_uc.npz", "uc"),
                                       ("slam120_uc_nearest.npz", "uc")])
def test_structured_oracle_reproduces_golden_slam_runs(name, mode, oracle_lib):
    g = load(name)
    e = StructuredEKF(128, mode, Rc=g["Rc"])
    poses = replay_slam(e, SyntheticLandmark(), g)
    assert e.N == int(g["counts"][-1])
    assert rel_err(poses, g["poses"]) < 1e-10
    assert rel_err(e.x, g["x"]) < 1e-10
    assert rel_err(e.P, g["P"]) < 1e-10
    np.testing.assert_array_equal(e.s, g["s"])


def test_structured_oracle_reproduces_append3(oracle_lib):
    g = load("append3.npz")
    e = StructuredEKF(4, "known")
    worst = replay_append3(g, e.predict, e.append, lambda z, R, idx: e.correct(z, R, idx), lambda: (e.x, e.P))
    assert worst < 1e-12
