#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the literal-dense NumPy restatement (oracle/ekf_dense.py).

The reference (pure MATLAB, not runnable here) holds no golden vectors, so these pin the *restatement's*
outputs -- PARITY UNPINNED with respect to MATLAB itself.  Inputs are stored next to the outputs so the
fixtures are self-contained.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ekf_slam_amd.world import SyntheticLandmark, make_run  # noqa: E402
from oracle import ekf_dense as D  # noqa: E402


def pack_run(run):
    u = np.array([r[0] for r in run])
    ptr = np.cumsum([0] + [len(r[1]) for r in run])
    data = np.array([list(t) for r in run for t in r[1]], dtype=np.float64).reshape(-1, 3)
    return u, ptr, data


def slam_fixture(name, cls, n_lm, seed, steps, policy, m=8):
    _, run = make_run(n_lm, seed, steps, policy=policy, m=m)
    e, lm = cls(), SyntheticLandmark()
    poses, counts = [], []
    for u, scan in run:
        e.predict(u)
        e.measure(scan, u, lm)
        poses.append(e.x[:3].copy())
        counts.append((len(e.x) - 3) // 2)
    u, ptr, data = pack_run(run)
    np.savez_compressed(os.path.join(HERE, name), u=u, scan_ptr=ptr, scan_data=data, x=e.x, P=e.P,
                        s=np.array(e.s, dtype=np.float64), poses=np.array(poses), counts=np.array(counts),
                        Rc=np.array(e.Rc, dtype=np.float64))
    print(name, "N =", counts[-1], "pose", poses[-1])


def append3_fixture():
    """Constructor state -> predict -> 3 appends interleaved with corrections; every intermediate state kept."""
    e = D.EKF_SLAM()
    ops, xs, Ps = [], [], []
    rng = np.random.default_rng(42)

    def snap(op):
        ops.append(op); xs.append(e.x.copy()); Ps.append(e.P.copy())

    snap([0, 0, 0, 0, 0, 0, 0, 0])
    for k in range(3):
        u = [0.1 + 0.01 * k, 3.0 + k]
        e.predict(u); snap([1, u[0], u[1], 0, 0, 0, 0, 0])
        r, b = rng.uniform(1, 5), rng.uniform(10, 170)
        R = np.diag([r * .01, b * 5.0])
        pos = [e.x[0] + r * np.cos(np.deg2rad(b + e.x[2])), e.x[1] + r * np.sin(np.deg2rad(b + e.x[2]))]
        e.append(u, R, pos, k + 1); snap([2, u[0], u[1], R[0, 0], R[1, 1], pos[0], pos[1], k + 1])
        for idx in range(1, k + 2):
            z = [rng.uniform(1, 5), rng.uniform(10, 170)]
            R = np.diag([z[0] * .01, z[1] * 5.0])
            e._correct(z, R, idx); snap([3, z[0], z[1], R[0, 0], R[1, 1], idx, 0, 0])
    n = len(xs[-1])
    X = np.full((len(xs), n), np.nan)
    PP = np.full((len(xs), n, n), np.nan)
    for i, (x, P) in enumerate(zip(xs, Ps)):
        X[i, :len(x)] = x
        PP[i, :len(x), :len(x)] = P
    np.savez_compressed(os.path.join(HERE, "append3.npz"), ops=np.array(ops, dtype=np.float64), x=X, P=PP)
    print("append3.npz", len(ops), "ops")


if __name__ == "__main__":
    append3_fixture()
    slam_fixture("slam20_known.npz", D.EKF_SLAM, 20, 20260101, 50, "all")
    slam_fixture("slam20_uc.npz", D.EKF_SLAM_UC, 20, 20260101, 50, "all")
    slam_fixture("slam120_uc_nearest.npz", D.EKF_SLAM_UC, 120, 20260102, 30, "nearest", 8)
