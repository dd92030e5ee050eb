#!/usr/bin/env python3
"""k_gather's launch time against the landmark count around the point where its grid (one workgroup per 256 columns, one workgroup resident
per CU at 218 registers x 7 wavefronts) exceeds the CU count: N = 32 768 landmarks <-> 256 workgroups.  HIP events on the engine's stream,
few pending pairs (the first corrections of a batch) and many (the last).   python scripts/probe_gather_rounds.py [storage]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ekf_slam_amd import Engine, _lib as L  # noqa: E402
from ekf_slam_amd.world import World  # noqa: E402

storage = sys.argv[1] if len(sys.argv) > 1 else "f32_split"
for N in (16000, 30000, 32512, 33024, 36000, 40000, 50000):
    w = World(N + 1, 20260101 + 5)
    rng = np.random.default_rng(77)
    n = 3 + 2 * N
    x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks[:N].reshape(-1)])
    e = Engine(mode="known", capacity=N + 1, storage=storage, batch=64)
    e.load_lowrank_state(x, np.arange(1, N + 1.0), rng.uniform(0.01, 0.1, n), rng.normal(0.0, 0.01, (n, 8)))
    steps = []
    for t in range(128):
        u = w.step()
        k = (t * 37) % N
        (_, r, b), = w.observe([k])
        steps.append((u, [r, b], np.diag([r * .01, b * 5.0]), k))
    for (u, z, R, k) in steps[:64]:
        e.predict(u); e.correct(z, R, k)
    e.sync()
    per = []
    for (u, z, R, k) in steps[64:]:
        e.timing_enable(L.EKF_KERNEL_GATHER, True, launches=4)
        e.predict(u); e.correct(z, R, k)
        e.sync()
        per.append(e.timing_read(L.EKF_KERNEL_GATHER)[1] * 1e3)
    print("N %6d  workgroups %4d  k_gather us: pending 0-3 %.2f  28-35 %.2f  60-63 %.2f" % (N, -(-(2 * N) // 256), np.mean(per[:4]), np.mean(per[28:36]), np.mean(per[60:])), flush=True)
    e.close()
