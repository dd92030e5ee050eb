"""Sweep downdate launch variants in child processes (env-selected) and report GB/s at N landmarks."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json, time
sys.path.insert(0, %r)
import numpy as np
from ekf_slam_amd import Engine, _lib as L
N, tile = int(sys.argv[1]), int(sys.argv[2])
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
e = Engine(capacity=N, tile=tile)
e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
for i in range(10): e.correct([10.0, 100.0], R, (i * 37) %% N)
e.sync()
best = None
for rep in range(3):
    e.timing_enable(L.EKF_KERNEL_DOWNDATE, True)
    for i in range(40): e.correct([10.0, 100.0], R, (i * 37) %% N)
    nl, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE)
    t = ms / nl
    best = t if best is None else min(best, t)
B = e.downdate_algorithmic_bytes()
print(json.dumps({"ms": best, "GBs": B / (best * 1e-3) / 1e9, "frac": B / (best * 1e-3) / 8e12}))
""" % ROOT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
combos = [(64, s) for s in (8, 16, 32, 64)] + [(128, s) for s in (4, 8, 16, 32)] + [(32, 16), (32, 32), (16, 16)]
for tile, slab in combos:
    for grid in (0,):
        env = dict(os.environ, EKF_LIB_PATH=os.path.join(ROOT, 'ekf_slam_amd', 'libekfslam_tuning.so'),  # -DEKF_TUNING build: make -C ekf_slam_amd/csrc tuning
                   EKF_DOWNDATE_GRID=str(grid), EKF_DOWNDATE_SLAB=str(slab))
        out = subprocess.run([sys.executable, "-c", CHILD, str(N), str(tile)], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-300:]
        print("tile %3d slab %3d grid %5d : %s" % (tile, slab, grid, line), flush=True)
