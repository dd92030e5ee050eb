"""Sweep the deferred-downdate batch size and the flush kernel's slab height at N landmarks."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json, time
sys.path.insert(0, %r)
import numpy as np
from ekf_slam_amd import Engine, _lib as L
N, tile, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
e = Engine(capacity=N, tile=tile, batch=batch)
e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
def run(K):
    for i in range(K):
        e.predict([0.1, 3.0]); e.correct([10.0, 100.0], R, (i * 37) %% N)
    e.flush(); e.sync()
run(2 * batch)
K = max(4 * batch, 128)
e.timing_enable(L.EKF_KERNEL_DOWNDATE, True); e.timing_enable(L.EKF_KERNEL_GATHER, True)
t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
nl, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE); ng, mg = e.timing_read(L.EKF_KERNEL_GATHER)
B = e.downdate_algorithmic_bytes()
print(json.dumps({"steps_per_s": round(K / dt, 1), "us_per_step": round(dt / K * 1e6, 2), "flush_ms": round(ms / nl, 4),
                  "flush_frac": round(B / (ms / nl * 1e-3) / 8e12, 4), "gather_us": round(mg / ng * 1e3, 2)}))
""" % ROOT
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
plan = [(64, 1, (8,)), (128, 1, (4,)),
        (64, 8, (32, 64)), (128, 8, (8, 16, 32)),
        (64, 16, (64,)), (128, 16, (16, 32)),
        (64, 32, (64,)), (128, 32, (16, 32)),
        (128, 48, (32,)), (128, 64, (32,))]
if len(sys.argv) > 2:
    plan = eval(sys.argv[2])
for tile, batch, slabs in plan:
    for slab in slabs:
        env = dict(os.environ, EKF_LIB_PATH=os.path.join(ROOT, 'ekf_slam_amd', 'libekfslam_tuning.so'),  # -DEKF_TUNING build: make -C ekf_slam_amd/csrc tuning
                   EKF_DOWNDATE_SLAB_BATCH=str(slab), EKF_DOWNDATE_SLAB=str(slab))
        out = subprocess.run([sys.executable, "-c", CHILD, str(N), str(tile), str(batch)], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-400:]
        print("tile %3d batch %3d slab %3d : %s" % (tile, batch, slab, line), flush=True)
