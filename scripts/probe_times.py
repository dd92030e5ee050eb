"""Ad-hoc GPU probe: where does host time go, and a first downdate bandwidth number."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.time()
from ekf_slam_amd import Engine
from ekf_slam_amd import _lib as L
print("import %.2fs" % (time.time() - t0), flush=True)

def T(label, fn, reps=1):
    t = time.time(); 
    for _ in range(reps): r = fn()
    dt = (time.time() - t) / reps
    print("%-40s %10.3f ms" % (label, dt * 1e3), flush=True)
    return r

e = T("create cap=32", lambda: Engine(capacity=32, tile=16))
T("predict x100 + sync", lambda: ([e.predict([0.1, 3.0]) for _ in range(100)], e.sync()))
T("get_x", lambda: e.get_x(), 10)
T("get_P", lambda: e.get_P(), 10)
T("append", lambda: (e.append([0.1, 3], np.eye(2), [1, 1], 1), e.sync()))
T("correct+sync", lambda: (e.correct([1, 40], np.eye(2), 0), e.sync()), 10)
T("N prop", lambda: e.N, 100)
e.close()

for N, tile in ((1000, 64), (10000, 64), (10000, 128)):
    n = 3 + 2 * N
    rng = np.random.default_rng(1)
    x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
    d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
    e = T("create N=%d T=%d" % (N, tile), lambda: Engine(capacity=N, tile=tile))
    T("load_lowrank", lambda: e.load_lowrank_state(x, s, d, U))
    print("device bytes %.3f GB" % (e.device_bytes() / 1e9))
    R = np.diag([0.2, 50.0])
    for _ in range(3): e.correct([10.0, 100.0], R, 5)
    e.sync()
    for grid in (0,):
        e.timing_enable(L.EKF_KERNEL_DOWNDATE, True); e.timing_enable(L.EKF_KERNEL_GATHER, True)
        t = time.time()
        K = 50
        for i in range(K): e.correct([10.0, 100.0], R, (i * 37) % N)
        e.sync(); wall = time.time() - t
        nl, ms = e.timing_read(L.EKF_KERNEL_DOWNDATE); ng, msg = e.timing_read(L.EKF_KERNEL_GATHER)
        B = e.downdate_algorithmic_bytes()
        print("N=%d T=%d: wall/step %.3f ms | downdate %.4f ms -> %.1f GB/s (%.1f%% of 8 TB/s) | gather %.4f ms"
              % (N, tile, wall / K * 1e3, ms / nl, B / (ms / nl * 1e-3) / 1e9, B / (ms / nl * 1e-3) / 8e12 * 100, msg / ng), flush=True)
    print("digest", e.digest())
    e.close()
