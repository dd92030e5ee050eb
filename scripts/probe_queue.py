"""Does host run-ahead hurt?  Time N steps (predict+correct, batch 32 at 10k landmarks) with a sync every S steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine
N = 10000; n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
e = Engine(capacity=N, batch=32)
e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
def run(K, S):
    t0 = time.perf_counter(); tcall = 0.0; worst = 0.0
    for i in range(K):
        e.predict([0.1, 3.0])
        a = time.perf_counter(); e.correct([10.0, 100.0], R, (i * 37) % N); b = time.perf_counter() - a
        tcall += b; worst = max(worst, b)
        if S and (i + 1) % S == 0: e.sync()
    e.flush(); e.sync()
    dt = time.perf_counter() - t0
    print("K=%5d sync-every=%4d : %.1f us/step wall, mean correct() call %.1f us, worst %.0f us" % (K, S, dt / K * 1e6, tcall / K * 1e6, worst * 1e6), flush=True)
run(128, 0)
for K in (256, 512, 1024, 2048):
    for S in (0, 64, 128, 256):
        run(K, S)
