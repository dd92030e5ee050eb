"""Host-side cost of one update-step (predict + correct) through the Python binding, plain calls vs marshalled steps, on a map
small enough that the GPU is never the limit.  python scripts/probe_host_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine
N = 64
rng = np.random.default_rng(1)
n = 3 + 2 * N
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
P = np.diag(rng.uniform(0.01, 0.1, n))
steps = []
for t in range(20000):
    z = np.array([10.0 + t % 7, 100.0])
    steps.append(([0.1, 3.0], z, np.diag([z[0] * .01, z[1] * 5.0]), (t * 37) % N))
for mode in ("plain", "raw"):
    e = Engine(capacity=N, tile=16, batch=32)
    e.set_state(x, P, np.arange(1, N + 1.0))
    run = e.marshal_steps(steps)
    e.sync(); t0 = time.perf_counter()
    if mode == "plain":
        for (u, z, R, k) in steps:
            e.predict(u); e.correct(z, R, k)
    else:
        for i in range(run["m"]):
            e.step_raw(run, i)
    t1 = time.perf_counter(); e.flush(); e.sync(); t2 = time.perf_counter()
    print("%-5s: %.2f us per step issued by the host (%.2f us incl. the GPU draining)" % (mode, (t1 - t0) / len(steps) * 1e6, (t2 - t0) / len(steps) * 1e6), flush=True)
    e.close()
