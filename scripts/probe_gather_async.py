"""k_gather under a concurrent flush (cfg.async_flush): phase stamps of the last gather of a burst + mean launch duration.
Needs the -DEKF_GATHER_STAMPS library: make -C ekf_slam_amd/csrc stamps && EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_stamps.so python scripts/probe_gather_async.py [landmarks] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine, _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
R = np.diag([0.2, 50.0])
names = ["uniform loads issued", "all loads issued", "staged", "sincos+patch", "predict+Hs+atan2+solve", "outputs"]
for asy in (False, True):
    e = Engine(capacity=N, tile=128, batch=batch, async_flush=asy)
    e.load_lowrank_state(x, s, d, U)
    k = 0
    def burst(m):
        global k
        for _ in range(m):
            e.predict([0.1, 3.0]); e.correct([10.0, 100.0], R, (k * 37) % N); k += 1
    burst(4 * batch); e.flush(); e.sync()
    for extra in (batch // 4, batch // 2):
        e.timing_enable(L.EKF_KERNEL_GATHER, True)
        burst(2 * batch + extra)            # the last `extra` gathers run while the second flush is in flight (async)
        q = e.get_Q3().reshape(-1)[:7]
        ng, mg = e.timing_read(L.EKF_KERNEL_GATHER)
        e.timing_enable(L.EKF_KERNEL_GATHER, False)
        dq = np.diff(np.concatenate([[0.0], q]))[1:]
        print("async" if asy else "sync ", "last gather (%2d into the batch): " % extra + "  ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, dq)),
              "  total %5.0f clocks;  mean launch %.1f us over %d" % (q[-1], mg / ng * 1e3, ng), flush=True)
        e.flush(); e.sync()
    e.close()
