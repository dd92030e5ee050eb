"""Long-run check: deferred (batch 32) vs immediate over thousands of steps with interleaved appends and reads must stay
bit-identical; a 10k-landmark deferred run must stay finite.  python scripts/soak.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine

def state(N, seed):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.3, -0.2, 40.0], rng.uniform(-50, 50, 2 * N)])
    d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.02, (n, 6)); s = np.arange(1, N + 1.0)
    return x, s, d, U

N = 1500
x, s, d, U = state(N, 5)
imm = Engine(capacity=N + 200, tile=128, batch=1)
dfr = Engine(capacity=N + 200, tile=128, batch=32)
for e in (imm, dfr):
    e.load_lowrank_state(x, s, d, U)
rng = np.random.default_rng(9)
t0 = time.perf_counter()
for step in range(4000):
    u = [0.05, float(rng.uniform(-5, 5))]
    for e in (imm, dfr): e.predict(u)
    idx = int(rng.integers(0, imm.N))
    z = [float(rng.uniform(1, 60)), float(rng.uniform(1, 359))]
    R = np.diag([z[0] * .01, z[1] * 5.0])
    for e in (imm, dfr): e.correct(z, R, idx)
    if step % 97 == 13 and imm.N < N + 200:
        pos = rng.uniform(-50, 50, 2)
        for e in (imm, dfr): e.append(u, R, pos, imm.N + 1)
    if step % 501 == 500:
        assert np.array_equal(imm.get_x(), dfr.get_x()), step
Pi, Pd = imm.get_P(), dfr.get_P()
assert Pi.tobytes() == Pd.tobytes() and np.isfinite(Pi).all()
print("soak 1: %d landmarks, 4000 steps with appends: deferred == immediate bit for bit (%.1f s)" % (imm.N, time.perf_counter() - t0), flush=True)
imm.close(); dfr.close()

N = 10000
x, s, d, U = state(N, 6)
e = Engine(capacity=N, tile=128, batch=32)
e.load_lowrank_state(x, s, d, U)
t0 = time.perf_counter()
for step in range(9600):
    e.predict([0.05, float(rng.uniform(-5, 5))])
    z = [float(rng.uniform(1, 60)), float(rng.uniform(1, 359))]
    e.correct(z, np.diag([z[0] * .01, z[1] * 5.0]), int(rng.integers(0, N)))
e.flush(); e.sync()
dt = time.perf_counter() - t0
dg = e.digest()
assert np.isfinite(dg).all() and np.isfinite(e.get_x()).all()
print("soak 2: 10000 landmarks, 9600 random steps: finite, digest %s, %.0f update-steps/s incl. host RNG" % (dg, 9600 / dt), flush=True)
