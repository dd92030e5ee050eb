"""Where does the F32-tile engine differ most from the F64-tile one?  (diagnostic for tests/test_f32_drift_gpu.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine
from ekf_slam_amd.world import World
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
app = int(sys.argv[3]) if len(sys.argv) > 3 else 10
N0 = 2000
cap = N0 + steps // app + 1
w = World(cap, 20260101 + 5)
rng = np.random.default_rng(78)
n0 = 3 + 2 * N0
x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks[:N0].reshape(-1)])
d = rng.uniform(0.01, 0.1, n0); U = rng.normal(0.0, 0.01, (n0, 8)); s = np.arange(1, N0 + 1.0)
e64 = Engine(mode="known", capacity=cap, storage="f64", batch=batch)
e32 = Engine(mode="known", capacity=cap, storage="f32", batch=batch)
for e in (e64, e32):
    e.load_lowrank_state(x, s, d, U)
for t in range(steps):
    u = w.step(); k = (t * 37) % N0
    (_, r, b), = w.observe([k]); R = np.diag([r * .01, b * 5.0])
    for e in (e64, e32):
        e.predict(u)
        if app and t % app == app - 1:
            e.append(u, R, w.landmarks[e.N], e.N + 1)
        e.correct([r, b], R, k)
    if (t + 1) % 50 == 0 or t + 1 == steps:
        A, B = e32.get_P(), e64.get_P()
        E = np.abs(A - B)
        i, j = np.unravel_index(np.argmax(E), E.shape)
        rel_local = E / np.maximum(np.abs(B), 1e-300)
        print("step %4d  max|dP| %.3e at (%d,%d) [first appended row %d]  P64 there %.6e  max|P| %.4e  -> rel %.3e ; median local rel err %.2e; err excluding appended rows %.3e"
              % (t + 1, E[i, j], i, j, n0, B[i, j], np.abs(B).max(), E[i, j] / np.abs(B).max(), np.median(rel_local[np.abs(B) > 1e-9]),
                 E[:n0, :n0].max() / np.abs(B).max()), flush=True)
