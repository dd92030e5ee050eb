"""Where does a k_gather launch spend its time?  Needs a library built with -DEKF_GATHER_STAMPS=1 (column lane 0's view) or =2
(the chain wavefront's view): clock64() stamps of workgroup 0, returned through the Q slots.
    make -C ekf_slam_amd/csrc stamps
    EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_stamps.so  python scripts/probe_gather_phases.py [landmarks] [batch] columns
    EKF_LIB_PATH=$PWD/ekf_slam_amd/libekfslam_stamps2.so python scripts/probe_gather_phases.py [landmarks] [batch] chain"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
view = sys.argv[3] if len(sys.argv) > 3 else "columns"
names = {"columns": ["uniform loads issued + barrier 0", "column loads issued", "upatch staged by all", "pair operands + patches", "wait for the solve (barrier B)", "outputs"],
         "diag": ["loads issued + barrier 0", "own block arrived", "operands staged", "patch chain + flag", "small outputs + barrier B"],
         "chain": ["loads issued + barrier 0", "own operands arrived", "sincos", "predict entries", "H_s", "diag wait+GS+phi+inv", "publish + barrier B"]}[view]
NST = len(names) + 1
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
e = Engine(capacity=N, tile=128, batch=batch)
e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
rows = []
for i in range(3 * batch):
    e.predict([0.1, 3.0]); e.correct([10.0, 100.0], R, (i * 37) % N)
    q = e.get_Q3().reshape(-1)[:NST]        # stamps in shader clocks relative to kernel entry
    if i >= batch:
        rows.append((i % batch, q))

for lo, hi in ((0, 4), (batch // 2 - 2, batch // 2 + 2), (batch - 4, batch)):
    sel = np.array([q for (k, q) in rows if lo <= k < hi])
    dq = np.diff(sel, axis=1).mean(axis=0)
    print("npend %2d..%2d: " % (lo, hi - 1) + "  ".join("%s %5.0f" % (nm, v) for nm, v in zip(names, dq)) + "   total %5.0f clocks" % sel[:, -1].mean())
