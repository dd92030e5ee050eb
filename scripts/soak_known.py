#!/usr/bin/env python3
"""Long run in known-correspondence mode (predict + correction per step, an append every `app` steps) on several engines side by side:
deferred / immediate / asynchronous / sharded must stay bit-identical to each other, and the distance from the CPU oracle is logged.
Usage: soak_known.py [steps] [landmarks] [app]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ekf_slam_amd import Engine
from ekf_slam_amd.sharding import ShardGroup
from ekf_slam_amd.world import World
from oracle.ekf_structured import StructuredEKF
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
N0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
app = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cap = N0 + steps // app + 2
w = World(cap, 20260117)
rng = np.random.default_rng(5)
n0 = 3 + 2 * N0
x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks[:N0].reshape(-1)])
d = rng.uniform(0.01, 0.1, n0); U = rng.normal(0.0, 0.01, (n0, 8)); s = np.arange(1, N0 + 1.0)
eng = {"immediate": Engine(capacity=cap, batch=1), "deferred8": Engine(capacity=cap, batch=8),
       "async16": Engine(capacity=cap, batch=16, async_flush=True), "shards4_b4": ShardGroup(4, capacity=cap, batch=4)}
ref = StructuredEKF(cap, "known")
P0 = np.diag(d) + U @ U.T
for e in eng.values():
    e.load_lowrank_state(x, s, d, U)
ref.set_state(x, P0, s)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for t in range(steps):
    u = w.step()
    k = (t * 37) % N0
    (_, r, b), = w.observe([k])
    R = np.diag([r * .01, b * 5.0])
    for e in list(eng.values()) + [ref]:
        e.predict(u)
    if t % app == app - 1:
        N = eng["immediate"].N
        for e in list(eng.values()) + [ref]:
            e.append(u, R, w.landmarks[N], N + 1)
    for e in eng.values():
        e.correct([r, b], R, k)
    ref.correct([r, b], R, k + 1)
    if (t + 1) % (steps // 8) == 0:
        xi = eng["immediate"].get_x()
        same = {n: bool(np.array_equal(e.get_x(), xi)) for n, e in eng.items() if n != "immediate"}
        print("step %5d: %s | rel err x vs oracle %.3e | heading %.3f" % (t + 1, same, rel(xi, ref.x), xi[2]), flush=True)
Pi = eng["immediate"].get_P()
print("P: deferred8 == immediate %s, async16 %s, shards4 %s | rel err P vs oracle %.3e" % (
    bool(np.array_equal(eng["deferred8"].get_P(), Pi)), bool(np.array_equal(eng["async16"].get_P(), Pi)),
    bool(np.array_equal(eng["shards4_b4"].get_P(), Pi)), rel(Pi, ref.P)))
