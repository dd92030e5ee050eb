"""Cost of one prefetch (ekf_prefetch_rows: base row-panels of `batch` landmarks + exchange) and of the corrections that use
it, on a shard group driven by one process (ekf_exchange_local) on one GPU.  python scripts/probe_prefetch.py [N] [world] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd.sharding import ShardGroup
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
g = ShardGroup(world, capacity=N, tile=128, batch=batch)
for e in g.shards:
    e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
def sync():
    for e in g.shards: e.sync()
k = 0
for rep in range(4):
    idx = [((k + i) * 37) % N for i in range(batch)]
    sync(); t0 = time.perf_counter()
    g.prefetch_rows(sorted(set(idx)))
    sync(); t1 = time.perf_counter()
    for i in idx:
        g.predict([0.1, 3.0]); g.correct_local([10.0, 100.0], R, i)
    sync(); t2 = time.perf_counter()
    k += batch
    print("rep %d: prefetch of %d rows %.1f us; %d predict+correct %.1f us (%.1f us each, flush included)" %
          (rep, batch, (t1 - t0) * 1e6, batch, (t2 - t1) * 1e6, (t2 - t1) * 1e6 / batch), flush=True)
g.close()
