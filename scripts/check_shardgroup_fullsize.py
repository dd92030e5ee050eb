"""10k landmarks, 8 shards in one process (ekf_exchange_local) on one GPU, deferred batch 32 with prefetched row-panels, against the
unsharded engine: x must match bit for bit, the P digest to summation order.  python scripts/check_shardgroup_fullsize.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd import Engine
from ekf_slam_amd.sharding import ShardGroup
N, world, batch = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 32
n = 3 + 2 * N
rng = np.random.default_rng(1)
x = np.concatenate([[0, 0, 0], rng.uniform(-100, 100, 2 * N)])
d = rng.uniform(0.01, 0.1, n); U = rng.normal(0, 0.01, (n, 8)); s = np.arange(1, N + 1.0)
one = Engine(capacity=N, tile=128, batch=batch)
one.load_lowrank_state(x, s, d, U)
g = ShardGroup(world, capacity=N, tile=128, batch=batch)
for e in g.shards:
    e.load_lowrank_state(x, s, d, U)
R = np.diag([0.2, 50.0])
k = 0
for rep in range(3):
    idx = [((k + i) * 37) % N for i in range(batch)]
    g.prefetch_rows(sorted(set(idx)))
    for i in idx:
        z = [10.0 + (i % 7), 100.0 + (i % 11)]
        one.predict([0.1, 3.0]); one.correct(z, R, i)
        g.predict([0.1, 3.0]); g.correct_local(z, R, i)
    k += batch
# three more steps through the per-step exchange (pending pairs patched in k_rowpanel)
for i in (5, 4242, 9999):
    z = [12.0, 77.0]
    one.predict([0.1, 3.0]); one.correct(z, R, i)
    g.predict([0.1, 3.0]); g.correct(z, R, i)
xs = g.get_x()
assert np.array_equal(xs, one.get_x()), "x differs"
dg = sum(np.asarray(e.digest()) for e in g.shards)
d1 = one.digest()
print("world %d: x bit-identical; digest sharded %s vs unsharded %s" % (world, dg, d1))
assert np.allclose(dg, d1, rtol=1e-12)
print("ok")
