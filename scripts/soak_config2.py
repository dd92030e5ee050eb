#!/usr/bin/env python3
"""Long run of configs[1]'s workload: the device-resident loop must stay bit-identical to the host-decided mode however long the run; the
distance of the GPU state from the CPU oracle is logged every `every` iterations with the place of the largest difference.
Usage: soak_config2.py [iterations] [landmarks] [every]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
from ekf_slam_amd.world import SyntheticLandmark, make_run
from oracle.ekf_structured import StructuredEKF
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 250
_, run = make_run(N, 20260102, 2 + iters, policy="nearest", m=8)
dev = EKF_SLAM_UC(capacity=N, batch=8)
host = EKF_SLAM_UC(capacity=N, batch=1, device_assoc=0)          # host-decided AND immediate: two independent switches away from `dev`
ref = StructuredEKF(N, "uc")
ld, lh, lr = Landmark('SYNTHETIC'), Landmark('SYNTHETIC'), SyntheticLandmark()
for t, (u, scan) in enumerate(run):
    for e, l in ((dev, ld), (host, lh), (ref, lr)):
        e.predict(u); e.measure(scan, u, l)
    if t >= 2 and ((t - 2) % every == 0 or t == len(run) - 1):
        xd, xh, xr = dev.x, host.x, ref.x
        dx = np.abs(xd - xr)
        i = int(np.argmax(dx))
        obs = np.asarray(dev.observed)
        Prr = dev._e.get_P_block(0, 0, 3, 3); Prr_o = ref.P[:3, :3]
        asym_g, asym_o = float(np.abs(Prr - Prr.T).max()), float(np.abs(Prr_o - Prr_o.T).max())
        print("iteration %5d: dev == host bitwise %s | max|dx| %.3e at state %d (landmark %d) value %.4f | rel %.3e | asym(Prr) gpu %.2e oracle %.2e | max|dPrr| %.2e | bearing obs %s" %
              (t, bool(np.array_equal(xd, xh)), dx[i], i, (i - 3) // 2 if i >= 3 else -1, xr[i], dx.max() / np.abs(xr).max(), asym_g, asym_o, float(np.abs(Prr - Prr_o).max()),
               np.round(obs[:2, 1], 1).tolist()), flush=True)
Pd = dev.P
print("P: dev == host bitwise %s, rel err P vs oracle %.3e" % (bool(np.array_equal(Pd, host.P)), float(np.abs(Pd - ref.P).max() / np.abs(ref.P).max())))
