#!/usr/bin/env python3
"""Long UC run in which landmarks are DISCOVERED along the way (no warm-up sweep: every scan sights the 8 landmarks nearest to the
true pose, new ones are appended when first seen): device-resident loop (batch 8) beside host-decided immediate, waited and the oracle.
Usage: soak_uc_discovery.py [iterations] [world landmarks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ekf_slam_amd.slam import EKF_SLAM_UC, Landmark
from ekf_slam_amd.world import SyntheticLandmark, World
from oracle.ekf_structured import StructuredEKF
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 400
w = World(NW, 20260119, u_nominal=(0.35, 2.0))
run = []
for t in range(iters):
    u = w.step()
    ids = [w.nearest(1)[0]] if t == 0 else w.nearest(8)
    run.append((u, w.observe(ids)))
eng = {"dev_b8": EKF_SLAM_UC(capacity=NW, batch=8), "host_b1": EKF_SLAM_UC(capacity=NW, batch=1, device_assoc=0),
       "waited_b4": EKF_SLAM_UC(capacity=NW, batch=4, device_assoc=1), "verified_async16": EKF_SLAM_UC(capacity=NW, batch=16, device_assoc=2, async_flush=True)}
lms = {k: Landmark('SYNTHETIC') for k in eng}
ref, lr = StructuredEKF(NW, "uc"), SyntheticLandmark()
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for t, (u, scan) in enumerate(run):
    for k, e in eng.items():
        e.predict(u); e.measure(scan, u, lms[k])
    ref.predict(u); ref.measure(scan, u, lr)
    if (t + 1) % (iters // 8) == 0:
        xd = eng["dev_b8"].x
        same = {k: bool(np.array_equal(e.x, xd)) for k, e in eng.items() if k != "dev_b8"}
        print("iteration %5d: N %3d (oracle %3d) | %s | rel err x vs oracle %.3e" % (t + 1, eng["dev_b8"]._e.N, ref.N, same, rel(xd, ref.x)), flush=True)
Pd = eng["dev_b8"].P
print("P equal bitwise: %s | rel err P vs oracle %.3e | s equal %s" % ({k: bool(np.array_equal(e.P, Pd)) for k, e in eng.items() if k != "dev_b8"},
      rel(Pd, ref.P), bool(np.array_equal(eng["dev_b8"].s, ref.s))))
