#!/usr/bin/env python3
"""SURVEY.md 8d "CPU baseline beside it": both CPU restatements of the reference timed on this host's cores.

  (i)  literal-dense NumPy/OpenBLAS transcription (oracle/ekf_dense.py: what the reference's MATLAB/MKL executes, O(n^3) per
       update-step) at N = 20 (BASELINE.json configs[0]: the example world, known correspondence, 100 SLAM iterations) and at
       N = 1 000 (a few update-steps; one (I - K H) P is 1.6e10 flops there);
  (ii) structured C/OpenMP O(n^2) restatement (oracle/ekf_structured.c) at N = 20, 1 000 and 10 000.

Restatements, not MATLAB (none is available).  CPU only; prints one JSON object.
    python scripts/bench_cpu_restatements.py [--out profiles/round1_cpu_baselines.json]
"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EKF_NO_TORCH", "1")
from oracle import ekf_dense                                   # noqa: E402
from oracle.ekf_structured import StructuredEKF, available_cores   # noqa: E402
from ekf_slam_amd.world import make_run, SyntheticLandmark     # noqa: E402


def slam_run(engine, run):
    """Returns (seconds, update-steps): every observed row is either an append or a correction."""
    lm = SyntheticLandmark('SYNTHETIC')
    rows = 0
    t0 = time.perf_counter()
    for (u, scan) in run:
        engine.predict(u)
        engine.measure(scan, u, lm)
        rows += len(scan)
    dt = time.perf_counter() - t0
    return dt, rows - (len(engine.x) - 3) // 2


def lowrank_state(N, seed):
    rng = np.random.default_rng(seed)
    n = 3 + 2 * N
    x = np.concatenate([[0.0, 0.0, 0.0], rng.uniform(-50, 50, 2 * N)])
    U = rng.normal(0, 0.01, (n, 8))
    P = U @ U.T
    P[np.arange(n), np.arange(n)] += rng.uniform(0.01, 0.1, n)
    return x, P


def steps_for(N, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(count):
        z = [float(rng.uniform(5, 60)), float(rng.uniform(1, 359))]
        out.append(([0.1, 3.0], z, np.diag([z[0] * .01, z[1] * 5.0]), (t * 37) % N))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--gpu", action="store_true", help="also time libekfslam on the configs[0] run (needs an MI355X)")
    args = ap.parse_args()
    res = {"note": "CPU restatements of the reference's path timed on this host (SURVEY.md 8d); restatements, not MATLAB",
           "cores": available_cores()}

    # configs[0]: 20 landmarks, known correspondence, every landmark sighted every iteration
    _, run = make_run(20, 20260101, 100, policy="all")
    e = ekf_dense.EKF_SLAM()
    dt, upd = slam_run(e, run)
    res["config0_literal_dense_numpy"] = {"landmarks": 20, "slam_iterations": len(run), "update_steps": upd, "seconds": dt,
                                          "update_steps_per_s": upd / dt, "slam_iterations_per_s": len(run) / dt}
    s = StructuredEKF(32, "known")
    dt, upd = slam_run(s, run)
    res["config0_structured_c"] = {"landmarks": 20, "slam_iterations": len(run), "update_steps": upd, "seconds": dt,
                                   "update_steps_per_s": upd / dt, "slam_iterations_per_s": len(run) / dt}

    if args.gpu:
        os.environ.pop("EKF_NO_TORCH", None)
        from ekf_slam_amd.slam import EKF_SLAM as GpuEkfSlam
        for batch in (1, 8):
            g = GpuEkfSlam(capacity=32, batch=batch)
            slam_run(g, run[:10])                      # warm-up on a throw-away filter
            g = GpuEkfSlam(capacity=32, batch=batch)
            dt, upd = slam_run(g, run)
            g._e.sync()
            res["config0_libekfslam_gpu_batch%d" % batch] = {"landmarks": 20, "slam_iterations": len(run), "update_steps": upd,
                                                            "seconds": dt, "update_steps_per_s": upd / dt,
                                                            "slam_iterations_per_s": len(run) / dt,
                                                            "note": "latency-bound: one ~7 us kernel chain per update-step whatever the map size"}

    # 1 000 landmarks: literal-dense (n = 2003: every eye(n), zeros(n) and n x n x n product executed as written)
    N = 1000
    x, P = lowrank_state(N, 3)
    e = ekf_dense.EKF_SLAM()
    e.x, e.P, e.s = x.copy(), P.copy(), list(range(1, N + 1))
    st = steps_for(N, 6, 4)
    t0 = time.perf_counter()
    for (u, z, R, k) in st:
        e.predict(u)
        e._correct(z, R, k + 1)
    dt = time.perf_counter() - t0
    res["literal_dense_numpy_1k"] = {"landmarks": N, "update_steps": len(st), "seconds": dt, "update_steps_per_s": len(st) / dt,
                                     "step": "1 predict (2 dense n^3 products) + 1 correction ((I - K H) P dense)"}
    for N, count in ((1000, 400), (10000, 40)):
        x, P = lowrank_state(N, 3)
        s = StructuredEKF(N, "known")
        s.set_state(x, P, np.arange(1, N + 1.0))
        st = steps_for(N, count, 4)
        t0 = time.perf_counter()
        for (u, z, R, k) in st:
            s.predict(u)
            s.correct(z, R, k + 1)
        dt = time.perf_counter() - t0
        res["structured_c_%dk" % (N // 1000)] = {"landmarks": N, "update_steps": len(st), "seconds": dt,
                                                "update_steps_per_s": len(st) / dt, "step": "1 predict + 1 correction"}
        del s
    # literal-dense at 10 k: not run (one (I - K H) P is 1.6e13 flops); n^3 extrapolation from the 1 k timing
    r = res["literal_dense_numpy_1k"]
    res["literal_dense_numpy_10k_extrapolated"] = {"landmarks": 10000, "update_steps_per_s": r["update_steps_per_s"] / (20003 / 2003) ** 3,
                                                   "basis": "n^3 scaling of the measured 1 k figure (not run)"}
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt + "\n")


if __name__ == "__main__":
    main()
