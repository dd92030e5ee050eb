"""GPU: what F32 tile storage costs in accuracy at configs[4]'s real LENGTH (BASELINE.json configs[4]: 40 k -> 50 k landmarks =
10 000 update-steps, every one rounding each landmark-block entry it touches to float once per pass over P; the reference's
arithmetic is F64 throughout, EKF_SLAM.m:141-145).  The dozen-step checks of tests/test_f32_storage_gpu.py say nothing about
that, so this runs the F32-tile engine and the F64-tile engine (same GPU, same inputs: predict + correction, a streaming
append every 10th step) side by side for 2 000 update-steps on a 2 000-landmark map and records the max-norm relative error of
x and P every 250 steps.

What it shows (profiles/round3_f32_drift.json): the error of P does NOT grow like a random walk of roundings (sqrt of the passes)
but LINEARLY in the update-steps, at ~1.6e-9 per step, whether every step rewrites P or only every 12th.  It sits on large,
rarely touched entries (the diagonal blocks of appended, not yet re-observed landmarks: ~17 against the bulk's 0.1): each
correction lowers them by far less than half a float ulp, so rounding the tile back to float returns the old value and the
decrements are lost one after the other (stagnation) -- a bias, not noise.  x, computed in F64 from F64 gains, stays 2 orders
of magnitude better.  The bound asserted here, and stated for configs[4] in DESIGN.md section 5:

    after K update-steps   rel err(P) <= 6e-8 + 3e-9 K      rel err(x) <= 6e-8 + 2e-10 K      (max-norm, against F64 tiles)

i.e. F32 tiles hold BASELINE.json's 1e-6 for ~300 update-steps, 6e-6 at 2 000 and 3e-5 over configs[4]'s full 10 000 -- the
F64 tile store (25.6 GB at 40 k landmarks, well inside one MI355X's 288 GB) is the mode that meets 1e-6 at that length."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS32 = 6e-8
STEPS, EVERY, N0 = 2000, 250, 2000


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def bound_P(k):
    return EPS32 + 3e-9 * k


def bound_x(k):
    return EPS32 + 2e-10 * k


@pytest.mark.parametrize("batch", [1, 12])
def test_f32_tiles_drift_over_two_thousand_update_steps(batch):
    import bench
    from ekf_slam_amd import Engine
    from ekf_slam_amd.world import World
    cap = N0 + STEPS // 10 + 1
    w = World(cap, 20260101 + 5)
    rng = np.random.default_rng(78)
    n0 = 3 + 2 * N0
    x = np.concatenate([[0.0, 0.0, 0.0], w.landmarks[:N0].reshape(-1)])
    d = rng.uniform(0.01, 0.1, n0)
    U = rng.normal(0.0, 0.01, (n0, 8))
    s = np.arange(1, N0 + 1.0)
    e64 = Engine(mode="known", capacity=cap, storage="f64", batch=batch)
    e32 = Engine(mode="known", capacity=cap, storage="f32", batch=batch)
    for e in (e64, e32):
        e.load_lowrank_state(x, s, d, U)
    Rc = [.01, 5.0]
    log = []
    worst_x = worst_P = 0.0
    for t in range(STEPS):
        u = w.step()
        k = (t * 37) % N0
        (_, r, b), = w.observe([k])
        R = np.diag([r * Rc[0], b * Rc[1]])
        for e in (e64, e32):
            e.predict(u)
            if t % 10 == 9:
                e.append(u, R, w.landmarks[e.N], e.N + 1)
            e.correct([r, b], R, k)
        if (t + 1) % EVERY == 0:
            ex, eP = rel_err(e32.get_x(), e64.get_x()), rel_err(e32.get_P(), e64.get_P())      # get_P flushes both
            passes = (t + 1) if batch == 1 else (t + 1) / batch + (t + 1) // EVERY              # + the flush each read forces
            log.append({"update_steps": t + 1, "passes_over_P": passes, "rel_err_x": ex, "rel_err_P": eP,
                        "bound_x": bound_x(t + 1), "bound_P": bound_P(t + 1)})
            worst_x, worst_P = max(worst_x, ex), max(worst_P, eP)
    assert e32.N == e64.N == N0 + STEPS // 10
    tr32, tr64 = e32.digest()[0], e64.digest()[0]
    print("f32 drift, batch %d: %s" % (batch, json.dumps(log)))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f32_drift_batch%d.json" % batch), "w") as fh:
            json.dump({"landmarks": [N0, e32.N], "batch": batch, "log": log}, fh)
    for rec in log:
        assert rec["rel_err_x"] <= rec["bound_x"] and rec["rel_err_P"] <= rec["bound_P"], rec
    assert abs(tr32 - tr64) / abs(tr64) <= bound_P(STEPS)
    assert log[-1]["rel_err_P"] > 1e-6          # the point of the statement: 1e-6 does NOT hold at this length with F32 tiles
    e64.close(); e32.close()
