"""GPU: what F32 tile storage costs in accuracy at configs[4]'s real LENGTH (BASELINE.json configs[4]: 40 k -> 50 k landmarks =
10 000 update-steps, every one rounding each landmark-block entry it touches to float once per pass over P; the reference's
arithmetic is F64 throughout, EKF_SLAM.m:141-145).  The dozen-step checks of tests/test_f32_storage_gpu.py say nothing about
that, so this runs the F32-tile engine and the F64-tile engine (same GPU, same inputs: predict + correction, a streaming
append every 10th step) side by side for 2 000 update-steps on a 2 000-landmark map and records the max-norm relative error of
x and P every 250 steps (profiles/round3_f32_drift.json).

ROUND TO NEAREST (cfg.f32_rounding = 1, what rounds 1-2 did): the error of P grows LINEARLY in the update-steps, ~1.6e-9 per step,
whether every step rewrites P or only every 12th.  It sits on large, rarely touched entries (the diagonal blocks of appended, not
yet re-observed landmarks: ~17 against the bulk's 0.1): each correction lowers them by far less than half a float ulp, so rounding
the tile back to float returns the old value and the decrements are lost one after the other (stagnation) -- a bias, not noise.
Asserted: rel err(P) <= 6e-8 + 3e-9 K after K update-steps -- 6e-6 at 2 000, 3e-5 over configs[4]'s 10 000.

STOCHASTIC ROUNDING (the default since round 3; kernels.hip::round_tile: up with probability (v - lo) / (hi - lo), seeded by a hash of
(row, column, number of the pass): unbiased, deterministic, the same bits from every kernel instance and shard layout): the lost
decrements now accumulate on average, the error is a random walk over the PASSES -- 1.2-1.5e-6 after 2 000 passes (batch 1),
0.6-1.1e-6 after 175 (batch 12; the max-norm is carried by a handful of large entries, so one seed's realisation differs from another's
by up to 2x) -- so the deferred mode is also the more accurate one.  Asserted, and stated for configs[4] in DESIGN.md 5:

    rel err(P) <= 6e-8 * (2 + 2.5 sqrt(passes over P)),      rel err(x) <= 6e-8 + 2e-10 K        (max-norm, against F64 tiles)

i.e. with batch 12 F32 tiles stay near BASELINE.json's 1e-6 for ~2 000 update-steps (bound 2.1e-6) and at ~2.5e-6 over configs[4]'s
10 000 (834 passes; bound 4.5e-6) where round-to-nearest reaches 1.6e-5; x, computed in F64 from F64 gains, stays two orders of magnitude better either way."""
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


def bound_P(k, passes, rounding):
    return EPS32 + 3e-9 * k if rounding == "nearest" else EPS32 * (2.0 + 2.5 * np.sqrt(passes))


def bound_x(k):
    return EPS32 + 2e-10 * k


@pytest.mark.parametrize("rounding", ["stochastic", "nearest"])
@pytest.mark.parametrize("batch", [1, 12])
def test_f32_tiles_drift_over_two_thousand_update_steps(batch, rounding):
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
    e32 = Engine(mode="known", capacity=cap, storage="f32", batch=batch, f32_rounding=1 if rounding == "nearest" else 0)
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
                        "bound_x": bound_x(t + 1), "bound_P": bound_P(t + 1, passes, rounding)})
            worst_x, worst_P = max(worst_x, ex), max(worst_P, eP)
    assert e32.N == e64.N == N0 + STEPS // 10
    tr32, tr64 = e32.digest()[0], e64.digest()[0]
    print("f32 drift, batch %d, %s: %s" % (batch, rounding, json.dumps(log)))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f32_drift_batch%d_%s.json" % (batch, rounding)), "w") as fh:
            json.dump({"landmarks": [N0, e32.N], "batch": batch, "rounding": rounding, "log": log}, fh)
    for rec in log:
        assert rec["rel_err_x"] <= rec["bound_x"] and rec["rel_err_P"] <= rec["bound_P"], rec
    assert abs(tr32 - tr64) / abs(tr64) <= log[-1]["bound_P"]
    if rounding == "nearest":
        assert log[-1]["rel_err_P"] > 1e-6      # the point of the statement: 1e-6 does NOT hold at this length with round-to-nearest
    elif batch == 12:
        assert log[-1]["rel_err_P"] < 1.5e-6    # deferred, stochastic: half of round-to-nearest's 2.8e-6 or better, and no longer growing linearly
    e64.close(); e32.close()
