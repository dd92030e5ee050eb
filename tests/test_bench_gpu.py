"""GPU: the default single-GPU bench line as the driver runs it (a fresh child process, never an exec from this one): ONE JSON line that carries the headline
with `roofline` and -- measured in the same invocation, each in a child process of its own -- BASELINE.json's other single-GPU configurations under
`other_configs`: configs[1] (1 000 landmarks, unknown correspondence, the device-resident measure loop) and the whole configs[4] workload on one GPU
(40 000 -> 50 000 landmarks, float tiles, the pass in F32 arithmetic and in split arithmetic).  Short legs here (the figures to quote come from the default run); no CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_carries_the_other_single_gpu_configurations():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "64", "--warmup", "16", "--deferred-steps", "80", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    assert b["n_gpus"] == 1 and b["unit"] == "update-steps/s" and b["value"] > 0 and b["config"]["state_finite"]
    assert b["roofline"]["bound"] == "hbm" and 0.5 < b["roofline"]["frac"] < 1.0 and "k_downdate_w" in b["roofline"]["kernel"]
    # a timed region under 0.2 s: the block is repeated, the median quoted, and the line says so
    assert "note" in b["config"] and b["repeats"] >= 3 and b["repeats"] % 2 == 1
    assert b["ms_per_step_min"] <= b["ms_per_step"] <= b["ms_per_step_max"]
    assert b["deferred"]["deferred_batch"] == 20 and b["deferred_b32"]["deferred_batch"] == 32
    oc = b["other_configs"]
    c1, c4 = oc["configs[1]"], oc["configs[4] on one GPU"]
    assert "error" not in c1 and "error" not in c4, oc
    assert c1["value"] > 5e4 and c1["config"]["state_finite"] and "device-resident loop" in c1["config"]["device_association"]
    assert c4["value"] > 1e3 and c4["config"]["state_finite"] and c4["roofline"]["kernel"].startswith("k_flush_strip32<")
    r4 = c4["roofline"]
    assert r4["bound"] in ("hbm", "mfma") and r4["frac"] == max(r4["roofs"]["hbm"]["frac"], r4["roofs"]["mfma"]["frac"])
    assert r4["pairs_per_launch"] == 64 and 0.3 < r4["roofs"]["hbm"]["frac"] < 1.0 and 0.4 < r4["roofs"]["mfma"]["frac"] < 1.0
    assert c4["steps"] == 9936 and "50000 landmarks" in c4["config"]["workload"]
    # the same workload with the pass in split arithmetic: bound by HBM, and faster than the F32-arithmetic leg
    c4s = oc["configs[4] on one GPU, split arithmetic"]
    assert "error" not in c4s, oc
    r4s = c4s["roofline"]
    assert c4s["config"]["state_finite"] and r4s["kernel"] == "k_flush_split3<2>" and r4s["pairs_per_launch"] == 64
    assert r4s["bound"] == "hbm" and r4s["roofs"]["mfma"]["executed_over_algorithmic_flops"] == 6.0 and r4s["roofs"]["mfma"]["peak"] == 2500.0
    assert c4s["steps"] == 9936 and c4s["value"] > 1.1 * c4["value"]
    # both once more with the pass beside the next batch's appends and corrections (cfg.async_flush): present, finite, not slower than 0.9 x
    for key, syn in (("configs[4] on one GPU, asynchronous pass", c4), ("configs[4] on one GPU, split arithmetic, asynchronous pass", c4s)):
        ca = oc[key]
        assert "error" not in ca, oc
        assert ca["config"]["async_flush"] and ca["config"]["state_finite"] and ca["steps"] == 9936 and ca["value"] > 0.9 * syn["value"]
