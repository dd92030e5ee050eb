"""CPU: bench.py fails loudly where it cannot run -- there is no CPU fallback to measure instead.  On a machine without a HIP
device the single-process run stops at ekf_create (EKF_ERR_NO_DEVICE), and `bench.py --gpus 2` (which spawns its own ranks as
child processes before touching HIP or torch) relays the failure: non-zero exit, no JSON line, no hang."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT a HIP device")
def test_single_process_bench_refuses_to_run_without_a_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--landmarks", "64", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "0 GPU(s) visible" in r.stderr or "no HIP device" in r.stderr or "EkfError" in r.stderr


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT a HIP device")
def test_self_spawned_ranks_fail_loudly_without_a_device():
    env = dict(os.environ, EKF_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--landmarks", "64", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank run failed" in r.stderr


def test_launcher_and_gpus_must_agree():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       cwd=ROOT, env=env)
    assert r.returncode != 0 and "launcher started 3 rank(s)" in r.stderr


def test_committed_pmc_summary_covers_the_default_bench_shapes():
    """`roofline.traffic` of the default `python bench.py` comes from the newest committed PMC summary for exactly the launch shape
    that ran (bench.load_committed_pmc); the default run launches the one-pair pass and the 20-pair flush at 10 000 landmarks,
    tile 128 -- both must be in profiles/, name the kernel the launcher reports, and hold traffic within a few per cent of the
    algorithmic bytes (a stale or missing summary would silently turn `traffic` into null)."""
    sys.path.insert(0, ROOT)
    import bench
    n = 3 + 2 * 10000
    b_alg = 8 * n * (n + 1)
    for pairs, kernel in ((1, "k_downdate_w"), (20, "k_flush_mfma")):
        rec = bench.load_committed_pmc(10000, 128, pairs)
        assert rec is not None, "no committed PMC summary for %d pair(s) per launch" % pairs
        assert rec["kernel"] and kernel in rec["kernel"]
        assert 1.0 <= rec["hbm_bytes_per_launch"] / b_alg < 1.10, rec
    assert bench.load_committed_pmc(10000, 128, 7) is None          # a shape nobody measured: null, not a neighbour's figure
