"""GPU: the multi-rank flow the scaling run will use, under the driver's eyes although only ONE GPU is available to the tests.

`python bench.py --gpus N` is started as a FRESH CHILD process (never a re-exec of this pytest process, which has initialised the
GPU); without a launcher in its environment that child spawns its own ranks (a `torch.distributed.run` grandchild, started
before the child touches HIP or torch).  With EKF_BENCH_BACKEND=gloo every rank runs on device 0 and the per-step exchange is
staged through the host -- the same shard plan (tile (I,J) on rank (I+J) mod N), the same kernels (k_rowpanel, k_gather<sharded>,
the pass over the owned tiles), the same bench code path as the RCCL run.  1, 2 and 4 ranks must end with the same state digest
(to summation order), report their world size and carry the `deferred_lookahead` leg; one more child runs ONE rank on the nccl
backend with the sharded code path forced (--force-sharded: cfg.force_sharded) and must report the library's own RCCL transport."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--landmarks", "1500", "--steps", "64", "--warmup", "16", "--batch", "8", "--deferred-steps", "64", "--no-cpu-baseline", "--no-other-configs"]


def _bench(n, extra_env=None, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + ARGS + list(extra_args), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "bench.py --gpus %d failed (rc %d):\n%s" % (n, r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_one_two_and_four_ranks_end_with_the_same_state():
    ref = None
    for n in (1, 2, 4):
        b = _bench(n, {"EKF_BENCH_BACKEND": "gloo"})
        assert b["n_gpus"] == n and b["steps"] == 64 and b["scaling"] == "strong"
        assert b["config"]["state_finite"] and b["deferred"]["state_finite"]
        dg, dd = np.array(b["config"]["state_digest"]), np.array(b["deferred"]["state_digest"])
        if n == 1:
            ref = (dg, dd)
            assert b["config"]["transport"] == "none" and "deferred_lookahead" not in b
            assert b["roofline"]["pairs_per_launch"] == 1 and b["deferred"]["roofline"]["pairs_per_launch"] == 8
        else:
            assert b["config"]["transport"].startswith("torch.distributed") and b["config"]["backend"] == "gloo"
            look = b.get("deferred_lookahead")
            assert look and look["state_finite"] and look["value"] > 0       # one exchange per batch: the leg ran
            np.testing.assert_allclose(np.array(look["state_digest"]), ref[1], rtol=1e-10)
        np.testing.assert_allclose(dg, ref[0], rtol=1e-10, err_msg="%d ranks: state digest differs from the single-process run" % n)
        np.testing.assert_allclose(dd, ref[1], rtol=1e-10, err_msg="%d ranks: deferred leg's digest differs" % n)


def test_one_rank_on_the_rccl_transport():
    plain = _bench(1)
    b = _bench(1, extra_args=["--force-sharded"])
    assert b["n_gpus"] == 1 and b["config"]["transport"] == "rccl-native" and b["config"]["backend"] == "nccl"
    np.testing.assert_allclose(np.array(b["config"]["state_digest"]), np.array(plain["config"]["state_digest"]), rtol=1e-12)
    assert b.get("deferred_lookahead") and b["deferred_lookahead"]["state_finite"]
