"""bench.py as its own launcher, without a GPU: `python3 bench.py --gpus N` (no WORLD_SIZE) must start N ranks as child processes before
anything heavy is imported, relay what rank 0 prints, and -- when the ranks fail, as they do here for want of a GPU -- still print ONE
JSON line with an "error" field and return a non-zero code quickly.  (The GPU tests run the same path to a successful line.)"""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="on a GPU box the ranks would start a real run: covered by tests/test_bench_gpu.py")
def test_launcher_reports_ranks_that_cannot_start():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SLURM_NTASKS")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "100000", "--time-budget", "60"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and time.time() - t0 < 120
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and "error" in d
    assert "launcher: rank exit codes" in out.stderr


def test_launcher_is_not_taken_under_a_launcher():
    """With WORLD_SIZE set (torch.distributed.run, SLURM) bench.py is a rank: a mismatch with --gpus is an error, not a second launch."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", COALA_NUMA="off")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "100000"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "does not match --gpus 2" in (out.stderr + out.stdout)
    assert "launcher: rank exit codes" not in out.stderr
