"""bench.py starts its own ranks when `--gpus N` comes without a launcher's environment."""

from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LAUNCHER_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                 "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")


def _run(*flags, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], env=env, cwd=str(ROOT),
                          capture_output=True, text=True, timeout=timeout)


def test_launcher_returns_the_failure_of_its_ranks():
    """Without two GPUs the nccl ranks refuse to start; the parent (which never touches the GPU)
    ends the job and hands their exit status on instead of hanging."""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the ranks would start")
    res = _run("--gpus", "2", "--shows", "4096", "--steps", "1", "--no-cpu-baseline", timeout=300)
    assert res.returncode != 0
    assert "needs 2 GPUs" in res.stderr or "NativeLibraryError" in res.stderr or "no GPU" in res.stderr.lower() \
        or "HIP" in res.stderr or "cuda" in res.stderr.lower()
    assert res.stdout.strip() == ""


@pytest.mark.gpu
@pytest.mark.parametrize("z_exchange", ["auto", "gather"])
def test_bench_gpus_2_launches_itself(device, z_exchange):
    """`python bench.py --gpus 2` with no WORLD_SIZE: two rank processes share the one GPU (gloo
    rehearsal), rank 0 prints one JSON line with n_gpus 2, every rank's lists equal the plain sweep --
    with the normalised matrix built on every rank (the exact path's default) and with the shards
    all-gathered."""
    res = _run("--gpus", "2", "--backend", "gloo", "--shows", "8192", "--steps", "2", "--warmup", "1", "--check",
               "--no-cpu-baseline", "--z-exchange", z_exchange)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert 0 < line["roofline"]["frac"] <= 1.0
    assert line["config"]["z_exchange"] == ("local" if z_exchange == "auto" else "gather")
    assert ("gather_wait_ms" in line["phases"]) == (z_exchange == "gather")
    if z_exchange == "auto":  # the all-gather variant is timed beside the resident one, with its own phases
        assert line["gather_step"]["value"] > 0 and "gather_wait_ms" in line["gather_step"]["phases"]
    else:
        assert line["gather_step"] is None
    assert res.stderr.count("lists equal to the plain sweep: True") == 2
