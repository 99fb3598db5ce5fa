"""bench.py's multi-rank control plane on CPU (`--dry-run`: spawn, TCP
rendezvous, ncclUniqueId broadcast, shard-size sum, barrier, max-over-ranks
time, exactly one JSON line from rank 0; no GPU work, no PyTorch):
  * self-launched: `python bench.py --gpus N --dry-run` for N = 1, 2, 4, 8
  * under the driver's launcher: `python -m torch.distributed.run ...`
  * a failing rank fails the whole run (non-zero exit code, no JSON on stdout)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_self_launched_dry_run(world):
    res = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps",
                          "3", "--warmup", "1", "--dry-run"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=clean_env())
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1
    assert out["max_elapsed"] >= 0.01 * world - 1e-3     # the slowest rank's time


def test_failing_rank_fails_the_run():
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=clean_env(ESQ_BENCH_DRY_FAIL_RANK="2"))
    assert res.returncode != 0
    assert res.stdout.strip() == ""
    assert "rank 2 exited with 3" in res.stderr


def test_two_rank_dry_run_under_torchrun():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH,
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=clean_env())
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["max_elapsed"] >= 0.02          # the slower rank's time


def test_bench_does_not_import_torch():
    src = open(BENCH).read()
    assert "import torch" not in src
    pkg = os.path.join(ROOT, "extensisq_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            assert "import torch" not in open(os.path.join(pkg, name)).read(), name
