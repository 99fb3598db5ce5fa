"""bench.py's multi-rank control plane under the driver's launcher, on CPU:
`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --dry-run`
(gloo rendezvous, ncclUniqueId broadcast, shard-size all-reduce, barrier,
max-over-ranks time, exactly one JSON line from rank 0)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_dry_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["max_elapsed"] >= 0.02          # the slower rank's time


def test_single_process_dry_run():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--dry-run"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads(res.stdout.strip())["n_gpus"] == 1
