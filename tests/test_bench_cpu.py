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


def _shape(obj, path=""):
    """the set of key paths of a JSON value (kernel labels excluded: they depend
    on the method, not on the number of ranks)"""
    out = set()
    if isinstance(obj, dict):
        for k, v in obj.items():
            if path.endswith("roofline/kernels"):
                out.add(path + "/<label>")
                out |= {p.replace(f"/{k}/", "/<label>/") for p in
                        _shape(v, path + "/" + k)}
            else:
                out.add(path + "/" + k)
                out |= _shape(v, path + "/" + k)
    return out


def test_json_line_has_the_same_shape_for_one_and_eight_ranks():
    """the N = 1 line of `--gpus 1`, of a self-launched 1-rank world and the
    N = 8 line are built by the same function (`assemble`) and carry the same
    fields -- so the first contact with an 8-GPU node cannot fail on a missing
    key of the line the driver parses"""
    def run(world, env):
        res = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps",
                              "3", "--warmup", "1", "--dry-run"],
                             capture_output=True, text=True, timeout=300, cwd=ROOT,
                             env=env)
        assert res.returncode == 0, res.stderr[-2000:]
        (line,) = [ln for ln in res.stdout.splitlines() if ln.strip()]
        return json.loads(line)

    one = run(1, clean_env())
    one_world = run(1, clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                                 MASTER_ADDR="127.0.0.1",
                                 MASTER_PORT=str(free_port())))
    eight = run(8, clean_env())
    # replicas_no_collective is measured only where there is a collective to drop
    extra = {"/config/replicas_no_collective/value",
             "/config/replicas_no_collective/ms_per_step"} | {
        "/config/config5_pr9_lockstep/" + k for k in (
            "workload", "value", "ms_per_step", "ms_per_step_rank_min",
            "rejected_steps_in_timed_region")} | {
        "/config/allreduce_us/" + k for k in ("median", "p99", "median_without_collective",
                                              "collective_median", "calls", "path")}
    assert _shape(one) == _shape(one_world) == _shape(eight) - extra
    # the lock-step line carries the price of one reduction and of lock-step itself
    assert eight["config"]["allreduce_us"]["p99"] >= eight["config"]["allreduce_us"]["median"]
    assert eight["config"]["lockstep_minus_replicas_ms"] is not None
    assert one["config"]["allreduce_us"] is None
    assert one["config"]["lockstep_minus_replicas_ms"] is None
    assert "pinned" in eight["config"]["cpu_affinity"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in eight
    assert eight["n_gpus"] == 8 and one["n_gpus"] == 1
    assert eight["config"]["rccl_nranks"] == 8
    assert eight["config"]["rccl_preflight"] is not None
    assert eight["config"]["ms_per_step_rank_min"] <= eight["config"]["ms_per_step_rank_max"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in eight["roofline"]


def test_eight_rank_lines_of_both_multi_gpu_configs():
    """the first lease of an 8-GPU node: `bench.py --gpus 8` (the metric workload) carries
    BASELINE.json configs[4] -- Pr9, one heat IVP per GPU, lock-step -- as
    `config.config5_pr9_lockstep` of the same line, and `--config pr9` is that
    configuration as the headline; a communicator that does not span the world is a
    failed run (rc != 0, no JSON)"""
    def run(extra_args, env):
        return subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3",
                               "--warmup", "1", "--dry-run"] + extra_args,
                              capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    res = run([], clean_env())
    assert res.returncode == 0, res.stderr[-2000:]
    pr8 = json.loads(res.stdout.strip().splitlines()[-1])
    c5 = pr8["config"]["config5_pr9_lockstep"]
    assert "Pr9" in c5["workload"] and c5["ms_per_step"] > 0
    assert "Pr8" in pr8["config"]["workload"] and "lockstep x8" in pr8["config"]["parallelism"]
    res = run(["--config", "pr9"], clean_env())
    assert res.returncode == 0, res.stderr[-2000:]
    pr9 = json.loads(res.stdout.strip().splitlines()[-1])
    assert "Pr9" in pr9["config"]["workload"] and pr9["n_gpus"] == 8
    assert pr9["config"]["config5_pr9_lockstep"] is None          # it IS the headline
    assert pr9["config"]["rccl_nranks"] == 8 and "lockstep x8" in pr9["config"]["parallelism"]
    res = run(["--config", "pr9"], clean_env(ESQ_BENCH_DRY_RCCL_NRANKS="7"))
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "RCCL sees 7 ranks, expected 8" in res.stderr
