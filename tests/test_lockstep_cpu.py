"""Multi-process CPU tests of the lock-step path (world_size 2):
  (1) the TCP rendezvous that distributes the ncclUniqueId and sums shard sizes;
  (2) the sharding mathematics: per-shard error sums of squares, all-reduced
      (gloo here, RCCL on the GPUs) and divided by the total size, reproduce the
      reference's run on the concatenated state (golden: lockstep.npz) step for
      step -- with the CPU oracle standing in for the per-GPU kernels."""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest
from numpy.testing import assert_allclose

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rendezvous_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from extensisq_amd import lockstep
    ident, total = lockstep.rendezvous(
        rank, world, 1000 + rank, lambda: bytes(range(128)), "127.0.0.1", port,
        timeout=60)
    q.put((rank, ident, total))


@pytest.mark.parametrize("world", [1, 2, 4])
def test_rendezvous(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_total = sum(1000 + r for r in range(world))
    for rank, ident, total in got:
        assert ident == bytes(range(128)) and total == want_total


def _lockstep_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from oracle import problems as pb
    from oracle import rk_oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "lockstep.npz"))
    N = int(g["N"])
    n = N * N
    per = 8 // world
    seeds = [int(s) for s in g["seeds"]][rank * per:(rank + 1) * per]
    f1 = pb.heat2d_rhs(N)
    y0 = np.concatenate([pb.heat2d_y0(N, seed=s) for s in seeds])
    n_total = 8 * n

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(per)])

    class Sharded(rk_oracle.Pr9):
        def _estimate_error_norm(self, K, h, scale):
            r = self._estimate_error(K, h) / scale
            ss = torch.tensor([float(np.real(r @ r.conjugate()))],
                              dtype=torch.float64)
            dist.all_reduce(ss)                   # the ONE exchange per step
            return (float(ss[0]) / n_total) ** 0.5

    s = Sharded(fun, 0.0, y0, float(g["t_end"]), first_step=float(g["h0"]),
                rtol=1e-6, atol=1e-9, nfev_stiff_detect=0)
    ts, errs = [], []
    while s.status == "running":
        s.step()
        ts.append(s.t)
        errs.append(s.error_norm_old)
    q.put((rank, ts, errs, s.y, s.nfev))
    dist.barrier()
    dist.destroy_process_group()


def test_lockstep_sharding_equals_concatenated_reference(golden_dir):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_lockstep_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = np.load(os.path.join(golden_dir, "lockstep.npz"))
    y_all = np.concatenate([r[3] for r in got])
    for rank, ts, errs, _y, nfev in got:
        assert_allclose(ts, g["t"], rtol=1e-10)      # same steps on every rank
        assert_allclose(errs, g["err"], rtol=1e-6)
        # the golden run estimated its first step itself (4 RHS evaluations in
        # h_start, common.py:519-763); the shards were handed that step size
        assert nfev == int(g["nfev"]) - 4
    assert got[0][1] == got[1][1]                     # bitwise identical t_k
    assert_allclose(y_all, g["y_end"], rtol=1e-9, atol=1e-12)
