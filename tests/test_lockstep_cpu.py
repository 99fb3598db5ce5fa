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


def _control_group_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from extensisq_amd import lockstep
    ctl = lockstep.ControlGroup(rank, world, "127.0.0.1", port, timeout=60)
    ident, total, offset = ctl.exchange(lambda: bytes(range(128)), 1000 + rank)
    assert offset == sum(1000 + r for r in range(rank))
    ctl.barrier()
    mx = ctl.allreduce([float(rank), -float(rank)], "max")
    sm = ctl.allreduce([1.5], "sum")
    blob = ctl.broadcast(b"x" * 300 if rank == 0 else None)
    ctl.barrier()
    ctl.close()
    q.put((rank, ident, total, mx, sm, blob))


@pytest.mark.parametrize("world", [1, 2, 4])
def test_control_group(world):
    """the persistent TCP control plane bench.py uses instead of PyTorch"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_control_group_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ident, total, mx, sm, blob in got:
        assert ident == bytes(range(128))
        assert total == sum(1000 + r for r in range(world))
        assert mx == [float(world - 1), 0.0]
        assert sm == [1.5 * world]
        assert blob == b"x" * 300


def test_control_group_skips_a_taken_port():
    """the port behind MASTER_PORT may belong to somebody else: a foreign
    listener sits on the first port of the range, the group must form on the
    next one"""
    port = free_port()
    squatter = socket.socket()
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    squatter.bind(("127.0.0.1", port))
    squatter.listen(8)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_control_group_worker, args=(r, 2, port, q))
                 for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert [g[2] for g in got] == [2001, 2001]
    finally:
        squatter.close()


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


# ------------------------------------------------- SSV2stab in a lock-step batch
def _rho_jac(t, y):
    """a y-dependent bound of the spectral radius: differs between shards"""
    return 300.0 * (1.0 + float(np.max(np.abs(y))))


def _rkc_lockstep_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from extensisq_amd.common import LockstepGroup
    from extensisq_amd.sommeijer import SSV2stab
    from oracle import problems as pb
    from oracle import rkc_oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def reduce_scalars(values, op):
        tt = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(tt, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN,
                                "sum": dist.ReduceOp.SUM}[op])
        return tt.tolist()

    N = 12
    n = N * N
    y0 = (1.0 + rank) * pb.heat2d_y0(N, seed=40 + rank)    # rank-dependent max|y|
    out = {}

    # (1) the PRODUCT's host logic (no device): the spectral radius a rank
    #     uses is the maximum over the batch, not its own rho_jac(t, y)
    s = object.__new__(SSV2stab)
    s._dev = None
    s._y_host = y0
    s.rho_jac = _rho_jac
    s._lockstep = LockstepGroup(None, world * n, reduce_scalars=reduce_scalars)
    out["local"] = _rho_jac(0.0, y0)
    out["used"] = s._spectral_radius(0.0)
    s._lockstep = None
    out["alone"] = s._spectral_radius(0.0)

    # (2) debug cross-check of the scalars that must agree on every rank
    grp = LockstepGroup(None, world * n, reduce_scalars=reduce_scalars)
    grp.debug = True
    grp.check_identical(None, "(t, h, m)", (0.5, 1e-3, 7))
    try:
        grp.check_identical(None, "(t, h, m)", (0.5, 1e-3, 7 + rank))
        out["caught"] = False
    except RuntimeError as exc:
        out["caught"] = "left lock-step" in str(exc)

    # (3) the sharding mathematics with the oracle standing in for the kernels:
    #     all-reduced error norm + max-reduced spectral radius reproduce the run
    #     on the concatenated state
    f1 = pb.heat2d_rhs(N)

    class Sharded(rkc_oracle.SSV2stab):
        def _err_norm(self, r):
            ss = reduce_scalars([float(r @ r)], "sum")[0]
            return (ss / (world * n)) ** 0.5

        def _rho_user(self, t, yn):
            return reduce_scalars([self.rho_jac(t, yn)], "max")[0]

    sh = Sharded(f1, 0.0, y0, 2e-3, rtol=1e-4, atol=1e-7, rho_jac=_rho_jac,
                 first_step=1e-5)
    ts, ms = [], []
    while sh.status == "running":
        assert sh.step() is None
        ts.append(sh.t)
        ms.append(int(rkc_oracle.maxm[()]))
    out.update(ts=ts, ms=ms, y=sh.y, nfev=sh.nfev)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_ssv2stab_lockstep_with_y_dependent_spectral_radius():
    """SSV2stab in a lock-step batch with a y-dependent `rho_jac`
    (reference sommeijer.py:174-204 takes it from the WHOLE state): every rank
    must use the batch maximum, else the ranks pick different m and h."""
    from oracle import problems as pb
    from oracle import rkc_oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rkc_lockstep_worker, args=(r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    a, b = got[0], got[1]
    assert a["local"] != b["local"]                       # the hole: they differ
    assert a["used"] == b["used"] == max(a["local"], b["local"])
    assert a["alone"] == a["local"] and b["alone"] == b["local"]
    assert a["caught"] and b["caught"]
    assert a["ts"] == b["ts"] and a["ms"] == b["ms"]      # bitwise lock-step
    # the same integration on the concatenated state, one process
    N = 12
    n = N * N
    f1 = pb.heat2d_rhs(N)
    y_all = np.concatenate([(1.0 + r) * pb.heat2d_y0(N, seed=40 + r)
                            for r in range(world)])

    def fun(t, y):
        return np.concatenate([f1(t, y[k * n:(k + 1) * n]) for k in range(world)])

    ref = rkc_oracle.SSV2stab(fun, 0.0, y_all, 2e-3, rtol=1e-4, atol=1e-7,
                              rho_jac=_rho_jac, first_step=1e-5)
    ts = []
    while ref.status == "running":
        assert ref.step() is None
        ts.append(ref.t)
    assert_allclose(a["ts"], ts, rtol=1e-10)
    assert_allclose(np.concatenate([a["y"], b["y"]]), ref.y, rtol=1e-9, atol=1e-12)
    assert a["nfev"] == ref.nfev
