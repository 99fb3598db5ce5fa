"""Lock-step batched integration across the GPUs of one node
(BASELINE.json configs[4]; not present in the reference).

Each rank (one process per GPU) integrates its own independent IVP shard with
the SAME step size: the weighted error sum of squares is all-reduced (one fp64,
RCCL over xGMI, inside the C library) so that every rank computes the error norm
of the concatenated state, `sqrt(sum_g sumsq_g / sum_g n_g)` -- exactly what the
reference computes on the concatenated vector (common.py:64-66) -- and takes the
same accept/reject decision.  No other data moves between GPUs.

The rendezvous below only distributes the 128-byte ncclUniqueId and sums the
shard sizes; it uses a plain TCP socket on MASTER_ADDR:MASTER_PORT+1 so that the
package itself needs no PyTorch.
"""
import ctypes as C
import os
import socket
import struct
import time

from . import _lib
from .common import LockstepGroup

_ID_BYTES = 128


def _recv_exact(sock, count):
    buf = b""
    while len(buf) < count:
        chunk = sock.recv(count - len(buf))
        if not chunk:
            raise ConnectionError("lock-step rendezvous: peer closed")
        buf += chunk
    return buf


def rendezvous(rank, world_size, n_local, make_id, addr=None, port=None,
               timeout=120.0):
    """Rank 0 creates an id with `make_id()` (bytes) and serves it; every rank
    returns (id_bytes, n_total).  Pure host code (testable without a GPU)."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + 1)
    if world_size == 1:
        return make_id(), int(n_local)
    if rank == 0:
        ident = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world_size)
        srv.settimeout(timeout)
        peers, total = [], int(n_local)
        try:
            for _ in range(world_size - 1):
                conn, _a = srv.accept()
                conn.settimeout(timeout)
                (n_peer,) = struct.unpack("<q", _recv_exact(conn, 8))
                total += n_peer
                peers.append(conn)
            for conn in peers:
                conn.sendall(struct.pack("<q", total) + ident)
        finally:
            for conn in peers:
                conn.close()
            srv.close()
        return ident, total
    deadline = time.time() + timeout
    while True:
        try:
            sock = socket.create_connection((addr, port), timeout=timeout)
            break
        except OSError:
            if time.time() > deadline:
                raise
            time.sleep(0.05)
    try:
        sock.sendall(struct.pack("<q", int(n_local)))
        payload = _recv_exact(sock, 8 + _ID_BYTES)
    finally:
        sock.close()
    (total,) = struct.unpack("<q", payload[:8])
    return payload[8:], total


class ControlGroup:
    """Host-side control plane of a multi-rank run (bench.py, tests): a TCP star
    on MASTER_ADDR:MASTER_PORT+1 with rank 0 at the centre.  It carries what the
    data path never does -- the 128-byte ncclUniqueId, shard sizes, barriers and
    max-over-ranks timings -- so neither the package nor the benchmark needs
    PyTorch.  Every call is collective (all ranks, same order)."""

    _OPS = {"sum": sum, "max": max, "min": min}
    _MAGIC = b"ESQCTL02"
    _PORT_SPAN = 16

    def __init__(self, rank, world_size, addr=None, port=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world_size)
        self._peers = []          # rank 0: sockets of ranks 1..world-1, by rank
        self._sock = None         # other ranks: socket to rank 0
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("ESQ_CTL_PORT") or
                   int(os.environ.get("MASTER_PORT", "29500")) + 1)
        # the launcher owns MASTER_PORT; the ports behind it may be taken too:
        # rank 0 binds the first free one of a small range, the others find it by
        # a handshake (magic + world size + their rank)
        ports = range(port, port + self._PORT_SPAN)
        if self.rank == 0:
            srv = None
            for cand in ports:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    srv.bind((addr, cand))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError(f"ControlGroup: no free port in {ports}")
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            try:
                while len(by_rank) < self.world - 1:
                    conn, _a = srv.accept()
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    try:
                        hello = _recv_exact(conn, 24)
                    except (OSError, ConnectionError):
                        conn.close()
                        continue
                    magic, world, peer = struct.unpack("<8sqq", hello)
                    if magic != self._MAGIC or world != self.world or \
                            not 0 < peer < self.world or peer in by_rank:
                        conn.close()                  # a stranger: not ours
                        continue
                    conn.sendall(self._MAGIC)
                    by_rank[peer] = conn
            finally:
                srv.close()
            self._peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            sock = None
            while sock is None:
                for cand in ports:
                    try:
                        trial = socket.create_connection((addr, cand), timeout=5.0)
                    except OSError:
                        continue
                    try:
                        trial.settimeout(5.0)
                        trial.sendall(struct.pack("<8sqq", self._MAGIC, self.world,
                                                  self.rank))
                        if _recv_exact(trial, 8) == self._MAGIC:
                            sock = trial
                            break
                    except (OSError, ConnectionError):
                        pass
                    trial.close()
                if sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("ControlGroup: rank 0 not found on "
                                           f"{addr}:{ports.start}-{ports.stop - 1}")
                    time.sleep(0.05)
            sock.settimeout(timeout)
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._sock = sock

    # -- framing: 8-byte length + payload
    @staticmethod
    def _send(sock, payload):
        sock.sendall(struct.pack("<q", len(payload)) + payload)

    @staticmethod
    def _recv(sock):
        (size,) = struct.unpack("<q", _recv_exact(sock, 8))
        return _recv_exact(sock, size) if size else b""

    def allreduce(self, values, op="sum"):
        """element-wise reduction of a list of floats over all ranks"""
        values = [float(v) for v in values]
        if self.world == 1:
            return values
        fmt = "<%dd" % len(values)
        if self.rank == 0:
            rows = [values] + [list(struct.unpack(fmt, self._recv(c)))
                               for c in self._peers]
            out = [self._OPS[op](col) for col in zip(*rows)]
            blob = struct.pack(fmt, *out)
            for c in self._peers:
                self._send(c, blob)
            return out
        self._send(self._sock, struct.pack(fmt, *values))
        return list(struct.unpack(fmt, self._recv(self._sock)))

    def broadcast(self, payload=None):
        """bytes from rank 0 to everyone"""
        if self.world == 1:
            return payload
        if self.rank == 0:
            for c in self._peers:
                self._send(c, payload)
            return payload
        return self._recv(self._sock)

    def barrier(self):
        self.allreduce([0.0])

    def exchange(self, make_id, n_local):
        """the `exchange` hook of `init_lockstep`: ncclUniqueId from rank 0, the
        summed shard size and this shard's offset in the concatenated state"""
        ident = self.broadcast(make_id() if self.rank == 0 else None)
        sizes = [0.0] * self.world
        sizes[self.rank] = float(n_local)
        sizes = [int(round(v)) for v in self.allreduce(sizes, "sum")]
        return ident, sum(sizes), sum(sizes[:self.rank])

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None


def _rccl_unique_id():
    buf = C.create_string_buffer(_ID_BYTES)
    _lib.check(_lib.load().esq_comm_unique_id(buf), None, "esq_comm_unique_id")
    return buf.raw


def init_lockstep(rank, world_size, device, n_local, addr=None, port=None,
                  exchange=None):
    """Create the RCCL communicator of this rank and return a `LockstepGroup`
    to pass as `lockstep=` to a solver constructor.

    `exchange(make_id, n_local) -> (id_bytes, n_total)` may replace the built-in
    one-shot TCP rendezvous (bench.py passes `ControlGroup.exchange`, which rides
    on its persistent control connections)."""
    lib = _lib.load()
    offset = None
    if exchange is not None:
        got = exchange(_rccl_unique_id, n_local)
        ident, n_total = got[0], got[1]
        if len(got) > 2:
            offset = got[2]
    else:
        ident, n_total = rendezvous(rank, world_size, n_local, _rccl_unique_id,
                                    addr, port)
    comm = C.c_void_p()
    buf = C.create_string_buffer(ident, _ID_BYTES)
    _lib.check(lib.esq_comm_init_rank(C.byref(comm), world_size, buf, rank,
                                      device), None, "esq_comm_init_rank")
    return LockstepGroup(comm, n_total, offset=offset)


def preflight(group, rank, world, device, ctl=None):
    """Check the lock-step collectives before anything depends on them: every
    rank all-reduces `rank + 1` through `esq_allreduce_scalars` (sum, max, min ->
    N(N+1)/2, N, 1) and one weighted-norm reduction through the error-norm path
    (`finish_reduction`: k_final_sum -> ncclAllReduce -> pinned slot), which is
    compared with the sum the TCP control plane computes from the per-rank
    values.  Raises on any mismatch; returns "ok"."""
    import numpy as np
    from ._lib import SLOT_Y, VEC_NONE, VEC_Y
    from .device import DeviceContext
    dev = DeviceContext(1024, 2, False, device)
    try:
        dev._chk(dev.lib.esq_set_comm(dev.handle, group.comm), "esq_set_comm")
        group.attach(dev)
        want = {"sum": world * (world + 1) / 2.0, "max": float(world), "min": 1.0}
        for op, expect in want.items():
            got = group.allreduce(dev, [rank + 1.0, 2.0 * (rank + 1.0)], op)
            if got != [expect, 2.0 * expect]:
                raise RuntimeError(f"lock-step pre-flight: all-reduce({op}) of rank+1 "
                                   f"over {world} ranks gave {got}, expected "
                                   f"{[expect, 2.0 * expect]}")
        dev.upload(SLOT_Y, 0, np.full(1024, rank + 1.0))
        out = C.c_double()
        dev._chk(dev.lib.esq_vec_sumsq(dev.handle, VEC_Y, VEC_NONE, C.byref(out)),
                 "esq_vec_sumsq")
        local = 1024.0 * (rank + 1.0) ** 2
        expect = ctl.allreduce([local], "sum")[0] if ctl is not None else \
            1024.0 * sum((r + 1.0) ** 2 for r in range(world))
        if out.value != expect:
            raise RuntimeError(f"lock-step pre-flight: all-reduced sum of squares "
                               f"{out.value} != {expect} (control plane)")
    finally:
        group._contexts = [d for d in group._contexts if d is not dev]
        dev._chk(dev.lib.esq_set_comm(dev.handle, None), "esq_set_comm")
        dev.close()
    return "ok"


def time_allreduce(group, device, ctl=None, count=200):
    """What one lock-step reduction costs on this communicator: `count` error-norm
    reductions of a 1024-element vector (k_sumsq -> k_final_sum -> ncclAllReduce of
    ONE double -> publish into the pinned slot -> the spinning host: exactly the
    path every step attempt takes) timed call by call, then the same calls on the
    same context without the communicator.  Returns microseconds: median and p99
    with the all-reduce, the median without it, and their difference -- the price
    of the collective itself."""
    import time

    import numpy as np
    from ._lib import SLOT_Y, VEC_NONE, VEC_Y
    from .device import DeviceContext
    dev = DeviceContext(1024, 2, False, device)

    def run():
        out = C.c_double()
        stamps = []
        for _ in range(count + 20):
            t0 = time.perf_counter()
            dev._chk(dev.lib.esq_vec_sumsq(dev.handle, VEC_Y, VEC_NONE, C.byref(out)),
                     "esq_vec_sumsq")
            stamps.append(time.perf_counter() - t0)
        us = np.sort(np.array(stamps[20:])) * 1e6          # 20 warm-up calls
        return float(np.median(us)), float(us[min(len(us) - 1, int(0.99 * len(us)))])

    try:
        dev.upload(SLOT_Y, 0, np.ones(1024))
        dev._chk(dev.lib.esq_set_comm(dev.handle, group.comm), "esq_set_comm")
        group.attach(dev)
        if ctl is not None:
            ctl.barrier()
        med, p99 = run()
        group._contexts = [d for d in group._contexts if d is not dev]
        dev._chk(dev.lib.esq_set_comm(dev.handle, None), "esq_set_comm")
        if ctl is not None:
            ctl.barrier()
        base, _ = run()
    finally:
        group._contexts = [d for d in group._contexts if d is not dev]
        dev.close()
    return {"median": med, "p99": p99, "median_without_collective": base,
            "collective_median": med - base, "calls": count,
            "path": "error-norm reduction of a 1024-element vector: k_sumsq + "
                    "k_final_sum + ncclAllReduce(1 double) + publish to the pinned "
                    "slot, host-timed call by call"}


def comm_size(group):
    """number of ranks RCCL reports for the group's communicator"""
    out = C.c_int(0)
    _lib.check(_lib.load().esq_comm_count(group.comm, C.byref(out)), None,
               "esq_comm_count")
    return out.value


def abort_lockstep(group):
    """ncclCommAbort: called by a rank that fails outside a collective so that
    its peers' pending all-reduce errors out instead of blocking"""
    if group is not None and not group.sync_aborted():
        _lib.load().esq_comm_abort(group.comm)
        group.comm = None


def destroy_lockstep(group):
    if group is not None and not group.sync_aborted():
        _lib.load().esq_comm_destroy(group.comm)
        group.comm = None
