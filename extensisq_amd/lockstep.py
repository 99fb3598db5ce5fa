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


def _rccl_unique_id():
    buf = C.create_string_buffer(_ID_BYTES)
    _lib.check(_lib.load().esq_comm_unique_id(buf), None, "esq_comm_unique_id")
    return buf.raw


def init_lockstep(rank, world_size, device, n_local, addr=None, port=None,
                  exchange=None):
    """Create the RCCL communicator of this rank and return a `LockstepGroup`
    to pass as `lockstep=` to a solver constructor.

    `exchange(make_id, n_local) -> (id_bytes, n_total)` may replace the built-in
    TCP rendezvous (bench.py passes one that rides on its torch.distributed
    gloo group, so no second port is needed)."""
    lib = _lib.load()
    if exchange is not None:
        ident, n_total = exchange(_rccl_unique_id, n_local)
    else:
        ident, n_total = rendezvous(rank, world_size, n_local, _rccl_unique_id,
                                    addr, port)
    comm = C.c_void_p()
    buf = C.create_string_buffer(ident, _ID_BYTES)
    _lib.check(lib.esq_comm_init_rank(C.byref(comm), world_size, buf, rank,
                                      device), None, "esq_comm_init_rank")
    return LockstepGroup(comm, n_total)


def destroy_lockstep(group):
    if group is not None and group.comm:
        _lib.load().esq_comm_destroy(group.comm)
        group.comm = None
