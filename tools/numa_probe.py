#!/usr/bin/env python3
"""Where do the host pages of a state download have to live?  (GPU box)
    python tools/numa_probe.py [n]
Prints the NUMA node of the visible GPU, the CPUs / memory nodes this process may use,
and the rate of an n-double (default 1e7 = 80 MB) device-to-host copy through
esq_snapshot_begin / esq_snapshot_copy into page-locked buffers BOUND to each memory
node (mbind before first touch; placement verified with move_pages), with the calling
thread on the CPUs of each node."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from extensisq_amd import _lib                                   # noqa: E402
from extensisq_amd.device import DeviceContext                   # noqa: E402

libc = C.CDLL(None, use_errno=True)
libc.syscall.restype = C.c_long
SYS_MBIND, SYS_MOVE_PAGES = 237, 279
MPOL_BIND = 2


def cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def mbind(addr, length, node):
    mask = (C.c_ulong * 16)()
    mask[node // 64] = 1 << (node % 64)
    page = addr & ~4095
    r = libc.syscall(C.c_long(SYS_MBIND), C.c_void_p(page), C.c_ulong(length + (addr - page)),
                     C.c_int(MPOL_BIND), mask, C.c_ulong(16 * 64), C.c_uint(0))
    return r if r == 0 else -C.get_errno()


def node_of(addr):
    pages = (C.c_void_p * 1)(addr & ~4095)
    status = (C.c_int * 1)(-99)
    r = libc.syscall(C.c_long(SYS_MOVE_PAGES), C.c_int(0), C.c_ulong(1), pages, None, status,
                     C.c_int(0))
    return status[0] if r == 0 else -C.get_errno()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    lib = _lib.load()
    bdf = _lib.device_pci_bus_id(0)
    try:
        gpu_node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except OSError:
        gpu_node = None
    print(f"HIP device 0 = {bdf}, NUMA node {gpu_node}")
    for ln in open("/proc/self/status"):
        if ln.startswith(("Cpus_allowed_list", "Mems_allowed_list")):
            print("  " + ln.strip())
    nodes = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node")
                   if d.startswith("node") and d[4:].isdigit())
    node_cpus = {k: cpulist(open(f"/sys/devices/system/node/node{k}/cpulist").read())
                 for k in nodes}
    print("  nodes:", {k: len(v) for k, v in node_cpus.items()})
    allowed = os.sched_getaffinity(0)
    dev = DeviceContext(n, 2)
    dev.upload(_lib.SLOT_Y, 0, np.arange(n, dtype=float))
    nbytes = 8 * n
    for cpu_node in nodes:
        cpus = node_cpus[cpu_node] & allowed
        if not cpus:
            print(f"thread on node {cpu_node}: no allowed CPU")
            continue
        os.sched_setaffinity(0, cpus)
        for mem_node in nodes + [None]:
            buf = np.empty(nbytes, dtype=np.uint8)
            addr = buf.ctypes.data
            rb = mbind(addr, nbytes, mem_node) if mem_node is not None else "first touch"
            C.memset(addr, 0, nbytes)
            where = (node_of(addr), node_of(addr + nbytes // 2), node_of(addr + nbytes - 1))
            pin = lib.esq_host_pin(C.c_void_p(addr), nbytes)
            ts = []
            for _ in range(8):
                token = C.c_void_p()
                assert lib.esq_snapshot_begin(dev.handle, _lib.SLOT_Y, 0, C.byref(token)) == 0
                t0 = time.perf_counter()
                rc = lib.esq_snapshot_copy(token, C.c_void_p(addr), 1 if pin == 0 else 0)
                ts.append(time.perf_counter() - t0)
                assert rc == 0, rc
            # (esq_snapshot_copy releases the page lock of a buffer handed in locked;
            # lock again for the next round)
                if pin == 0:
                    pin = lib.esq_host_pin(C.c_void_p(addr), nbytes)
            ok = bool(np.array_equal(buf.view(np.float64)[:5], np.arange(5.0)))
            best = min(ts[1:])
            print(f"thread on node {cpu_node}, pages bound to {mem_node} (mbind {rb}, pages on "
                  f"{where}): {best * 1e3:.2f} ms = {nbytes / best / 1e9:.1f} GB/s, "
                  f"median {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms, data ok {ok}")
            if pin == 0:
                lib.esq_host_unpin(C.c_void_p(addr))
            del buf
    dev.close()


if __name__ == "__main__":
    main()


def under_compute():
    """the same copy while a solver takes Pr8 steps on the device (another thread)"""
    import threading
    import bench
    lib = _lib.load()
    w = bench.make_workload("pr8", None, 0)
    n = w["y0"].size
    solver = w["cls"](w["rhs"], 0.0, w["y0"], 1e9, device=0, **w["kw"])
    stop = threading.Event()
    count = [0]

    def run():
        while not stop.is_set():
            solver.step()
            count[0] += 1
    dev = DeviceContext(n, 2)
    dev.upload(_lib.SLOT_Y, 0, np.arange(n, dtype=float))
    nbytes = 8 * n
    buf = np.empty(nbytes, dtype=np.uint8)
    addr = buf.ctypes.data
    C.memset(addr, 0, nbytes)
    for busy in (False, True, False, True):
        if busy:
            stop.clear()
            th = threading.Thread(target=run)
            c0, t_start = count[0], time.perf_counter()
            th.start()
            time.sleep(0.05)
        ts = []
        for _ in range(12):
            pin = lib.esq_host_pin(C.c_void_p(addr), nbytes)
            token = C.c_void_p()
            assert lib.esq_snapshot_begin(dev.handle, _lib.SLOT_Y, 0, C.byref(token)) == 0
            t0 = time.perf_counter()
            rc = lib.esq_snapshot_copy(token, C.c_void_p(addr), 1 if pin == 0 else 0)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        extra = ""
        if busy:
            stop.set()
            th.join()
            extra = ", steps at %.3f ms each meanwhile" % (
                1e3 * (time.perf_counter() - t_start) / max(1, count[0] - c0))
        ts = sorted(ts[1:])
        print(f"copy of {nbytes >> 20} MiB, device {'stepping' if busy else 'idle'}: min "
              f"{ts[0] * 1e3:.2f} ms, median {ts[len(ts) // 2] * 1e3:.2f} ms{extra}")
    dev.close()


def huge_pages():
    """does the page size behind the destination matter?  (the IOMMU / the DMA engine
    translate per page: 4 KiB pages vs transparent 2 MiB ones)"""
    lib = _lib.load()
    for f in ("enabled", "defrag", "shmem_enabled"):
        try:
            print(f"  transparent_hugepage/{f}:",
                  open(f"/sys/kernel/mm/transparent_hugepage/{f}").read().strip())
        except OSError as exc:
            print("  transparent_hugepage:", exc)

    def anon_huge_kb():
        for ln in open("/proc/self/smaps_rollup"):
            if ln.startswith("AnonHugePages"):
                return int(ln.split()[1])
        return -1
    n = 10_000_000
    nbytes = 8 * n
    dev = DeviceContext(n, 2)
    dev.upload(_lib.SLOT_Y, 0, np.arange(n, dtype=float))
    libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    for name, advice in (("default", None), ("MADV_HUGEPAGE", 14), ("MADV_NOHUGEPAGE", 15),
                         ("default", None), ("MADV_NOHUGEPAGE", 15), ("MADV_HUGEPAGE", 14)):
        before = anon_huge_kb()
        buf = np.empty(nbytes + (4 << 20), dtype=np.uint8)
        addr = (buf.ctypes.data + (2 << 20) - 1) & ~((2 << 20) - 1)      # 2 MiB aligned
        r = libc.madvise(addr, nbytes, advice) if advice is not None else 0
        C.memset(addr, 0, nbytes)
        huge = anon_huge_kb() - before
        ts = []
        for _ in range(6):
            pin = lib.esq_host_pin(C.c_void_p(addr), nbytes)
            token = C.c_void_p()
            assert lib.esq_snapshot_begin(dev.handle, _lib.SLOT_Y, 0, C.byref(token)) == 0
            t0 = time.perf_counter()
            assert lib.esq_snapshot_copy(token, C.c_void_p(addr), 1 if pin == 0 else 0) == 0
            ts.append(time.perf_counter() - t0)
        print(f"destination {name} (madvise {r}; {huge >> 10} MiB of it in huge pages): "
              f"min {min(ts[1:]) * 1e3:.2f} ms, median {sorted(ts)[3] * 1e3:.2f} ms")
        del buf
    dev.close()


if __name__ == "__main__" and os.environ.get("ESQ_NUMA_PROBE_COMPUTE", "1") != "0":
    under_compute()
    huge_pages()
