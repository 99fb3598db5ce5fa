"""Probe of host-memory costs behind `solver.y` at n = 1e7 (80 MB per state):
what a fresh destination costs (page faults, pinning) against the copy itself.
Run on the GPU box: python tools/probe_pinned.py"""
import ctypes as C
import mmap
import threading
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
libc = C.CDLL("libc.so.6")
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
n = 10_000_000
nbytes = n * 8
dev = C.c_void_p()
hip.hipMalloc(C.byref(dev), C.c_size_t(nbytes))
hip.hipMemset(dev, 1, C.c_size_t(nbytes))
hip.hipDeviceSynchronize()


def t(f, reps=5):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.2f" % x for x in out)


keep = []


def d2h(a):
    hip.hipMemcpy(a.ctypes.data_as(C.c_void_p), dev, C.c_size_t(nbytes), 2)


def fresh_thp(touch=True):
    m = mmap.mmap(-1, nbytes + (2 << 20))
    a = np.frombuffer(m, dtype=np.float64, count=n, offset=0)
    addr = a.ctypes.data
    al = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
    libc.madvise(C.c_void_p(al), C.c_size_t(nbytes - (2 << 20)), 14)   # MADV_HUGEPAGE
    if touch:
        libc.memset(C.c_void_p(addr), 0, C.c_size_t(nbytes))
    keep.append(m)
    return a


warm = np.empty(n)
warm[:] = 0
print("D2H into WARM pageable np         ms:", t(lambda: d2h(warm)))
def fresh_4k():
    b = np.empty(n)
    libc.memset(C.c_void_p(b.ctypes.data), 0, C.c_size_t(nbytes))
    keep.append(b)


print("np.empty + memset (4K faults)     ms:", t(fresh_4k))
print("mmap + MADV_HUGEPAGE + memset     ms:", t(lambda: fresh_thp()))
a = fresh_thp()
print("D2H into warm THP buffer          ms:", t(lambda: d2h(a)))


def reg_copy(buf):
    p = C.c_void_p(buf.ctypes.data)
    hip.hipHostRegister(p, C.c_size_t(nbytes), 0)
    hip.hipMemcpy(p, dev, C.c_size_t(nbytes), 2)
    hip.hipHostUnregister(p)


print("register+D2H+unregister warm 4K   ms:", t(lambda: reg_copy(warm)))
print("register+D2H+unregister warm THP  ms:", t(lambda: reg_copy(a)))
print("fresh THP untouched + D2H         ms:", t(lambda: d2h(fresh_thp(False))))
print("fresh THP untouched + reg + D2H   ms:", t(lambda: reg_copy(fresh_thp(False))))


def par(k):
    bufs = [np.empty(n) for _ in range(k)]
    th = [threading.Thread(target=libc.memset, args=(C.c_void_p(b.ctypes.data), 0,
                                                     C.c_size_t(nbytes))) for b in bufs]
    for x in th:
        x.start()
    for x in th:
        x.join()
    keep.extend(bufs)


for k in (1, 2, 4, 8):
    print(f"prefault {k} x 80 MB in {k} threads      ms:", t(lambda: par(k), reps=3))
