#!/usr/bin/env python3
"""Device-to-host copy rate by SOURCE allocation, raw HIP (GPU box): python tools/alloc_probe.py
hipMalloc's several buffers, copies 76 MiB from the start / middle / end of each through
hipMemcpyAsync into one pinned host buffer, frees and re-allocates, prints addresses and ms."""
import ctypes as C
import time

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
MIB = 1 << 20


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hip error {e}")


def main():
    nbytes = 76 * MIB
    chk(hip.hipSetDevice(0), "hipSetDevice")
    host = vp()
    chk(hip.hipHostMalloc(C.byref(host), C.c_size_t(nbytes), C.c_uint(0)), "hipHostMalloc")
    s = vp()
    chk(hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)), "stream")

    def alloc(mib, flags=None):
        p = vp()
        if flags is None:
            chk(hip.hipMalloc(C.byref(p), C.c_size_t(mib * MIB)), "hipMalloc")
        else:
            chk(hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(mib * MIB), C.c_uint(flags)),
                "hipExtMallocWithFlags")
        chk(hip.hipMemset(p, 1, C.c_size_t(mib * MIB)), "memset")
        chk(hip.hipDeviceSynchronize(), "sync")
        return p

    def rate(p, off):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            chk(hip.hipMemcpyAsync(host, vp(p.value + off), C.c_size_t(nbytes), C.c_int(2), s), "cpy")
            chk(hip.hipStreamSynchronize(s), "sync")
            ts.append(time.perf_counter() - t0)
        return 1e3 * min(ts)

    def show(name, p, mib):
        offs = [0, ((mib * MIB - nbytes) // 2) & ~4095, mib * MIB - nbytes]
        print(f"{name}: {mib} MiB at {p.value:#x}: " + "  ".join(f"{rate(p, o):.2f}" for o in offs) + " ms")

    a = alloc(1200); show("A (first)", a, 1200)
    b = alloc(1200); show("B (A alive)", b, 1200)
    c = alloc(1200); show("C (A, B alive)", c, 1200)
    show("A again", a, 1200)
    chk(hip.hipFree(a), "free"); chk(hip.hipFree(b), "free")
    d = alloc(1200); show("D (A, B freed; C alive)", d, 1200)
    e = alloc(1200); show("E (C, D alive)", e, 1200)
    small = [alloc(80) for _ in range(6)]
    for k, p in enumerate(small):
        show(f"small {k}", p, 80)
    big = alloc(8000); show("8000 MiB", big, 8000)
    for p in [c, d, e, big] + small:
        chk(hip.hipFree(p), "free")
    f = alloc(1200); show("F (everything freed)", f, 1200)
    g = alloc(1200); show("G (F alive)", g, 1200)
    chk(hip.hipFree(f), "free"); chk(hip.hipFree(g), "free")
    # hipDeviceMallocContiguous = 0x4: physically contiguous
    h = alloc(1200, 4); show("H (recycled range, hipDeviceMallocContiguous)", h, 1200)
    i = alloc(1200, 4); show("I (H alive, contiguous)", i, 1200)
    j = alloc(1200); show("J (default again)", j, 1200)
    k = alloc(11000, 4); show("K (11 GB contiguous)", k, 11000)


if __name__ == "__main__":
    main()
