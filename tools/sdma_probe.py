#!/usr/bin/env python3
"""Device-to-host copy rate by HIP stream (GPU box):  python tools/sdma_probe.py [MiB] [streams]
Creates <streams> non-blocking streams and times an 80 MB hipMemcpyAsync device -> pinned host
on each one alone (three rounds), then on pairs (half of the buffer each).  Which DMA engine
a stream's copies use is the runtime's choice; this shows whether the rate is a property of
the stream."""
import ctypes as C
import sys
import time

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hip error {e}")


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 76
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nbytes = mib << 20
    chk(hip.hipSetDevice(0), "hipSetDevice")
    dev, host = vp(), vp()
    chk(hip.hipMalloc(C.byref(dev), C.c_size_t(nbytes)), "hipMalloc")
    chk(hip.hipHostMalloc(C.byref(host), C.c_size_t(nbytes), C.c_uint(0)), "hipHostMalloc")
    chk(hip.hipMemset(dev, 1, C.c_size_t(nbytes)), "hipMemset")
    chk(hip.hipDeviceSynchronize(), "sync")
    streams = []
    for _ in range(ns):
        s = vp()
        chk(hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)), "stream")   # non-blocking
        streams.append(s)
    D2H = 2

    def copy(s, off, n):
        chk(hip.hipMemcpyAsync(vp(host.value + off), vp(dev.value + off), C.c_size_t(n),
                               C.c_int(D2H), s), "memcpy")

    for rnd in range(3):
        row = []
        for s in streams:
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                copy(s, 0, nbytes)
                chk(hip.hipStreamSynchronize(s), "sync")
                ts.append(time.perf_counter() - t0)
            row.append(min(ts))
        print(f"round {rnd}, one stream each, ms: " + " ".join(f"{1e3 * t:.2f}" for t in row))
    half = (nbytes // 2) & ~4095
    row = []
    for i in range(0, ns - 1, 2):
        a, b = streams[i], streams[i + 1]
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            copy(a, 0, half)
            copy(b, half, nbytes - half)
            chk(hip.hipStreamSynchronize(a), "sync")
            chk(hip.hipStreamSynchronize(b), "sync")
            ts.append(time.perf_counter() - t0)
        row.append(min(ts))
    print("pairs (half each), ms: " + " ".join(f"{1e3 * t:.2f}" for t in row))
    # the null stream and a blocking stream for comparison
    t0 = time.perf_counter()
    chk(hip.hipMemcpy(host, dev, C.c_size_t(nbytes), C.c_int(D2H)), "hipMemcpy")
    print(f"hipMemcpy (null stream): {1e3 * (time.perf_counter() - t0):.2f} ms")


if __name__ == "__main__":
    main()
