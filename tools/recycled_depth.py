#!/usr/bin/env python3
"""How much of the device's memory reads slow for the DMA engines right now?  (GPU box)
    python tools/recycled_depth.py [blocks] [MiB per block]
hipMalloc's <blocks> blocks (kept until the end), copies 16 MiB of each to the host through
hipMemcpyAsync and prints the rates run-length encoded (F: >= 45 GB/s, s: below)."""
import ctypes as C
import sys
import time

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
MIB = 1 << 20


def chk(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hip error {e}")


def main():
    blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    nbytes = 16 * MIB
    chk(hip.hipSetDevice(0), "hipSetDevice")
    host, s = vp(), vp()
    chk(hip.hipHostMalloc(C.byref(host), C.c_size_t(nbytes), C.c_uint(0)), "hipHostMalloc")
    chk(hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)), "stream")
    marks, rates = [], []
    for _ in range(blocks):
        p = vp()
        chk(hip.hipMalloc(C.byref(p), C.c_size_t(mib * MIB)), "hipMalloc")
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            chk(hip.hipMemcpyAsync(host, p, C.c_size_t(nbytes), C.c_int(2), s), "copy")
            chk(hip.hipStreamSynchronize(s), "sync")
            best = max(best, nbytes / (time.perf_counter() - t0) / 1e9)
        rates.append(best)
        marks.append("F" if best >= 45.0 else "s")
    out, k = [], 0
    while k < len(marks):
        j = k
        while j < len(marks) and marks[j] == marks[k]:
            j += 1
        out.append(f"{marks[k]}x{j - k}")
        k = j
    print(f"{blocks} blocks of {mib} MiB in allocation order: " + " ".join(out))
    print(f"rates: min {min(rates):.1f}, max {max(rates):.1f} GB/s; slow blocks: "
          f"{marks.count('s')} = {marks.count('s') * mib / 1024:.1f} GiB")


if __name__ == "__main__":
    main()
