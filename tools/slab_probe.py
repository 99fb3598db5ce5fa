#!/usr/bin/env python3
"""Is the rate of a device-to-host copy a property of the SOURCE allocation?  (GPU box)
    python tools/slab_probe.py [contexts]
Makes Pr8-sized contexts (n = 1e7, 14 rows: a 1.2 GB slab each) one after the other, the
previous one destroyed when the next is made (as consecutive solve_ivp calls do), and times
engine copies of the state and of three rows of K of each."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from extensisq_amd import _lib                                   # noqa: E402
from extensisq_amd.device import DeviceContext                   # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lib = _lib.load()
    n = 9_999_392
    nbytes = 8 * n
    buf = np.zeros(nbytes, dtype=np.uint8)
    addr = buf.ctypes.data
    prev = None
    for k in range(count):
        dev = DeviceContext(n, 14)
        if prev is not None:
            prev.close()
        prev = dev
        dev.upload(_lib.SLOT_Y, 0, np.zeros(n))
        row = []
        for slot, r in ((_lib.SLOT_Y, 0), (_lib.SLOT_YNEW, 0), (_lib.SLOT_K, 0), (_lib.SLOT_K, 6),
                        (_lib.SLOT_K, 13)):
            ts = []
            for _ in range(3):
                pin = lib.esq_host_pin(C.c_void_p(addr), nbytes)
                token = C.c_void_p()
                rc = lib.esq_snapshot_begin(dev.handle, slot, r, C.byref(token))
                if rc != 0:
                    ts.append(float("nan"))
                    if pin == 0:
                        lib.esq_host_unpin(C.c_void_p(addr))
                    continue
                t0 = time.perf_counter()
                assert lib.esq_snapshot_copy(token, C.c_void_p(addr), 1 if pin == 0 else 0) == 0
                ts.append(time.perf_counter() - t0)
            row.append(min(ts))
        print(f"context {k}: y {row[0] * 1e3:.2f}  y_new {row[1] * 1e3:.2f}  K0 {row[2] * 1e3:.2f}  "
              f"K6 {row[3] * 1e3:.2f}  K13 {row[4] * 1e3:.2f} ms")
    prev.close()


if __name__ == "__main__":
    main()
