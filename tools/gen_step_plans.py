#!/usr/bin/env python3
"""Writes tests/golden/step_plans.json: the step programs (launch plans) of every
built-in tableau on the built-in plugins, for every combination of chain-entry
capabilities and with / without lazily written rows, as esq_plan_describe prints
them (no GPU needed: the plans are built on a detached context).

    python tools/gen_step_plans.py            # rewrite the golden file
    python tools/gen_step_plans.py --check    # compare, exit 1 on a difference
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden", "step_plans.json")

FUSE_ALL, FUSE_SRC, FUSE_QUERY, CAP_QUERY = 0x5e, 0x20, 0x80, 16
# round 6: an early estimate whose y_pre is not stored; the FSAL end-point stage and the
# error norm inside a chain sweep (include/extensisq_amd.h)
CAP_PRE, CAP_ERRNORM = 32, 64
CAPS_R6 = [15 | CAP_PRE, 15 | CAP_ERRNORM, 15 | CAP_PRE | CAP_ERRNORM]
METHODS = ["BS5", "Ts5", "Pr7", "Pr8", "Pr9", "CK5", "Me4", "CFMR7osc"]
PLUGINS = [("bruss2d", 2236), ("heat2d", 2236), ("heat2d", 1000), ("diff3d", 159),
           ("plain", 100)]


def heun():
    class Heun:
        n_stages = 2
        A = np.array([[0.0, 0.0], [1.0, 0.0]])
        B = np.array([0.5, 0.5])
        C = np.array([0.0, 1.0])
        E = np.array([0.5, -0.5, 0.0])
    return Heun


def early_estimate(cls):
    """(e_pre, b_scale_pre) of the pairs that test an early estimate (the classes'
    `_early_estimate`), or None"""
    if cls.__name__ == "BS5":
        return cls.E_pre, cls.B_scale_pre
    if cls.__name__ == "CFMR7osc":
        s = cls.n_stages
        return cls.E[:s - 1], cls.A[s - 1, :s - 1]
    return None


def describe(lib, as_ptr, cls, plugin, N, caps, fuse, lazy, depth=4, src=0, pre=None):
    s = cls.n_stages
    arrs = [np.ascontiguousarray(getattr(cls, k), dtype=float) for k in "ABCE"]
    fsal = int(arrs[3][s] != 0)
    buf = C.create_string_buffer(1 << 15)
    e, b = ([np.ascontiguousarray(x, dtype=float) for x in pre] if pre is not None
            else (np.zeros(1), np.zeros(1)))
    r = lib.esq_plan_describe(plugin.encode(), N, s, *[as_ptr(a) for a in arrs], fsal, caps,
                              fuse, lazy, depth, src, as_ptr(e), as_ptr(b),
                              len(e) if pre is not None else 0, buf, len(buf))
    if r:
        raise RuntimeError(f"esq_plan_describe({cls}, {plugin}) -> {r}")
    return buf.value.decode().strip().split("\n")


def table():
    import extensisq_amd as esq
    from extensisq_amd import _lib
    lib = _lib.load()
    out = {}
    classes = [(m, getattr(esq, m)) for m in METHODS] + [("Heun", heun())]
    for name, cls in classes:
        for plugin, N in PLUGINS:
            # (the 3-D plugin's chain entry: plain chains, from the state, rows unwritten)
            all_caps = (range(16) if plugin in ("bruss2d", "heat2d")
                        else [0, 1, 2, 3] if plugin == "diff3d" else [0])
            for caps in all_caps:
                for lazy in (0, 1):
                    # working sets inside the Infinity Cache take stage 1 from the state
                    src = 1 if (plugin, N) == ("heat2d", 1000) else 0
                    key = f"{name}/{plugin}{N}/caps{caps}/lazy{lazy}"
                    out[key] = describe(lib, _lib.as_ptr, cls, plugin, N, caps | CAP_QUERY,
                                        FUSE_ALL | FUSE_SRC | FUSE_QUERY, lazy, 4, src)
            # round 6: the new chain forms (2-D plugins), and -- key suffix /pre -- the
            # WHOLE-STEP programs of the pairs with an early estimate (esq_rk_set_pre)
            pre = early_estimate(cls)
            src = 1 if (plugin, N) == ("heat2d", 1000) else 0
            r6 = CAPS_R6 if plugin in ("bruss2d", "heat2d") else []
            for caps in r6:
                for lazy in (0, 1):
                    key = f"{name}/{plugin}{N}/caps{caps}/lazy{lazy}"
                    out[key] = describe(lib, _lib.as_ptr, cls, plugin, N, caps | CAP_QUERY,
                                        FUSE_ALL | FUSE_SRC | FUSE_QUERY, lazy, 4, src)
            if pre is not None:
                for caps in [c for c in all_caps if c in (0, 3, 15)] + r6:
                    for lazy in (0, 1):
                        key = f"{name}/{plugin}{N}/caps{caps}/lazy{lazy}/pre"
                        out[key] = describe(lib, _lib.as_ptr, cls, plugin, N,
                                            caps | CAP_QUERY, FUSE_ALL | FUSE_SRC | FUSE_QUERY,
                                            lazy, 4, src, pre=pre)
    return out


def main():
    got = table()
    if "--check" in sys.argv:
        with open(GOLD) as fh:
            want = json.load(fh)
        bad = [k for k in sorted(set(got) | set(want)) if got.get(k) != want.get(k)]
        for k in bad[:20]:
            print(k, "\n  got ", got.get(k), "\n  want", want.get(k))
        sys.exit(1 if bad else 0)
    with open(GOLD, "w") as fh:
        json.dump(got, fh, indent=0, sort_keys=True)
    print(len(got), "plans ->", GOLD, os.path.getsize(GOLD), "bytes")


if __name__ == "__main__":
    main()
