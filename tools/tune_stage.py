#!/usr/bin/env python3
"""GPU micro-benchmark of the stage-accumulate kernel (run on the MI355X box):
per-stage time / algorithmic GB/s of Pr8's 12 stage kernels at n ~ 1e7 for a
set of launch-geometry / cache-policy settings given through the ESQ_*
environment knobs the library reads at context creation.

    python tools/tune_stage.py [--n 9999392] [--reps 20] KEY=VAL[,VAL...] ...
e.g. python tools/tune_stage.py ESQ_BLOCKS_PER_CU=4,8,16 ESQ_STAGE_POLICY=00,10
"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def main():
    n, reps = 9_999_392, 20
    grid = {}
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--n":
            n = int(args.pop(0))
        elif a == "--reps":
            reps = int(args.pop(0))
        else:
            k, v = a.split("=")
            grid[k] = v.split(",")
    import extensisq_amd as esq
    from extensisq_amd._lib import PROF_STAGE, SLOT_K, SLOT_Y
    cls = esq.Pr8
    s = cls.n_stages
    rng = np.random.default_rng(0)
    row = rng.standard_normal(n)
    keys = sorted(grid)
    for combo in itertools.product(*(grid[k] for k in keys)):
        for k, v in zip(keys, combo):
            os.environ[k] = v
        ctx = esq.DeviceContext(n, s + 1)
        ctx.set_tableau(cls.A, cls.B, cls.C, cls.E, 0)
        ctx.set_tol(1e-6, 1e-9)
        ctx.upload(SLOT_Y, 0, row)
        for r in range(s + 1):
            ctx.upload(SLOT_K, r, row)
        lib, h = ctx.lib, ctx.handle
        out = {"settings": dict(zip(keys, combo)), "stages": []}
        tot_ms = tot_b = 0.0
        for i in range(1, s):
            for _ in range(3):
                lib.esq_rk_stage_accumulate(h, i, 1e-3)
            ctx.profile_reset()
            ctx.profile_enable([PROF_STAGE])
            for _ in range(reps):
                lib.esq_rk_stage_accumulate(h, i, 1e-3)
            ctx.profile_enable(None)
            ms, cnt, by = ctx.profile_read(PROF_STAGE)
            nnz = int(np.count_nonzero(cls.A[i, :i]))
            out["stages"].append({"i": i, "nnz": nnz, "us": 1e3 * ms / cnt,
                                  "gbs": by / ms / 1e6})
            tot_ms += ms / cnt
            tot_b += by / cnt
        out["sum_us"] = 1e3 * tot_ms
        out["gbs"] = tot_b / tot_ms / 1e6
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
