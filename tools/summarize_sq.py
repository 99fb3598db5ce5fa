#!/usr/bin/env python3
"""condense the SQ counter passes of tools/r06_sq_counters.sh (gpurun_out/prof_<cfg>_sq*/)
into profiles/r06_sq_counters.json: per kernel the per-launch averages of every counter and
the launch duration under the profiler."""
import collections
import csv
import json
import os
import sys

root = os.path.join(os.path.dirname(__file__), "..")
out = {}
for cfg in sys.argv[1:] or ["ts5", "pr8", "rkc"]:
    for suf in ("sq", "sq2"):
        path = os.path.join(root, "gpurun_out", f"prof_{cfg}_{suf}", "bench_counter_collection.csv")
        if not os.path.exists(path):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        dur = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in agg.items():
            if "esq::" not in k:
                continue
            n = len(dur[k])
            e = out.setdefault(cfg, {}).setdefault(k, {})
            e.setdefault("launches", {})[suf] = n
            e.setdefault("avg_ns_under_profiler", {})[suf] = sum(dur[k].values()) / n
            for c, x in v.items():
                e[c] = x / n
json.dump(out, open(os.path.join(root, "profiles", "r06_sq_counters.json"), "w"), indent=1, sort_keys=True)
print("profiles/r06_sq_counters.json:", {c: len(v) for c, v in out.items()})
