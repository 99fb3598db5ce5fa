#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_*)
into small tracked files under profiles/:
    profiles/rNN_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary
    profiles/rNN_pmc_traffic.json      per-kernel HBM traffic from the PMC passes
Usage: python tools/summarize_profiles.py r01
FETCH_SIZE is doubled (gfx950 reports exactly half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact; both are in KiB.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(G, "prof_stats", "bench_kernel_stats.csv"),
                os.path.join(P, f"{tag}_kernel_stats.csv"))
    fetch = mean_counter(os.path.join(G, "prof_fetch", "bench_counter_collection.csv"))
    write = mean_counter(os.path.join(G, "prof_write", "bench_counter_collection.csv"))
    stats = {r["Name"]: r for r in csv.DictReader(
        open(os.path.join(G, "prof_stats", "bench_kernel_stats.csv")))}
    out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- "
                      "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "note": "bytes per launch; fetch = 2 x FETCH_SIZE KiB (gfx950 "
                   "correction), write = WRITE_SIZE KiB",
           "kernels": {}}
    st_bytes = st_launch = 0.0
    st_ns = st_calls = 0.0
    for k in sorted(fetch):
        if "rocclr" in k:
            continue
        f_b = 2.0 * fetch[k][0] * 1024.0
        w_b = write.get(k, (0.0, 0))[0] * 1024.0
        rec = {"fetch_bytes": f_b, "write_bytes": w_b, "hbm_bytes": f_b + w_b,
               "launches_sampled": fetch[k][1]}
        if k in stats:
            rec["avg_ns_kernel_trace"] = float(stats[k]["AverageNs"])
            rec["calls_kernel_trace"] = int(stats[k]["Calls"])
        out["kernels"][short(k)] = rec
        if ("k_lincomb" in k or "k_block_acc" in k or "_chain<" in k) and k in stats:
            st_bytes += (f_b + w_b) * int(stats[k]["Calls"])
            st_launch += int(stats[k]["Calls"])
            st_ns += float(stats[k]["TotalDurationNs"])
            st_calls += int(stats[k]["Calls"])
    out["stage_accumulate"] = {
        "hbm_bytes_per_launch": st_bytes / st_launch if st_launch else None,
        "avg_launch_ns_kernel_trace": st_ns / st_calls if st_calls else None,
    }
    with open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["stage_accumulate"]))


if __name__ == "__main__":
    main()
