#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_*)
into small tracked files under profiles/ and print the roofline figure they give:
    profiles/rNN_kernel_stats_<config>.csv   rocprofv3 --kernel-trace --stats summary
    profiles/rNN_pmc_traffic_<config>.json   per-kernel HBM traffic (PMC passes) +
                                             the dominant class's bytes, time, rate
    profiles/rNN_bench_<config>.json         bench.py's JSON line of the same command
Usage: python tools/summarize_profiles.py r02 [config ...]      (default: pr8)

FETCH_SIZE is doubled (gfx950 reports exactly half of the bytes of wide coalesced
reads, MI355X_MICROARCH.md §HBM); WRITE_SIZE is exact; both are in KiB.

Kernel names are folded onto the labels of bench.py's `roofline.kernels` table so
that the two sources can be laid side by side: the figure printed here
(PMC bytes / rocprof kernel time / 8 TB/s) must lie between bench.py's
`roofline.frac` (compulsory bytes / event time / 8 TB/s) and the same with the
designed L2-side bytes (halo re-reads included).
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
PEAK = 8.0e12

STAGE = ("k_lincomb", "k_block_acc", "rhs+stage", "rhs1+stage", "rhs+block") + tuple(
    f"chain{d}{suffix}" for d in range(2, 7) for suffix in ("", "+solerr", "+errnorm", "+pre"))
RKC = ("rhs_rkc", "k_rkc_first", "k_rkc_stage") + tuple(f"rkc_chain{d}" for d in range(2, 9))


def label(name):
    """rocprof kernel name -> bench.py kernel label"""
    # marching chain sweeps (round 3): k_chain2d<NF, PERIODIC, D, NU, KINDLAST, Fn>
    m = re.search(r"k_chain2d<\d+, (?:true|false), (\d+), (\d+), (\d+)", name)
    if m:
        # (kind 3 also carries an early estimate -- "+pre" for bench.py: folded below;
        # kind 4, round 6: the chain runs through the end of an FSAL step)
        sol = {"3": "+solerr", "4": "+errnorm"}.get(m.group(3), "")
        return f"chain{m.group(1)}{sol}<{m.group(2)}>"
    # 3-D chain sweeps of the explicit pairs (round 5): k_chain3d<D, NU, JT, NW, KINDLAST, St>
    m = re.search(r"k_chain3d<(\d+), (\d+), \d+, \d+, (\d+)", name)
    if m:
        sol = "+solerr" if m.group(3) == "3" else ""
        return f"chain{m.group(1)}{sol}<{m.group(2)}>"
    m = re.search(r"k_rkc3d_chain<(\d+)", name)
    if m:                                   # 3-D Chebyshev chain sweeps (round 4)
        return f"rkc_chain{m.group(1)}"
    if re.search(r"_sweep<.*EpiRkcErr", name):
        return "rhs+rkcerr"
    if re.search(r"_sweep<.*EpiRkc\b", name):
        return "rhs_rkc"
    # one-stage sweeps of the 2-D stencil plugins (round 4: esq_stencil2d.hpp) and the
    # 3-D pair / row sweeps: k_stencil2d_sweep<NF, PERIODIC, Fn, Epi...<NT>, Src>
    m = re.search(r"k_(?:stencil2d_sweep|diff3d_pairs|diff3d_sweep|stencil3d_pairs|"
                  r"stencil3d_march|stencil3d_points)<.*?Epi(\w+?)(?:<(\d+)|[,>])", name)
    if m:
        kind = m.group(1).lower()
        if kind == "none":
            return "rhs_plugin"
        if kind == "rkcerr":
            return "rhs+rkcerr"
        if kind == "rkc":
            return "rhs_rkc"
        first = "1" if "SrcAxpy" in name else ""
        return f"rhs{first}+{kind}<{m.group(2)}>" if m.group(2) else f"rhs{first}+{kind}"
    m = re.search(r"k_(bruss2d|heat2d|diff3d|diag)_sweep<.*Epi(\w+)<(\d+)", name)
    if m:                                   # fused sweeps (round 2)
        first = "1" if "SrcAxpy" in name else ""
        return f"rhs{first}+{m.group(2).lower()}<{m.group(3)}>"
    m = re.search(r"k_diff3d_v2<\d+, (\d+)>", name)
    if m:
        return {"0": "rhs_plugin", "1": "rhs_rkc", "2": "rhs+rkcerr"}[m.group(1)]
    if re.search(r"k_(bruss2d|heat2d|diff3d|diag)", name):
        return "rhs_plugin"
    m = re.search(r"(k_[a-z0-9_]+)<(\d+)", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def mean_counter(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[label(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def one(tag, cfg):
    stats_csv = os.path.join(G, f"prof_{cfg}_stats", "bench_kernel_stats.csv")
    shutil.copy(stats_csv, os.path.join(P, f"{tag}_kernel_stats_{cfg}.csv"))
    bench = None
    try:
        with open(os.path.join(G, f"prof_{cfg}_bench.json")) as fh:
            bench = json.loads(fh.read().strip().splitlines()[-1])
        with open(os.path.join(P, f"{tag}_bench_{cfg}.json"), "w") as fh:
            json.dump(bench, fh, indent=1)
    except Exception as exc:                               # noqa: BLE001
        print(f"[{cfg}] no bench JSON: {exc}")
    fetch = mean_counter(os.path.join(G, f"prof_{cfg}_fetch",
                                      "bench_counter_collection.csv"))
    write = mean_counter(os.path.join(G, f"prof_{cfg}_write",
                                      "bench_counter_collection.csv"))
    stats = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats_csv)):
        if "rocclr" in r["Name"]:
            continue
        s = stats[label(r["Name"])]
        s[0] += int(r["Calls"])
        s[1] += float(r["TotalDurationNs"])
    # "pr8_7070": tools/profile_bench.sh pr8@7070; "pr8_diff3d[_400]": pr8:diff3d[@400]
    parts = cfg.split("_")
    base = parts[0]
    plugin = next((x for x in parts[1:] if not x.isdigit()), "")
    grid = next((x for x in parts[1:] if x.isdigit()), "")
    flags = (f"--config {base}" + (f" --plugin {plugin}" if plugin else "")
             + (f" --grid {grid}" if grid else ""))
    out = {"command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- "
                      f"python3 bench.py {flags} --steps 3 --warmup 1; "
                      f"kernel times: rocprofv3 --kernel-trace --stats -- python3 "
                      f"bench.py {flags} --steps {20 if not grid else 10} --warmup 5",
           "note": "bytes per launch; fetch = 2 x FETCH_SIZE KiB (gfx950 "
                   "correction), write = WRITE_SIZE KiB",
           "kernels": {}}
    dom = STAGE if base != "rkc" else RKC
    d_bytes = d_ns = d_calls = 0.0
    all_bytes = all_ns = 0.0
    # bench.py labels a chain sweep that does not write its K rows "...-K<n>" and a
    # step's last Chebyshev chain "...-last"; each is the same kernel for rocprofv3
    events = {}
    for k, v in (bench or {}).get("roofline", {}).get("kernels", {}).items():
        e = events.setdefault(k.replace("-K<", "<").replace("-last", "").replace("+pre", "+solerr"),
                              {"launches": 0, "us": 0.0, "moved": 0.0, "floor": 0.0})
        e["launches"] += v["launches"]
        e["us"] += v["avg_us"] * v["launches"]
        e["moved"] += v["moved_bytes_per_launch"] * v["launches"]
        e["floor"] += v.get("floor_bytes_per_launch", v["moved_bytes_per_launch"]) * v["launches"]
    for k in sorted(stats):
        calls, ns = stats[k]
        f_b = 2.0 * fetch.get(k, (0.0, 0))[0] * 1024.0
        w_b = write.get(k, (0.0, 0))[0] * 1024.0
        rec = {"calls_kernel_trace": calls, "avg_ns_kernel_trace": ns / calls,
               "fetch_bytes": f_b, "write_bytes": w_b, "hbm_bytes": f_b + w_b,
               "gbs": (f_b + w_b) / (ns / calls) if calls else None}
        if k in events and events[k]["launches"]:
            e = events[k]
            rec["bench_avg_us_events"] = e["us"] / e["launches"]
            rec["bench_designed_bytes"] = e["moved"] / e["launches"]
            rec["bench_floor_bytes"] = e["floor"] / e["launches"]
        out["kernels"][k] = rec
        all_bytes += (f_b + w_b) * calls
        all_ns += ns
        if k.split("<")[0] in dom:
            d_bytes += (f_b + w_b) * calls
            d_ns += ns
            d_calls += calls
    out["dominant_class"] = {
        "members": [k for k in out["kernels"] if k.split("<")[0] in dom],
        "hbm_bytes_per_launch": d_bytes / d_calls if d_calls else None,
        "avg_launch_ns_kernel_trace": d_ns / d_calls if d_calls else None,
        "gbs_pmc": d_bytes / d_ns if d_ns else None,
        "frac_of_8TBs_pmc": d_bytes / d_ns * 1e9 / PEAK if d_ns else None,
    }
    out["all_kernels"] = {"gbs_pmc": all_bytes / all_ns if all_ns else None,
                          "frac_of_8TBs_pmc": all_bytes / all_ns * 1e9 / PEAK
                          if all_ns else None}
    if bench:
        out["bench_roofline"] = {k: bench["roofline"].get(k) for k in (
            "achieved", "frac", "avg_launch_us", "moved_bytes_per_launch",
            "lower_bound_bytes_per_launch", "l2_side_gbs", "algorithmic_gbs",
            "whole_step_gbs")}
        out["bench_value"] = bench["value"]
        out["bench_ms_per_step"] = bench["ms_per_step"]
    with open(os.path.join(P, f"{tag}_pmc_traffic_{cfg}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"[{cfg}] dominant class from profiles/: "
          f"{json.dumps(out['dominant_class'])}")
    if bench:
        print(f"[{cfg}] bench.py roofline: {json.dumps(out['bench_roofline'])}")


def driver(tag):
    """the exact command the driver runs (tools/profile_bench.sh driver): kernel
    stats + the JSON line, no counter passes"""
    shutil.copy(os.path.join(G, "prof_driver_stats", "bench_kernel_stats.csv"),
                os.path.join(P, f"{tag}_kernel_stats_driver_command.csv"))
    with open(os.path.join(G, "prof_driver_bench.json")) as fh:
        bench = json.loads(fh.read().strip().splitlines()[-1])
    with open(os.path.join(P, f"{tag}_bench_driver_command.json"), "w") as fh:
        json.dump(bench, fh, indent=1)
    print(f"[driver] value {bench['value']:.4g}  ms/step {bench['ms_per_step']:.4f}  "
          f"frac {bench['roofline']['frac']:.3f}")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    os.makedirs(P, exist_ok=True)
    for cfg in (sys.argv[2:] or ["pr8"]):
        if cfg == "driver":
            driver(tag)
        else:
            one(tag, cfg)


if __name__ == "__main__":
    main()
