#!/usr/bin/env python3
"""idle time between consecutive kernels of a rocprofv3 kernel trace, by (previous kernel ->
next kernel): python tools/gap_report.py <bench_kernel_trace.csv> [skip_fraction]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4


def short(n):
    m = re.search(r"k_chain2d<(\d), \w+, (\d), (\d), (\d)", n)
    if m:
        return "chain2d D=%s NU=%s kind=%s" % (m.group(2), m.group(3), m.group(4))
    m = re.search(r"k_rkc3d_chain<(\d), \d+, \d+,.*?(true|false), (true|false)>", n)
    if m:
        return "rkc3d D=%s first=%s last=%s" % m.groups()
    m = re.search(r"(k_\w+)", n)
    return m.group(1) if m else n[:40]


ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ks = ks[int(len(ks) * skip):]
gaps = collections.defaultdict(list)
busy = 0
for (pn, ps, pe), (n, s, e) in zip(ks, ks[1:]):
    gaps[(pn, n)].append((s - pe) / 1e3)
    busy += e - s
span = ks[-1][2] - ks[0][1]
print("kernels %d, span %.1f us, busy %.1f %%" % (len(ks), span / 1e3, 100.0 * busy / span))
for (pn, n), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print("  %-34s -> %-34s x%-4d mean gap %6.2f us  max %7.2f" % (pn, n, len(g), sum(g) / len(g), max(g)))
