#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch / occupancy / LDS of the kernels of one translation unit:
    python tools/kernel_resources.py esq_rhs_diff3d.hip [name filter] [extra hipcc flags ...]
(hipcc -Rpass-analysis=kernel-resource-usage on extensisq_amd/csrc/<unit>; no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSRC = os.path.join(ROOT, "extensisq_amd", "csrc")


def main():
    unit = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c",
           os.path.join(CSRC, unit), "-o", "/dev/null"] + extra
    txt = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    if "error:" in txt:
        print(txt[-3000:])
        sys.exit(1)
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].strip()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and flt not in dem:
            continue

        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append((dem.split("(")[0][:110], g("VGPRs"), g("AGPRs"), g("SGPRs"),
                     g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g(r"LDS Size \[bytes/block\]")))
    print(f"{'kernel':110s} VGPR AGPR SGPR scratch occ LDS")
    for r in sorted(rows):
        print(f"{r[0]:110s} {r[1]:4d} {r[2]:4d} {r[3]:4d} {r[4]:7d} {r[5]:3d} {r[6]}")


if __name__ == "__main__":
    main()
