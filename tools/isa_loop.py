#!/usr/bin/env python3
"""Instruction mix of a kernel's largest loop (the marching loop of a sweep):
    python tools/isa_loop.py <unit.hip> <mangled-name prefix> [extra hipcc flags ...]
compiles extensisq_amd/csrc/<unit> to gfx950 assembly (device only) and counts the
instructions between the loop's label and its backward branch, by mnemonic."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSRC = os.path.join(ROOT, "extensisq_amd", "csrc")


def kernel_body(txt, prefix):
    m = re.search(r"^(" + re.escape(prefix) + r"[^:\n]*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M)
    if not m:
        names = sorted(set(re.findall(r"^(_Z\w+):", txt, re.M)))
        raise SystemExit("no kernel starts with %r; kernels:\n  %s" % (
            prefix, "\n  ".join(n for n in names if "k_" in n)[:4000]))
    return m.group(1), m.group(2)


def main():
    unit, prefix = sys.argv[1], sys.argv[2]
    out = "/tmp/isa_%s.s" % os.path.basename(unit)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-ffp-contract=off", "--cuda-device-only", "-S", os.path.join(CSRC, unit), "-o",
           out] + sys.argv[3:]
    if not (os.environ.get("ISA_REUSE") and os.path.exists(out)):
        subprocess.run(cmd, check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    name, body = kernel_body(open(out).read(), prefix)
    lines = [ln.strip() for ln in body.split("\n")
             if ln.strip() and not ln.strip().startswith((";", ".amd", ".p2align", ".section"))]
    lines = [re.sub(r"\s*;.*$", "", ln) for ln in lines]
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if re.match(r"^\.?LBB\d+_\d+:$", ln)}
    best = None
    for i, ln in enumerate(lines):
        m = re.match(r"s_c?branch\w* (\.?LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    loop = [ln for ln in lines[best[0]:best[1] + 1] if not ln.endswith(":")]
    count = collections.Counter(ln.split()[0] for ln in loop)
    valu = sum(v for k, v in count.items() if k.startswith("v_"))
    print(name[:100])
    print(f"instructions in the kernel {len(lines)}, in its largest loop {len(loop)}: "
          f"vector ALU {valu}, scalar {sum(v for k, v in count.items() if k.startswith('s_'))}, "
          f"memory {sum(v for k, v in count.items() if k.startswith(('buffer_', 'global_', 'ds_', 'scratch_', 'flat_')))}")
    for k, v in count.most_common(int(os.environ.get("ISA_TOP", "45"))):
        print(f"{v:6d} {k}")


if __name__ == "__main__":
    main()
