#!/usr/bin/env python3
"""Per-kernel resource usage of the shipped build, read from the <name>.res files the Makefile leaves next to every object
(hipcc -Rpass-analysis=kernel-resource-usage): kernel (demangled) -> VGPRs, AGPRs, spilled VGPRs / SGPRs, scratch bytes per lane,
occupancy, LDS.  `python tools/kernel_resources.py [pattern]` prints the kernels that spill (or match the pattern)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fusion_amd", "csrc")
FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.strip("\n").split("\n")


def load(csrc=CSRC):
    """{file: {demangled kernel name: {field: int}}}"""
    res = {}
    for path in sorted(glob.glob(os.path.join(csrc, "*.res"))):
        kernels, cur = [], None
        for line in open(path, errors="replace"):
            m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass-analysis=kernel-resource-usage\]", line)
            if not m:
                continue
            key, val = m.group(1).strip(), m.group(2)
            if key == "Function Name":
                cur = {"mangled": val}
                kernels.append(cur)
            elif cur is not None and key in FIELDS:
                cur[FIELDS[key]] = int(val)
        names = demangle([k["mangled"] for k in kernels]) if kernels else []
        res[os.path.basename(path)[:-4]] = {n: k for n, k in zip(names, kernels)}
    return res


def short(name):
    """fz::sort_rows_kernel<1024, 28, 2, false, true>(fz::SortArgs) -> sort_rows_kernel<1024, 28, 2, false, true>"""
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name) if not name.endswith(">") else name
    depth, cut = 0, len(name)
    for i, c in enumerate(name):          # cut the argument list: the first '(' at template depth 0
        if c == "<": depth += 1
        elif c == ">": depth -= 1
        elif c == "(" and depth == 0:
            cut = i
            break
    return name[:cut].replace("fz::", "")


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else None
    for f, ks in load().items():
        for n, k in ks.items():
            if (pat and pat in n) or (not pat and (k.get("vgpr_spill", 0) or k.get("scratch", 0))):
                print(f"{f:8s} {short(n):70s} vgprs {k.get('vgprs'):3d} agprs {k.get('agprs'):3d} vgpr_spill {k.get('vgpr_spill'):3d} "
                      f"scratch {k.get('scratch'):4d} B/lane occupancy {k.get('occupancy')} lds {k.get('lds')}")
