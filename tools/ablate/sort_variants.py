#!/usr/bin/env python3
"""Ablation builds of sort.hip (diagnostics only).  Usage (build container): python tools/ablate/sort_variants.py
Then on the GPU box: python tools/sort_pass_cost.py fusion_amd/libfusion_hip.so tools/ablate/libfusion_abl_sort_<name>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fusion_amd", "csrc")
SRC = open(os.path.join(CSRC, "sort.hip")).read()
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math".split()

SW = "((X) ^ (((X) >> 5) & 31u))"
sw = lambda x: SW.replace("X", x)
VARIANTS = {
    # XOR-swizzled slot layout of the radix pass's exchange: bank = (slot ^ (slot >> 5)) & 31.  The striped read side
    # stays conflict-free (a 32-slot group is XORed with one constant); the scatter side's banks are as random as before.
    "swz": [("            exch[dst] = kw;\n", f"            exch[{sw('dst')}] = kw;\n"),
            ("            ks[i] = exch[(slot0 + i * 64)];\n            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);",
             f"            ks[i] = exch[{sw('(uint32_t)(slot0 + i * 64)')}];\n            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);"),
            ("            exch[meta[i] >> 16] = meta[i] & 0xffffu;\n            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);",
             f"            exch[{sw('(meta[i] >> 16)')}] = meta[i] & 0xffffu;\n            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);"),
            ("            meta[i] = exch[(slot0 + i * 64)];\n            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);",
             f"            meta[i] = exch[{sw('(uint32_t)(slot0 + i * 64)')}];\n            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);")],
}


def main():
    objs = [os.path.join(CSRC, f) for f in "util.o fuse.o tables.o score.o maxsim.o bm25.o sparse.o tune.o encoder.o".split()]
    for name, patches in VARIANTS.items():
        s = SRC
        for a, b in patches:
            assert s.count(a) == 1, (name, a, s.count(a))
            s = s.replace(a, b)
        src, obj = f"/tmp/sort_abl_{name}.hip", f"/tmp/sort_abl_{name}.o"
        open(src, "w").write(s)
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-c", src, "-o", obj])
        out = os.path.join(ROOT, "tools", "ablate", f"libfusion_abl_sort_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *objs, "-ldl"])
        print("built", out)


if __name__ == "__main__":
    main()
