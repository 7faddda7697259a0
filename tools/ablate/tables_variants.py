#!/usr/bin/env python3
"""Ablation builds of tables.hip (diagnostics only; wrong results by design): each variant is a textual patch of the shipped source
compiled into tools/ablate/libfusion_abl_<name>.so.  Usage (build container): python tools/ablate/tables_variants.py
Then on the GPU box: python tools/run_tables_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fusion_amd", "csrc")
SRC = open(os.path.join(CSRC, "tables.hip")).read()
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math".split()

T1024 = "constexpr int BT_T = 1024;"
E7 = "constexpr int BT_E4 = 7;"
R1 = "constexpr int BT_R = 1;"
I1A, I1B = "fuse_nsf_bigtab_kernel<false, 1, BT_R>", "fuse_nsf_bigtab_kernel<true, 1, BT_R>"
VARIANTS = {
    # 8 waves x 256 VGPRs: two items per table residency (half the swaps), one or two float4 searched in lockstep
    "t512_r2": [(T1024, "constexpr int BT_T = 512;"), (E7, "constexpr int BT_E4 = 14;"), (R1, "constexpr int BT_R = 2;")],
    "t512_r2_ilv2": [(T1024, "constexpr int BT_T = 512;"), (E7, "constexpr int BT_E4 = 14;"), (R1, "constexpr int BT_R = 2;"),
                     (I1A, "fuse_nsf_bigtab_kernel<false, 2, BT_R>"), (I1B, "fuse_nsf_bigtab_kernel<true, 2, BT_R>")],
    "ilv2": [(I1A, "fuse_nsf_bigtab_kernel<false, 2, BT_R>"), (I1B, "fuse_nsf_bigtab_kernel<true, 2, BT_R>")],
    "nosched": [("                        __builtin_amdgcn_sched_barrier(0);   // one group's searches at a time", "                        // (no sched barrier)")],
    "lut8k": [("    for (int lutb = 16384; lutb >= 2048 && !p.ok; lutb >>= 1) {", "    for (int lutb = 8192; lutb >= 2048 && !p.ok; lutb >>= 1) {")],
    "t512_ilv1": [(T1024, "constexpr int BT_T = 512;"), (E7, "constexpr int BT_E4 = 14;")],
    # no HBM reads of scores after the first step (lookups + swaps + stores only)
    "nostream": [("        return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(base + min(my_off() + 4 * BT_T * i, lim)));   // streamed once",
                  "        f4v r = {1.f, 2.f, 3.f, 4.f}; asm volatile(\"\" : \"+v\"(r)); return r;")],
    # no searches at all: streaming, swaps and accumulation only
    "nolookup": [("                bt_lookup<W, STEPS>(tab, lut, lo_v, inv_w, top, steps, last_pair, x, best);",
                  "                for (int e = 0; e < W; ++e) best[e] = (int)x[e] & 1023;")],
    # no table swaps (the first system's table stays): what the LDS-DMA phases cost
    "noswap": [("            const bool swap = cur != s;", "            const bool swap = cur < 0;")],
}


def main():
    os.makedirs(os.path.join(ROOT, "tools", "ablate"), exist_ok=True)
    objs = [os.path.join(CSRC, f) for f in "util.o fuse.o sort.o score.o maxsim.o bm25.o sparse.o tune.o encoder.o".split()]
    for name, patches in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        s = SRC
        for a, b in patches:
            assert s.count(a) >= 1, (name, a)
            s = s.replace(a, b)
        src = f"/tmp/tables_abl_{name}.hip"
        open(src, "w").write(s)
        obj = f"/tmp/tables_abl_{name}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, "-I", CSRC, "-c", src, "-o", obj])
        out = os.path.join(ROOT, "tools", "ablate", f"libfusion_abl_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj, *objs, "-ldl"])
        print("built", out)


if __name__ == "__main__":
    main()
