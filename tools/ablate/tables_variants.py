#!/usr/bin/env python3
"""Ablation builds of tables.hip (diagnostics only; wrong results by design): each variant is a textual patch of the shipped source
compiled into tools/ablate/libfusion_abl_<name>.so.  Usage (build container): python tools/ablate/tables_variants.py
Then on the GPU box: python tools/run_tables_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "fusion_amd", "csrc")
SRC = open(os.path.join(CSRC, "tables.hip")).read()
FLAGS = "-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math".split()

VARIANTS = {
    # the value: one multiply instead of int -> double -> multiply -> float
    "novalue": [("const float tr = (float)((double)best[e] * invP);", "const float tr = (float)best[e] * 3.5e-5f;")],
    # no neighbour decision: the count alone
    "nofinal": [("    float tl[4], th[4], tm[4];", "    for (int e = 0; e < 4; ++e) best[e] = pos[e];\n    return;\n    float tl[4], th[4], tm[4];")],
    # no probes: the bucket start is the answer
    "noprobe": [("        for (int st = STEPS - 1; st >= 0; --st)\n#pragma unroll\n            for (int e = 0; e < 4; ++e) pos[e] +=",
                 "        for (int st = -1; st >= 0; --st)\n#pragma unroll\n            for (int e = 0; e < 4; ++e) pos[e] +=")],
    # no LDS at all in the search: streaming, swaps and accumulation only
    "nolookup": [("        bt_lookup4<STEPS>(tab, lut, lo_v, inv_w, top, gsteps, v[i], best);",
                  "        for (int e = 0; e < 4; ++e) best[e] = (int)v[i][e];")],
    # no HBM reads of scores after the first step (lookups + swaps + stores only)
    "nostream": [("            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim)));   // streamed once\n            v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;",
                  "            v[i][0] += 1e-3f;")],
    # the HBM reads are issued but their data is never used (no register dependence on them): is it the waits or the traffic?
    "streamdiscard": [("            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim)));   // streamed once\n            v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;",
                       "            { f4v f; asm volatile(\"global_load_dwordx4 %0, %1, off nt\" : \"=v\"(f) : \"v\"(nxt + min(toff + 4 * BT_T * i, nlim))); }\n            v[i][0] += 1e-3f;")],
    # every prefetch reads row 0 of plane 0 (served by L2): same instructions, no HBM traffic
    "streaml2": [("            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim)));",
                  "            const f4v f = *reinterpret_cast<const f4v*>(x00 + min(toff + 4 * BT_T * i, nlim));"),
                 ("                                          const float* __restrict__ nxt, int nlim, int toff) {", "                                          const float* __restrict__ nxt, int nlim, int toff, const float* x00) {"),
                 ("idx, nxt, nlim, toff); break;", "idx, nxt, nlim, toff, a.planes[0]); break;")],
    # regular (temporal) loads instead of nt
    "streamtemporal": [("            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim)));",
                        "            const f4v f = *reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim));")],
    # neither swaps nor streaming: the lookups alone (+ stores)
    "lookuponly": [("            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(nxt + min(toff + 4 * BT_T * i, nlim)));   // streamed once\n            v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;",
                    "            v[i][0] += 1e-3f;"), ("            const bool swap = cur != s;", "            const bool swap = cur < 0;")],
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
