#!/usr/bin/env python3
"""Times the percentile-rank fusion (S = 4, Q = 1024, N = 27,942, P = 27,943) with the shipped library and with every ablation
build under tools/ablate/ (tools/ablate/tables_variants.py), each in its own process (FUSION_AMD_LIB), alternating, 3 rounds."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from fusion_amd import ops
from bench_kernels import lleqa_planes, quantile_table, timeit
Q, N, P, S = 1024, 27942, 27943, 4
g = torch.Generator(device="cuda").manual_seed(7)
planes = lleqa_planes(Q, N, g)
distr = [quantile_table(p, P) for p in planes]
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
res = {}
for norm in ("percentile-rank", "normal-curve-equivalent"):
    prep = ops.nsf_tables_prepare(distr, norm)
    res[norm] = round(timeit(lambda: ops.fuse_nsf(planes, None, [0.25] * S, norm, distr, out=out, tables=prep), n=20), 4)
prep = ops.nsf_tables_prepare(distr[:1], "percentile-rank")
res["pr_S1"] = round(timeit(lambda: ops.fuse_nsf(planes[:1], None, [1.0], "percentile-rank", distr[:1], out=out, tables=prep), n=20), 4)
print(res)
''' % (ROOT, ROOT)
libs = {"shipped": os.path.join(ROOT, "fusion_amd", "libfusion_hip.so")}
for f in sorted(glob.glob(os.path.join(ROOT, "tools", "ablate", "libfusion_abl_*.so"))):
    libs[os.path.basename(f)[len("libfusion_abl_"):-3]] = f
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for name, path in libs.items():
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, FUSION_AMD_LIB=path), capture_output=True, text=True)
        print(json.dumps({"build": name, "round": rnd, "ms": r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-300:]}), flush=True)
