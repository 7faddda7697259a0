import sys, json; sys.path.insert(0, "/root/repo")
import torch, numpy as np, bench
from fusion_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
Q, N = 1024, 27942
planes = [bench.rand_plane(ops, Q, N, g, s + 1.0, float(s)) for s in range(4)]
ranks = []
for i, p in enumerate(planes):
    od, sk, rk = ops.sort_rows_desc(p, want_rank=True)
    if i == 3:
        rk = torch.where(rk < int(0.6 * N), rk, torch.full_like(rk, -1))
    ranks.append(rk)
w = [0.25] * 4
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
for norm in ("min-max", "z-score", "arctan"):
    for name, r in (("no validity", None), ("colbert rank plane", [None, None, None, ranks[3]]), ("all rank planes", ranks)):
        ms = bench.timeit_ms(lambda: ops.fuse_nsf(planes, r, w, norm, out=out), n=20)
        nplanes = 5 + (0 if r is None else sum(x is not None for x in r))
        print(norm, name, round(ms, 4), "ms", round(nplanes * Q * N * 4 / ms / 1e6, 0), "GB/s actual-traffic", flush=True)
