import sys, json; sys.path.insert(0, "/root/repo")
import torch, numpy as np, bench
from fusion_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
Q, N = 1024, 27942
planes = [bench.rand_plane(ops, Q, N, g, s + 1.0, float(s)) for s in range(4)]
ranks, orders = [], []
for i, p in enumerate(planes):
    od, sk, rk = ops.sort_rows_desc(p, want_rank=True)
    if i == 3:
        k = int(0.6 * N)
        rk = torch.where(rk < k, rk, torch.full_like(rk, -1)); od = od.clone(); od[:, k:] = -1
    ranks.append(rk); orders.append(od)
lens = torch.full((4, Q), N, dtype=torch.int32, device="cuda"); lens[3] = int(0.6 * N)
bits = [None, None, None, ops.rank_to_bitmap(ranks[3])]
w = [0.25] * 4
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
for norm in ("min-max", "z-score", "arctan"):
    for name, kw in (("no validity", dict(ranks=None)), ("colbert rank plane", dict(ranks=[None, None, None, ranks[3]])),
                     ("colbert bitmap", dict(ranks=None, valid_bits=bits))):
        ms = bench.timeit_ms(lambda: ops.fuse_nsf(planes, kw.get("ranks"), w, norm, out=out, valid_bits=kw.get("valid_bits")), n=20)
        print(norm, name, round(ms, 4), "ms", flush=True)
for name, kw in (("no validity", dict(ranks=None)), ("colbert rank plane", dict(ranks=[None, None, None, ranks[3]])), ("colbert bitmap", dict(ranks=None, valid_bits=bits))):
    ms = bench.timeit_ms(lambda: ops.fuse_nsf(planes, kw.get("ranks"), w, "min-max", out=out, orders=orders, lens=lens, valid_bits=kw.get("valid_bits")), n=20)
    print("min-max (list ends + flat pass)", name, round(ms, 4), "ms", flush=True)
