import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fusion_amd import ops
from tools.bench_kernels import timeit, rand_plane
Q, N, S = 1024, 27942, 4
g = torch.Generator(device="cuda").manual_seed(0)
planes = [rand_plane(Q, N, g, s + 1, s) for s in range(S)]
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
w = [0.25] * S
for P in (101, 1001, 28001):
    distr = [torch.quantile(p[:8].flatten()[:1000000].double(), torch.linspace(0, 1, P, device="cuda", dtype=torch.float64)).float().contiguous() for p in planes]
    for norm in ("percentile-rank", "normal-curve-equivalent"):
        ms = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, distr, out=out))
        print(norm, "P =", P, round(ms, 4), "ms", round((S + 1) * Q * N * 4 / ms / 1e6, 1), "GB/s")
