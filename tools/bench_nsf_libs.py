import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import _lib, ops
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
Q, N, S = 1024, 27942, 4
g = torch.Generator(device="cuda").manual_seed(0)
planes = []
for s in range(S):
    p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(torch.randn((Q, N), generator=g, device="cuda") * (s + 1) + s); planes.append(p)
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
for lib in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(lib)
    print(os.path.basename(lib), {n: round(timeit(lambda: ops.fuse_nsf(planes, None, [0.25] * S, n, out=out)), 4) for n in ("min-max", "z-score", "arctan")}, flush=True)
    import numpy as np
    P = 1001
    distr = [torch.quantile(p[:64].flatten()[:1 << 20].double(), torch.linspace(0, 1, P, device="cuda", dtype=torch.float64)).float() for p in planes]
    print(os.path.basename(lib), {n: round(timeit(lambda: ops.fuse_nsf(planes, None, [0.25] * S, n, distr, out=out)), 4)
                                  for n in ("percentile-rank", "normal-curve-equivalent")}, flush=True)
