import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import ops
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
for M in (36864, 8192):
    for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        X = torch.randn((M, K), generator=g, device="cuda"); W = torch.randn((N, K), generator=g, device="cuda"); b = torch.randn(N, generator=g, device="cuda")
        out = ops.alloc_plane(M, N, torch.float32, "cuda")
        t_mine = timeit(lambda: ops.dot_scores(X, W, out=out))
        t_blas = timeit(lambda: torch.nn.functional.linear(X, W, b))
        fl = 2.0 * M * N * K
        err = (ops.dot_scores(X[:256], W) - X[:256] @ W.t()).abs().max().item()
        print(f"M={M:6d} N={N:5d} K={K:5d}  mine {t_mine:.3f} ms {fl/t_mine/1e9:6.1f} TF | hipBLASLt {t_blas:.3f} ms {fl/t_blas/1e9:6.1f} TF | maxdiff {err:.2e}", flush=True)
