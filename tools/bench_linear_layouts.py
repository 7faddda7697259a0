"""hipBLASLt / rocBLAS fp32 GEMM rates at the packed encoder's shapes: TN (F.linear) vs NN (pre-transposed weight), M variants."""
import sys, torch
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
for lib in ("hipblaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for M in (36176, 36864, 32768):
        for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
            X = torch.randn((M, K), generator=g, device="cuda"); W = torch.randn((N, K), generator=g, device="cuda"); b = torch.randn(N, generator=g, device="cuda")
            Wt = W.t().contiguous()
            fl = 2.0 * M * N * K
            t_tn = timeit(lambda: torch.nn.functional.linear(X, W, b))
            t_nn = timeit(lambda: torch.addmm(b, X, Wt))
            print(f"{lib:9s} M={M:6d} N={N:5d} K={K:5d}  TN {t_tn:.3f} ms {fl/t_tn/1e9:6.1f} TF | NN {t_nn:.3f} ms {fl/t_nn/1e9:6.1f} TF", flush=True)
