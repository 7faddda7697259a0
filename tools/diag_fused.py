"""The filter-epilogue GEMM alone on one 229,376-document chunk (thresholds at a given survivor rate), per library build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fusion_amd import _lib, ops
from tools.diag_gemm import timeit

Q, N, d, cap = 1024, 229376, 768, 7168
g = torch.Generator(device="cuda").manual_seed(0)
Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
S = ops.dot_scores(Qn, Dn)
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
for lib in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(lib)
    L = _lib.lib()
    ms0 = timeit(lambda: ops.dot_scores(Qn, Dn, out=out), n=10)
    for rate in (0.001, 0.0044):
        kth = max(1, int(rate * N))
        tau = torch.full((1024,), float("inf"), device="cuda"); tau[:Q] = torch.topk(S, kth, dim=1).values[:, -1]
        cs = torch.empty((Q, cap), device="cuda"); ci = torch.empty((Q, cap), dtype=torch.int64, device="cuda")
        ln = torch.zeros(Q, dtype=torch.int32, device="cuda"); ov = torch.zeros(1, dtype=torch.int32, device="cuda")
        def f():
            ln.zero_()
            ops.check(L.fz_dot_scores_filter_f32(ops._ptr(Qn), Qn.stride(0), ops._ptr(Dn), Dn.stride(0), Q, N, d, 0, ops._ptr(tau), ops._ptr(cs), ops._ptr(ci),
                                                 ops._ptr(ln), cap, ops._ptr(ov), None), "gemm filter")
        ms = timeit(f, n=10)
        print(f"{os.path.basename(lib)}: plain {ms0:.3f} ms | filter epilogue at {rate}: {ms:.3f} ms (+{100 * (ms / ms0 - 1):.1f} %)  mean candidates {float(ln.float().mean()):.0f}", flush=True)
