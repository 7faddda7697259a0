"""Threshold-filter kernel alone: one 229,376-column chunk, 1024 rows, thresholds at a given survivor rate (timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fusion_amd import _lib, ops
from tools.diag_gemm import timeit

rows, n, k, cap = 1024, 229376, 1000, 7168
g = torch.Generator(device="cuda").manual_seed(0)
S = ops.as_plane(torch.rand((rows, n), generator=g, device="cuda"))
for lib in sys.argv[1:]:
    _lib._lib = None; _lib.LIB_PATH = os.path.abspath(lib)
    L = _lib.lib()
    for rate in (0.0, 0.001, 0.004, 0.0076):
        tau = torch.full((rows,), 1.0 - rate, device="cuda")
        cs = torch.empty((rows, cap), device="cuda"); ci = torch.empty((rows, cap), dtype=torch.int64, device="cuda")
        ln = torch.zeros(rows, dtype=torch.int32, device="cuda"); ov = torch.zeros(1, dtype=torch.int32, device="cuda")
        def f():
            ln.zero_()
            ops.check(L.fz_topk_filter_append_f32(ops._ptr(S), rows, n, S.stride(0), 0, ops._ptr(tau), ops._ptr(cs), ops._ptr(ci), ops._ptr(ln), cap, ops._ptr(ov), None), "filter")
        ms = timeit(f, n=20)
        print(f"{os.path.basename(lib)} rate {rate}: {ms:.3f} ms  {rows * n * 4 / ms / 1e6:.0f} GB/s  overflow {int(ov.item())}", flush=True)
