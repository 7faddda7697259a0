#!/usr/bin/env python3
"""Pieces of the top-k form of the final ordering (ops.select_topk) against the full row sort, Q = 1024 x N = 27,942, k = 1000."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd import _lib, ops
from tools.bench_kernels import timeit

Q, N, k = 1024, 27942, 1000
g = torch.Generator(device="cuda").manual_seed(0)
for dt in (torch.float32, torch.float64):
    fused = ops.alloc_plane(Q, N, dt, "cuda"); fused.copy_(torch.randn((Q, N), generator=g, device="cuda").to(dt))
    _, _, rank = ops.sort_rows_desc(ops.alloc_plane(Q, N, torch.float32, "cuda").copy_(torch.randn((Q, N), generator=g, device="cuda")), want_rank=True)
    cap = 2048
    cols = ops.alloc_plane(Q, cap, torch.int32, "cuda"); vals = ops.alloc_plane(Q, cap, dt, "cuda"); negp = ops.alloc_plane(Q, cap, torch.float32, "cuda")
    clen = torch.empty(Q, dtype=torch.int32, device="cuda"); over = torch.zeros(1, dtype=torch.int32, device="cuda")
    L = _lib.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sel = lambda: L.fz_select_topk_f(P(fused), 32 if dt == torch.float32 else 64, P(rank), Q, N, fused.stride(0), k, cap, P(cols), P(vals), P(negp), P(clen), P(over), st)
    rec = {"dtype": str(dt), "select_kernel_ms": timeit(sel)}
    rec["sort_by_pos_ms"] = timeit(lambda: ops.sort_rows_desc(negp, row_len=clen, want_keys=False))
    bp, _, _ = ops.sort_rows_desc(negp, row_len=clen, want_keys=False)
    rec["sort_by_score_ms"] = timeit(lambda: ops.sort_rows_desc(vals, init_order=bp, row_len=clen))
    rec["select_topk_total_ms"] = timeit(lambda: ops.select_topk(fused, rank, k))
    rec["full_sort_placed_ms"] = timeit(lambda: ops.sort_rows_desc(fused, init_rank=rank))
    rec["mean_candidates"] = float(clen.float().mean())
    print(json.dumps(rec), flush=True)
