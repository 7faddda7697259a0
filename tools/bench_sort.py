#!/usr/bin/env python3
"""Micro-benchmark of the row sort (and ablation builds under tools/ablate/). Usage: python tools/bench_sort.py [lib.so ...]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd import _lib, ops

def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def run(tag):
    Q, N = 1024, 27942
    g = torch.Generator(device="cuda").manual_seed(0)
    k32 = ops.alloc_plane(Q, N, torch.float32, "cuda"); k32.copy_(torch.rand((Q, N), generator=g, device="cuda") * 2 - 1)
    k64 = ops.alloc_plane(Q, N, torch.float64, "cuda"); k64.copy_(torch.rand((Q, N), generator=g, device="cuda", dtype=torch.float64))
    o, _, _ = ops.sort_rows_desc(k32, want_keys=False)
    res = {
        "f32 order+rank": timeit(lambda: ops.sort_rows_desc(k32, want_keys=False, want_rank=True)),
        "f32 order only": timeit(lambda: ops.sort_rows_desc(k32, want_keys=False, want_rank=False)),
        "f32 order+keys": timeit(lambda: ops.sort_rows_desc(k32, want_keys=True, want_rank=False)),
        "f64 order+rank": timeit(lambda: ops.sort_rows_desc(k64, want_keys=False, want_rank=True)),
        "f64 gathered order+keys": timeit(lambda: ops.sort_rows_desc(k64, init_order=o, want_keys=True)),
    }
    for n in (1000, 4096, 8192, 16384):
        kk = ops.alloc_plane(4096, n, torch.float32, "cuda"); kk.copy_(torch.rand((4096, n), generator=g, device="cuda"))
        res[f"f32 4096x{n} order+rank"] = timeit(lambda: ops.sort_rows_desc(kk, want_keys=False, want_rank=True))
    print(tag, {k: round(v, 4) for k, v in res.items()}, flush=True)

if __name__ == "__main__":
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    for p in libs:
        _lib._lib = None
        _lib.LIB_PATH = os.path.abspath(p)
        run(os.path.basename(p))
