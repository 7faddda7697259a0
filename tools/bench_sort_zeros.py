#!/usr/bin/env python3
"""What a BM25-like float64 row costs the row sort as a function of its share of exact zeros, and what the non-zero keys alone would cost:
the measurement behind the zero-compaction of the float64 ranking sort (round 6).  Usage: python tools/bench_sort_zeros.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def bm25_like(Q, N, zero_frac, g, dtype=torch.float64):
    x = torch.distributions.Gamma(0.8, 0.25).sample((Q, N)).to("cuda").to(dtype) + 0.01     # heavy-tailed positive scores
    z = torch.rand((Q, N), generator=g, device="cuda") < zero_frac
    x[z] = 0.0
    p = ops.alloc_plane(Q, N, dtype, "cuda"); p.copy_(x)
    return p

if __name__ == "__main__":
    Q, N = 1024, 27942
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    for zf in (0.0, 0.2, 0.4, 0.6, 0.8, 0.95):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            k = bm25_like(Q, N, zf, g, dt)
            out[f"{tag} zeros={zf}"] = round(timeit(lambda: ops.sort_rows_desc(k, want_keys=False, want_rank=True)), 4)
            if dt == torch.float64:
                out[f"{tag} zeros={zf} lexical"] = round(timeit(lambda: ops.sort_rows_desc(k, want_keys=False, want_rank=True, lexical=True)), 4)
    for n in (5600, 11200, 16384, 22400):
        k = bm25_like(Q, n, 0.0, g)
        out[f"f64 non-zero keys only n={n}"] = round(timeit(lambda: ops.sort_rows_desc(k, want_keys=False, want_rank=True)), 4)
    print(json.dumps(out, indent=1))
