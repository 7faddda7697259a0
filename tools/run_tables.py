#!/usr/bin/env python3
"""A few launches of the percentile-rank / NCE fusion at the table size the reference reads (S = 4, Q = 1024, N = 27,942, P = 27,943),
for rocprofv3 (tools/pmc_tables.sh).  Usage: python3 tools/run_tables.py [norm] [S]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fusion_amd import ops
from bench_kernels import lleqa_planes, quantile_table

norm = sys.argv[1] if len(sys.argv) > 1 else "percentile-rank"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
Q, N, P = 1024, 27942, 27943
g = torch.Generator(device="cuda").manual_seed(7)
planes = lleqa_planes(Q, N, g)[:S]
distr = [quantile_table(p, P) for p in planes]
out = ops.alloc_plane(Q, N, torch.float32, "cuda")
prep = ops.nsf_tables_prepare(distr, norm)
for _ in range(5):
    ops.fuse_nsf(planes, None, [1.0 / S] * S, norm, distr, out=out, tables=prep)
torch.cuda.synchronize()
print("ok", ops.last_tables_path, prep.search_info())
