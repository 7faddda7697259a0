#!/usr/bin/env python3
"""Diagnostic: the fp64 row sort on the bench's BM25 score plane -- time, rows flagged for the generic launch, run statistics."""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from fusion_amd import ops, _lib
args = types.SimpleNamespace(queries=1024, corpus=27942, dim=768, no_encode=True, encoder_size="base", encode_buckets=8, encode_mode="packed",
                             overlap_bm25=False, no_gemm_tuning=True)
st = bench.build_lleqa(args, torch.device("cuda", 0), 0)
b = st["bm25"]; Q, N = st["Q"], st["N"]
B = ops.bm25_scores(b["toff"], b["pdoc"], b["ptf"], b["idf"], b["doc_len"], b["avgdl"], 2.5, 0.2, b["qoff"], b["qterms"], Q, N, doc_norm=b["doc_norm"])
ms = bench.timeit_ms(lambda: ops.sort_rows_desc(B, want_keys=False, want_rank=True), n=10)
print("bm25 sort ms", round(ms, 4))
lib = _lib.lib()
order = torch.empty((Q, B.stride(0)), dtype=torch.int32, device="cuda"); rank = torch.empty_like(order)
ws = torch.full((Q,), -7, dtype=torch.int32, device="cuda")
rc = lib.fz_sort_rows_desc(ops._ptr(B), 64, None, None, Q, N, B.stride(0), ops._ptr(order), None, ops._ptr(rank), None, None, ops._ptr(ws), Q * 4, ops._stream(B))
torch.cuda.synchronize()
f = ws.cpu().numpy(); print("rc", rc, "flag values", np.unique(f, return_counts=True))
h = B.cpu().numpy()
hi = (h.view(np.uint64) >> 32).astype(np.uint32)
r = 0
for r in (0, 1, 2):
    v, c = np.unique(hi[r], return_counts=True)
    d = [len(np.unique(h[r][hi[r] == x])) for x in v[c > 1]]
    print("row", r, "zeros", int((h[r] == 0).sum()), "neg", int((h[r] < 0).sum()), "distinct hi", len(v), "multi-hi groups", int((c > 1).sum()),
          "max group", int(c.max()), "groups with >1 distinct value", int(sum(x > 1 for x in d)), "max distinct in a group", max(d) if d else 0,
          "groups >17 with >1 distinct", int(sum(1 for x, cc in zip(d, c[c > 1]) if x > 1 and cc > 17)))
