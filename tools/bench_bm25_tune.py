#!/usr/bin/env python3
"""The BM25 lexical module on the device at LLeQA's shape (27,942 articles of ~150 lemmas, the dev split's 201 questions): index build,
search_device, and the k1 x b grid search of bm25.py:221-237 (17 x 11 = 187 pairs) as a device sweep (BM25.tune) -- against the same search
with the per-posting float64 expression (USE_POSTING_VALUES = False).  Usage: python tools/bench_bm25_tune.py [Q]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import ops
from fusion_amd.retrievers.bm25 import BM25


def main():
    Q = int(sys.argv[1]) if len(sys.argv) > 1 else 201
    N, V = 27942, 20000
    rng = np.random.default_rng(0)
    p = 1.0 / np.arange(30, V + 30) ** 1.05; p /= p.sum()          # a Zipf vocabulary WITHOUT its 30 most frequent (stop-word-like) ranks
    vocab = np.array([f"m{i}" for i in range(V)])
    lens = np.clip(rng.normal(150, 60, N), 16, 512).astype(np.int64)
    toks = vocab[rng.choice(V, size=int(lens.sum()), p=p)]
    off = np.concatenate([[0], np.cumsum(lens)])
    docs = [" ".join(toks[off[i]:off[i + 1]]) for i in range(N)]
    queries = [" ".join(vocab[rng.choice(V, size=int(rng.integers(4, 12)), p=p)]) for _ in range(Q)]
    gold = [sorted(rng.choice(N, size=int(rng.integers(1, 5)), replace=False).tolist()) for _ in range(Q)]
    t0 = time.perf_counter(); m = BM25(docs, 2.5, 0.2); torch.cuda.synchronize(); t_index = time.perf_counter() - t0

    def timed(f, n=5):
        f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    res = dict(Q=Q, N=N, index_build_s=round(t_index, 2), postings=int(m.pdoc.numel()))
    ops.sort_zero_compact_rows(reset=True)
    rs = m.search_device(queries)
    res["rows_compacted_by_the_ranking_sort"] = ops.sort_zero_compact_rows(reset=True)[0]
    res["zero_share_mean"] = float((rs.scores64 == 0).double().mean().item())
    res["search_device_ms"] = round(timed(lambda: m.search_device(queries)), 3)
    m.USE_POSTING_VALUES = False
    res["search_device_ms_per_posting_expression"] = round(timed(lambda: m.search_device(queries)), 3)
    m.USE_POSTING_VALUES = True
    t0 = time.perf_counter(); rows = m.tune(queries, gold); torch.cuda.synchronize(); res["grid_search_187_pairs_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    t0 = time.perf_counter(); rows = m.tune(queries, gold); torch.cuda.synchronize(); res["grid_search_187_pairs_ms_second_run"] = round((time.perf_counter() - t0) * 1e3, 1)
    res["best"] = max(rows, key=lambda r: r["recall@100"])
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
