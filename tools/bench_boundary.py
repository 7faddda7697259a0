#!/usr/bin/env python3
"""Host cost of the reference-typed boundary of Aggregator.fuse (hybrid.py:66-75,170-220): packing the reference's
list[Q] of list[<= N] of {'corpus_id', 'score'} into planes (pack_ranked_lists) and turning a fused result back into it
(FusedResult.to_lists) -- no device involved.  S = 4, N = 27,942, ColBERT lists cut to 60 %.  Usage: python tools/bench_boundary.py [Q]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd.planes import FusedResult
from fusion_amd.retrievers.hybrid import pack_ranked_lists


def synth_lists(Q, N=27942, S=4, seed=0):
    rng = np.random.default_rng(seed)
    ids = rng.permutation(np.arange(1, 4 * N + 1))[:N]
    lists = {}
    for s in range(S):
        per = []
        for q in range(Q):
            sc = rng.normal(size=N).astype(np.float32); o = np.argsort(-sc)
            keep = N if s < S - 1 else int(.6 * N)
            per.append([{"corpus_id": int(ids[i]), "score": float(sc[i])} for i in o[:keep]])
        lists[f"s{s}"] = per
    return lists


def best_of(f, n=3):
    best = None
    for _ in range(n):
        t = time.perf_counter(); r = f(); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, r


def main():
    Q = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lists = synth_lists(Q)
    t_pack, (ids, N, packed) = best_of(lambda: pack_ranked_lists(lists))
    sc64, rk, od, ln, _ = packed["s0"]
    out = {"Q": Q, "N": N, "S": len(lists), "pack_ms_per_query": round(t_pack / Q * 1e3, 2)}
    for name, sc in (("float32 (nsf)", sc64[:, :N].astype(np.float32)), ("float64 (rrf / bcf / none)", sc64[:, :N].copy())):
        fr = FusedResult(order=torch.from_numpy(od[:, :N].copy()), scores=torch.from_numpy(sc), lens=torch.from_numpy(ln), ids=ids)
        t, L = best_of(fr.to_lists)
        out[f"unpack_ms_per_query {name}"] = round(t / Q * 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
