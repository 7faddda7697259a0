"""Evaluation metrics with the reference's exact semantics (src/utils/metrics.py:25-162), including its
non-standard nDCG (position 0 undiscounted, 1/log2(i+1) from i=1, ideal DCG over ALL gold ids).
Host-side: Q lists of ids, negligible next to scoring."""
from __future__ import annotations

from statistics import mean

import numpy as np


class Metrics:
    def __init__(self, recall_at_k: list[int], map_at_k: list[int] = (), mrr_at_k: list[int] = (), ndcg_at_k: list[int] = ()):
        self.recall_at_k = list(recall_at_k)
        self.map_at_k = list(map_at_k)
        self.mrr_at_k = list(mrr_at_k)
        self.ndcg_at_k = list(ndcg_at_k)

    def compute_all_metrics(self, all_ground_truths: list[list[int]], all_results: list[list[int]]) -> dict:
        table = [(f"recall@{k}", self.recall, k) for k in self.recall_at_k]
        table += [(f"map@{k}", self.average_precision, k) for k in self.map_at_k]
        table += [(f"mrr@{k}", self.reciprocal_rank, k) for k in self.mrr_at_k]
        table += [(f"ndcg@{k}", self.ndcg, k) for k in self.ndcg_at_k]
        table += [("r-precision", self.r_precision, None)]
        return {name: self.compute_mean_score(fn, all_ground_truths, all_results, k) for name, fn, k in table}

    def compute_mean_score(self, score_func, all_ground_truths, all_results, k: int = None):
        return mean([score_func(g, r, k) for g, r in zip(all_ground_truths, all_results)])

    @staticmethod
    def _hits(gold, res, k):
        g = set(gold)
        return [1 if d in g else 0 for d in (res if k is None else res[:k])]

    def recall(self, ground_truths, results, k: int = None):
        return sum(self._hits(ground_truths, results, k)) / len(ground_truths)

    def precision(self, ground_truths, results, k: int = None):
        h = self._hits(ground_truths, results, k)
        return sum(h) / len(h)

    def average_precision(self, ground_truths, results, k: int = None):
        h = self._hits(ground_truths, results, k)
        run, total = 0, 0.0
        for i, rel in enumerate(h):
            run += rel
            if rel:
                total += run / (i + 1)   # precision@(i+1) at each relevant position
        return total / len(ground_truths)

    def reciprocal_rank(self, ground_truths, results, k: int = None):
        h = self._hits(ground_truths, results, k)
        if not h:   # the reference raises on max([]): guarded here
            return 0.0
        return max(1 / (i + 1) if rel else 0.0 for i, rel in enumerate(h))

    def ndcg(self, ground_truths, results, k: int = None):
        h = self._hits(ground_truths, results, k)
        if not h:
            return 0
        dcg = h[0] + sum(h[i] / np.log2(i + 1) for i in range(1, len(h)))
        idcg = 1 + sum(1 / np.log2(i + 1) for i in range(1, len(ground_truths)))
        return (dcg / idcg) if idcg != 0 else 0

    def r_precision(self, ground_truths, results, R: int = None):
        R = len(ground_truths)
        return sum(self._hits(ground_truths, results, R)) / R

    def fscore(self, ground_truths, results, k: int = None):
        p, r = self.precision(ground_truths, results, k), self.recall(ground_truths, results, k)
        return (2 * p * r) / (p + r) if (p != 0.0 or r != 0.0) else 0.0


RECALL_KS, MAP_KS, MRR_KS, NDCG_KS = [5, 10, 20, 50, 100, 200, 500, 1000], [10, 100], [10, 100], [10, 100]   # hybrid.py:28


def gold_rank_tables(n_gold: np.ndarray):
    """Host-side constants of the gold-rank metrics, computed the way metrics.py:100-118 computes them: the discount table
    (1 at position 0, 1/log2(i+1) from position 1), the ideal DCG per query, and the metric names in output order."""
    top = int(max(NDCG_KS))                                               # only ranks below the largest cut-off are ever discounted
    table = np.ones(top + 1)
    table[1:] = 1.0 / np.log2(np.arange(1, top + 1, dtype=np.float64) + 1.0)   # metrics.py:108: 1/log2(i+1) from position 1
    idcg = np.array([1 + sum(1 / np.log2(i + 1) for i in range(1, int(n))) if n > 0 else 1.0 for n in n_gold], dtype=np.float64)
    names = ([f"recall@{k}" for k in RECALL_KS] + [f"map@{k}" for k in MAP_KS] + [f"mrr@{k}" for k in MRR_KS]
             + [f"ndcg@{k}" for k in NDCG_KS] + ["r-precision"])
    return table, idcg, names


def metrics_from_gold_ranks(ranks: np.ndarray, n_gold: np.ndarray, list_len: np.ndarray, recall_ks=None, only=None) -> list[dict]:
    """All metrics of run_evaluation (hybrid.py:24-42) from the 0-based ranks of the gold documents.
    ranks [W, Q, G] int64 (np.iinfo(int64).max = never retrieved), n_gold [Q] = len(ground_truths) as the reference
    divides by it, list_len [Q] = length of the fused list.  Same per-query formulas and summation order as
    metrics.py:72-136; means over queries use an exactly rounded sum (statistics.mean differs by <= 1 ulp).
    Everything is laid out [gold, query, weight vector] so that every slice below is contiguous (W = 1771 in the 4-system
    sweep: this function, not the counting kernel, would otherwise dominate the sweep).
    recall_ks: the recall cut-offs (default: run_evaluation's); only: the metric names to compute and return (default: all) -- the BM25
    grid search (bm25.py:223) evaluates recall@{10,100,200,500,1000} and r-precision only."""
    want = (lambda name: True) if only is None else (lambda name: name in set(only))
    W, Q, G = ranks.shape
    INF = np.iinfo(np.int64).max
    r = np.ascontiguousarray(np.sort(ranks, axis=2).transpose(2, 1, 0))   # [G, Q, W], ascending gold ranks per (q, w)
    have = r < INF
    ng = np.maximum(n_gold, 1).astype(np.float64)[:, None]                # [Q, 1]
    table, idcg, _ = gold_rank_tables(n_gold)
    top = len(table) - 1
    idcg = idcg[:, None]
    per_query: dict[str, np.ndarray] = {}                                 # each [Q, W]
    for k in (RECALL_KS if recall_ks is None else recall_ks):
        if want(f"recall@{k}"):
            per_query[f"recall@{k}"] = (r < k).sum(0) / ng                # r < k implies retrieved
    for k in MAP_KS:
        if not want(f"map@{k}"):
            continue
        ap = np.zeros((Q, W))
        for i in range(G):                                                # i-th gold hit sits at rank r[i]: precision = (i+1)/(rank+1)
            ok = r[i] < k
            ap = ap + np.where(ok, (i + 1) / (np.where(ok, r[i], 0).astype(np.float64) + 1.0), 0.0)
        per_query[f"map@{k}"] = ap / ng
    for k in MRR_KS:
        if not want(f"mrr@{k}"):
            continue
        ok = r[0] < k
        per_query[f"mrr@{k}"] = np.where(ok, 1.0 / (np.where(ok, r[0], 0).astype(np.float64) + 1.0), 0.0)
    for k in NDCG_KS:
        if not want(f"ndcg@{k}"):
            continue
        head = (r == 0).sum(0).astype(np.float64)                         # relevances[0]
        tail = np.zeros((Q, W))
        for i in range(G):                                                # sum(...) for positions >= 1, in rank order
            ok = (r[i] >= 1) & (r[i] < k)
            tail = tail + np.where(ok, table[np.minimum(r[i], top)], 0.0)
        per_query[f"ndcg@{k}"] = (head + tail) / idcg
    if want("r-precision"):
        per_query["r-precision"] = (r < n_gold[None, :, None]).sum(0) / ng
    # mean over the queries with an exactly rounded sum, all (metric, weight vector) pairs at once: error-free TwoSum
    # accumulation (hi + lo carries the sum to ~106 bits; all terms are >= 0), one rounding at the end -- what
    # math.fsum(row) / Q gives, without 15 * W Python-level calls
    names = list(per_query)
    hi = np.zeros((len(names), W))
    lo = np.zeros((len(names), W))
    for q in range(Q):
        x = np.stack([per_query[n][q] for n in names])
        t = hi + x
        bb = t - hi
        lo += (hi - (t - bb)) + (x - bb)
        hi = t
    means = (hi + lo) / Q
    return [{n: float(means[i, w]) for i, n in enumerate(names)} for w in range(W)]
