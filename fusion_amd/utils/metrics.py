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
