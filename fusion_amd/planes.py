"""Device-resident ranked lists.

The reference passes `list[Q] of list[<=N] of {'corpus_id', 'score'}` between Ranker and Aggregator
(hybrid.py:66-75,93-106,137,170-179).  On the GPU the same information is one `RankedSystem` per
retrieval system: dense planes indexed by corpus POSITION (column j = j-th entry of the corpus dict),
which is what the fusion kernels read with coalesced 16-byte accesses.  The list-of-dicts form is
rebuilt only when a caller asks for it (`to_lists`).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch


class _gc_paused:
    """Building Q x N small dicts trips the cyclic collector's generation thresholds again and again, and every full collection walks
    all the dicts made so far (they are containers): paused for the duration, the lists cost their allocations only."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def _dict_list(cids: list, scores: list) -> list[dict]:
    """[{'corpus_id': c, 'score': s}, ...] -- the reference's list entry (hybrid.py:75,307); ids arrive as Python objects
    (ndarray.tolist(): ints for integer tables, the original objects for an object table)."""
    from . import _pyhost
    return _pyhost.build(cids, scores)


@dataclass
class RankedSystem:
    scores: torch.Tensor          # [Q, N] float32 plane: score of doc j for query q (undefined where rank < 0)
    order: torch.Tensor           # [Q, N] int32 plane: order[q, r] = corpus position at rank r (r < lens[q]), else -1
    rank: torch.Tensor            # [Q, N] int32 plane: rank[q, j] = list position of doc j, -1 if absent
    lens: torch.Tensor            # [Q] int32: list length per query
    ids: np.ndarray               # [N] corpus position -> corpus_id (hybrid.py:66 idx2id)
    sorted_scores: torch.Tensor | None = None   # [Q, N] scores in rank order (same dtype as the ranking keys)
    full: bool = True             # every list covers all N docs (lens == N)
    meta: dict = field(default_factory=dict)
    scores64: torch.Tensor | None = None   # [Q, N] float64 plane when the raw scores are not float32 values (BM25's Python
                                           # floats, host lists): the 'none' passthrough keeps them unrounded (hybrid.py:280)
    score_sorted: bool = False    # every list is in descending order of its float32 scores (rankers: yes; host lists: checked)
    stats4: torch.Tensor | None = None   # [4, Q] fp32: mean | unbiased std | min | max of every list's float32 scores (over the LISTED
                                         # documents), a by-product of the ranking sort; min-max / z-score fusion reads it in place

    @property
    def zstats(self):
        """(mean [Q], unbiased std [Q]) when the ranking sort produced them, else None."""
        return None if self.stats4 is None else (self.stats4[0], self.stats4[1])

    def stats(self, norm: str):
        """(a, b) = (min, max) for 'min-max', (mean, unbiased std) for 'z-score': [Q] fp32 each, over the listed documents of every
        query (hybrid.py:254-262).  From the ranking sort when it produced them; otherwise computed ONCE per system -- the two ends of
        a score-sorted list, or one reduction (fz_row_stats_f32) -- and kept: a system's statistics do not depend on the fusion weights,
        so Aggregator.fuse, the float64 fusion of the tuning grid and Aggregator.tune all normalise with the same bits."""
        if self.stats4 is not None:
            return (self.stats4[2], self.stats4[3]) if norm == "min-max" else (self.stats4[0], self.stats4[1])
        key = "stats_" + norm
        if self.meta.get(key) is None:
            from . import ops
            if norm == "min-max" and self.score_sorted:
                self.meta[key] = ops.minmax_from_order(self.scores, self.order, self.lens)
            else:
                self.meta[key] = ops.row_stats(self.scores, None if self.full else self.rank, norm)
        return self.meta[key]

    def valid_bits(self) -> torch.Tensor | None:
        """Validity of a partial system as a bitmap (1 bit per document instead of the 4-byte rank), built once and kept."""
        if self.full:
            return None
        if self.meta.get("valid_bits") is None:
            from . import ops
            self.meta["valid_bits"] = ops.rank_to_bitmap(self.rank)
        return self.meta["valid_bits"]

    @property
    def Q(self) -> int:
        return self.scores.shape[0]

    @property
    def N(self) -> int:
        return self.scores.shape[1]

    def list_scores(self) -> torch.Tensor:
        """[Q, N] the scores in LIST order (rank r -> score of the document there; -inf past the list's end), in the ranking keys' dtype.
        The rankers do not have the sort write this plane any more (round 5: nothing on the device reads it -- every fusion kernel works
        on the planes by corpus position -- and for float64 keys it cost 8 B per document in two partial-line passes): it is one gather,
        made when the reference-typed lists are asked for."""
        if self.sorted_scores is not None:
            return self.sorted_scores
        src = self.scores64 if self.scores64 is not None else self.scores
        got = torch.gather(src, 1, self.order.clamp(min=0).long())
        return torch.where(self.order >= 0, got, torch.full_like(got, float("-inf")))

    def to_lists(self, topk: int | None = None) -> list[list[dict]]:
        """-> the reference's RankedLists (hybrid.py:75,106,137)."""
        order = self.order.cpu().numpy()
        lens = self.lens.cpu().numpy()
        ss = self.list_scores().cpu().numpy()
        out = []
        with _gc_paused():
            for q in range(self.Q):
                n = int(lens[q]) if topk is None else min(int(lens[q]), topk)
                out.append(_dict_list(self.ids[order[q, :n]].tolist(), ss[q, :n].astype(np.float64).tolist()))   # Python floats (float(s))
        return out


@dataclass
class FusedResult:
    """Output of Aggregator.fuse on the device: fused lists over the union of ids per query."""
    order: torch.Tensor           # [Q, N] int32: corpus position at fused rank r (r < lens[q])
    scores: torch.Tensor          # [Q, N] fused score at fused rank r (float64 for rrf/bcf/none, float32 otherwise)
    lens: torch.Tensor            # [Q] int32 = |union of ids|
    ids: np.ndarray

    def to_lists(self) -> list[list[dict]]:
        order = self.order.cpu().numpy()
        sc = self.scores.cpu().numpy()
        lens = self.lens.cpu().numpy()
        f32 = sc.dtype == np.float32
        out = []
        # nsf scores leave the reference as np.float32 scalars (hybrid.py:258 ... zip(keys, scores.cpu().numpy())), rrf / bcf / 'none' as
        # Python floats: iterating a float32 array yields exactly those scalars, .tolist() the Python floats -- one C loop each
        with _gc_paused():
            for q in range(order.shape[0]):
                n = int(lens[q])
                out.append(_dict_list(self.ids[order[q, :n]].tolist(), list(sc[q, :n]) if f32 else sc[q, :n].tolist()))
        return out

    def predictions(self, topk: int | None = None) -> list[list]:
        order = self.order.cpu().numpy()
        lens = self.lens.cpu().numpy()
        return [self.ids[order[q, :(int(lens[q]) if topk is None else min(int(lens[q]), topk))]].tolist() for q in range(order.shape[0])]
