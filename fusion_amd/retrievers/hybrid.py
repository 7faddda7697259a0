"""MI355X-native drop-in for the reference's `src/retrievers/hybrid.py` (488 lines): same class names,
method signatures, CLI flags, result types and error behaviour; the arithmetic runs in the HIP kernels of
fusion_amd/csrc behind include/fusion_hip.h.

Reference map (file:line in the reference tree)
  run_evaluation            hybrid.py:24-42
  Ranker.*_search           hybrid.py:49-75, 77-106, 108-137, 139-163
  Aggregator.fuse           hybrid.py:170-220 (+ convert2dict :222-233, transform_scores :235-280,
                            weight_scores :282-291, aggregate_scores :293-307)
  main / argparse           hybrid.py:310-468, 471-488

Interchange type: the reference's RankedLists (list[Q] of list[<=N] of {'corpus_id','score'}) is accepted and
returned everywhere the reference does; additionally every Ranker method takes `as_device=True` to hand over
device-resident `RankedSystem` planes, which `Aggregator.fuse` consumes without a host round trip.

Documented deviations: D1 own SPLADE wrapper; D2 working cross-encoder rerank; D4 the weights
check really checks (KeyError); ties inside a system are broken by ascending corpus position; percentile tables
must be ascending (they are quantiles, hybrid.py:391-397).
"""
from __future__ import annotations

import argparse
import itertools
import os
from os.path import join

import numpy as np
import torch

from .. import ops
from ..planes import FusedResult, RankedSystem

RankedLists = list  # list[list[dict]]


def run_evaluation(predictions: list[list[int]], labels: list[list[int]], print2console: bool = True, log2wandb: bool = False,
                   args: argparse.Namespace = None):
    """hybrid.py:24-42 (wandb logging is out of scope: observability to an external SaaS)."""
    from ..utils.metrics import Metrics
    evaluator = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    scores = evaluator.compute_all_metrics(all_ground_truths=labels, all_results=predictions)
    if print2console:
        for metric, score in scores.items():
            print(f"- {metric.capitalize()}: {score:.3f}")
    return scores


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("fusion_amd needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _rank_scores(scores: torch.Tensor, ids: np.ndarray, return_topk: int | None) -> RankedSystem:
    """Full ranking of a [Q, N] score plane: what util.semantic_search(top_k=N) + sorted() produce (hybrid.py:103)."""
    Q, N = scores.shape
    k = N if return_topk is None else min(return_topk, N)
    lens = torch.full((Q,), k, dtype=torch.int32, device=scores.device)
    full = k == N
    # single-workgroup rows: the sort has the row in registers, so the list's mean / std / min / max -- over its k listed documents --
    # come for free (a truncated float64 ranking takes one reduction later instead: RankedSystem.stats)
    stats4 = None
    if N <= ops.sort_max_n(scores.dtype) and Q > 0 and (full or scores.dtype == torch.float32):
        stats4 = torch.empty((4, Q), dtype=torch.float32, device=scores.device)
    order, _, rank = ops.sort_rows_desc(scores, want_keys=False, want_rank=True, stats_out=stats4, stats_len=None if full or stats4 is None else lens)
    if not full:  # lists truncated to top-k: docs beyond rank k are absent from the list (in place: the planes keep their padded rows)
        rank.masked_fill_(rank >= k, -1)
        order[:, k:] = -1
    return RankedSystem(scores=scores, order=order, rank=rank, lens=lens, ids=ids, full=full, score_sorted=True, stats4=stats4)   # (list-order scores: RankedSystem.list_scores, on demand)


class Ranker:
    """Produce per-system ranked lists (hybrid.py:45-163)."""

    @staticmethod
    def bm25_search(queries: list[str], corpus: dict[int, str], do_preprocessing: bool, k1: float, b: float, return_topk: int = None,
                    *, as_device: bool = False):
        """hybrid.py:49-75. `do_preprocessing=True` runs the reference's TextPreprocessor recipe (src/data/preprocessor.py: spaCy
        fr_core_news_md -- punctuation, numbers and stop words dropped, lemmas, lower case) when that third-party model is installed and
        raises RuntimeError when it is not (offline: pass lemmatised whitespace text with do_preprocessing=False); never a silent no-op."""
        from .bm25 import BM25, preprocess
        documents = list(corpus.values())
        if do_preprocessing:
            documents, queries = preprocess(documents), preprocess(queries)
        ids = np.array(list(corpus.keys()))
        retriever = BM25(corpus=documents, k1=k1, b=b, device=_device())
        rs = retriever.search_device(queries, ids=ids)
        if return_topk is not None and return_topk < rs.N:
            k = return_topk
            rs.rank.masked_fill_(rs.rank >= k, -1)          # in place: the planes keep their padded, 16-byte aligned rows
            rs.order[:, k:] = -1
            rs.lens = torch.full_like(rs.lens, k); rs.full = False
            rs.stats4 = None   # the sort's statistics covered all N documents; the cut list's are taken over its k entries (RankedSystem.stats)
        return rs if as_device else rs.to_lists()

    SPARSE_DENSITY = 0.05     # SPLADE corpora with at most this fraction of non-zeros are scored through the inverted index

    @staticmethod
    def single_vector_search(queries: list[str], corpus: dict[int, str], model_name_or_path: str, return_topk: int = None,
                             *, encoder=None, as_device: bool = False, cache_dir: str | None = None):
        """hybrid.py:77-106: encode docs + queries (batch 64), cosine similarity, full ranking.
        `encoder` (optional) injects an already-built fusion_amd.encoders module (e.g. random_init for synthetic runs);
        `cache_dir` (optional) keeps the encoded corpus on disk, keyed by (checkpoint, corpus text) -- the sweep of
        scripts/run_hybrid.sh otherwise re-encodes the corpus in each of its 99 processes."""
        from .. import encoders
        documents = list(corpus.values())
        ids = np.array(list(corpus.keys()))
        kind = "splade" if "splade" in model_name_or_path.lower() else "dpr"
        model = encoder if encoder is not None else encoders.from_pretrained(model_name_or_path, kind, device=_device())
        key = encoders.corpus_cache_key(model_name_or_path, documents, kind) if cache_dir else ""
        if kind == "splade":
            # SPLADE vectors are a few hundred non-zeros of 32,005: the corpus side is kept (and cached) as an inverted index of its
            # L2-normalised rows -- tens of MB instead of the dense 3.6 GB -- and scored by fz_sparse_dot_f32: the products of the dense
            # cos_sim minus its exact zeros.  A corpus that is not sparse (SPARSE_DENSITY) takes the dense GEMM like DPR.
            def build():
                dn = ops.normalize_rows(ops.pad_dim(model.encode(documents, batch_size=64, query_mode=False).to(_device())))
                V = int(getattr(model, "dim", dn.shape[1]))
                if ops.density(dn[:, :V]) > Ranker.SPARSE_DENSITY:
                    return dn, torch.zeros(0, dtype=torch.int32), torch.zeros(0), torch.tensor([dn.shape[0], V, 0])
                ix = ops.sparse_index(dn, V)
                return ix.toff, ix.pdoc, ix.pw, torch.tensor([ix.N, ix.V, 1])
            a, b, c, shape = encoders.cached_tensors(cache_dir, key, ["sp_a", "sp_b", "sp_c", "sp_shape"], build)
            q_embs = model.encode(queries, batch_size=64, query_mode=True)
            if int(shape[2]) == 1:
                index = ops.SparseIndex(a.to(_device()), b.to(_device()), c.to(_device()), int(shape[0]), int(shape[1]))
                scores = ops.sparse_cos_scores(q_embs, index)
                del index
            else:
                scores = ops.dot_scores(ops.normalize_rows(q_embs), a.to(_device()))
            rs = _rank_scores(scores, ids, return_topk)
            del a, b, c, q_embs
            if encoder is None:
                del model
                torch.cuda.empty_cache()
            return rs if as_device else rs.to_lists()
        (d_embs,) = encoders.cached_tensors(cache_dir, key, ["emb"], lambda: (model.encode(documents, batch_size=64, query_mode=False),))
        d_embs = d_embs.to(_device())
        q_embs = model.encode(queries, batch_size=64, query_mode=True)
        scores = ops.cos_scores(q_embs, d_embs)   # hybrid.py:103 always uses cos_sim
        rs = _rank_scores(scores, ids, return_topk)
        del d_embs, q_embs
        if encoder is None:
            del model
            torch.cuda.empty_cache()
        return rs if as_device else rs.to_lists()

    @staticmethod
    def multi_vector_search(queries: list[str], corpus: dict[int, str], model_name_or_path: str, output_dir: str = "output",
                            return_topk: int = None, *, encoder=None, as_device: bool = False, cache_dir: str | None = None):
        """hybrid.py:108-137.  The reference goes through colbert-ai's PLAID index (approximate, candidate-pruned);
        here every (query, document) pair gets its exact MaxSim score on the device, a superset ranking of PLAID's."""
        from .. import encoders
        documents = list(corpus.values())
        ids = np.array(list(corpus.keys()))
        model = encoder if encoder is not None else encoders.from_pretrained(model_name_or_path, "colbert", device=_device())
        # the reference reuses its on-disk PLAID index when it exists (hybrid.py:126-130); the analogue here is the
        # cached exact token matrix
        # ... keyed by the encoder's precision too: token matrices of a float32 forward and of the mixed-precision one differ by a few
        # float16 steps, and a cache filled by one must not be scored against queries encoded by the other
        precision = "colbert-amp16" if getattr(model, "amp", False) else "colbert-fp32"
        key = encoders.corpus_cache_key(model_name_or_path, documents, precision) if cache_dir else ""
        Dtok, Doff = encoders.cached_tensors(cache_dir, key, ["tok", "off"], lambda: model.encode_docs(documents, batch_size=64))
        Dtok, Doff = Dtok.to(_device()), Doff.to(_device())
        Qtok = model.encode_queries(queries, batch_size=64)
        scores = ops.maxsim(Qtok, Dtok, Doff, max_doc_len=model.max_doc_length)
        rs = _rank_scores(scores, ids, return_topk)
        del Dtok, Qtok
        if encoder is None:
            del model
            torch.cuda.empty_cache()
        return rs if as_device else rs.to_lists()

    @staticmethod
    def cross_encoder_search(queries: list[str], candidates: list, model_name_or_path: str, return_topk: int = None, *, model=None,
                             corpus: dict[int, str] = None):
        """monoBERT rerank (hybrid.py:139-163).  The reference's version is dead code (`docs` undefined at :159, and main passes
        fused lists where it expects dicts, :462).  Working form: `candidates[i]` is a dict id->text, or -- as main
        passes it -- a fused ranked list of {'corpus_id', 'score'} together with `corpus` to look the texts up.
        `model` (optional) = an already-built fusion_amd.encoders.CrossEncoder (PyTorch-ROCm forward), else the checkpoint at
        `model_name_or_path` is loaded; ties keep the candidate order (stable sort)."""
        from .. import encoders
        own = model is None
        if own:   # CrossEncoderCustom(model_name_or_path), hybrid.py:151: a local checkpoint directory / HF cache entry (no network here)
            model = encoders.from_pretrained(model_name_or_path, "monobert", device=_device())
        ranked_lists = []
        for query, cands in zip(queries, candidates):
            if isinstance(cands, dict):
                cids, docs = list(cands.keys()), list(cands.values())
            else:
                cids = [x["corpus_id"] for x in cands]
                docs = [corpus[c] for c in cids]
            scores = model.predict([(query, d) for d in docs]).cpu().tolist() if docs else []
            order = sorted(range(len(docs)), key=lambda i: scores[i], reverse=True)[: return_topk or len(docs)]
            ranked_lists.append([{"corpus_id": cids[i], "score": scores[i]} for i in order])
        if own:
            del model
            torch.cuda.empty_cache()
        return ranked_lists


def _pack_by_dicts(ranked_lists: dict):
    """The reference's own data flow, entry by entry (any hashable id): corpus position = first-seen order of the ids."""
    names = list(ranked_lists.keys())
    Q = len(next(iter(ranked_lists.values())))
    pos: dict = {}
    for n in names:
        for lst in ranked_lists[n]:
            for x in lst:
                if x["corpus_id"] not in pos:
                    pos[x["corpus_id"]] = len(pos)
    N = max(len(pos), 1)
    ids = np.empty(N, dtype=object)
    for k, v in pos.items():
        ids[v] = k
    try:
        if all(type(k) is int for k in pos):     # (a bool or a float id stays what it is: astype would turn True into 1)
            ids = ids.astype(np.int64)
    except (TypeError, ValueError, OverflowError):
        pass
    ld = ops.round_up(N, 64)
    packed = {}
    for n in names:
        sc64 = np.zeros((Q, ld), dtype=np.float64)
        rk = np.full((Q, ld), -1, dtype=np.int32)
        od = np.full((Q, ld), -1, dtype=np.int32)
        ln = np.zeros(Q, dtype=np.int32)
        is_sorted = True
        for q, lst in enumerate(ranked_lists[n]):
            d = {}
            for x in lst:            # convert2dict (hybrid.py:231): first position kept, last score wins
                d[x["corpus_id"]] = x["score"]
            if d:
                j = np.fromiter((pos[c] for c in d.keys()), dtype=np.int64, count=len(d))
                v = np.fromiter(d.values(), dtype=np.float64, count=len(d))
                sc64[q, j] = v
                rk[q, j] = np.arange(len(d), dtype=np.int32)
                od[q, : len(d)] = j
                v32 = v.astype(np.float32)
                is_sorted = is_sorted and bool(np.all(v32[:-1] >= v32[1:]))   # False for NaN as well
            ln[q] = len(d)
        packed[n] = (sc64, rk, od, ln, is_sorted)
    return ids, N, packed


def pack_ranked_lists(ranked_lists: dict):
    """The reference's RankedLists (hybrid.py:66-75: system -> list[Q] of list[<= N] of {'corpus_id', 'score'}) -> per system the host
    planes a RankedSystem is made of: (scores64 [Q, ld], rank [Q, ld], order [Q, ld], lens [Q], every list sorted by score?), plus the
    corpus position -> id table.  Pure host work, no device involved (bench.py times it as the boundary's packing cost).

    Integer ids (the reference's: dataset article ids, hybrid.py:66) take a vectorised route: ids and scores of a list are pulled out by
    one C loop over the dicts (csrc/pyhost.c), positions come from a direct table over the id range (any bijection id <-> position
    serves: ties are broken by list order, never by position), duplicates inside a list -- convert2dict keeps the FIRST position and the
    LAST score, hybrid.py:231 -- are detected by a scatter/gather round trip and sent, list by list, through the dict.  Everything else
    (string ids, ids that do not fit int64, booleans ...) goes the reference's own entry-by-entry way."""
    from .. import _pyhost
    names = list(ranked_lists.keys())
    Q = len(next(iter(ranked_lists.values())))
    raw = {}
    for n in names:
        per_q = []
        for lst in ranked_lists[n]:
            got = _pyhost.extract(lst) if type(lst) is list else None
            if got is None:
                return _pack_by_dicts(ranked_lists)
            per_q.append(got)
        raw[n] = per_q
    # the id table: the union of all lists' ids, sorted; position of an id = its index there.  Ids in a modest range (dataset article
    # ids are) map through a direct table -- one gather per list; a sparse id space falls back to searchsorted
    lo = min((int(c.min()) for n in names for c, _ in raw[n] if c.size), default=0)
    hi = max((int(c.max()) for n in names for c, _ in raw[n] if c.size), default=0)
    n_ids = sum(int(c.size) for n in names for c, _ in raw[n])            # ids seen, with repeats: an upper bound of the distinct ones
    direct = hi - lo < (1 << 26) and hi - lo <= 64 * max(n_ids, 1024)    # a few thousand ids spread over 2^26 values: not a 320 MB table
    if direct:
        seen = np.zeros(hi - lo + 1, dtype=bool)
        for n in names:
            for cid, _ in raw[n]:
                seen[cid - lo] = True
        table = np.flatnonzero(seen).astype(np.int64) + lo
        where = np.cumsum(seen, dtype=np.int32) - 1          # id - lo -> position
        locate = lambda cid: where[cid - lo]
    else:
        table = np.empty(0, dtype=np.int64)
        for n in names:
            for cid, _ in raw[n]:
                if cid.size == 0:
                    continue
                if table.size:
                    j = np.minimum(np.searchsorted(table, cid), table.size - 1)
                    if np.array_equal(table[j], cid):
                        continue
                table = np.union1d(table, cid)
        locate = lambda cid: np.searchsorted(table, cid)
    N = max(int(table.size), 1)
    ids = table if table.size else np.zeros(1, dtype=np.int64)
    ld = ops.round_up(N, 64)
    packed = {}
    for n in names:
        sc64 = np.zeros((Q, ld), dtype=np.float64)
        rk = np.full((Q, ld), -1, dtype=np.int32)
        od = np.full((Q, ld), -1, dtype=np.int32)
        ln = np.zeros(Q, dtype=np.int32)
        is_sorted = True
        for q, (cid, val) in enumerate(raw[n]):
            m = cid.size
            if m == 0:
                continue
            j = locate(cid)
            ar = np.arange(m, dtype=np.int32)
            rk[q, j] = ar
            if not np.array_equal(rk[q, j], ar):     # an id occurs twice: first position kept, last score wins (hybrid.py:231)
                d = {}
                for c, v_ in zip(cid.tolist(), val.tolist()):
                    d[c] = v_
                cid = np.fromiter(d.keys(), dtype=np.int64, count=len(d))
                val = np.fromiter(d.values(), dtype=np.float64, count=len(d))
                m = cid.size
                rk[q, j] = -1
                j = locate(cid)
                ar = np.arange(m, dtype=np.int32)
                rk[q, j] = ar
            sc64[q, j] = val
            od[q, :m] = j
            ln[q] = m
            v32 = val.astype(np.float32)
            is_sorted = is_sorted and bool(np.all(v32[:-1] >= v32[1:]))   # False for NaN as well
        packed[n] = (sc64, rk, od, ln, is_sorted)
    return ids, N, packed


class Aggregator:
    """Normalise + fuse ranked lists (hybrid.py:166-307)."""

    # NumPy scalar promotion of `np.float32 score * w` (hybrid.py:291).  False (default): NumPy >= 2 -- a Python-float weight is a weak
    # scalar, product and running sum stay float32; only np.float64 weights (the tuning grid, hybrid.py:405-409) give float64.  That is
    # the arithmetic of the reference run under this image's NumPy 2.2, on whose outputs tests/golden is pinned.  True: NumPy 1.x value-
    # based promotion, the reference's own pinned environment (torch 2.1.2 / pandas 2.1.4 era): EVERY nsf product and sum is float64.
    # Scores differ by ~1e-8 relative; near-tied documents can swap.  Also switched on by FUSION_AMD_NUMPY1_PROMOTION=1.
    NUMPY1_PROMOTION = os.environ.get("FUSION_AMD_NUMPY1_PROMOTION", "0") == "1"
    # fuse_device(topk=k): from this many queries on, float64 fused PLANES over full lists ('none' sums, np.float64 weights; rrf / bcf only
    # for rows beyond one workgroup -- shorter ones are fused inside the sort, round 5) are SELECTED (ops.select_topk) instead of sorted
    # and cut; below it the selection's extra launches and its flag read cost more than the sort (bench.py: 0.26 vs 0.14 ms at Q = 195)
    SELECT_MIN_Q = 512
    last_topk_path = None     # "select" | "sort": which of the two the last fuse_device(topk=...) took (tests pin it)
    last_rank_fused_sort = None   # True: the last rrf / bcf fusion ran as the load phase of the final sort (one kernel, no float64 plane)

    @classmethod
    def _wide(cls, w) -> bool:
        return cls.NUMPY1_PROMOTION or ops.is_wide_weight(w)

    @classmethod
    def fuse(cls, ranked_lists: dict, method: str, normalization: str = None, linear_weights: dict[str, float] = None,
             percentile_distributions: dict[str, np.ndarray] = None, return_topk: int = 1000, *, as_device: bool = False):
        """hybrid.py:170-220.  `ranked_lists`: system -> RankedLists (reference format) or system -> RankedSystem (device)."""
        fused = cls.fuse_device(cls._to_device(ranked_lists), method, normalization, linear_weights, percentile_distributions)
        if as_device:
            return fused
        return fused.to_lists()[:return_topk]   # slices QUERIES, as hybrid.py:220 does

    # -- device pipeline -----------------------------------------------------------------
    @classmethod
    def fuse_device(cls, systems: dict[str, RankedSystem], method: str, normalization: str = None,
                    linear_weights: dict[str, float] = None, percentile_distributions: dict[str, np.ndarray] = None,
                    topk: int | None = None) -> FusedResult:
        """The fusion on the device.  topk=k: only the first k entries of every fused list are produced (what main() reads of them:
        predictions(1000), hybrid.py:537) -- the rows are selected, not sorted (ops.select_topk); the entries and their order are those
        of the full lists, bit for bit.  Aggregator.fuse always returns the full lists, as the reference does."""
        names = list(systems.keys())
        S = [systems[n] for n in names]
        Q = S[0].Q
        assert all(s.Q == Q for s in S), (
            "Ranked results from different retrieval systems have varying lenghts across systems (i.e., some systems have been run on more queries).")
        N = S[0].N
        if any(s.N != N for s in S):
            raise ValueError("device systems must be planes over the same corpus")
        dev = S[0].scores.device
        all_full = all(s.full for s in S)
        ranks = None if all_full else [None if s.full else s.rank for s in S]   # validity: only the partial lists carry any
        vbits = None if all_full else [s.valid_bits() for s in S]              # ... and the nsf passes read it as 1 bit per document

        rank_fused_sort = False
        if method in ("bcf", "rrf"):
            lens = torch.stack([s.lens for s in S]).contiguous()
            # the fusion is the LOAD PHASE of the final sort (ops.sort_rank_fused: no float64 plane between two kernels) wherever one
            # workgroup holds a row -- also when only the first k entries are wanted: the full fused sort (0.46 ms per 1024 x 27,942)
            # now costs less than fuse + selection + two small sorts (0.50), so rrf / bcf lists are sorted and cut; longer rows keep
            # the two calls (and the selection)
            rank_fused_sort = 0 < N <= ops.sort_max_n(torch.float64) and Q > 0
            fused = None if rank_fused_sort else ops.fuse_rank([s.rank for s in S], lens, method)
        elif method == "nsf":
            if percentile_distributions is None:            # the reference calls .get() on it for every system (hybrid.py:213)
                raise AttributeError("'NoneType' object has no attribute 'get'")
            w = [linear_weights[n] for n in names]          # KeyError when a system has no weight (hybrid.py:214)
            wide = [cls._wide(x) for x in w]                # np.float64 weights (the tuning grid): NumPy promotes to float64
            if normalization in ("min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"):
                distr = None
                if normalization in ("percentile-rank", "normal-curve-equivalent"):
                    distr = [cls._table(percentile_distributions.get(n), dev) for n in names]
                st = [s.stats(normalization) for s in S] if normalization in ("min-max", "z-score") else None   # kept per system
                if any(wide):   # transform every system in float32 (weight 1: fl32(t * 1) == t), then weight + sum as NumPy does
                    T = cls._normalised_planes(S, normalization, distr, st)
                    fused = ops.fuse_wsum(T, ranks, w, narrow=[not x for x in wide])
                else:           # one flat pass: statistics in hand, validity of the partial lists as bitmaps
                    fused = ops.fuse_nsf([s.scores for s in S], ranks, w, normalization, distr, stats=st, valid_bits=vbits)
            else:                                           # 'none' / unknown string: raw Python floats, float64 (hybrid.py:280)
                fused = ops.fuse_wsum([s.scores if s.scores64 is None else s.scores64 for s in S], ranks, w)
        else:                                               # unknown method: raw scores are summed (hybrid.py:203-218)
            fused = ops.fuse_wsum([s.scores if s.scores64 is None else s.scores64 for s in S], ranks, [1.0] * len(S))

        # topk: float64 fused rows (rrf / bcf / 'none' / np.float64 weights) over full lists are selected, not sorted -- two thirds of the
        # full float64 sort's time at N = 27,942.  float32 rows (the selection costs what their four-pass sort costs) and partial lists
        # (the inverse insertion order costs more than the selection saves) are sorted and cut
        select = (topk is not None and topk < N and fused is not None and fused.dtype == torch.float64 and all_full
                  and Q >= cls.SELECT_MIN_Q)   # (small batches: launch- and sync-bound, the sort wins)
        cls.last_topk_path = "sort"
        cls.last_rank_fused_sort = rank_fused_sort
        if rank_fused_sort:
            if all_full:    # first-insertion order == system 0's ranking, whose rank plane is also the first fusion operand: read once
                lens_out = torch.full((Q,), N, dtype=torch.int32, device=dev)
                order, sk, _ = ops.sort_rank_fused([s.rank for s in S], lens, method, init_rank=S[0].rank, covers_all=True)
            else:
                ins, U = ops.insertion_order([s.order for s in S], lens, N)
                lens_out = U
                order, sk, _ = ops.sort_rank_fused([s.rank for s in S], lens, method, init_order=ins, row_len=U)
        elif all_full:
            # first-insertion order == system 0's ranking: its rank plane places every doc (coalesced, no gather)
            if select:
                sel = ops.select_topk(fused, S[0].rank, topk)
                if sel is not None:
                    cls.last_topk_path = "select"
                    return FusedResult(order=sel[0], scores=sel[1], lens=sel[2], ids=S[0].ids)
            lens_out = torch.full((Q,), N, dtype=torch.int32, device=dev)
            order, sk, _ = ops.sort_rows_desc(fused, init_rank=S[0].rank, covers_all=True)   # every list is full: every slot gets written
        else:
            lens = torch.stack([s.lens for s in S]).contiguous()
            ins, U = ops.insertion_order([s.order for s in S], lens, N)
            lens_out = U
            order, sk, _ = ops.sort_rows_desc(fused, init_order=ins, row_len=U)
        if topk is not None and topk < N:
            return FusedResult(order=order[:, :topk], scores=sk[:, :topk], lens=torch.clamp(lens_out, max=topk), ids=S[0].ids)
        return FusedResult(order=order, scores=sk, lens=lens_out, ids=S[0].ids)

    # -- N1: the whole weight grid in one pass (hybrid.py:404-426) ----------------------------------
    @classmethod
    def tune(cls, ranked_lists: dict, normalization: str, weight_combinations: list[dict[str, float]], labels: list[list],
             percentile_distributions: dict[str, np.ndarray] = None) -> list[dict]:
        """For every weight vector: the metrics run_evaluation() would report on Aggregator.fuse(method='nsf', ...).
        The reference re-normalises, re-fuses, re-sorts and re-evaluates per weight vector (hybrid.py:416-425); here the
        systems are normalised once and one counting kernel yields the fused ranks of the gold documents for all weight
        vectors (csrc/tune.hip) -- every metric is a function of those ranks."""
        from ..utils.metrics import metrics_from_gold_ranks
        systems = cls._to_device(ranked_lists)
        names = list(systems.keys())
        S = [systems[n] for n in names]
        # NumPy promotion (hybrid.py:291,304): the reference's grid (np.arange, :405-409) holds np.float64 weights -> float64
        # products and sums; a grid of Python floats fuses in float32.  A grid that mixes the two kinds goes the generic way.
        kinds = {cls._wide(x) for w in weight_combinations for x in w.values()}
        if (normalization not in ("min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent") or len(S) > 4
                or len(kinds) > 1):
            return cls._tune_by_fusing(systems, normalization, weight_combinations, labels, percentile_distributions)
        wide = kinds == {True}
        Q, N = S[0].Q, S[0].N
        dev = S[0].scores.device
        all_full = all(s.full for s in S)
        distr = None
        if normalization in ("percentile-rank", "normal-curve-equivalent"):
            distr = [cls._table(percentile_distributions.get(n), dev) for n in names]
        st = [s.stats(normalization) for s in S] if normalization in ("min-max", "z-score") else None
        T = cls._normalised_planes(S, normalization, distr, st, zero_unlisted=True)   # the very planes fuse_device's float64 path sums
        if all_full:
            pos = S[0].rank
        else:
            lens = torch.stack([s.lens for s in S]).contiguous()
            ins, U, pos = ops.insertion_order([s.order for s in S], lens, N, want_pos=True)
        weights = torch.tensor([[float(w[n]) for n in names] for w in weight_combinations],      # KeyError as hybrid.py:214
                               dtype=torch.float64).to(torch.float64 if wide else torch.float32).to(dev)
        id2pos = {cid: j for j, cid in enumerate(S[0].ids.tolist())}
        gold_pos = [[id2pos.get(g, -1) for g in dict.fromkeys(gl)] for gl in labels]   # unique, order kept
        G = int(ops._lib.lib().fz_tune_max_gold())
        Gmax = max((len(g) for g in gold_pos), default=0)
        W = len(weight_combinations)
        n_gold = np.array([len(gl) for gl in labels], dtype=np.int64)          # the reference divides by len(ground_truths)
        if Gmax <= G:
            # every gold list fits one counting launch: ranks -> metrics stay on the device (csrc/tune.hip, tune_metrics_kernel);
            # [W, 15] float64 come back instead of [W, Q, G] ranks and a NumPy evaluation that cost more than the sweep itself
            from ..utils.metrics import MAP_KS, MRR_KS, NDCG_KS, RECALL_KS, gold_rank_tables
            gold = np.full((Q, G), -1, dtype=np.int32)
            for q, gl in enumerate(gold_pos):
                gold[q, :len(gl)] = gl
            gold_dev = torch.from_numpy(gold).to(dev)
            table, idcg, mnames = gold_rank_tables(n_gold)
            rk = ops.gold_ranks(T, pos, weights, gold_dev)
            means = ops.tune_metrics(rk, gold_dev, pos, torch.from_numpy(n_gold.astype(np.int32)).to(dev), torch.from_numpy(idcg).to(dev),
                                     torch.from_numpy(table).to(dev), dict(recall=RECALL_KS, map=MAP_KS, mrr=MRR_KS, ndcg=NDCG_KS)).cpu().numpy()
            return [dict(zip(mnames, row)) for row in means.tolist()]   # Python floats, as run_evaluation returns
        ranks = np.full((W, Q, max(Gmax, 1)), np.iinfo(np.int64).max, dtype=np.int64)
        pos_host = None
        for g0 in range(0, Gmax, G):
            gold = np.full((Q, G), -1, dtype=np.int32)
            for q, gl in enumerate(gold_pos):
                chunk = gl[g0:g0 + G]
                gold[q, :len(chunk)] = chunk
            out = ops.gold_ranks(T, pos, weights, torch.from_numpy(gold).to(dev)).cpu().numpy().astype(np.int64)
            if pos_host is None:
                pos_host = pos.cpu().numpy()
            listed = (gold >= 0) & (np.take_along_axis(pos_host, np.maximum(gold, 0).astype(np.int64), axis=1) >= 0)
            blk = np.where(listed[None, :, :], out, np.iinfo(np.int64).max)   # never retrieved -> rank = infinity
            ranks[:, :, g0:g0 + G] = blk[:, :, : ranks.shape[2] - g0]
        list_len = (pos_host >= 0).sum(1) if pos_host is not None else np.zeros(Q, dtype=np.int64)
        return metrics_from_gold_ranks(ranks, n_gold, list_len)

    @staticmethod
    def _normalised_planes(S, normalization, distr, st, zero_unlisted=False):
        """Every system's transformed scores as its own float32 plane (fusion kernel with weight 1: fl32(t * 1) + 0 == t), with the SAME
        per-system statistics the one-pass fusion uses.  zero_unlisted: 0 instead of -inf where a system does not list a document (what
        the sweep kernel multiplies by the weights; a system adds nothing for such a document, hybrid.py:301-304)."""
        T = []
        for i, s in enumerate(S):
            t = ops.fuse_nsf([s.scores], None if s.full else [s.rank], [1.0], normalization, None if distr is None else [distr[i]],
                             stats=None if st is None else [st[i]], valid_bits=None if s.full else [s.valid_bits()])
            if zero_unlisted and not s.full:
                ops.zero_unlisted_(t, ops.harmonise([t, s.rank])[1])
            T.append(t)
        return T

    @classmethod
    def _tune_by_fusing(cls, systems, normalization, weight_combinations, labels, percentile_distributions):
        """Generic path (float64 'none' arithmetic, > 4 systems): one device fusion + evaluation per weight vector."""
        out = []
        for w in weight_combinations:
            fused = cls.fuse_device(systems, "nsf", normalization, w, percentile_distributions, topk=1000)   # every cut-off is <= 1000
            out.append(run_evaluation(fused.predictions(1000), labels, print2console=False))
        return out

    @staticmethod
    def _table(distr, dev) -> torch.Tensor:
        t = np.asarray(distr, dtype=np.float64).astype(np.float32)   # torch.tensor(distr, dtype=float32), hybrid.py:272
        if t.ndim != 1 or t.size == 0:
            raise ValueError("percentile distribution must be a non-empty 1-D table")
        if np.any(np.diff(t) < 0):
            raise ValueError("percentile distribution must be ascending (it is a quantile table, hybrid.py:391-397)")
        return torch.from_numpy(t).to(dev)

    @staticmethod
    def _to_device(ranked_lists: dict) -> dict[str, RankedSystem]:
        first = next(iter(ranked_lists.values()))
        if isinstance(first, RankedSystem):
            return ranked_lists
        dev = _device()
        Q = len(first)
        assert all(len(v) == Q for v in ranked_lists.values()), (
            "Ranked results from different retrieval systems have varying lenghts across systems (i.e., some systems have been run on more queries).")
        ids, N, packed = pack_ranked_lists(ranked_lists)
        out = {}
        t = lambda a: torch.from_numpy(a).to(dev)[:, :N]
        for n, (sc64, rk, od, ln, is_sorted) in packed.items():
            sc = sc64.astype(np.float32)     # torch.tensor(list(values), dtype=float32) of the normalisations (hybrid.py:255)
            exact32 = bool(np.array_equal(sc.astype(np.float64), sc64, equal_nan=True))
            out[n] = RankedSystem(scores=t(sc), order=t(od), rank=t(rk), lens=torch.from_numpy(ln).to(dev), ids=ids,
                                  full=bool((ln == N).all()), scores64=None if exact32 else t(sc64), score_sorted=is_sorted)
        return out

    # -- the reference's small helpers, kept for API compatibility ---------------------------
    @staticmethod
    def convert2dict(results: list[dict]) -> dict:
        """hybrid.py:222-233."""
        return {res["corpus_id"]: res["score"] for res in results}

    @staticmethod
    def transform_scores(results: dict, transformation: str, percentile_distr: np.ndarray = None) -> dict:
        """hybrid.py:235-280 for one list; the statistics and the transform run in the fusion kernel (weight 1)."""
        n = len(results)
        if transformation == "borda-count":
            return {pid: (n - idx + 1) / n for idx, pid in enumerate(results.keys())}
        if transformation == "reciprocal-rank":
            return {pid: 1 / (60 + idx + 1) for idx, pid in enumerate(results.keys())}
        if transformation not in ("min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"):
            return results
        dev = _device()
        plane = ops.alloc_plane(1, n, torch.float32, dev)
        plane.copy_(torch.tensor(list(results.values()), dtype=torch.float64).to(torch.float32).unsqueeze(0))
        distr = None
        if transformation in ("percentile-rank", "normal-curve-equivalent"):
            distr = [Aggregator._table(percentile_distr, dev)]
        out = ops.fuse_nsf([plane], None, [1.0], transformation, distr).cpu().numpy()[0]
        return {pid: s for pid, s in zip(results.keys(), out)}

    @staticmethod
    def weight_scores(results: dict, w: float) -> dict:
        """hybrid.py:282-291."""
        return {cid: s * w for cid, s in results.items()}

    @staticmethod
    def aggregate_scores(*args: dict) -> list[dict]:
        """hybrid.py:293-307 for already-transformed dicts (host; the device path is fuse())."""
        agg: dict = {}
        for res in args:
            for pid, s in res.items():
                agg[pid] = agg.get(pid, 0.0) + s
        return [{"corpus_id": p, "score": s} for p, s in sorted(agg.items(), key=lambda kv: kv[1], reverse=True)]


# ---------------------------------------------------------------------------------------------------
# driver (hybrid.py:310-468) -- same flags; data comes from local files because there is no network
# ---------------------------------------------------------------------------------------------------
MODEL_CKPTS = {
    "dpr": {"general": "antoinelouis/biencoder-camembert-base-mmarcoFR", "legal": "maastrichtlawtech/dpr-legal-french"},
    "splade": {"general": "antoinelouis/spladev2-camembert-base-mmarcoFR", "legal": "maastrichtlawtech/splade-legal-french"},
    "colbert": {"general": "antoinelouis/colbertv1-camembert-base-mmarcoFR", "legal": "maastrichtlawtech/colbert-legal-french"},
    "monobert": {"general": "antoinelouis/crossencoder-camembert-base-mmarcoFR", "legal": "maastrichtlawtech/monobert-legal-french"},
}


def weight_grid(system_names: list[str], step: float = 0.05) -> list[dict[str, float]]:
    """hybrid.py:405-409: every weight vector on the `step` lattice that sums to 1 (np.isclose)."""
    grid = np.arange(0, 1 + step, step)
    return [dict(zip(system_names, comb)) for comb in itertools.product(grid, repeat=len(system_names)) if np.isclose(sum(comb), 1.0)]


def load_data(args):
    """LLeQA corpus / questions.  The reference pulls maastrichtlawtech/lleqa from the HF hub (hybrid.py:336-339);
    offline, `--synthetic N,Q` generates an LLeQA-shaped corpus, `--data_dir` reads corpus.jsonl / questions_{split}.jsonl."""
    import json
    if getattr(args, "synthetic", None):
        n, q = (int(x) for x in args.synthetic.split(","))
        rng = np.random.default_rng(0)
        vocab = np.array([f"mot{i}" for i in range(5000)])
        p = 1.0 / np.arange(1, 5001); p /= p.sum()
        corpus = {int(i + 1): " ".join(rng.choice(vocab, size=int(rng.integers(20, 200)), p=p)) for i in range(n)}
        queries = [" ".join(rng.choice(vocab, size=int(rng.integers(4, 16)), p=p)) for _ in range(q)]
        pos = [sorted(rng.choice(np.arange(1, n + 1), size=int(rng.integers(1, 5)), replace=False).tolist()) for _ in range(q)]
        return corpus, queries, pos
    d = args.data_dir
    if not d or not os.path.isdir(d):
        raise FileNotFoundError("no network: pass --data_dir with corpus.jsonl + questions_<split>.jsonl, or --synthetic N,Q")
    corpus = {}
    with open(join(d, "corpus.jsonl")) as f:
        for line in f:
            r = json.loads(line); corpus[r["id"]] = r["article"]
    split = "validation" if args.data_split == "dev" else args.data_split
    queries, pos = [], []
    with open(join(d, f"questions_{split}.jsonl")) as f:
        for line in f:
            r = json.loads(line); queries.append(r["question"]); pos.append(r["article_ids"])
    return corpus, queries, pos


def main(args):
    import pandas as pd
    from .. import encoders
    sep = f"#{'-' * 40}#"
    os.makedirs(args.output_dir, exist_ok=True)
    if os.environ.get("FUSION_AMD_TUNE_GEMMS") == "1":   # opt-in: process-wide PyTorch switch (see encoders.enable_gemm_tuning)
        encoders.enable_gemm_tuning()
    args.eval_type = ("in" if args.models_domain == "legal" else "out") + "domain"
    print("Loading corpus and queries...")
    corpus, queries, pos_pids = load_data(args)
    results: dict[str, RankedSystem] = {}
    synth = bool(getattr(args, "synthetic", None))

    cache = join(args.output_dir, "corpus_cache") if not getattr(args, "no_corpus_cache", False) else None

    def enc(kind):
        if synth:   # random-init CamemBERT-shaped encoder (no checkpoints offline)
            return encoders.random_init(kind, device=_device(), size=getattr(args, "synthetic_model", "tiny"))
        return None
    if args.run_bm25:
        print(f"{sep}\n# Ranking with BM25\n{sep}")
        results["bm25"] = Ranker.bm25_search(queries, corpus, do_preprocessing=False, k1=2.5, b=0.2, as_device=True)
    if args.run_dpr:
        print(f"{sep}\n# Ranking with DPR\n{sep}")
        results["dpr"] = Ranker.single_vector_search(queries, corpus, MODEL_CKPTS["dpr"][args.models_domain], encoder=enc("dpr"), as_device=True, cache_dir=cache)
    if args.run_splade:
        print(f"{sep}\n# Ranking with SPLADE\n{sep}")
        results["splade"] = Ranker.single_vector_search(queries, corpus, MODEL_CKPTS["splade"][args.models_domain], encoder=enc("splade"), as_device=True, cache_dir=cache)
    if args.run_colbert:
        print(f"{sep}\n# Ranking with ColBERT\n{sep}")
        results["colbert"] = Ranker.multi_vector_search(queries, corpus, MODEL_CKPTS["colbert"][args.models_domain], encoder=enc("colbert"), as_device=True, cache_dir=cache)

    if args.analyze_score_distributions:
        print(f"{sep}\n# Analyzing the score distributions per system\n{sep}")
        return analyze_score_distributions(args, results, corpus, pos_pids)

    def distributions():
        if args.normalization in ("percentile-rank", "normal-curve-equivalent"):
            df = pd.read_csv(join(args.output_dir, f"score_distributions_raw_{args.eval_type}_28k.csv"))
            return {k: np.array(v) for k, v in df.to_dict("series").items()}
        return {}

    if args.fusion == "nsf" and args.tune_linear_fusion_weight:
        combos = weight_grid(list(results.keys()))
        distr = distributions()
        print(f"{sep}\n# Tuning the weights of convex combination between systems: {len(combos)} permutations\n{sep}")
        perfs = Aggregator.tune(results, args.normalization, combos, pos_pids, distr)
        rows = [{**perf, **{f"weight_{k}": v for k, v in weights.items()}} for perf, weights in zip(perfs, combos)]
        pd.DataFrame(rows).to_csv(join(args.output_dir, f"nsf_{args.normalization}_{args.eval_type}.csv"), index=False)   # schema of hybrid.py:420-425
        print("Done.")
        return rows

    weights = {s: 1 / len(results) for s in results} if args.fusion == "nsf" else {}
    distr = distributions() if args.fusion == "nsf" else {}
    print(f"{sep}\n# Fusing results with {args.fusion.upper()}{' (' + args.normalization + ')' if args.fusion == 'nsf' else ''}\n{sep}")
    # main() reads 1000 entries of every fused list (hybrid.py:467 via Metrics' largest cut-off): the rows are selected, not sorted
    fused = Aggregator.fuse_device(Aggregator._to_device(results), args.fusion, args.normalization, weights, distr, topk=1000)
    predictions = fused.predictions(1000)
    if args.run_monobert:   # hybrid.py:460-462, with the argument bug fixed: rerank the fused top-k with the cross-encoder
        print(f"{sep}\n# Re-ranking with monoBERT \n{sep}")
        ce = encoders.random_cross_encoder(device=_device(), size=getattr(args, "synthetic_model", "tiny")) if synth else None
        k = getattr(args, "rerank_topk", 100)
        reranked = Ranker.cross_encoder_search(queries, [[{"corpus_id": c} for c in p[:k]] for p in predictions],
                                               MODEL_CKPTS["monobert"][args.models_domain], model=ce, corpus=corpus)
        predictions = [[x["corpus_id"] for x in r] + p[k:] for r, p in zip(reranked, predictions)]
    print(f"{sep}\n# Evaluation \n{sep}")
    return run_evaluation(predictions=predictions, labels=pos_pids, args=args)


def analyze_score_distributions(args, results: dict[str, RankedSystem], corpus: dict, pos_pids: list[list], table_sizes=None):
    """hybrid.py:363-402: per-system transformed scores of every (query, document), quantile tables of 1k / 10k / 100k /
    |corpus| points (the input of the percentile-rank / NCE normalisations, hybrid.py:412,451) and the scores of the
    positives vs as many random negatives.  The transform runs once per system on the device (one fusion-kernel pass
    with weight 1) instead of once per (query, system) with a host round trip (hybrid.py:378)."""
    import random
    import pandas as pd
    names = list(results.keys())
    dev = results[names[0]].scores.device
    distr = {}
    if args.normalization in ("percentile-rank", "normal-curve-equivalent"):
        df = pd.read_csv(join(args.output_dir, f"score_distributions_raw_{args.eval_type}_10k.csv"))   # hybrid.py:374
        distr = {k: np.array(v) for k, v in df.to_dict("series").items()}
    random.seed(42)
    max_pid = max(corpus.keys())
    neg_pids = [random.sample(list(set(range(1, max_pid + 1)) - set(x)), k=len(x)) for x in pos_pids]   # hybrid.py:368
    transformed, in_list_order = {}, {}
    for n in names:
        rs = results[n]
        if args.normalization in ("min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"):
            d = [Aggregator._table(distr.get(n), dev)] if n in distr else None
            t = ops.fuse_nsf([rs.scores], None if rs.full else [rs.rank], [1.0], args.normalization, d)
        else:
            t = rs.scores if rs.scores64 is None else rs.scores64                # 'none': the raw Python floats (hybrid.py:280)
        lens = rs.lens.cpu().numpy()
        by_rank = torch.gather(t, 1, rs.order.clamp(min=0).long()).cpu().numpy().astype(np.float64)   # [q, r] = score at list position r
        in_list_order[n] = [by_rank[q, : lens[q]] for q in range(rs.Q)]
        t = t.cpu().numpy().astype(np.float64)
        listed = np.ones_like(t, dtype=bool) if rs.full else (rs.rank.cpu().numpy() >= 0)
        transformed[n] = (t, listed)
    ids = results[names[0]].ids
    id2pos = {c: j for j, c in enumerate(ids.tolist())}
    # scores_{norm}_{eval}_{split}.csv: one row per (query, system, listed document), in the reference's order: query by query,
    # system by system, each list from its head (hybrid.py:370-379,387)
    Q = results[names[0]].Q
    all_scores_df = pd.DataFrame({
        "system": np.concatenate([np.repeat(n, len(in_list_order[n][q])) for q in range(Q) for n in names]) if Q else np.array([], dtype=str),
        "score": np.concatenate([in_list_order[n][q] for q in range(Q) for n in names]) if Q else np.array([], dtype=np.float64)})
    all_scores_df.to_csv(join(args.output_dir, f"scores_{args.normalization}_{args.eval_type}_{args.data_split}.csv"), index=False)
    # quantile tables (hybrid.py:390-397): drop zeros and each system's two smallest distinct scores, then N+1 quantiles
    for N in (table_sizes or [1000, 10000, 100000, len(corpus)]):
        cols = {}
        for n, (t, l) in transformed.items():
            v = t[l]
            two = np.unique(v)[:2]
            v = v[(v != 0.0) & ~np.isin(v, two)]
            cols[n] = np.quantile(v, np.linspace(0, 1, N + 1)) if v.size else np.full(N + 1, np.nan)
        pd.DataFrame(cols).to_csv(join(args.output_dir, f"score_distributions_{args.normalization}_{args.eval_type}_{round(N / 1e3)}k.csv"), index=False)
    # labelled scores of positives / sampled negatives, 0 when a system does not list the document (hybrid.py:381-383,400)
    rows = []
    for q, (pos, neg) in enumerate(zip(pos_pids, neg_pids)):
        for label, pids in (("positive", pos), ("negative", neg)):
            for pid in pids:
                j = id2pos.get(pid)
                rows.append({"label": label, **{n: (float(t[q, j]) if j is not None and l[q, j] else 0) for n, (t, l) in transformed.items()}})
    pd.DataFrame(rows, columns=["label"] + names).to_csv(
        join(args.output_dir, f"labeled_scores_{args.normalization}_{args.eval_type}_{args.data_split}.csv"), index=False)
    print("Done.")
    return all_scores_df


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--data_split", type=str, choices=["dev", "test", "train"])
    parser.add_argument("--models_domain", type=str, choices=["general", "legal"])
    for name in ("bm25", "dpr", "splade", "colbert", "monobert"):
        parser.add_argument(f"--run_{name}", action="store_true", default=False)
    parser.add_argument("--fusion", type=str, choices=["bcf", "rrf", "nsf"])
    parser.add_argument("--normalization", type=str, choices=["none", "min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"])
    parser.add_argument("--tune_linear_fusion_weight", action="store_true", default=False)
    parser.add_argument("--analyze_score_distributions", action="store_true", default=False)
    parser.add_argument("--output_dir", type=str)
    # offline additions (no HF hub): local data or synthetic LLeQA-shaped data
    parser.add_argument("--data_dir", type=str, default=os.environ.get("LLEQA_DIR"))
    parser.add_argument("--synthetic", type=str, default=None, help="N,Q: synthetic corpus/queries of that size")
    parser.add_argument("--synthetic_model", type=str, default="tiny", choices=["tiny", "base"])
    parser.add_argument("--no_corpus_cache", action="store_true", default=False, help="do not keep encoded corpora under <output_dir>/corpus_cache")
    parser.add_argument("--rerank_topk", type=int, default=100, help="candidates per query handed to monoBERT")
    return parser


if __name__ == "__main__":
    a, _ = build_parser().parse_known_args()   # unknown flags ignored, as hybrid.py:487
    main(a)
