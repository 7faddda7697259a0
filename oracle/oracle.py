"""CPU oracle -- Python face of oracle/fusion_oracle.c.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under fusion_amd/ does.  Parity status is stated in fusion_oracle.c's
header: fuse / BM25 / Metrics / cos-sim + dot scoring / chunked top-k search / SPLADE pooling /
the weight-grid loop / the score-distribution tables are PINNED on tests/golden (made by the
reference's own classes via oracle/gen_golden.py); MaxSim is UNPINNED (colbert-ai is absent
from the reference tree and from the image).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from statistics import mean

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NORMS = {None: 0, "none": 0, "min-max": 1, "z-score": 2, "arctan": 3, "percentile-rank": 4, "normal-curve-equivalent": 5}
RANK_METHODS = {"rrf": 0, "bcf": 1}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfusion_oracle.so")
    src = os.path.join(_HERE, "fusion_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("FUSION_ORACLE_SO")   # e.g. build_asan/libfusion_oracle_asan.so (make -C oracle asan; tests/test_sanitizers_cpu.py)
        if not so:
            so = os.path.join(_HERE, "libfusion_oracle.so")
            if not os.path.exists(so):
                build()
        _LIB = C.CDLL(so)
    return _LIB


def _p(a, t=C.c_void_p):
    return None if a is None else a.ctypes.data_as(t)


def _ptr_array(arrs):
    """array of pointers (NULL for None entries)"""
    A = (C.c_void_p * len(arrs))()
    for i, a in enumerate(arrs):
        A[i] = None if a is None else a.ctypes.data
    return A


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed: {rc}")


def num_threads() -> int:
    return int(lib().fzo_num_threads())


def set_threads(n: int):
    lib().fzo_set_threads(int(n))


# ---------------------------------------------------------------------------------------
# scoring
# ---------------------------------------------------------------------------------------
def normalize_rows(X: np.ndarray) -> np.ndarray:
    """F.normalize(x, p=2, dim=-1)  (splade/base.py:195-196; util.cos_sim of ST 2.2.2)"""
    X = np.ascontiguousarray(X, dtype=np.float32)
    Y = np.empty_like(X)
    _chk(lib().fzo_normalize_rows_f32(_p(X), X.shape[0], X.shape[1], X.shape[1], _p(Y), X.shape[1]), "normalize")
    return Y


def dot_scores(Qn: np.ndarray, Dn: np.ndarray, fma_chain: bool = False) -> np.ndarray:
    """torch.mm(q, d.t())  (splade/base.py:197)"""
    Qn = np.ascontiguousarray(Qn, dtype=np.float32)
    Dn = np.ascontiguousarray(Dn, dtype=np.float32)
    Q, d = Qn.shape
    N = Dn.shape[0]
    out = np.empty((Q, N), dtype=np.float32)
    f = lib().fzo_dot_scores_f32_fma if fma_chain else lib().fzo_dot_scores_f32
    _chk(f(_p(Qn), _p(Dn), Q, N, d, _p(out), N), "dot_scores")
    return out


def cos_scores(Qe: np.ndarray, De: np.ndarray, fma_chain: bool = False) -> np.ndarray:
    """util.cos_sim as called at hybrid.py:103"""
    return dot_scores(normalize_rows(Qe), normalize_rows(De), fma_chain)


def maxsim(Qtok: np.ndarray, Dtok: np.ndarray, Doff: np.ndarray) -> np.ndarray:
    """exact ColBERT late interaction. Qtok [Q,Lq,dim], Dtok [sumL,dim] packed, Doff [N+1]"""
    Qtok = np.ascontiguousarray(Qtok, dtype=np.float32)
    Dtok = np.ascontiguousarray(Dtok, dtype=np.float32)
    Doff = np.ascontiguousarray(Doff, dtype=np.int64)
    Q, Lq, dim = Qtok.shape
    N = len(Doff) - 1
    out = np.empty((Q, N), dtype=np.float32)
    _chk(lib().fzo_maxsim_f32(_p(Qtok), _p(Dtok), _p(Doff), Q, Lq, N, dim, _p(out), N), "maxsim")
    return out


def search(Qe: np.ndarray, De: np.ndarray, k: int, similarity: str = "cos_sim"):
    """BaseModel.search (splade/base.py:199-251) == util.semantic_search (hybrid.py:103): scores -> per-query top-k,
    sorted by score descending.  Chunking does not change the result set; ties (implementation-defined in the
    reference: unsorted topk + heap order) -> ascending document index.  Returns (scores [Q,k], ids [Q,k])."""
    S = cos_scores(Qe, De) if similarity == "cos_sim" else dot_scores(Qe, De)
    return topk_rows(S, min(k, S.shape[1]))


def splade_pool(logits: np.ndarray, lens: np.ndarray, pooling: str = "max") -> np.ndarray:
    """SPLADE.forward (splade/splade.py:88-99): amax (or sum) over the sequence of log1p(relu(logits * mask)).
    logits [B,L,V] fp32, lens [B] = number of attended tokens per sequence (mask = position < len)."""
    logits = np.asarray(logits, dtype=np.float32)
    B, L, V = logits.shape
    mask = (np.arange(L)[None, :] < np.asarray(lens)[:, None]).astype(np.float32)[:, :, None]
    act = np.log1p(np.maximum(logits * mask, np.float32(0.0)), dtype=np.float32)
    return act.sum(axis=1, dtype=np.float32) if pooling == "sum" else act.max(axis=1)


# ---------------------------------------------------------------------------------------
# ordering
# ---------------------------------------------------------------------------------------
def sort_rows_desc(keys: np.ndarray, init_order: np.ndarray | None = None, row_len: np.ndarray | None = None,
                   want_rank: bool = False):
    """Stable descending sort per row (Python sorted(reverse=True): bm25.py:104, hybrid.py:306).
    Returns (order, sorted_keys[, rank])."""
    assert keys.dtype in (np.float32, np.float64) and keys.ndim == 2
    keys = np.ascontiguousarray(keys)
    rows, n = keys.shape
    order = np.full((rows, n), -1, dtype=np.int32)
    sk = np.full((rows, n), -np.inf, dtype=keys.dtype)
    rank = np.full((rows, n), -1, dtype=np.int32) if want_rank else None
    if init_order is not None:
        init_order = np.ascontiguousarray(init_order, dtype=np.int32)
    if row_len is not None:
        row_len = np.ascontiguousarray(row_len, dtype=np.int32)
    _chk(lib().fzo_sort_rows_desc(_p(keys), 32 if keys.dtype == np.float32 else 64, _p(init_order), _p(row_len),
                                  rows, n, n, _p(order), _p(sk), _p(rank)), "sort_rows_desc")
    return (order, sk, rank) if want_rank else (order, sk)


def insertion_order(orders: list[np.ndarray], lens: np.ndarray, N: int):
    S = len(orders)
    Q = orders[0].shape[0]
    ld = orders[0].shape[1]
    orders = [np.ascontiguousarray(o, dtype=np.int32) for o in orders]
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    ins = np.full((Q, ld), -1, dtype=np.int32)
    U = np.zeros((Q,), dtype=np.int32)
    _chk(lib().fzo_insertion_order(_ptr_array(orders), _p(lens), S, Q, N, ld, _p(ins), _p(U)), "insertion_order")
    return ins, U


def topk_rows(scores: np.ndarray, k: int, id_base: int = 0):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, n = scores.shape
    os_ = np.empty((rows, k), dtype=np.float32)
    oi = np.empty((rows, k), dtype=np.int64)
    _chk(lib().fzo_topk_rows_f32(_p(scores), rows, n, n, k, C.c_int64(id_base), _p(os_), _p(oi)), "topk_rows")
    return os_, oi


def topk_merge(in_scores: np.ndarray, in_ids: np.ndarray):
    """[G,rows,k] x2 -> [rows,k] x2"""
    in_scores = np.ascontiguousarray(in_scores, dtype=np.float32)
    in_ids = np.ascontiguousarray(in_ids, dtype=np.int64)
    G, rows, k = in_scores.shape
    os_ = np.empty((rows, k), dtype=np.float32)
    oi = np.empty((rows, k), dtype=np.int64)
    _chk(lib().fzo_topk_merge(_p(in_scores), _p(in_ids), G, rows, k, _p(os_), _p(oi)), "topk_merge")
    return os_, oi


# ---------------------------------------------------------------------------------------
# fusion on dense planes
# ---------------------------------------------------------------------------------------
def fuse_rank(ranks: list[np.ndarray], lens: np.ndarray, method: str) -> np.ndarray:
    """rrf/bcf (hybrid.py:206-211,248-252,301-304). ranks[s] [Q,N] int32 (-1 absent), lens [S,Q]."""
    S = len(ranks)
    Q, N = ranks[0].shape
    ranks = [np.ascontiguousarray(r, dtype=np.int32) for r in ranks]
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    fused = np.empty((Q, N), dtype=np.float64)
    _chk(lib().fzo_fuse_rank_f64(_ptr_array(ranks), _p(lens), S, Q, N, N, RANK_METHODS[method], _p(fused)), "fuse_rank")
    return fused


def row_stats(scores: np.ndarray, rank: np.ndarray | None, norm: str):
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows, N = scores.shape
    a = np.empty(rows, dtype=np.float32)
    b = np.empty(rows, dtype=np.float32)
    if rank is not None:
        rank = np.ascontiguousarray(rank, dtype=np.int32)
    _chk(lib().fzo_row_stats_f32(_p(scores), _p(rank), rows, N, N, NORMS[norm], _p(a), _p(b)), "row_stats")
    return a, b


def fuse_nsf(planes: list[np.ndarray], ranks: list[np.ndarray | None] | None, weights, norm: str,
             distr: list[np.ndarray | None] | None = None) -> np.ndarray:
    """normalise -> weight -> sum (hybrid.py:212-214,254-280,291,301-304)."""
    S = len(planes)
    Q, N = planes[0].shape
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    if ranks is not None:
        ranks = [None if r is None else np.ascontiguousarray(r, dtype=np.int32) for r in ranks]
    w = np.asarray(weights, dtype=np.float64).astype(np.float32)  # NumPy-2: python float weight -> fp32
    fused = np.empty((Q, N), dtype=np.float32)
    dptr, P = None, None
    if distr is not None:
        distr = [np.zeros(1, np.float32) if d is None else np.ascontiguousarray(d, dtype=np.float32) for d in distr]
        dptr = _ptr_array(distr)
        P = np.array([len(d) for d in distr], dtype=np.int32)
    _chk(lib().fzo_fuse_nsf_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), _p(w), S, Q, N, N,
                                NORMS[norm], dptr, _p(P), _p(fused), None, None), "fuse_nsf")
    return fused


def fuse_none(planes, ranks, weights) -> np.ndarray:
    """nsf + 'none'/unknown normalisation: float64 passthrough (hybrid.py:280,291,304)."""
    S = len(planes)
    Q, N = planes[0].shape
    planes = [np.ascontiguousarray(p, dtype=np.float32) for p in planes]
    if ranks is not None:
        ranks = [None if r is None else np.ascontiguousarray(r, dtype=np.int32) for r in ranks]
    w = np.asarray(weights, dtype=np.float64)
    fused = np.empty((Q, N), dtype=np.float64)
    _chk(lib().fzo_fuse_none_f64(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), _p(w), S, Q, N, N,
                                 _p(fused)), "fuse_none")
    return fused


def is_wide_weight(w) -> bool:
    """NumPy-2 promotion of `np.float32 score * w` (hybrid.py:291): float64 only for an np.float64 weight (a Python
    float / int is a weak scalar, np.float32 / np.float16 stay float32).  np.float64 subclasses float: test it first."""
    return isinstance(w, (np.float64, np.longdouble))


def fuse_wsum(planes, ranks, weights, narrow) -> np.ndarray:
    """Weight-and-sum with per-system promotion (see fzo_fuse_wsum_f64): planes fp32 or fp64, -> fp64."""
    S = len(planes)
    Q, N = planes[0].shape
    planes = [np.ascontiguousarray(p, dtype=(np.float64 if p.dtype == np.float64 else np.float32)) for p in planes]
    if ranks is not None:
        ranks = [None if r is None else np.ascontiguousarray(r, dtype=np.int32) for r in ranks]
    w = np.asarray([float(x) for x in weights], dtype=np.float64)
    pf = np.array([p.dtype == np.float64 for p in planes], dtype=np.int32)
    nr = np.array([bool(n) for n in narrow], dtype=np.int32)
    fused = np.empty((Q, N), dtype=np.float64)
    _chk(lib().fzo_fuse_wsum_f64(_ptr_array(planes), _p(pf), None if ranks is None else _ptr_array(ranks), _p(w), _p(nr),
                                 S, Q, N, N, _p(fused)), "fuse_wsum")
    return fused


# ---------------------------------------------------------------------------------------
# list-of-dict adapter: the reference's Aggregator.fuse signature (hybrid.py:170-220)
# ---------------------------------------------------------------------------------------
def lists_to_planes(ranked_lists: dict[str, list[list[dict]]]):
    """RankedLists -> dense planes (+ the raw float64 scores).  convert2dict (hybrid.py:222-233) collapses duplicate ids:
    the FIRST position is kept, the LAST score wins; rank = position among unique ids."""
    systems = list(ranked_lists.keys())
    Q = len(ranked_lists[systems[0]])
    assert all(len(v) == Q for v in ranked_lists.values()), "varying number of queries across systems"  # hybrid.py:192
    # corpus position = first-seen order over (system, query, rank); any bijection works
    pos: dict = {}
    for s in systems:
        for q in range(Q):
            for x in ranked_lists[s][q]:
                if x["corpus_id"] not in pos:
                    pos[x["corpus_id"]] = len(pos)
    N = max(1, len(pos))
    ids = np.empty(N, dtype=object)
    for k, v in pos.items():
        ids[v] = k
    S = len(systems)
    planes = [np.zeros((Q, N), dtype=np.float32) for _ in range(S)]
    planes64 = [np.zeros((Q, N), dtype=np.float64) for _ in range(S)]   # the raw Python floats ('none' keeps them, hybrid.py:280)
    ranks = [np.full((Q, N), -1, dtype=np.int32) for _ in range(S)]
    orders = [np.full((Q, N), -1, dtype=np.int32) for _ in range(S)]
    lens = np.zeros((S, Q), dtype=np.int32)
    for si, s in enumerate(systems):
        for q in range(Q):
            d: dict = {}
            for x in ranked_lists[s][q]:
                d[x["corpus_id"]] = x["score"]  # python dict: first position kept, last value wins
            for r, (cid, sc) in enumerate(d.items()):
                j = pos[cid]
                planes[si][q, j] = np.float32(sc)      # torch.tensor(list(values), dtype=float32), hybrid.py:255
                planes64[si][q, j] = float(sc)
                ranks[si][q, j] = r
                orders[si][q, r] = j
            lens[si, q] = len(d)
    return systems, ids, planes, ranks, orders, lens, planes64


def fuse_lists(ranked_lists, method, normalization=None, linear_weights=None, percentile_distributions=None,
               return_topk: int = 1000):
    """Same signature and result type as the reference's Aggregator.fuse (hybrid.py:170-220)."""
    systems, ids, planes, ranks, orders, lens, raw64 = lists_to_planes(ranked_lists)
    Q, N = planes[0].shape
    if method in ("rrf", "bcf"):
        fused = fuse_rank(ranks, lens, method)
    elif method == "nsf":
        w = [linear_weights[s] for s in systems]  # KeyError if a system lacks a weight (hybrid.py:214)
        distr = [percentile_distributions.get(s) for s in systems]  # AttributeError on None (hybrid.py:213)
        wide = [is_wide_weight(x) for x in w]
        if normalization in ("min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"):
            tabled = normalization in ("percentile-rank", "normal-curve-equivalent")
            if not any(wide):
                fused = fuse_nsf(planes, ranks, w, normalization, distr if tabled else None)
            else:   # np.float64 weights (the tuning grid): transform in fp32, weight and sum with NumPy's promotion
                T = [fuse_nsf([planes[i]], [ranks[i]], [1.0], normalization, [distr[i]] if tabled else None) for i in range(len(systems))]
                fused = fuse_wsum(T, ranks, w, [not x for x in wide])
        else:       # 'none' / unknown: the raw Python floats, float64 throughout (hybrid.py:280)
            fused = fuse_wsum(raw64, ranks, w, [False] * len(systems))
    else:
        # hybrid.py:203-216: unknown method -> raw scores summed (no transform, no weights)
        fused = fuse_wsum(raw64, ranks, [1.0] * len(systems), [False] * len(systems))
    ins, U = insertion_order(orders, lens, N)
    order, sk = sort_rows_desc(fused, init_order=ins, row_len=U)
    out = []
    for q in range(Q):
        out.append([{"corpus_id": ids[order[q, r]], "score": (float(sk[q, r]) if fused.dtype == np.float64 else np.float32(sk[q, r]))}
                    for r in range(U[q])])
    return out[:return_topk]  # slices QUERIES (hybrid.py:220)


def tune_lists(ranked_lists, normalization, weight_combinations, labels, percentile_distributions=None):
    """The weight-grid loop (hybrid.py:404-426): one full fuse + run_evaluation (hybrid.py:24-29) per weight vector."""
    ev = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    out = []
    for w in weight_combinations:
        fused = fuse_lists(ranked_lists, "nsf", normalization, w, percentile_distributions)
        out.append(ev.compute_all_metrics(labels, [[x["corpus_id"] for x in r] for r in fused]))
    return out


def transform_lists(lists: list[list[dict]], normalization: str, distr=None) -> list[dict]:
    """Aggregator.transform_scores(convert2dict(list)) per query (hybrid.py:378): {corpus_id: transformed score}."""
    if normalization in (None, "none"):
        return [{x["corpus_id"]: x["score"] for x in l} for l in lists]
    out = []
    for l in lists:
        _, ids, planes, ranks, orders, lens, _raw = lists_to_planes({"s": [l]})
        t = fuse_nsf(planes, ranks, [1.0], normalization, None if distr is None else [distr])   # fl32(t * 1) + 0 == t
        out.append({ids[j]: t[0, j] for j in orders[0][0, : lens[0, 0]]})
    return out


def score_tables(ranked_lists, normalization, n_points, distributions=None):
    """The quantile tables of hybrid.py:390-397: per system, all transformed scores of all queries, minus the zeros and
    the two smallest distinct values, at the quantiles linspace(0, 1, n_points + 1) (float64, linear interpolation).
    Returns ({system: all scores}, {system: table})."""
    scores, tables = {}, {}
    for s, lists in ranked_lists.items():
        tr = transform_lists(lists, normalization, None if not distributions else distributions.get(s))
        v = np.array([float(x) for t in tr for x in t.values()], dtype=np.float64)
        scores[s] = v
        two = np.unique(v)[:2]
        kept = v[(v != 0.0) & ~np.isin(v, two)]
        tables[s] = np.quantile(kept, np.linspace(0, 1, n_points + 1)) if kept.size else np.full(n_points + 1, np.nan)
    return scores, tables


# ---------------------------------------------------------------------------------------
# BM25 (bm25.py:33-161)
# ---------------------------------------------------------------------------------------
class BM25:
    """Restates TFIDF.__init__ index building (bm25.py:37-43,52-83) + BM25 (bm25.py:129-156)."""

    def __init__(self, corpus: list[str], k1: float, b: float):
        self.k1, self.b = k1, b
        self.N = len(corpus)
        toks = [doc.split() for doc in corpus]
        self.vocab: dict[str, int] = {}
        for t in toks:
            for w in t:
                if w not in self.vocab:
                    self.vocab[w] = len(self.vocab)
        V = len(self.vocab)
        post: list[dict[int, int]] = [dict() for _ in range(V)]
        for i, t in enumerate(toks):
            for w in t:
                d = post[self.vocab[w]]
                d[i] = d.get(i, 0) + 1
        df = np.array([len(p) for p in post], dtype=np.int64)
        self.df = df
        self.idf = np.array([self._compute_idf(int(x)) for x in df], dtype=np.float64)
        self.doc_len = np.array([len(t) for t in toks], dtype=np.int32)
        self.avgdl = float(mean(self.doc_len.tolist()))  # bm25.py:138 statistics.mean
        self.toff = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(df, out=self.toff[1:])
        self.pdoc = np.empty(int(self.toff[-1]), dtype=np.int32)
        self.ptf = np.empty(int(self.toff[-1]), dtype=np.int32)
        for t, p in enumerate(post):
            o = int(self.toff[t])
            for k, (dj, tf) in enumerate(sorted(p.items())):
                self.pdoc[o + k] = dj
                self.ptf[o + k] = tf

    TFIDF_SCORE = False

    def _compute_idf(self, df: int) -> float:
        return math.log10((self.N - df + 0.5) / (df + 0.5))          # bm25.py:145-147

    def update_params(self, k1: float, b: float) -> None:              # bm25.py:158-161
        self.k1, self.b = k1, b

    def scores(self, queries: list[str]) -> np.ndarray:
        qt = [[self.vocab.get(w, -1) for w in q.split()] for q in queries]
        qoff = np.zeros(len(qt) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in qt], out=qoff[1:])
        qterms = np.array([t for x in qt for t in x] or [0], dtype=np.int32)
        Q = len(queries)
        out = np.empty((Q, self.N), dtype=np.float64)
        if self.TFIDF_SCORE:
            _chk(lib().fzo_tfidf_scores_f64(_p(self.toff), _p(self.pdoc), _p(self.ptf), _p(self.idf), _p(qoff), _p(qterms), Q, self.N, _p(out), self.N), "tfidf")
            return out
        _chk(lib().fzo_bm25_scores_f64(_p(self.toff), _p(self.pdoc), _p(self.ptf), _p(self.idf), _p(self.doc_len),
                                       C.c_double(self.avgdl), C.c_double(self.k1), C.c_double(self.b), _p(qoff), _p(qterms),
                                       Q, self.N, _p(out), self.N), "bm25")
        return out

    def search_all(self, queries: list[str], top_k: int):
        """bm25.py:90-106: every doc scored, stable sort desc, [:top_k]"""
        sc = self.scores(queries)
        order, sk = sort_rows_desc(sc)
        return [[{"corpus_id": int(order[q, r]), "score": float(sk[q, r])} for r in range(min(top_k, self.N))]
                for q in range(len(queries))]


class TFIDF(BM25):
    """TFIDF (bm25.py:33-127): score += tf * idf with idf = log10((N + 1) / (df + 1)) (bm25.py:86-88, 108-115).  (In the reference BM25
    derives from TFIDF; the restatement shares the index building the other way round.)"""
    TFIDF_SCORE = True

    def __init__(self, corpus: list[str]):
        super().__init__(corpus, 0.0, 0.0)

    def _compute_idf(self, df: int) -> float:
        return math.log10((self.N + 1) / (df + 1))


class AtireBM25(BM25):
    """AtireBM25 (bm25.py:164-173): BM25's score with TFIDF's idf."""

    def _compute_idf(self, df: int) -> float:
        return math.log10((self.N + 1) / (df + 1))


# ---------------------------------------------------------------------------------------
# Metrics (metrics.py:25-162) -- restated 1:1 including its non-standard nDCG
# ---------------------------------------------------------------------------------------
class Metrics:
    def __init__(self, recall_at_k, map_at_k=(), mrr_at_k=(), ndcg_at_k=()):
        self.recall_at_k, self.map_at_k, self.mrr_at_k, self.ndcg_at_k = list(recall_at_k), list(map_at_k), list(mrr_at_k), list(ndcg_at_k)

    def compute_all_metrics(self, all_ground_truths, all_results):
        sc = {}
        for k in self.recall_at_k:
            sc[f"recall@{k}"] = mean(self.recall(g, r, k) for g, r in zip(all_ground_truths, all_results))
        for k in self.map_at_k:
            sc[f"map@{k}"] = mean(self.average_precision(g, r, k) for g, r in zip(all_ground_truths, all_results))
        for k in self.mrr_at_k:
            sc[f"mrr@{k}"] = mean(self.reciprocal_rank(g, r, k) for g, r in zip(all_ground_truths, all_results))
        for k in self.ndcg_at_k:
            sc[f"ndcg@{k}"] = mean(self.ndcg(g, r, k) for g, r in zip(all_ground_truths, all_results))
        sc["r-precision"] = mean(self.r_precision(g, r) for g, r in zip(all_ground_truths, all_results))
        return sc

    @staticmethod
    def recall(gold, res, k):  # metrics.py:125-136
        return sum(1 for d in res[:k] if d in gold) / len(gold)

    @staticmethod
    def precision(gold, res, k):  # metrics.py:138-149
        return sum(1 for d in res[:k] if d in gold) / len(res[:k])

    def average_precision(self, gold, res, k):  # metrics.py:72-83
        return sum(self.precision(gold, res, i + 1) if d in gold else 0 for i, d in enumerate(res[:k])) / len(gold)

    @staticmethod
    def reciprocal_rank(gold, res, k):  # metrics.py:85-95
        return max([1 / (i + 1) if d in gold else 0.0 for i, d in enumerate(res[:k])])

    @staticmethod
    def ndcg(gold, res, k):  # metrics.py:97-110 (position 0 undiscounted; 1/log2(i+1) for i>=1; idcg over all gold)
        rel = [1 if d in gold else 0 for d in res[:k]]
        dcg = rel[0] + sum(rel[i] / np.log2(i + 1) for i in range(1, len(rel)))
        idcg = 1 + sum(1 / np.log2(i + 1) for i in range(1, len(gold)))
        return (dcg / idcg) if idcg != 0 else 0

    @staticmethod
    def r_precision(gold, res):  # metrics.py:112-123
        R = len(gold)
        return sum(1 for d in res[:R] if d in gold) / R
