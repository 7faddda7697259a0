"""MI355X-native drop-in for the reference's lexical module `src/retrievers/bm25.py` (291 lines): the three retrieval classes --
TFIDF (:33-127), BM25 (:129-161), AtireBM25 (:164-173) -- with the reference's exact arithmetic, and the working form of its driver
(`main`, :176-272; `scripts/run_bm25.sh`): the k1 x b grid search, the evaluation run and the hard-negatives extraction.

The inverted index is built on the host from whitespace tokens (as the reference builds it: the text is pre-processed BEFORE it gets
here, bm25.py:216-220); every (query, document) pair is scored on the device in float64 in the reference's expression and addition order
(csrc/bm25.hip) and ranked by the stable row sort (csrc/sort.hip): scores, lists and therefore every metric are bit-identical to the
Python loops (tests/golden/bm25.json, bm25_family.json -- outputs of the reference itself).

The grid search (`BM25.tune`) is to BM25 what `Aggregator.tune` is to the fusion weights: the index and the queries' postings stay
resident, each of the 187 (k1, b) pairs is one norm pass + one scoring launch + one row sort, and only the ranks of the gold documents
come back -- every metric of bm25.py:223,230 is a function of them.

Reference defects in the driver, not reproduced (documented in DESIGN.md):
  * bm25.py:235 `scores.pop('recall')` raises KeyError with the Metrics of src/utils/metrics.py (its results are flat), :236
    `DataFrame.append` left pandas in 2.0: the grid loop dies on its first row.  The CSV the loop was meant to write -- one row per
    (k1, b) with the recall columns -- is written here;
  * the k1 = 0 column: with np.float64 parameters (np.arange, :226) a document that lacks a query term scores idf * 0 / 0 = NaN (:154) and
    sorted() is handed NaN keys.  Here a posting-less (term, document) pair adds nothing, i.e. k1 = 0 is the binary model sum(idf).
"""
from __future__ import annotations

import argparse
import itertools
import json
import math
import os
import pickle
import time
from collections import Counter
from os.path import join
from statistics import mean

import numpy as np
import torch

from .. import ops
from ..planes import RankedSystem


LEXICAL_MIN_ZERO_SHARE = 0.3   # expected share of exact zeros per row from which the ranking sort's zero-compacting instantiation is asked for


def expected_zero_share(df: np.ndarray, n_docs: int, query_terms: list[list[int]]) -> float:
    """Mean over the queries of prod over a query's DISTINCT in-vocabulary terms of (1 - df_t / N): the share of documents expected to hold
    none of its terms (terms taken as independent) -- i.e. to score exactly 0.0.  Host arithmetic on the index's df table (one np.unique over
    the batch's (query, term) pairs), no device work."""
    Q = len(query_terms)
    if Q == 0 or n_docs <= 0:
        return 0.0
    lens = np.fromiter((len(t) for t in query_terms), dtype=np.int64, count=Q)
    flat = np.fromiter((x for t in query_terms for x in t), dtype=np.int64, count=int(lens.sum()))
    qid = np.repeat(np.arange(Q, dtype=np.int64), lens)
    keep = flat >= 0
    V = max(int(df.shape[0]), 1)
    pairs = np.unique(qid[keep] * V + flat[keep])                     # distinct (query, term) pairs
    miss = 1.0 - df.astype(np.float64) / float(n_docs)
    with np.errstate(divide="ignore"):
        logs = np.log(miss[pairs % V])                                # (-inf for a term every document holds: that query expects no zeros)
    per_query = np.exp(np.bincount(pairs // V, weights=logs, minlength=Q)) if pairs.size else np.ones(Q)
    per_query = np.nan_to_num(per_query, nan=0.0)
    return float(per_query.mean())


class TFIDF:
    """bm25.py:33-127: score(q, d) = sum over q.split(), in order, of tf(t, d) * idf(t), idf = log10((N + 1) / (df + 1))."""

    def __init__(self, corpus: list[str], device="cuda"):
        self.corpus = corpus
        self.corpus_size = len(corpus)
        self.device = torch.device(device)
        toks = [doc.split() for doc in corpus]
        self.vocab: dict[str, int] = {}
        for t in toks:
            for w in t:
                self.vocab.setdefault(w, len(self.vocab))
        V = len(self.vocab)
        # tf postings (bm25.py:58-65) and df (bm25.py:67-75)
        tid = np.fromiter((self.vocab[w] for t in toks for w in t), dtype=np.int64, count=sum(len(t) for t in toks))
        did = np.repeat(np.arange(len(toks), dtype=np.int64), [len(t) for t in toks])
        key = tid * max(self.corpus_size, 1) + did
        uniq, tf = np.unique(key, return_counts=True)
        pt, pd = uniq // max(self.corpus_size, 1), uniq % max(self.corpus_size, 1)
        self.df_host = np.bincount(pt, minlength=V).astype(np.int64)
        self.idf_host = np.array([self._compute_idf(int(x)) for x in self.df_host], dtype=np.float64)
        self.doc_len_host = np.array([len(t) for t in toks], dtype=np.int32)
        toff = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(self.df_host, out=toff[1:])
        self._toff_host, self._pdoc_host, self._ptf_host = toff, pd.astype(np.int32), tf.astype(np.int32)
        d = self.device
        self.toff = torch.from_numpy(toff).to(d)
        self.pdoc = torch.from_numpy(self._pdoc_host).to(d)
        self.ptf = torch.from_numpy(self._ptf_host).to(d)
        self.idf = torch.from_numpy(self.idf_host).to(d)
        self.doc_len = torch.from_numpy(self.doc_len_host).to(d)
        # where every term's postings cross the document slices one workgroup scores: per index, like the idf table
        self.slice_off = ops.bm25_slice_offsets(self.toff, self.pdoc, self.corpus_size) if self.device.type == "cuda" and V > 0 else None
        self._qcache = None
        self.zero_share_estimate = 0.0   # of the last list of queries (set by _query_csr)

    def __repr__(self):
        return f"{self.__class__.__name__}".lower()                       # bm25.py:45-46: names the pickles of save_indexes

    def get_vocab(self):
        """bm25.py:48-50: the vocabulary in alphabetical order."""
        return sorted(self.vocab)

    def _compute_idf(self, df: int) -> float:
        return math.log10((self.corpus_size + 1) / (df + 1))              # bm25.py:86-88

    # -- queries -> CSR of term ids (kept for the last list of queries: the grid search scores the same queries 187 times) --
    def _query_csr(self, queries: list[str]):
        if self._qcache is not None and (self._qcache[0] is queries or self._qcache[0] == queries):
            return self._qcache[1], self._qcache[2]
        qt = [[self.vocab.get(w, -1) for w in q.split()] for q in queries]   # query terms are NOT de-duplicated (bm25.py:112,152)
        qoff = np.zeros(len(qt) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in qt], out=qoff[1:])
        flat = np.array([t for x in qt for t in x] or [0], dtype=np.int32)
        qoff_d, flat_d = torch.from_numpy(qoff).to(self.device), torch.from_numpy(flat).to(self.device)
        self._qcache = (list(queries), qoff_d, flat_d)   # (a copy: a caller that edits its list in place gets a fresh CSR)
        self.zero_share_estimate = expected_zero_share(self.df_host, self.corpus_size, qt)
        return qoff_d, flat_d

    def scores(self, queries: list[str], want_f32: bool = False):
        """[Q, N] float64 plane (want_f32: and its float32 rounding, from the same launch)."""
        qoff, flat = self._query_csr(queries)
        return ops.tfidf_scores(self.toff, self.pdoc, self.ptf, self.idf, qoff, flat, len(queries), self.corpus_size,
                                slice_off=self.slice_off, want_f32=want_f32)

    def search_device(self, queries: list[str], ids: np.ndarray | None = None) -> RankedSystem:
        sc64, sc32 = self.scores(queries, want_f32=True)
        Q, N = sc64.shape
        stats4 = None   # mean | std | min | max of every list's float32 scores: by-products of the ranking sort (rows that fit one workgroup)
        if N <= ops.sort_max_n(torch.float64) and Q > 0:
            stats4 = torch.empty((4, Q), dtype=torch.float32, device=self.device)
        # ranks from the float64 scores, ties -> ascending index.  lexical: when the batch's rows are expected to be mostly exact zeros (documents
        # that share no term with the query) the sort's zero-compacting instantiation orders only the non-zero keys -- a host-side ESTIMATE picks
        # the instantiation (it costs rows without zeros 1-7 %), the kernel decides row by row from the actual count; same outputs either way
        order, _, rank = ops.sort_rows_desc(sc64, want_keys=False, want_rank=True, stats_out=stats4, lexical=self.zero_share_estimate >= LEXICAL_MIN_ZERO_SHARE)
        lens = torch.full((Q,), N, dtype=torch.int32, device=self.device)
        # float32 plane for the normalisations (torch.tensor(scores, dtype=float32), hybrid.py:255), written by the scoring kernel from
        # the accumulators it holds (no conversion pass); the float64 scores stay for the 'none' passthrough, which keeps BM25's Python
        # floats (hybrid.py:280).  Rounding is monotone: the float32 scores are in descending order along the float64 ranking too.
        return RankedSystem(scores=sc32, order=order, rank=rank, lens=lens,
                            ids=np.arange(N, dtype=np.int64) if ids is None else ids, full=True,
                            scores64=sc64, score_sorted=True, stats4=stats4)

    def ranked_positions(self, queries: list[str], top_k: int, budget_bytes: int = 48 << 30) -> np.ndarray:
        """[Q, min(top_k, N)] corpus positions of the first top_k entries of every ranked list (what the driver keeps of search_all,
        bm25.py:248-249), the queries taken in chunks whose planes fit the budget.  A corpus that fits one workgroup's row (28,672 documents)
        is ranked in full (search_device); a longer one -- mMARCO's 8.8 M passages -- is CUT, not ranked: every 28,672-document stretch of the
        float64 score row is sorted on its own, its first top_k entries survive, and the survivors (kept in corpus order, so that the stable
        sort breaks ties by ascending index as the full sort does) go round again until one row holds them -- the same first top_k entries as
        the full ranking, without its cross-chunk ranking of all N documents."""
        N, k = self.corpus_size, min(top_k, self.corpus_size)
        W = ops.sort_max_n(torch.float64)
        per_pair = 20 if N <= W else 12 + 24 * min(1.0, (k + 1) / W)      # planes alive per (query, document)
        step = max(1, int(budget_bytes // max(1, int(per_pair * ops.round_up(max(N, 1), 64)))))
        out = np.empty((len(queries), k), dtype=np.int64)
        for lo in range(0, len(queries), step):
            chunk = queries[lo:lo + step]
            if N <= W:
                out[lo:lo + len(chunk)] = self.search_device(chunk).order[:, :k].cpu().numpy()
                continue
            sc = self.scores(chunk)                                        # [q, N] float64
            q = sc.shape[0]
            ids = None                                                     # [q, n] corpus positions of the surviving columns (None: the identity)
            while True:
                n = sc.shape[1]
                C = -(-n // W)
                if C > 1:                                                  # pad to whole stretches: -inf never survives a real score, NaN sorts first as everywhere
                    pad = C * W - n
                    if pad:
                        sc = torch.cat([sc, torch.full((q, pad), float("-inf"), dtype=sc.dtype, device=sc.device)], 1)
                        if ids is not None:
                            ids = torch.cat([ids, torch.full((q, pad), -1, dtype=torch.int64, device=sc.device)], 1)
                    rows = sc.reshape(q * C, W)
                else:
                    rows = sc
                order, keys, _ = ops.sort_rows_desc(ops.as_plane(rows.contiguous()), want_keys=True)
                kk = min(k, rows.shape[1])
                order, keys = order[:, :kk].long(), keys[:, :kk]
                if C > 1:
                    base = (torch.arange(C, device=sc.device) * W).repeat(q)[:, None]
                    cols = (order + base).reshape(q, C * kk)               # columns of this level's row, stretch by stretch: corpus order is kept
                    ids = cols if ids is None else torch.gather(ids, 1, cols)
                    sc = keys.reshape(q, C * kk)
                    continue
                pos = order if ids is None else torch.gather(ids, 1, order)
                out[lo:lo + q] = pos[:, :k].cpu().numpy()
                break
        return out

    def search_all(self, queries: list[str], top_k: int) -> list:
        """bm25.py:90-106: every document scored (zero scores included), stable sort desc, [:top_k]."""
        t0 = time.perf_counter()
        out = self.search_device(queries).to_lists(top_k)
        if queries:
            print(f"Avg. latency (ms/quey): {((time.perf_counter() - t0) / len(queries)) * 1000}")   # (sic) bm25.py:97
        return out

    def search(self, query: str, top_k: int) -> list:
        return self.search_device([query]).to_lists(top_k)[0]

    def save_indexes(self, output_dir: str, dataset: str) -> None:
        """bm25.py:117-126: the four indexes as pickles, in the reference's own Python types (vocab: set, tf: {word: {doc: count}},
        df: Counter, idf: {word: float}) so that whatever read the reference's files reads these."""
        words = list(self.vocab)
        tf = {}
        for w, t in self.vocab.items():
            lo, hi = int(self._toff_host[t]), int(self._toff_host[t + 1])
            tf[w] = dict(zip(self._pdoc_host[lo:hi].tolist(), self._ptf_host[lo:hi].tolist()))
        payload = {"vocab": set(words), "tf": tf, "df": Counter({w: int(self.df_host[t]) for w, t in self.vocab.items()}),
                   "idf": {w: float(self.idf_host[t]) for w, t in self.vocab.items()}}
        for name, obj in payload.items():
            with open(join(output_dir, f"{self.__repr__()}_{name}_{dataset}.pkl"), "wb") as f:
                pickle.dump(obj, f)


class BM25(TFIDF):
    """bm25.py:129-161: score(q, d) = sum of idf * tf (k1 + 1) / (tf + k1 (1 - b + b |d| / avgdl)), idf = log10((N - df + .5) / (df + .5))
    (can be <= 0), avgdl = statistics.mean(doc_len)."""

    USE_POSTING_VALUES = True   # False: the per-posting expression (fz_bm25_scores_f64_f32): A/B runs and tests of the two forms

    def __init__(self, corpus: list[str], k1: float, b: float, device="cuda"):
        self.k1, self.b = k1, b
        self._pval = None
        super().__init__(corpus, device=device)
        self.avgdl = float(mean(self.doc_len_host.tolist())) if self.corpus_size else 0.0   # bm25.py:138
        self._norm_key, self._norm = None, None

    def _compute_idf(self, df: int) -> float:
        N = self.corpus_size
        return math.log10((N - df + 0.5) / (df + 0.5))                    # bm25.py:145-147

    def update_params(self, k1: float, b: float) -> None:
        self.k1, self.b = k1, b

    def _doc_norm(self) -> torch.Tensor:
        key = (float(self.k1), float(self.b))
        if self._norm_key != key:
            self._norm = ops.bm25_doc_norms(self.doc_len, self.avgdl, self.k1, self.b)
            # every posting's whole term idf * (tf (k1 + 1)) / (tf + norm_d) for this (k1, b): per index, like the idf table -- the scoring
            # walk then only adds (no float64 division per (query, posting)); same bits
            self._pval = ops.bm25_posting_values(self.toff, self.pdoc, self.ptf, self.idf, self._norm, self.k1) if self.pdoc.numel() else None
            self._norm_key = key
        return self._norm

    def scores(self, queries: list[str], want_f32: bool = False):
        """[Q, N] float64 plane (want_f32: and its float32 rounding, from the same launch); query terms are NOT de-duplicated (bm25.py:152)."""
        qoff, flat = self._query_csr(queries)
        norm = self._doc_norm()
        return ops.bm25_scores(self.toff, self.pdoc, self.ptf, self.idf, self.doc_len, self.avgdl, self.k1, self.b,
                               qoff, flat, len(queries), self.corpus_size,
                               doc_norm=norm, slice_off=self.slice_off, want_f32=want_f32, pval=self._pval if self.USE_POSTING_VALUES else None)

    # -- the grid search of bm25.py:221-237 on the device --------------------------------------------------------------
    def tune(self, queries: list[str], labels: list[list], ids=None, k1_range=None, b_range=None,
             recall_at_k=(10, 100, 200, 500, 1000), top_k: int = 1000) -> list[dict]:
        """For every (k1, b) of itertools.product(k1_range, b_range) (bm25.py:226-228): what
        `update_params(k1, b); search_all(queries, top_k); Metrics(recall_at_k).compute_all_metrics(labels, ids of the lists)` reports
        (bm25.py:231-234) -- recall@k for the given cut-offs and r-precision -- as one dict per pair: {'k1', 'b', 'recall@10', ...}.
        The lists are never built: one scoring launch + one row sort per pair, and the ranks of the gold documents (a gold document at
        rank >= top_k was cut from the list: never retrieved) come back.  ids: corpus position -> dataset id (idx2id, bm25.py:214).
        The model's own k1 / b are restored afterwards."""
        from ..utils.metrics import metrics_from_gold_ranks
        k1_range = np.arange(0., 8.5, 0.5) if k1_range is None else k1_range      # bm25.py:226
        b_range = np.arange(0., 1.1, 0.1) if b_range is None else b_range        # bm25.py:227
        combos = list(itertools.product(*[k1_range, b_range]))
        Q, N = len(queries), self.corpus_size
        ids = np.arange(N, dtype=np.int64) if ids is None else np.asarray(ids)
        id2pos = {c: j for j, c in enumerate(ids.tolist())}
        gold_pos = [[id2pos.get(g, -1) for g in dict.fromkeys(gl)] for gl in labels]
        G = max(1, max((len(g) for g in gold_pos), default=1))
        gp = np.full((Q, G), -1, dtype=np.int64)
        for q, gl in enumerate(gold_pos):
            gp[q, :len(gl)] = gl
        gp_dev = torch.from_numpy(np.maximum(gp, 0)).to(self.device)
        n_gold = np.array([len(gl) for gl in labels], dtype=np.int64)             # the reference divides by len(ground_truths)
        list_len = np.full(Q, min(top_k, N), dtype=np.int64)
        INF = np.iinfo(np.int64).max
        keep = (self.k1, self.b)
        got = torch.empty((len(combos), Q, G), dtype=torch.int32, device=self.device)
        for w, (k1, b) in enumerate(combos):
            self.update_params(k1, b)
            _, _, rank = ops.sort_rows_desc(self.scores(queries), want_keys=False, want_rank=True, lexical=self.zero_share_estimate >= LEXICAL_MIN_ZERO_SHARE)
            got[w] = torch.gather(rank, 1, gp_dev)
        self.update_params(*keep)
        ranks = got.cpu().numpy().astype(np.int64)
        ranks = np.where((gp[None] >= 0) & (ranks < top_k), ranks, INF)
        names = [f"recall@{k}" for k in recall_at_k] + ["r-precision"]
        perfs = metrics_from_gold_ranks(ranks, n_gold, list_len, recall_ks=list(recall_at_k), only=names)
        return [{"k1": float(k1), "b": float(b), **p} for (k1, b), p in zip(combos, perfs)]


class AtireBM25(BM25):
    """bm25.py:164-173 (https://www.cs.otago.ac.nz/homepages/andrew/papers/2014-2.pdf): BM25's score with TFIDF's idf."""

    def _compute_idf(self, df: int) -> float:
        return math.log10((self.corpus_size + 1) / (df + 1))


# ---------------------------------------------------------------------------------------------------
# driver (bm25.py:176-291) -- same flags; data comes from local files because there is no network
# ---------------------------------------------------------------------------------------------------
def preprocess(texts: list[str], lemmatize: bool = True) -> list[str]:
    """src/data/preprocessor.py:15-76 (spaCy fr_core_news_md: punctuation, numbers and stop words dropped, lemmas, lower case).  The model is
    third-party and not installed offline: asking for it without it is an error, not a silent no-op."""
    try:
        import spacy
        nlp = spacy.load("fr_core_news_md")
    except Exception as ex:   # noqa: BLE001
        raise RuntimeError("--do_preprocessing needs spaCy with fr_core_news_md (requirements.txt:25-26 of the reference); it is not installed here. "
                           "Pass text that is already pre-processed (lower-cased lemmas, whitespace-separated) and drop the flag.") from ex
    import re
    out = []
    for doc in nlp.pipe(texts):
        toks = [(t.lemma_ if lemmatize else t.text) for t in doc
                if not t.is_punct and not (t.is_digit or t.like_num or re.match(r".*\d+", t.text)) and not t.is_stop]
        out.append(" ".join(toks).lower())
    return out


def load_data(args):
    """bm25.py:181-212 offline: `--data_dir` with corpus.jsonl ({'id', 'article'}) + questions_<split>.jsonl ({'id', 'question',
    'article_ids'}) -- the split main() picks: train for the negatives, validation for the grid search, test otherwise -- or
    `--synthetic N,Q`.  -> corpus {id: text}, qids, queries, pos_pids."""
    if getattr(args, "synthetic", None):
        n, q = (int(x) for x in args.synthetic.split(","))
        rng = np.random.default_rng(0)
        vocab = np.array([f"mot{i}" for i in range(5000)])
        p = 1.0 / np.arange(1, 5001); p /= p.sum()
        corpus = {int(i + 1): " ".join(rng.choice(vocab, size=int(rng.integers(20, 200)), p=p)) for i in range(n)}
        queries = [" ".join(rng.choice(vocab, size=int(rng.integers(4, 16)), p=p)) for _ in range(q)]
        pos = [sorted(rng.choice(np.arange(1, n + 1), size=int(rng.integers(1, 5)), replace=False).tolist()) for _ in range(q)]
        return corpus, list(range(q)), queries, pos
    d = args.data_dir
    if not d or not os.path.isdir(d):
        raise FileNotFoundError("no network: pass --data_dir with corpus.jsonl + questions_<split>.jsonl, or --synthetic N,Q")
    corpus = {}
    with open(join(d, "corpus.jsonl")) as f:
        for line in f:
            r = json.loads(line); corpus[r["id"]] = r["article"]
    if args.dataset == "lleqa":
        split = "train" if args.do_negatives_extraction else ("validation" if args.do_hyperparameter_tuning else "test")   # bm25.py:189
    else:
        split = "train" if args.do_negatives_extraction else "dev_small"                                                  # bm25.py:199
    qids, queries, pos = [], [], []
    with open(join(d, f"questions_{split}.jsonl")) as f:
        for i, line in enumerate(f):
            r = json.loads(line); qids.append(r.get("id", i)); queries.append(r["question"]); pos.append(r["article_ids"])
    return corpus, qids, queries, pos


def main(args):
    import pandas as pd
    from ..utils.metrics import Metrics
    os.makedirs(args.output_dir, exist_ok=True)
    print("Loading documents and queries...")
    corpus, qids, queries, pos_pids = load_data(args)
    documents = list(corpus.values())
    ids = np.array(list(corpus.keys()))                                   # idx2id (bm25.py:214)
    if args.do_preprocessing:
        print("Preprocessing documents and queries (lemmatizing=True)...")
        documents, queries = preprocess(documents), preprocess(queries)

    if args.do_hyperparameter_tuning:
        print("Starting hyperparameter tuning...")
        retriever = BM25(corpus=documents, k1=0., b=0.)
        rows = retriever.tune(queries, pos_pids, ids=ids)                  # bm25.py:221-237, all 187 pairs
        grid_df = pd.DataFrame(rows)
        grid_df.to_csv(join(args.output_dir, "bm25_tuning_results.csv"), sep=",", float_format="%.5f", index=False)
        heat = grid_df.pivot_table(values="recall@100", index="k1", columns="b")[::-1] * 100    # bm25.py:240
        try:   # bm25.py:241-242 draws it with seaborn; matplotlib alone gives the same annotated heat map
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots(figsize=(8, 9))
            ax.imshow(heat.values, cmap="YlOrBr", vmin=40, vmax=60, aspect="auto")
            ax.set_xticks(range(len(heat.columns))); ax.set_xticklabels([f"{c:.1f}" for c in heat.columns]); ax.set_xlabel("b")
            ax.set_yticks(range(len(heat.index))); ax.set_yticklabels([f"{r:.1f}" for r in heat.index]); ax.set_ylabel("k1")
            for (i, j), v in np.ndenumerate(heat.values):
                ax.text(j, i, f"{v:.1f}", ha="center", va="center", fontsize=7)
            fig.savefig(join(args.output_dir, "bm25_tuning_heatmap.pdf"))
            plt.close(fig)
        except Exception as ex:   # noqa: BLE001  (a plot is not worth a failed sweep)
            print(f"(no heat map: {type(ex).__name__}: {ex})")
        print("Done.")
        return rows

    print("Initializing the BM25 retriever model...")
    retriever = BM25(corpus=documents, k1=args.k1, b=args.b)
    print("Running BM25 model on queries...")
    out = {}
    ranked_lists = [ids[row].tolist() for row in retriever.ranked_positions(queries, top_k=1000)]   # search_all(queries, top_k=1000) -> idx2id (bm25.py:248-249)
    if args.do_evaluation:
        print("Computing the retrieval scores...")
        evaluator = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
        out = evaluator.compute_all_metrics(all_ground_truths=pos_pids, all_results=ranked_lists)
        with open(join(args.output_dir, f"performance_bm25_{args.dataset}_dev.json"), "w") as f:   # (the reference names it _dev whatever the split)
            json.dump(out, f, indent=2)
    if args.do_negatives_extraction:
        print(f"Extracting top-{args.num_negatives} negatives for each question...")
        results = dict()
        for q_id, truths_i, preds_i in zip(qids, pos_pids, ranked_lists):
            truths = set(truths_i)
            results[q_id] = [y for y in preds_i if y not in truths][:args.num_negatives]
        results = dict(sorted(results.items()))
        with open(join(args.output_dir, "negatives_bm25.json"), "w") as f:
            json.dump(results, f, indent=2)
    retriever.save_indexes(output_dir=args.output_dir, dataset=args.dataset)
    print("Done.")
    return out


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--dataset", type=str, help="Dataset to use.", choices=["lleqa"] + [f"mmarco-{x}" for x in (
        "ar", "de", "en", "es", "fr", "hi", "id", "it", "ja", "nl", "pt", "ru", "vi", "zh")], default="lleqa")
    parser.add_argument("--do_preprocessing", action="store_true", default=False)
    parser.add_argument("--k1", type=float, default=1.5)
    parser.add_argument("--b", type=float, default=0.75)
    parser.add_argument("--do_evaluation", action="store_true", default=False)
    parser.add_argument("--do_negatives_extraction", action="store_true", default=False)
    parser.add_argument("--num_negatives", type=int, default=10)
    parser.add_argument("--do_hyperparameter_tuning", action="store_true", default=False)
    parser.add_argument("--output_dir", type=str)
    # offline additions (no HF hub / ir_datasets): local data or synthetic LLeQA-shaped data
    parser.add_argument("--data_dir", type=str, default=os.environ.get("LLEQA_DIR"))
    parser.add_argument("--synthetic", type=str, default=None, help="N,Q: synthetic corpus/queries of that size")
    return parser


if __name__ == "__main__":
    a, _ = build_parser().parse_known_args()   # unknown flags ignored, as bm25.py:290
    main(a)
