"""BM25 with the reference's exact arithmetic (src/retrievers/bm25.py:33-161): inverted index built on the
host from whitespace tokens, every (query, document) pair scored on the device in float64 in the
reference's expression and addition order (csrc/bm25.hip), full stable ranking on the device.

Only the retrieval classes are mirrored; the tuning / negatives-extraction CLI (bm25.py:176-291) is out of scope."""
from __future__ import annotations

import math
from statistics import mean

import numpy as np
import torch

from .. import ops
from ..planes import RankedSystem


class BM25:
    def __init__(self, corpus: list[str], k1: float, b: float, device="cuda"):
        self.k1, self.b = k1, b
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
        df = np.bincount(pt, minlength=V).astype(np.int64)
        N = self.corpus_size
        # bm25.py:145-147: log10((N - df + 0.5)/(df + 0.5)) -- can be <= 0
        self.idf_host = np.array([math.log10((N - int(x) + 0.5) / (int(x) + 0.5)) for x in df], dtype=np.float64)
        self.doc_len_host = np.array([len(t) for t in toks], dtype=np.int32)
        self.avgdl = float(mean(self.doc_len_host.tolist())) if N else 0.0   # bm25.py:138
        toff = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(df, out=toff[1:])
        d = self.device
        self.toff = torch.from_numpy(toff).to(d)
        self.pdoc = torch.from_numpy(pd.astype(np.int32)).to(d)
        self.ptf = torch.from_numpy(tf.astype(np.int32)).to(d)
        self.idf = torch.from_numpy(self.idf_host).to(d)
        self.doc_len = torch.from_numpy(self.doc_len_host).to(d)

        self._norm_key, self._norm = None, None
        # where every term's postings cross the document slices one workgroup scores: per index, like the idf table
        self.slice_off = ops.bm25_slice_offsets(self.toff, self.pdoc, self.corpus_size) if self.device.type == "cuda" and V > 0 else None

    def update_params(self, k1: float, b: float) -> None:
        self.k1, self.b = k1, b

    def _doc_norm(self) -> torch.Tensor:
        key = (self.k1, self.b)
        if self._norm_key != key:
            self._norm = ops.bm25_doc_norms(self.doc_len, self.avgdl, self.k1, self.b)
            self._norm_key = key
        return self._norm

    def scores(self, queries: list[str], want_f32: bool = False):
        """[Q, N] float64 plane (want_f32: and its float32 rounding, from the same launch); query terms are NOT de-duplicated (bm25.py:152)."""
        qt = [[self.vocab.get(w, -1) for w in q.split()] for q in queries]
        qoff = np.zeros(len(qt) + 1, dtype=np.int64)
        np.cumsum([len(x) for x in qt], out=qoff[1:])
        flat = np.array([t for x in qt for t in x] or [0], dtype=np.int32)
        return ops.bm25_scores(self.toff, self.pdoc, self.ptf, self.idf, self.doc_len, self.avgdl, self.k1, self.b,
                               torch.from_numpy(qoff).to(self.device), torch.from_numpy(flat).to(self.device), len(queries), self.corpus_size,
                               doc_norm=self._doc_norm(), slice_off=self.slice_off, want_f32=want_f32)

    def search_device(self, queries: list[str], ids: np.ndarray | None = None) -> RankedSystem:
        sc64, sc32 = self.scores(queries, want_f32=True)
        Q, N = sc64.shape
        stats4 = None   # mean | std | min | max of every list's float32 scores: by-products of the ranking sort (rows that fit one workgroup)
        if N <= ops.sort_max_n(torch.float64) and Q > 0:
            stats4 = torch.empty((4, Q), dtype=torch.float32, device=self.device)
        order, _, rank = ops.sort_rows_desc(sc64, want_keys=False, want_rank=True, stats_out=stats4)   # ranks from the float64 scores, ties -> ascending index
        lens = torch.full((Q,), N, dtype=torch.int32, device=self.device)
        # float32 plane for the normalisations (torch.tensor(scores, dtype=float32), hybrid.py:255), written by the scoring kernel from
        # the accumulators it holds (no conversion pass); the float64 scores stay for the 'none' passthrough, which keeps BM25's Python
        # floats (hybrid.py:280).  Rounding is monotone: the float32 scores are in descending order along the float64 ranking too.
        return RankedSystem(scores=sc32, order=order, rank=rank, lens=lens,
                            ids=np.arange(N, dtype=np.int64) if ids is None else ids, full=True,
                            scores64=sc64, score_sorted=True, stats4=stats4)

    def search_all(self, queries: list[str], top_k: int) -> list:
        """bm25.py:90-106: every document scored (zero scores included), stable sort desc, [:top_k]."""
        return self.search_device(queries).to_lists(top_k)

    def search(self, query: str, top_k: int) -> list:
        return self.search_all([query], top_k)[0]
