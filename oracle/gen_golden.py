#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE itself (test infrastructure).

Runs only in the build container, where /root/reference exists; the GPU box never
sees the reference, only the small fixtures this script writes to tests/golden/.

What is imported from the reference (unmodified, no bytecode written):
  * src.retrievers.hybrid.Aggregator   (hybrid.py:166-307)  -- fuse / transform_scores
  * src.retrievers.hybrid.run_evaluation (hybrid.py:24-42)  -- the metric set of the tuning loop
  * src.retrievers.bm25.BM25           (bm25.py:129-156)    -- search_all
  * src.utils.metrics.Metrics          (metrics.py:25-162)  -- compute_all_metrics
  * src.retrievers.splade.base.BaseModel (splade/base.py:186-251) -- compute_batchwise_similarity, search
    (the in-tree mirror of sentence-transformers' util.cos_sim / util.semantic_search used at hybrid.py:103)
  * src.retrievers.splade.splade.SPLADE (splade/splade.py:88-99) -- forward (max / sum pooling)

The splade package does not import as shipped: `transformers.file_utils.default_cache_path` and
`transformers.optimization.AdamW` left transformers 5, and `splade/__init__.py` exports neither `BaseModel` nor
`MmarcoReader` although `splade.py:11` imports them from it.  The harness sets those four NAMES (a path string,
torch's AdamW, the reference's own BaseModel class, a placeholder for the training-only reader); the methods are
then called UNBOUND on a SimpleNamespace carrying the attributes they read (`similarity`, `encode`, `model`,
`pooling`, `relu`, `pruning_topk`), so no checkpoint is needed and no arithmetic goes through a placeholder.

Absent third-party modules that those files import at module top but never touch
on the functions we call (dotenv, ir_datasets, seaborn, wandb, spacy) are replaced
by empty harness-side placeholder modules in sys.modules. No arithmetic goes
through a placeholder. sentence-transformers / colbert-ai are NOT stubbed:
their arithmetic is not reproduced here ("parity unpinned" at those call sites,
see DESIGN.md).

Fixtures are data only: seeded synthetic inputs + the reference's outputs.  A full run takes ~12 minutes, 11 of them in
gen_pr28k: the reference's percentile-rank transform materialises a [27,943, 27,942] float32 distance matrix (3.1 GB, twice)
per list (hybrid.py:273), eight lists in all; it needs ~8 GB of RAM.
Usage: python oracle/gen_golden.py [generator ...]  (rewrites tests/golden/*.npz|json; no argument = all of them)
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("FUSION_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_placeholders():
    for name in ["dotenv", "ir_datasets", "seaborn", "wandb", "spacy"]:
        if name in sys.modules:
            continue
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None
    tok = types.ModuleType("spacy.tokens")
    tok.Doc = object  # only used as a type annotation at preprocessor.py:43
    sys.modules["spacy"].tokens = tok
    sys.modules["spacy.tokens"] = tok


def load_reference():
    _install_placeholders()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from src.retrievers.hybrid import Aggregator
    from src.retrievers.bm25 import BM25
    from src.utils.metrics import Metrics
    return Aggregator, BM25, Metrics


def load_reference_splade():
    """splade/base.py + splade/splade.py, unmodified; see the module docstring for the four harness-side names."""
    import torch
    import transformers.file_utils as fu
    import transformers.optimization as opt
    if not hasattr(fu, "default_cache_path"):
        fu.default_cache_path = os.path.join(os.path.expanduser("~"), ".cache", "huggingface")   # base.py:12, a path only
    if not hasattr(opt, "AdamW"):
        opt.AdamW = torch.optim.AdamW                                                              # splade.py:9, training only
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import src.retrievers.splade as pkg
    from src.retrievers.splade.base import BaseModel
    if not hasattr(pkg, "BaseModel"):
        pkg.BaseModel = BaseModel               # splade.py:11 `from . import BaseModel, MmarcoReader`
    if not hasattr(pkg, "MmarcoReader"):
        pkg.MmarcoReader = object               # training-only data reader
    from src.retrievers.splade.splade import SPLADE
    return BaseModel, SPLADE


# --------------------------------------------------------------------------------------
# synthetic ranked lists (LLeQA-shaped score distributions)
# --------------------------------------------------------------------------------------
SYSTEMS = ["bm25", "dpr", "splade", "colbert"]


def synth_system_scores(rng, system, n, variant):
    """fp32-representable python floats, one per corpus position."""
    if variant == "const":
        s = np.full(n, 3.25, dtype=np.float32)
    elif system == "bm25":
        s = np.maximum(0.0, rng.gamma(0.5, 4.0, n) - 2.0).astype(np.float32)  # ~40% exact zeros
    elif system in ("dpr", "splade"):
        s = rng.uniform(-0.2, 0.9, n).astype(np.float32)
    else:
        s = rng.normal(20.0, 4.0, n).astype(np.float32)
    if variant == "ties" and n >= 8:
        # plant exact ties (and a negative-zero / positive-zero pair) at random places
        idx = rng.choice(n, size=min(n // 2, 64), replace=False)
        s[idx] = s[idx[0]]
        if system != "bm25":
            s[idx[1]] = 0.0
            s[idx[2]] = -0.0
    return s


def ranked_list_from_scores(scores, ids, keep):
    """Sort desc, ties -> ascending corpus position (the build's documented tie rule);
    keep only the first `keep` entries (ColBERT/PLAID lists are shorter than N)."""
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    order = order[:keep]
    return [{"corpus_id": int(ids[i]), "score": float(scores[i])} for i in order]


def make_case(rng, S, Q, N, variant):
    systems = SYSTEMS[:S] if variant != "colbert_first" else ["colbert", "bm25", "dpr", "splade"][:S]
    ids = rng.permutation(np.arange(1, 4 * N + 1))[:N]  # article ids are not positions
    lists, lens = {}, {}
    for s in systems:
        per_q = []
        for _q in range(Q):
            sc = synth_system_scores(rng, s, N, variant)
            keep = N
            if s == "colbert" and N >= 4:
                keep = max(1, int(0.6 * N))
            per_q.append(ranked_list_from_scores(sc, ids, keep))
        lists[s] = per_q
    return systems, ids, lists


def pack_lists(systems, lists, Q):
    """-> ids[S,Q,L] int64 (-1 pad), scores[S,Q,L] float64, lens[S,Q]"""
    L = max(len(lists[s][q]) for s in systems for q in range(Q))
    S = len(systems)
    ids = -np.ones((S, Q, L), dtype=np.int64)
    sc = np.zeros((S, Q, L), dtype=np.float64)
    ln = np.zeros((S, Q), dtype=np.int32)
    for si, s in enumerate(systems):
        for q in range(Q):
            l = lists[s][q]
            ln[si, q] = len(l)
            ids[si, q, : len(l)] = [x["corpus_id"] for x in l]
            sc[si, q, : len(l)] = [x["score"] for x in l]
    return ids, sc, ln


def pack_out(fused, Q):
    U = max(len(fused[q]) for q in range(Q))
    ids = -np.ones((Q, U), dtype=np.int64)
    sc = np.zeros((Q, U), dtype=np.float64)
    ln = np.zeros((Q,), dtype=np.int32)
    for q in range(Q):
        ln[q] = len(fused[q])
        ids[q, : ln[q]] = [x["corpus_id"] for x in fused[q]]
        sc[q, : ln[q]] = [float(x["score"]) for x in fused[q]]
    return ids, sc, ln


METHODS = [
    ("rrf", "none"), ("bcf", "none"),
    ("nsf", "none"), ("nsf", "min-max"), ("nsf", "z-score"), ("nsf", "arctan"),
    ("nsf", "percentile-rank"), ("nsf", "normal-curve-equivalent"),
]


def gen_fuse(Aggregator):
    import copy
    cases = []
    cfgs = [
        # seed, S, Q, N, variant
        (0, 2, 4, 257, "plain"), (1, 3, 4, 257, "plain"), (2, 4, 4, 257, "plain"),
        (3, 4, 3, 1000, "plain"), (4, 2, 3, 1000, "ties"), (5, 4, 3, 257, "ties"),
        (6, 3, 2, 2, "plain"), (7, 2, 2, 1, "plain"), (8, 4, 2, 64, "const"),
        (9, 4, 3, 257, "colbert_first"), (10, 3, 2, 5, "ties"),
    ]
    for seed, S, Q, N, variant in cfgs:
        rng = np.random.default_rng(seed)
        systems, ids, lists = make_case(rng, S, Q, N, variant)
        w = rng.dirichlet(np.ones(S))
        w = np.round(w / 0.05) * 0.05
        w[-1] = max(0.0, 1.0 - w[:-1].sum())
        weights = {s: float(x) for s, x in zip(systems, w)}
        # synthetic percentile table per system: quantiles of that system's pooled scores
        distr = {}
        for s in systems:
            pool = np.array([x["score"] for q in range(Q) for x in lists[s][q]], dtype=np.float64)
            P = min(101, max(3, len(pool)))
            distr[s] = np.quantile(pool, np.linspace(0, 1, P))
        in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
        blob = {
            "systems": np.array(systems), "in_ids": in_ids, "in_scores": in_sc, "in_len": in_len,
            "weights": np.array([weights[s] for s in systems], dtype=np.float64),
        }
        for s in systems:
            blob[f"distr_{s}"] = distr[s]
        for method, norm in METHODS:
            fused = Aggregator.fuse(copy.deepcopy(lists), method=method, normalization=norm,
                                    linear_weights=weights, percentile_distributions=distr)
            o_ids, o_sc, o_len = pack_out(fused, Q)
            key = f"{method}__{norm}"
            blob[f"out_ids__{key}"] = o_ids
            blob[f"out_scores__{key}"] = o_sc
            blob[f"out_len__{key}"] = o_len
        name = f"fuse_seed{seed}_S{S}_Q{Q}_N{N}_{variant}.npz"
        np.savez_compressed(os.path.join(OUT, name), **blob)
        cases.append(name)
    return cases


def gen_kat(Aggregator):
    """Known-answer tests of SURVEY.md §8c, regenerated from the reference."""
    L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]
    kat = {}
    # KAT-1 tie-break = first-insertion order
    a = {"s1": [L([(100, 2.0), (200, 1.0)])], "s2": [L([(200, 2.0), (100, 1.0)])]}
    b = {"s2": a["s2"], "s1": a["s1"]}
    kat["kat1_s1s2"] = Aggregator.fuse(a, method="rrf")
    kat["kat1_s2s1"] = Aggregator.fuse(b, method="rrf")
    # KAT-2 mixed coverage
    lists = {"bm25": [L([(10, 7.5), (11, 3.0), (12, 0.0)])], "dpr": [L([(12, .9), (10, .5), (13, .1)])]}
    w = {"bm25": .5, "dpr": .5}
    for m, n in [("rrf", "none"), ("bcf", "none"), ("nsf", "min-max"), ("nsf", "z-score"), ("nsf", "arctan"), ("nsf", "none")]:
        import copy
        kat[f"kat2_{m}_{n}"] = Aggregator.fuse(copy.deepcopy(lists), method=m, normalization=n, linear_weights=w, percentile_distributions={})
    # KAT-3 uneven lists
    lists = {"s1": [L([(1, 5.0), (2, 1.0)])], "s2": [L([(3, 9.0), (1, 8.0), (2, 7.0)])]}
    kat["kat3_zscore"] = Aggregator.fuse(lists, method="nsf", normalization="z-score", linear_weights={"s1": .5, "s2": .5}, percentile_distributions={})
    # KAT-4 constant rows / single element
    kat["kat4_minmax_const"] = [{"corpus_id": k, "score": float(v)} for k, v in Aggregator.transform_scores({1: 2.0, 2: 2.0, 3: 2.0}, "min-max").items()]
    kat["kat4_zscore_const"] = [{"corpus_id": k, "score": float(v)} for k, v in Aggregator.transform_scores({1: 2.0, 2: 2.0, 3: 2.0}, "z-score").items()]
    kat["kat4_zscore_single"] = [{"corpus_id": k, "score": float(v)} for k, v in Aggregator.transform_scores({1: 2.0}, "z-score").items()]
    # KAT-5 percentile rank
    kat["kat5_percentile"] = [{"corpus_id": k, "score": float(v)} for k, v in Aggregator.transform_scores({1: 0.2, 2: 4.6, 3: 99.0}, "percentile-rank", percentile_distr=np.linspace(0, 10, 11)).items()]
    # KAT-6 return_topk slices QUERIES (hybrid.py:220)
    three = {"s1": [L([(1, 1.0), (2, .5)])] * 3, "s2": [L([(2, 1.0), (1, .5)])] * 3}
    kat["kat6_topk2"] = Aggregator.fuse(three, method="rrf", return_topk=2)
    # duplicate ids inside one list (convert2dict collapse, hybrid.py:231)
    dup = {"s1": [L([(1, 3.0), (2, 2.0), (1, 1.0), (3, 0.5)])], "s2": [L([(3, 1.0), (2, .5)])]}
    kat["kat_dup_rrf"] = Aggregator.fuse(dup, method="rrf")
    kat["kat_dup_bcf"] = Aggregator.fuse(dup, method="bcf")

    def clean(v):
        return [[{"corpus_id": int(x["corpus_id"]), "score": float(x["score"])} for x in q] for q in v] if v and isinstance(v[0], list) else \
               [{"corpus_id": int(x["corpus_id"]), "score": float(x["score"])} for x in v]
    out = {}
    for k, v in kat.items():
        c = clean(v)
        # json has no NaN literal in strict mode; encode as string
        def enc(x):
            if isinstance(x, list):
                return [enc(y) for y in x]
            if isinstance(x, dict):
                return {kk: ("nan" if isinstance(vv, float) and vv != vv else vv) for kk, vv in x.items()}
            return x
        out[k] = enc(c)
    with open(os.path.join(OUT, "kat_fuse.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_bm25(BM25):
    rng = np.random.default_rng(123)
    vocab = [f"w{i}" for i in range(30)]
    p = 1.0 / np.arange(1, 31)
    p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 25)), p=p)) for _ in range(50)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 8)), p=p)) for _ in range(8)]
    queries += ["w0 w0 w1", "oov w3 oov2", "", "zzz"]  # repeated term, OOV terms, empty, all-OOV
    out = {"docs": docs, "queries": queries, "params": [], "results": []}
    for k1, b in [(2.5, 0.2), (1.2, 0.75), (0.9, 0.0)]:
        m = BM25(corpus=docs, k1=k1, b=b)
        res = m.search_all(queries, top_k=len(docs))
        out["params"].append([k1, b])
        out["results"].append([[[int(x["corpus_id"]), float(x["score"])] for x in r] for r in res])
        if k1 == 2.5:
            out["idf"] = {w: float(m.idf[w]) for w in sorted(m.idf)}   # sorted: the vocabulary is a set (hash-seed order)
            out["avgdl"] = float(m.avgdl)
    # KAT-8 of SURVEY.md
    kdocs = ["chat noir dormir", "chien noir courir courir", "loi article code civil", "chat chien"]
    m = BM25(corpus=kdocs, k1=2.5, b=0.2)
    out["kat8"] = {"docs": kdocs, "queries": ["chat noir", "courir inconnu"],
                   "results": [[[int(x["corpus_id"]), float(x["score"])] for x in r] for r in m.search_all(["chat noir", "courir inconnu"], top_k=4)],
                   "avgdl": float(m.avgdl), "idf_chat": float(m.idf["chat"])}
    with open(os.path.join(OUT, "bm25.json"), "w") as f:
        json.dump(out, f)


def gen_bm25_family(_unused=None):
    """The rest of the reference's lexical module (bm25.py): TFIDF (:33-127) and AtireBM25 (:164-173) through search_all, and the three
    things bm25.py's main() does with BM25 -- driven piece by piece, because main() itself cannot run (it downloads its data, :185-212,
    and its grid loop dies on its first row: `scores.pop('recall')` raises KeyError with today's Metrics, :235, and DataFrame.append is
    gone from pandas 2, :236):
      * the k1 x b grid of :221-233: update_params -> search_all(top_k=1000) -> idx2id -> Metrics(recall_at_k=[10,100,200,500,1000]);
        the k1 = 0 column is left out: with np.float64 parameters a document lacking a query term scores 0/0 = NaN there (:154) and the
        reference then sorts NaN keys -- an artefact of timsort's comparison sequence, not a ranking (cf. D16);
      * the evaluation of :246-252 (k1 = 2.5, b = 0.2, the LLeQA preset of run_bm25.sh:24-25);
      * the negatives of :254-261."""
    import itertools
    import src.retrievers.bm25 as ref
    from src.utils.metrics import Metrics
    rng = np.random.default_rng(2024)
    V, N, Q = 400, 1100, 10
    vocab = np.array([f"m{i}" for i in range(V)])
    p = 1.0 / np.arange(1, V + 1) ** 1.1
    p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 60)), p=p)) for _ in range(N)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(2, 9)), p=p)) for _ in range(Q - 3)]
    queries += ["m0 m0 m5 m5 m5", "horsvocab m7 inconnu", "m399 m398 m1"]
    ids = [1000 + 3 * i for i in range(N)]                                   # idx2id (bm25.py:214): dataset ids are not positions
    qids = [70 + q for q in range(Q)]
    gold = [sorted(int(x) for x in rng.choice(ids, size=int(rng.integers(1, 4)), replace=False)) for _ in range(Q)]
    # make the gold findable: every gold document gets its query's terms appended
    for q, gl in enumerate(gold):
        for g in gl:
            docs[ids.index(g)] += " " + queries[q]
    idx2id = dict(enumerate(ids))
    out = {"docs": docs, "queries": queries, "ids": ids, "qids": qids, "gold": gold}

    def lists(m, top_k):
        return [[[int(x["corpus_id"]), float(x["score"])] for x in r] for r in m.search_all(queries, top_k=top_k)]
    t = ref.TFIDF(corpus=docs)
    out["tfidf"] = {"repr": repr(t), "idf": {w: float(t.idf[w]) for w in sorted(t.idf)}, "results": lists(t, 50), "vocab_sorted_head": t.get_vocab()[:5]}
    a = ref.AtireBM25(corpus=docs, k1=1.2, b=0.75)
    out["atire"] = {"repr": repr(a), "k1": 1.2, "b": 0.75, "results": lists(a, 50)}
    # the grid (bm25.py:221-233)
    evaluator = Metrics(recall_at_k=[10, 100, 200, 500, 1000])
    retriever = ref.BM25(corpus=docs, k1=0., b=0.)
    k1_range = np.arange(0., 8.5, 0.5)
    b_range = np.arange(0., 1.1, 0.1)
    rows = []
    for k1, b in itertools.product(*[k1_range, b_range]):
        if k1 == 0.0:
            continue
        retriever.update_params(k1, b)
        ranked = retriever.search_all(queries, top_k=1000)
        ranked = [[idx2id.get(x["corpus_id"]) for x in results] for results in ranked]
        sc = evaluator.compute_all_metrics(all_ground_truths=gold, all_results=ranked)
        rows.append({"k1": float(k1), "b": float(b), **{k: float(v) for k, v in sc.items()}})
    out["grid"] = {"k1_range": [float(x) for x in k1_range], "b_range": [float(x) for x in b_range], "rows": rows}
    # evaluation + negatives at the LLeQA preset
    m = ref.BM25(corpus=docs, k1=2.5, b=0.2)
    ranked = [[idx2id.get(x["corpus_id"]) for x in r] for r in m.search_all(queries, top_k=1000)]
    ev = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    out["evaluation"] = {k: float(v) for k, v in ev.compute_all_metrics(all_ground_truths=gold, all_results=ranked).items()}
    neg = {}
    for q_id, truths_i, preds_i in zip(qids, gold, ranked):
        neg[q_id] = [y for y in preds_i if y not in truths_i][:10]
    out["negatives"] = {str(k): v for k, v in sorted(neg.items())}
    out["top1000_head"] = [r[:20] for r in ranked]
    with open(os.path.join(OUT, "bm25_family.json"), "w") as f:
        json.dump(out, f)


def gen_metrics(Metrics):
    rng = np.random.default_rng(7)
    cases = []
    ev = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    for Q, N in [(5, 40), (8, 1500), (3, 7)]:
        gold = [sorted(rng.choice(np.arange(1, N + 1), size=int(rng.integers(1, 6)), replace=False).tolist()) for _ in range(Q)]
        pred = [rng.permutation(np.arange(1, N + 1)).tolist() for _ in range(Q)]
        sc = ev.compute_all_metrics(all_ground_truths=gold, all_results=pred)
        cases.append({"gold": gold, "pred": pred, "scores": {k: float(v) for k, v in sc.items()}})
    kat9 = Metrics(recall_at_k=[1, 2, 500], map_at_k=[2], mrr_at_k=[2], ndcg_at_k=[2]).compute_all_metrics([[1, 2], [9]], [[1, 3, 2], [4, 9, 5]])
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump({"cases": cases, "kat9": {k: float(v) for k, v in kat9.items()}}, f)


# --------------------------------------------------------------------------------------
# round 2: scoring / search / SPLADE pooling pinned on the reference's in-tree splade package,
# the weight-grid loop, the score-distribution analysis, a full LLeQA row, unsorted lists
# --------------------------------------------------------------------------------------
def _fake_model(BaseModel, similarity, Qe=None, De=None):
    """The attributes BaseModel.compute_batchwise_similarity / search read from `self`."""
    from types import SimpleNamespace
    m = SimpleNamespace(similarity=similarity)
    m.compute_batchwise_similarity = lambda q_embs, d_embs: BaseModel.compute_batchwise_similarity(m, q_embs, d_embs)
    m.encode = lambda texts, query_mode, batch_size: (Qe if query_mode else De)
    return m


def gen_similarity(BaseModel):
    """compute_batchwise_similarity (splade/base.py:186-197), cos_sim and dot_score."""
    import torch
    torch.set_num_threads(1)   # one summation order, whatever box regenerates the fixtures
    rng = np.random.default_rng(2024)
    # DPR-shaped: un-normalised Gaussians, d = 768
    Qe = rng.normal(0, 1, (8, 768)).astype(np.float32)
    De = rng.normal(0, 1, (300, 768)).astype(np.float32)
    De[17] = De[3]                      # a duplicated document: exactly equal scores
    De[40] = 2.5 * De[41]               # cosine is scale invariant, the dot product is not
    out = {"Qe": Qe, "De": De}
    for sim in ("cos_sim", "dot_score"):
        m = _fake_model(BaseModel, sim)
        out[sim] = BaseModel.compute_batchwise_similarity(m, torch.from_numpy(Qe), torch.from_numpy(De)).numpy()
    np.savez_compressed(os.path.join(OUT, "sim_dpr_Q8_N300_d768.npz"), **out)
    # SPLADE-shaped: V = 32,005, sparse non-negative activations (log1p(relu) outputs), stored as COO triplets
    Q, N, V = 4, 257, 32005

    def sparse_rows(rows, nnz_lo, nnz_hi):
        r, c, v = [], [], []
        for i in range(rows):
            k = int(rng.integers(nnz_lo, nnz_hi))
            cols = rng.choice(V, size=k, replace=False)
            r += [i] * k; c += cols.tolist(); v += rng.gamma(2.0, 0.4, k).astype(np.float32).tolist()
        return np.array(r, np.int32), np.array(c, np.int32), np.array(v, np.float32)
    qr, qc, qv = sparse_rows(Q, 20, 60)
    dr, dc, dv = sparse_rows(N, 60, 240)
    Qs = np.zeros((Q, V), np.float32); Qs[qr, qc] = qv
    Ds = np.zeros((N, V), np.float32); Ds[dr, dc] = dv
    out = {"shape": np.array([Q, N, V]), "q_row": qr, "q_col": qc, "q_val": qv, "d_row": dr, "d_col": dc, "d_val": dv}
    for sim in ("cos_sim", "dot_score"):
        m = _fake_model(BaseModel, sim)
        out[sim] = BaseModel.compute_batchwise_similarity(m, torch.from_numpy(Qs), torch.from_numpy(Ds)).numpy()
    np.savez_compressed(os.path.join(OUT, "sim_splade_Q4_N257_V32005.npz"), **out)


def gen_search(BaseModel):
    """BaseModel.search (splade/base.py:199-251): chunked mm -> topk(sorted=False) -> heap -> sorted, with a stand-in
    `encode` that returns seeded embeddings.  Same algorithm as util.semantic_search (hybrid.py:103) and the chunked
    evaluator (src/utils/sentence_transformers.py:346-364)."""
    import torch
    torch.set_num_threads(1)
    rng = np.random.default_rng(77)
    Q, N, d = 6, 1000, 64
    Qe = rng.normal(0, 1, (Q, d)).astype(np.float32)
    De = rng.normal(0, 1, (N, d)).astype(np.float32)
    for a, b in [(20, 500), (21, 501), (22, 999), (700, 3)]:
        De[b] = De[a]                   # exact duplicates: tied scores inside a list
    De[123] = 0.0                       # a zero vector: F.normalize's eps clamp -> score exactly 0
    out = {"Qe": Qe, "De": De}
    cfgs = [("k10_qc4_dc300", 10, 4, 300), ("kN_qc100_dc500000", N, 100, 500000), ("k37_qc2_dc128", 37, 2, 128),
            ("k1000_qc3_dc333", 1000, 3, 333)]
    for sim in ("cos_sim", "dot_score"):
        m = _fake_model(BaseModel, sim, torch.from_numpy(Qe), torch.from_numpy(De))
        for name, k, qc, dc in cfgs:
            res = BaseModel.search(m, ["q"] * Q, ["d"] * N, batch_size=32, query_chunk_size=qc, doc_chunk_size=dc, topk=k)
            out[f"ids__{sim}__{name}"] = np.array([[x["doc_id"] for x in r] for r in res], dtype=np.int64)
            out[f"scores__{sim}__{name}"] = np.array([[x["score"] for x in r] for r in res], dtype=np.float32)
    out["configs"] = np.array([f"{n}:{k}:{qc}:{dc}" for n, k, qc, dc in cfgs])
    np.savez_compressed(os.path.join(OUT, "search_Q6_N1000_d64.npz"), **out)


def gen_splade_pool(SPLADE):
    """SPLADE.forward (splade/splade.py:88-99) on seeded MLM logits, 'max' (the hybrid path's default) and 'sum'."""
    import torch
    from types import SimpleNamespace
    torch.set_num_threads(1)
    rng = np.random.default_rng(5)
    B, L, V = 5, 24, 509
    logits = rng.normal(0, 2.0, (B, L, V)).astype(np.float32)
    logits[0, 3, :40] = 0.0
    logits[1, :, 7] = -1.0              # a vocabulary entry that never activates
    lens = np.array([24, 9, 1, 17, 2], dtype=np.int32)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    out = {"logits": logits, "lens": lens}
    for pooling in ("max", "sum"):
        fake = SimpleNamespace(model=lambda input_ids, attention_mask: SimpleNamespace(logits=torch.from_numpy(logits)),
                               pooling=pooling, relu=torch.nn.ReLU(), pruning_topk=None)
        out[pooling] = SPLADE.forward(fake, torch.zeros((B, L), dtype=torch.long), torch.from_numpy(mask)).numpy()
    np.savez_compressed(os.path.join(OUT, "splade_pool_B5_L24_V509.npz"), **out)


def _lattice(names, step=0.05):
    """hybrid.py:405-409, evaluated with the same numpy calls."""
    import itertools
    return [{n: w for n, w in zip(names, comb)} for comb in itertools.product(np.arange(0, 1 + step, step), repeat=len(names))
            if np.isclose(sum(comb), 1.0)]


def gen_tune(Aggregator):
    """The weight-grid loop of hybrid.py:404-426: per weight vector Aggregator.fuse(deepcopy(results)) -> run_evaluation."""
    import copy
    from src.retrievers.hybrid import run_evaluation
    for seed, S, Q, N, variant in [(20, 2, 4, 257, "ties"), (21, 3, 4, 257, "colbert_first")]:
        rng = np.random.default_rng(seed)
        systems, ids, lists = make_case(rng, S, Q, N, variant)
        # gold labels: 1-4 ids per query, mostly taken from the head of a mixture ranking so that the metrics move with
        # the weights; one label is never retrieved by any system (absent id), one query has a duplicated label
        labels = []
        for q in range(Q):
            pool = [x["corpus_id"] for s in systems for x in lists[s][q][:40]]
            g = rng.choice(pool, size=int(rng.integers(1, 5)), replace=False).tolist()
            labels.append([int(x) for x in g])
        labels[1].append(10 ** 7)
        labels[2].append(labels[2][0])
        distr = {}
        for s in systems:
            pool = np.array([x["score"] for q in range(Q) for x in lists[s][q]], dtype=np.float64)
            distr[s] = np.quantile(pool, np.linspace(0, 1, 101))
        combos = _lattice(systems)
        in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
        blob = {"systems": np.array(systems), "in_ids": in_ids, "in_scores": in_sc, "in_len": in_len,
                "labels": np.array([",".join(str(x) for x in g) for g in labels]),
                "weights": np.array([[w[s] for s in systems] for w in combos], dtype=np.float64)}
        for s in systems:
            blob[f"distr_{s}"] = distr[s]
        names = None
        for norm in ["min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent", "none"]:
            rows = []
            for w in combos:
                fused = Aggregator.fuse(copy.deepcopy(lists), method="nsf", normalization=norm, percentile_distributions=distr, linear_weights=w)
                perf = run_evaluation(predictions=[[x["corpus_id"] for x in r] for r in fused], labels=labels, print2console=False)
                names = names or list(perf.keys())
                assert list(perf.keys()) == names
                rows.append([float(perf[k]) for k in names])
            blob[f"metrics__{norm}"] = np.array(rows, dtype=np.float64)
        blob["metric_names"] = np.array(names)
        np.savez_compressed(os.path.join(OUT, f"tune_seed{seed}_S{S}_Q{Q}_N{N}_{variant}.npz"), **blob)


def gen_analysis(Aggregator):
    """The score-distribution analysis of hybrid.py:363-402, which lives inside main() and cannot be called: the
    per-(query, system) transform IS the reference's Aggregator.transform_scores(convert2dict(...)); the three table
    recipes around it are re-stated here with the same pandas operations (rows of {'system','score'} dicts ->
    DataFrame; per system drop zeros and the two smallest distinct scores; quantiles at linspace(0,1,n+1); labelled
    scores of the positives and of random.seed(42)-sampled negatives, 0 where a system does not list the document)."""
    import random
    import pandas as pd
    rng = np.random.default_rng(31)
    S, Q, N = 3, 3, 120
    systems, ids, lists = make_case(rng, S, Q, N, "colbert_first")   # colbert lists are 0.6 N long
    corpus_ids = sorted(int(i) for i in ids)
    max_pid = max(corpus_ids)
    pos_pids = [[int(x) for x in rng.choice(corpus_ids, size=int(rng.integers(1, 4)), replace=False)] for _ in range(Q)]
    random.seed(42)
    neg_pids = [random.sample(list(set(range(1, max_pid + 1)) - set(x)), k=len(x)) for x in pos_pids]
    in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
    blob = {"systems": np.array(systems), "in_ids": in_ids, "in_scores": in_sc, "in_len": in_len,
            "corpus_ids": np.array(corpus_ids, dtype=np.int64),
            "pos_pids": np.array([",".join(map(str, p)) for p in pos_pids]),
            "neg_pids": np.array([",".join(map(str, p)) for p in neg_pids])}
    raw_tables = None
    for norm in ["none", "min-max", "z-score", "arctan", "percentile-rank"]:
        distributions = {}
        if norm == "percentile-rank":
            distributions = {s: raw_tables[s] for s in systems}    # the reference reads the 'raw' tables back from CSV
        all_scores, labeled = [], []
        for i in range(Q):
            transformed = {}
            for s in systems:
                transformed[s] = Aggregator.transform_scores(results=Aggregator.convert2dict(lists[s][i]), transformation=norm,
                                                             percentile_distr=distributions.get(s))
                all_scores.extend({"system": s, "score": v} for v in transformed[s].values())
            for label, pids in (("positive", pos_pids[i]), ("negative", neg_pids[i])):
                for pid in pids:
                    labeled.append({"label": label, **{s: t.get(pid, 0) for s, t in transformed.items()}})
        df = pd.DataFrame(all_scores, columns=["system", "score"])
        for s in systems:
            blob[f"scores__{norm}__{s}"] = df.loc[df["system"] == s, "score"].to_numpy(dtype=np.float64)
        for n_pts in (10, 1000):
            for s in systems:
                g = df.loc[df["system"] == s, "score"]
                two_smallest = g.drop_duplicates().nsmallest(2)
                kept = g[(g != 0.0) & (~g.isin(two_smallest))]
                blob[f"table__{norm}__{n_pts}__{s}"] = kept.quantile(np.linspace(0, 1, n_pts + 1)).to_numpy(dtype=np.float64)
        if norm == "none":
            raw_tables = {s: blob[f"table__none__1000__{s}"] for s in systems}
        ldf = pd.DataFrame(labeled, columns=["label"] + systems)
        blob[f"labeled_label__{norm}"] = ldf["label"].to_numpy().astype(str)
        for s in systems:
            blob[f"labeled__{norm}__{s}"] = ldf[s].to_numpy(dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "analysis_seed31_S3_Q3_N120.npz"), **blob)


def gen_fullrow(Aggregator):
    """One full LLeQA row (N = 27,942), S = 4, Q = 2: the z-score / arctan tolerances at the real list length."""
    import copy
    rng = np.random.default_rng(40)
    S, Q, N = 4, 2, 27942
    systems, ids, lists = make_case(rng, S, Q, N, "plain")
    weights = {"bm25": 0.15, "dpr": 0.35, "splade": 0.3, "colbert": 0.2}
    distr = {}
    for s in systems:
        pool = np.array([x["score"] for q in range(Q) for x in lists[s][q]], dtype=np.float64)
        distr[s] = np.quantile(pool, np.linspace(0, 1, 1001))
    in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
    blob = {"systems": np.array(systems), "in_ids": in_ids.astype(np.int32), "in_scores": in_sc.astype(np.float32), "in_len": in_len,
            "weights": np.array([weights[s] for s in systems], dtype=np.float64)}
    assert np.array_equal(blob["in_scores"].astype(np.float64), in_sc)     # the inputs are fp32 values: nothing is lost
    for s in systems:
        blob[f"distr_{s}"] = distr[s]
    for method, norm in METHODS:
        fused = Aggregator.fuse(copy.deepcopy(lists), method=method, normalization=norm, linear_weights=weights, percentile_distributions=distr)
        o_ids, o_sc, o_len = pack_out(fused, Q)
        key = f"{method}__{norm}"
        blob[f"out_ids__{key}"] = o_ids.astype(np.int32)
        f64 = method in ("rrf", "bcf") or norm == "none"
        blob[f"out_scores__{key}"] = o_sc if f64 else o_sc.astype(np.float32)
        if not f64:
            assert np.array_equal(blob[f"out_scores__{key}"].astype(np.float64), o_sc, equal_nan=True)   # fp32 values (NumPy 2 promotion)
        blob[f"out_len__{key}"] = o_len
    np.savez_compressed(os.path.join(OUT, f"fuse_fullrow_seed40_S{S}_Q{Q}_N{N}.npz"), **blob)


def gen_unsorted(Aggregator):
    """Lists that are NOT sorted by score, and lists with duplicate ids, through the nsf normalisations: the reference
    takes min / max / mean / std over the VALUES (hybrid.py:255-262), whatever the list order."""
    import copy
    L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]
    rng = np.random.default_rng(50)
    out = {}
    cases = {
        "dup": {"s1": [L([(1, 5.0), (2, 4.0), (1, 1.0), (3, 4.5)])], "s2": [L([(3, 1.0), (2, .5), (4, .25)])]},
        "unsorted": {"s1": [L([(1, 0.5), (2, 7.0), (3, -2.0), (4, 3.0)])], "s2": [L([(4, 1.0), (3, 9.0), (2, 2.0), (5, 0.0)])]},
    }
    n = 64
    idp = rng.permutation(np.arange(1, n + 1))
    cases["unsorted_rand"] = {
        "a": [L([(int(i), float(np.float32(v))) for i, v in zip(idp, rng.normal(0, 3, n))]) for _ in range(2)],
        "b": [L([(int(i), float(np.float32(v))) for i, v in zip(idp[::-1][: n // 2], rng.uniform(-1, 1, n // 2))]) for _ in range(2)],
    }
    for cname, lists in cases.items():
        w = {s: x for s, x in zip(lists.keys(), (0.3, 0.7))}
        out[cname] = {"lists": lists, "weights": w, "out": {}}
        for norm in ["min-max", "z-score", "arctan", "none"]:
            fused = Aggregator.fuse(copy.deepcopy(lists), method="nsf", normalization=norm, linear_weights=w, percentile_distributions={})
            out[cname]["out"][norm] = [[{"corpus_id": int(x["corpus_id"]), "score": float(x["score"])} for x in r] for r in fused]
        for m in ["rrf", "bcf"]:
            fused = Aggregator.fuse(copy.deepcopy(lists), method=m)
            out[cname]["out"][m] = [[{"corpus_id": int(x["corpus_id"]), "score": float(x["score"])} for x in r] for r in fused]
    with open(os.path.join(OUT, "unsorted_fuse.json"), "w") as f:
        json.dump(out, f)


def _reference_tables(rng, systems, n_draws, n, n_points):
    """Quantile tables the way hybrid.py:389-397 makes them from the pooled scores of a run: zeros and each system's two smallest
    distinct scores dropped, then `quantile(np.linspace(0, 1, n_points + 1))` -- n_points = len(corpus) gives the `_28k` table
    hybrid.py:412,451 read (27,943 rows), 10,000 the `_10k` table of :374."""
    import pandas as pd
    out = {}
    for s in systems:
        pool = pd.Series(np.concatenate([synth_system_scores(rng, s, n, "plain") for _ in range(n_draws)]).astype(np.float64))
        kept = pool[(pool != 0.0) & (~pool.isin(pool.drop_duplicates().nsmallest(2)))]
        out[s] = kept.quantile(np.linspace(0, 1, n_points + 1)).to_numpy(dtype=np.float64)
    return out


def gen_pr28k(Aggregator):
    """percentile-rank / normal-curve-equivalent at the table size the reference READS (hybrid.py:412,451: the `_28k` table has
    len(corpus) + 1 = 27,943 rows per system): one full LLeQA row, S = 4, the ColBERT list cut to 60 %.  The reference's
    [P, N] distance matrix is 3.1 GB per list here.  Planted: a run of duplicated quantiles, scores that ARE table entries,
    scores exactly half-way between two entries, scores below and above the table."""
    import copy
    rng = np.random.default_rng(60)
    S, Q, N = 4, 1, 27942
    systems, ids, lists = make_case(rng, S, Q, N, "plain")
    distr = _reference_tables(rng, systems, 3, N, N)
    assert all(len(t) == N + 1 for t in distr.values())
    distr["dpr"][5000:5032] = distr["dpr"][5000]               # duplicated quantiles: the FIRST of them is the argmin (hybrid.py:274)
    distr["colbert"][20000:20003] = distr["colbert"][20000]
    for s in systems:                                             # planted scores, as float32 values like every other score
        t32 = distr[s].astype(np.float32)
        l = lists[s][0]
        for j, k in enumerate(range(100, 27000, 1500)):
            l[40 + 3 * j]["score"] = float(t32[k])                                        # a table entry itself
            l[41 + 3 * j]["score"] = float(np.float32((np.float64(t32[k]) + np.float64(t32[k + 1])) / 2))   # (rounded) midpoint
        l[7]["score"] = float(np.float32(t32[-1] + np.float32(3.5)))                      # above the table
        l[9]["score"] = float(np.float32(t32[0] - np.float32(1.25)))                      # below the table
        if s == "dpr":
            l[11]["score"] = float(t32[5010])                                             # inside the duplicated run
    weights = {"bm25": 0.15, "dpr": 0.35, "splade": 0.3, "colbert": 0.2}
    in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
    blob = {"systems": np.array(systems), "in_ids": in_ids.astype(np.int32), "in_scores": in_sc.astype(np.float32), "in_len": in_len,
            "weights": np.array([weights[s] for s in systems], dtype=np.float64)}
    assert np.array_equal(blob["in_scores"].astype(np.float64), in_sc)
    for s in systems:
        blob[f"distr_{s}"] = distr[s]
    for norm in ["percentile-rank", "normal-curve-equivalent"]:
        fused = Aggregator.fuse(copy.deepcopy(lists), method="nsf", normalization=norm, linear_weights=weights, percentile_distributions=distr)
        o_ids, o_sc, o_len = pack_out(fused, Q)
        blob[f"out_ids__nsf__{norm}"] = o_ids.astype(np.int32)
        blob[f"out_scores__nsf__{norm}"] = o_sc.astype(np.float32)
        assert np.array_equal(blob[f"out_scores__nsf__{norm}"].astype(np.float64), o_sc, equal_nan=True)
        blob[f"out_len__nsf__{norm}"] = o_len
    np.savez_compressed(os.path.join(OUT, f"pr28k_seed60_S{S}_Q{Q}_N{N}.npz"), **blob)


def gen_tune10k(Aggregator):
    """The weight-grid loop (hybrid.py:404-426) with 10,001-entry tables (the `_10k` size of hybrid.py:374), percentile-rank and NCE."""
    import copy
    from src.retrievers.hybrid import run_evaluation
    seed, S, Q, N = 22, 2, 4, 257
    rng = np.random.default_rng(seed)
    systems, ids, lists = make_case(rng, S, Q, N, "plain")
    labels = []
    for q in range(Q):
        pool = [x["corpus_id"] for s in systems for x in lists[s][q][:40]]
        labels.append([int(x) for x in rng.choice(pool, size=int(rng.integers(1, 5)), replace=False).tolist()])
    distr = _reference_tables(rng, systems, 60, N, 10000)
    combos = _lattice(systems)
    in_ids, in_sc, in_len = pack_lists(systems, lists, Q)
    blob = {"systems": np.array(systems), "in_ids": in_ids, "in_scores": in_sc, "in_len": in_len,
            "labels": np.array([",".join(str(x) for x in g) for g in labels]),
            "weights": np.array([[w[s] for s in systems] for w in combos], dtype=np.float64)}
    for s in systems:
        blob[f"distr_{s}"] = distr[s]
    names = None
    for norm in ["percentile-rank", "normal-curve-equivalent"]:
        rows = []
        for w in combos:
            fused = Aggregator.fuse(copy.deepcopy(lists), method="nsf", normalization=norm, percentile_distributions=distr, linear_weights=w)
            perf = run_evaluation(predictions=[[x["corpus_id"] for x in r] for r in fused], labels=labels, print2console=False)
            names = names or list(perf.keys())
            assert list(perf.keys()) == names
            rows.append([float(perf[k]) for k in names])
        blob[f"metrics__{norm}"] = np.array(rows, dtype=np.float64)
    blob["metric_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, f"tune10k_seed{seed}_S{S}_Q{Q}_N{N}.npz"), **blob)


def main():
    os.makedirs(OUT, exist_ok=True)
    # splade first: transformers' lazy imports probe optional packages with importlib.util.find_spec, which chokes on
    # the spec-less placeholder modules that load_reference() installs for hybrid.py / bm25.py
    BaseModel, SPLADE = load_reference_splade()
    Aggregator, BM25, Metrics = load_reference()
    only = set(sys.argv[1:])   # e.g. `python oracle/gen_golden.py pr28k tune10k`: those generators alone (default: all)
    names = []
    if not only or "fuse" in only:
        names = gen_fuse(Aggregator)
    for key, fn, arg in [("kat", gen_kat, Aggregator), ("bm25", gen_bm25, BM25), ("metrics", gen_metrics, Metrics),
                         ("similarity", gen_similarity, BaseModel), ("search", gen_search, BaseModel),
                         ("splade_pool", gen_splade_pool, SPLADE), ("tune", gen_tune, Aggregator), ("analysis", gen_analysis, Aggregator),
                         ("unsorted", gen_unsorted, Aggregator), ("fullrow", gen_fullrow, Aggregator), ("pr28k", gen_pr28k, Aggregator),
                         ("tune10k", gen_tune10k, Aggregator), ("bm25_family", gen_bm25_family, None)]:
        if not only or key in only:
            fn(arg)
    print("wrote", len(names), "fuse fixtures + kat_fuse.json, bm25.json, metrics.json, sim_*.npz, search_*.npz, splade_pool_*.npz, "
          "tune_*.npz, analysis_*.npz, unsorted_fuse.json, fuse_fullrow_*.npz, pr28k_*.npz, tune10k_*.npz, bm25_family.json ->", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
