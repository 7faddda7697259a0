#!/usr/bin/env python3
"""Generate golden input/output vectors from the REFERENCE itself (test infrastructure).

Runs only in the build container, where /root/reference exists; the GPU box never
sees the reference, only the small fixtures this script writes to tests/golden/.

What is imported from the reference (unmodified, no bytecode written):
  * src.retrievers.hybrid.Aggregator   (hybrid.py:166-307)  -- fuse / transform_scores
  * src.retrievers.bm25.BM25           (bm25.py:129-156)    -- search_all
  * src.utils.metrics.Metrics          (metrics.py:25-162)  -- compute_all_metrics

Absent third-party modules that those files import at module top but never touch
on the functions we call (dotenv, ir_datasets, seaborn, wandb, spacy) are replaced
by empty harness-side placeholder modules in sys.modules. No arithmetic goes
through a placeholder. sentence-transformers / colbert-ai are NOT stubbed:
their arithmetic is not reproduced here ("parity unpinned" at those call sites,
see DESIGN.md).

Fixtures are data only: seeded synthetic inputs + the reference's outputs.
Usage: python oracle/gen_golden.py  (rewrites tests/golden/*.npz|json)
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
            out["idf"] = {w: float(v) for w, v in m.idf.items()}
            out["avgdl"] = float(m.avgdl)
    # KAT-8 of SURVEY.md
    kdocs = ["chat noir dormir", "chien noir courir courir", "loi article code civil", "chat chien"]
    m = BM25(corpus=kdocs, k1=2.5, b=0.2)
    out["kat8"] = {"docs": kdocs, "queries": ["chat noir", "courir inconnu"],
                   "results": [[[int(x["corpus_id"]), float(x["score"])] for x in r] for r in m.search_all(["chat noir", "courir inconnu"], top_k=4)],
                   "avgdl": float(m.avgdl), "idf_chat": float(m.idf["chat"])}
    with open(os.path.join(OUT, "bm25.json"), "w") as f:
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


def main():
    os.makedirs(OUT, exist_ok=True)
    Aggregator, BM25, Metrics = load_reference()
    names = gen_fuse(Aggregator)
    gen_kat(Aggregator)
    gen_bm25(BM25)
    gen_metrics(Metrics)
    print("wrote", len(names), "fuse fixtures + kat_fuse.json, bm25.json, metrics.json ->", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
