"""Pins the CPU oracle on the golden vectors produced by the REFERENCE's own classes
(oracle/gen_golden.py -> tests/golden/).  CPU only; runs in seconds."""
import glob
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

FUSE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "fuse_*.npz")))
METHODS = [("rrf", "none"), ("bcf", "none"), ("nsf", "none"), ("nsf", "min-max"), ("nsf", "z-score"), ("nsf", "arctan"),
           ("nsf", "percentile-rank"), ("nsf", "normal-curve-equivalent")]
# fp64 rank fusion / fp64 passthrough: bit-exact.  fp32 elementwise transforms of exact stats
# (min-max, percentile-rank): bit-exact.  z-score (torch's mean/std summation order), arctan /
# NCE (libm vs SLEEF atan/erfinv): within the 1e-4 contract -- tolerance stated here.
EXACT = {("rrf", "none"), ("bcf", "none"), ("nsf", "none"), ("nsf", "min-max"), ("nsf", "percentile-rank")}
TOL = {("nsf", "z-score"): 2e-6, ("nsf", "arctan"): 1e-6, ("nsf", "normal-curve-equivalent"): 1e-4}


def load_case(path):
    z = np.load(path, allow_pickle=False)
    systems = [str(s) for s in z["systems"]]
    ids, sc, ln = z["in_ids"], z["in_scores"], z["in_len"]
    Q = ids.shape[1]
    lists = {s: [[{"corpus_id": int(ids[si, q, r]), "score": float(sc[si, q, r])} for r in range(ln[si, q])] for q in range(Q)]
             for si, s in enumerate(systems)}
    weights = {s: float(w) for s, w in zip(systems, z["weights"])}
    distr = {s: z[f"distr_{s}"] for s in systems}
    return z, systems, lists, weights, distr, Q


def test_fixtures_present():
    assert len(FUSE_FILES) >= 10
    for f in ("kat_fuse.json", "bm25.json", "metrics.json"):
        assert os.path.exists(os.path.join(GOLDEN, f))


@pytest.mark.parametrize("path", FUSE_FILES, ids=[os.path.basename(p)[:-4] for p in FUSE_FILES])
@pytest.mark.parametrize("method,norm", METHODS)
def test_fuse_matches_reference(oracle, path, method, norm):
    z, systems, lists, weights, distr, Q = load_case(path)
    got = oracle.fuse_lists(lists, method=method, normalization=norm, linear_weights=weights, percentile_distributions=distr)
    key = f"{method}__{norm}"
    e_ids, e_sc, e_len = z[f"out_ids__{key}"], z[f"out_scores__{key}"], z[f"out_len__{key}"]
    assert len(got) == Q
    for q in range(Q):
        n = int(e_len[q])
        assert len(got[q]) == n
        g_ids = np.array([x["corpus_id"] for x in got[q]], dtype=np.int64)
        g_sc = np.array([float(x["score"]) for x in got[q]], dtype=np.float64)
        if (method, norm) in EXACT:
            np.testing.assert_array_equal(g_ids, e_ids[q, :n])
            np.testing.assert_array_equal(g_sc, e_sc[q, :n])
        else:
            tol = TOL[(method, norm)]
            # same multiset of ids; scores (looked up by id) within tol; order may differ only among near-ties
            assert sorted(g_ids.tolist()) == sorted(e_ids[q, :n].tolist())
            exp = {int(i): s for i, s in zip(e_ids[q, :n], e_sc[q, :n])}
            ref = np.array([exp[int(i)] for i in g_ids])
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(g_sc), fin)
            assert np.array_equal(np.isnan(g_sc), np.isnan(ref))
            assert np.array_equal(g_sc[np.isinf(ref)], ref[np.isinf(ref)])
            assert np.max(np.abs(g_sc[fin] - ref[fin]), initial=0.0) <= tol
            # our own list is sorted desc (NaN first)
            d = g_sc[~np.isnan(g_sc)]
            assert np.all(d[:-1] >= d[1:])


def _dec(v):
    return float("nan") if v == "nan" else v


def test_kat_fuse(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "kat_fuse.json")))
    L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]
    a = {"s1": [L([(100, 2.0), (200, 1.0)])], "s2": [L([(200, 2.0), (100, 1.0)])]}
    b = {"s2": a["s2"], "s1": a["s1"]}
    # KAT-1 first-insertion tie-break
    assert oracle.fuse_lists(a, "rrf") == kat["kat1_s1s2"]
    assert oracle.fuse_lists(b, "rrf") == kat["kat1_s2s1"]
    assert [x["corpus_id"] for x in kat["kat1_s1s2"][0]] == [100, 200]
    assert [x["corpus_id"] for x in kat["kat1_s2s1"][0]] == [200, 100]
    lists = {"bm25": [L([(10, 7.5), (11, 3.0), (12, 0.0)])], "dpr": [L([(12, .9), (10, .5), (13, .1)])]}
    w = {"bm25": .5, "dpr": .5}
    for m, n, tol in [("rrf", "none", 0), ("bcf", "none", 0), ("nsf", "min-max", 0), ("nsf", "z-score", 1e-6), ("nsf", "arctan", 1e-6),
                      ("nsf", "none", 1e-7)]:
        got = oracle.fuse_lists(lists, m, n, w, {})
        exp = kat[f"kat2_{m}_{n}"]
        assert [x["corpus_id"] for x in got[0]] == [x["corpus_id"] for x in exp[0]]
        for g, e in zip(got[0], exp[0]):
            assert abs(float(g["score"]) - e["score"]) <= tol
    # KAT-3 uneven lists
    lists = {"s1": [L([(1, 5.0), (2, 1.0)])], "s2": [L([(3, 9.0), (1, 8.0), (2, 7.0)])]}
    got = oracle.fuse_lists(lists, "nsf", "z-score", {"s1": .5, "s2": .5}, {})
    assert [x["corpus_id"] for x in got[0]] == [x["corpus_id"] for x in kat["kat3_zscore"][0]]
    for g, e in zip(got[0], kat["kat3_zscore"][0]):
        assert abs(float(g["score"]) - e["score"]) <= 1e-6
    # KAT-4 constant rows / single element (through fuse with weight 1)
    const = {"s": [L([(1, 2.0), (2, 2.0), (3, 2.0)])]}
    assert [float(x["score"]) for x in oracle.fuse_lists(const, "nsf", "min-max", {"s": 1.0}, {})[0]] == [x["score"] for x in kat["kat4_minmax_const"]]
    assert [float(x["score"]) for x in oracle.fuse_lists(const, "nsf", "z-score", {"s": 1.0}, {})[0]] == [x["score"] for x in kat["kat4_zscore_const"]]
    single = oracle.fuse_lists({"s": [L([(1, 2.0)])]}, "nsf", "z-score", {"s": 1.0}, {})
    assert math.isnan(float(single[0][0]["score"])) and math.isnan(_dec(kat["kat4_zscore_single"][0]["score"]))
    # KAT-5 percentile rank
    got = oracle.fuse_lists({"s": [L([(1, 0.2), (2, 4.6), (3, 99.0)])]}, "nsf", "percentile-rank", {"s": 1.0}, {"s": np.linspace(0, 10, 11)})
    exp = {x["corpus_id"]: x["score"] for x in kat["kat5_percentile"]}
    for x in got[0]:
        assert float(x["score"]) == exp[x["corpus_id"]]
    # KAT-6 return_topk slices queries
    three = {"s1": [L([(1, 1.0), (2, .5)])] * 3, "s2": [L([(2, 1.0), (1, .5)])] * 3}
    assert oracle.fuse_lists(three, "rrf", return_topk=2) == kat["kat6_topk2"]
    # duplicate ids collapse
    dup = {"s1": [L([(1, 3.0), (2, 2.0), (1, 1.0), (3, 0.5)])], "s2": [L([(3, 1.0), (2, .5)])]}
    assert oracle.fuse_lists(dup, "rrf") == kat["kat_dup_rrf"]
    assert oracle.fuse_lists(dup, "bcf") == kat["kat_dup_bcf"]


def test_fuse_error_behaviour(oracle):
    L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]
    with pytest.raises(AssertionError):
        oracle.fuse_lists({"a": [L([(1, 1.0)])], "b": [L([(1, 1.0)])] * 2}, "rrf")
    with pytest.raises(KeyError):
        oracle.fuse_lists({"a": [L([(1, 1.0)])], "b": [L([(1, 1.0)])]}, "nsf", "min-max", {"a": 1.0}, {})
    with pytest.raises(AttributeError):
        oracle.fuse_lists({"a": [L([(1, 1.0)])]}, "nsf", "min-max", {"a": 1.0}, None)


def test_bm25_matches_reference(oracle):
    g = json.load(open(os.path.join(GOLDEN, "bm25.json")))
    for (k1, b), exp in zip(g["params"], g["results"]):
        m = oracle.BM25(g["docs"], k1=k1, b=b)
        got = m.search_all(g["queries"], top_k=len(g["docs"]))
        for gq, eq in zip(got, exp):
            assert [x["corpus_id"] for x in gq] == [e[0] for e in eq]
            assert [x["score"] for x in gq] == [e[1] for e in eq]  # float64 bit-exact
    m = oracle.BM25(g["docs"], 2.5, 0.2)
    assert m.avgdl == g["avgdl"]
    for w, v in g["idf"].items():
        assert m.idf[m.vocab[w]] == v
    k = g["kat8"]
    m = oracle.BM25(k["docs"], 2.5, 0.2)
    assert m.avgdl == k["avgdl"] == 3.25 and m.idf[m.vocab["chat"]] == k["idf_chat"] == 0.0
    got = m.search_all(k["queries"], top_k=4)
    assert [[[x["corpus_id"], x["score"]] for x in r] for r in got] == k["results"]
    assert [x["corpus_id"] for x in got[0]] == [0, 1, 2, 3]
    assert got[1][0]["corpus_id"] == 1 and abs(got[1][0]["score"] - 0.558098) < 1e-6


def test_metrics_match_reference(oracle):
    g = json.load(open(os.path.join(GOLDEN, "metrics.json")))
    ev = oracle.Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    for c in g["cases"]:
        got = ev.compute_all_metrics(c["gold"], c["pred"])
        assert set(got) == set(c["scores"])
        for k, v in c["scores"].items():
            assert float(got[k]) == v, k
    k9 = oracle.Metrics([1, 2, 500], [2], [2], [2]).compute_all_metrics([[1, 2], [9]], [[1, 3, 2], [4, 9, 5]])
    assert {k: float(v) for k, v in k9.items()} == g["kat9"]
    assert g["kat9"]["recall@1"] == .25 and g["kat9"]["recall@2"] == .75 and g["kat9"]["ndcg@2"] == .75
