"""The rest of the reference's lexical module on the CPU: the oracle's TFIDF / AtireBM25 restatements and its BM25 grid loop against
tests/golden/bm25_family.json (outputs of src/retrievers/bm25.py itself, oracle/gen_golden.py::gen_bm25_family), the drop-in module's
CLI surface (flags, shim, scripts/run_bm25.sh), and the host-side metric evaluation the device sweep ends in.  No GPU."""
import itertools
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def fam():
    return json.load(open(os.path.join(GOLDEN, "bm25_family.json")))


def _pairs(lists):
    return [[[x["corpus_id"], x["score"]] for x in r] for r in lists]


def test_oracle_tfidf_and_atire_match_the_reference(oracle, fam):
    t = oracle.TFIDF(fam["docs"])
    for w, v in fam["tfidf"]["idf"].items():
        assert t.idf[t.vocab[w]] == v                                     # log10((N + 1) / (df + 1)), bm25.py:86-88
    assert _pairs(t.search_all(fam["queries"], top_k=50)) == fam["tfidf"]["results"]          # ids and float64 scores bit for bit
    a = oracle.AtireBM25(fam["docs"], fam["atire"]["k1"], fam["atire"]["b"])
    assert _pairs(a.search_all(fam["queries"], top_k=50)) == fam["atire"]["results"]


def test_oracle_grid_rows_match_the_reference_loop(oracle, fam):
    """bm25.py:221-233 restated on the oracle: update_params -> search_all(top_k=1000) -> idx2id -> Metrics, every (k1, b) of the grid
    with k1 > 0 (the k1 = 0 column is NaN-sorted in the reference: see the generator)."""
    ids, gold = fam["ids"], fam["gold"]
    m = oracle.BM25(fam["docs"], 0.0, 0.0)
    ev = oracle.Metrics(recall_at_k=[10, 100, 200, 500, 1000])
    rows = {(r["k1"], r["b"]): r for r in fam["grid"]["rows"]}
    k1_range, b_range = np.arange(0., 8.5, 0.5), np.arange(0., 1.1, 0.1)
    assert [float(x) for x in k1_range] == fam["grid"]["k1_range"] and [float(x) for x in b_range] == fam["grid"]["b_range"]
    checked = 0
    for k1, b in list(itertools.product(k1_range, b_range))[11::7]:       # every 7th pair: the whole grid is the GPU test's job
        m.update_params(k1, b)
        ranked = [[ids[x["corpus_id"]] for x in r] for r in m.search_all(fam["queries"], top_k=1000)]
        got = ev.compute_all_metrics(gold, ranked)
        exp = rows[(float(k1), float(b))]
        for name, v in got.items():
            assert float(v) == exp[name], (k1, b, name)
        checked += 1
    assert checked >= 20
    m.update_params(2.5, 0.2)
    ranked = [[ids[x["corpus_id"]] for x in r] for r in m.search_all(fam["queries"], top_k=1000)]
    assert [r[:20] for r in ranked] == fam["top1000_head"]
    ev = oracle.Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    got = ev.compute_all_metrics(gold, ranked)
    assert {k: float(v) for k, v in got.items()} == fam["evaluation"]
    neg = {str(q): [y for y in p if y not in g][:10] for q, g, p in zip(fam["qids"], gold, ranked)}
    assert neg == fam["negatives"]


def test_gold_rank_metrics_with_chosen_cutoffs_equal_the_list_metrics():
    """BM25.tune ends in metrics_from_gold_ranks(recall_ks=..., only=...): the same numbers Metrics computes from the lists."""
    from fusion_amd.utils.metrics import Metrics, metrics_from_gold_ranks
    rng = np.random.default_rng(5)
    Q, N, top_k = 9, 400, 300
    perms = [rng.permutation(N) for _ in range(Q)]
    gold = [sorted(rng.choice(N, size=int(rng.integers(1, 5)), replace=False).tolist()) for _ in range(Q)]
    gold[3] = gold[3] + [N + 7]                                            # a gold id that is not in the corpus
    ranks = np.full((1, Q, 5), np.iinfo(np.int64).max, dtype=np.int64)
    for q in range(Q):
        pos = np.argsort(perms[q])
        for i, g in enumerate(gold[q]):
            if g < N and pos[g] < top_k:
                ranks[0, q, i] = pos[g]
    got = metrics_from_gold_ranks(ranks, np.array([len(g) for g in gold]), np.full(Q, top_k), recall_ks=[10, 100, 200, 500, 1000],
                                  only=[f"recall@{k}" for k in (10, 100, 200, 500, 1000)] + ["r-precision"])[0]
    exp = Metrics(recall_at_k=[10, 100, 200, 500, 1000]).compute_all_metrics(gold, [p[:top_k].tolist() for p in perms])
    assert list(got) == list(exp)
    for k in exp:
        assert abs(got[k] - float(exp[k])) <= 1e-15, k


def test_cli_surface_matches_the_reference():
    """Flags of bm25.py:275-291 (unknown ones ignored), the shim under src/retrievers/, and run_bm25.sh's positional interface."""
    from fusion_amd.retrievers.bm25 import build_parser
    a, rest = build_parser().parse_known_args("--dataset lleqa --do_preprocessing --do_hyperparameter_tuning --output_dir o --frobnicate 3".split())
    assert a.dataset == "lleqa" and a.do_preprocessing and a.do_hyperparameter_tuning and not a.do_evaluation and rest == ["--frobnicate", "3"]
    a, _ = build_parser().parse_known_args("--dataset mmarco-fr --do_evaluation --k1 0.9 --b 0.4 --output_dir o".split())
    assert (a.k1, a.b, a.num_negatives) == (0.9, 0.4, 10) and a.do_evaluation and not a.do_negatives_extraction
    assert build_parser().parse_known_args([])[0].k1 == 1.5 and build_parser().parse_known_args([])[0].b == 0.75      # bm25.py:282-283
    with pytest.raises(SystemExit):
        build_parser().parse_known_args(["--dataset", "msmarco"])
    env = dict(os.environ, DRY_RUN="1", BM25_EXTRA="--synthetic 300,4")
    sh = os.path.join(ROOT, "scripts", "run_bm25.sh")
    out = subprocess.run(["bash", sh, "tuning", "lleqa"], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "--do_hyperparameter_tuning" in out.stdout and "--output_dir output/tuning" in out.stdout and "--synthetic 300,4" in out.stdout
    out = subprocess.run(["bash", sh, "testing", "lleqa"], env=env, capture_output=True, text=True)
    assert "--do_evaluation" in out.stdout and "--k1 2.5 --b 0.2" in out.stdout                                       # run_bm25.sh:23-25
    out = subprocess.run(["bash", sh, "testing", "mmarco"], env=env, capture_output=True, text=True)
    assert "--k1 0.9 --b 0.4" in out.stdout and "--dataset mmarco-fr" in out.stdout                                   # run_bm25.sh:26-28
    for bad in (["train", "lleqa"], ["tuning", "trec"], []):
        r = subprocess.run(["bash", sh, *bad], env=env, capture_output=True, text=True)
        assert r.returncode == 1 and r.stdout.startswith("ERROR:")                                                      # run_bm25.sh:4-13
    shim = open(os.path.join(ROOT, "src", "retrievers", "bm25.py")).read()
    assert "from fusion_amd.retrievers.bm25 import" in shim and "TFIDF" in shim and "AtireBM25" in shim


def test_preprocessing_without_spacy_is_an_error_not_a_no_op():
    from fusion_amd.retrievers import bm25
    try:
        import spacy  # noqa: F401
        spacy.load("fr_core_news_md")
    except Exception:
        with pytest.raises(RuntimeError, match="fr_core_news_md"):
            bm25.preprocess(["Le chat dort."])
    else:
        assert bm25.preprocess(["Le chat dort 3 fois."])[0].islower()


def test_expected_zero_share_picks_the_ranking_sort():
    """The host-side estimate behind `lexical=`: per query prod over its DISTINCT known terms of (1 - df / N), averaged."""
    from fusion_amd.retrievers.bm25 import LEXICAL_MIN_ZERO_SHARE, expected_zero_share
    df = np.array([5, 0, 10, 1])
    assert expected_zero_share(df, 10, [[0, 0, 1], [2], [-1], []]) == pytest.approx((0.5 + 0.0 + 1.0 + 1.0) / 4)
    assert expected_zero_share(df, 10, []) == 0.0 and expected_zero_share(df, 0, [[0]]) == 0.0
    assert expected_zero_share(np.array([998, 3]), 1000, [[0, 1]]) < LEXICAL_MIN_ZERO_SHARE < expected_zero_share(np.array([998, 3]), 1000, [[1, 1]])
