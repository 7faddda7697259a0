"""The reference's lexical module on the MI355X (fusion_amd/retrievers/bm25.py): TFIDF / AtireBM25 through the device scoring + row sort
against the reference's own outputs (tests/golden/bm25_family.json), the k1 x b grid search against the reference's loop, and the driver
(`main`) end to end.  Bar: ids and float64 scores bit for bit; metric means <= 1e-12 (statistics.mean vs an exactly rounded sum)."""
import itertools
import json
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fam():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return json.load(open(os.path.join(GOLDEN, "bm25_family.json")))


def _pairs(lists):
    return [[[x["corpus_id"], x["score"]] for x in r] for r in lists]


def test_tfidf_and_atire_match_the_reference(fam):
    from fusion_amd.retrievers.bm25 import BM25, TFIDF, AtireBM25
    t = TFIDF(fam["docs"])
    assert repr(t) == fam["tfidf"]["repr"] and t.get_vocab()[:5] == fam["tfidf"]["vocab_sorted_head"]
    for w, v in fam["tfidf"]["idf"].items():
        assert t.idf_host[t.vocab[w]] == v
    assert _pairs(t.search_all(fam["queries"], top_k=50)) == fam["tfidf"]["results"]          # ids, float64 scores: bit for bit
    assert _pairs([t.search(fam["queries"][0], top_k=50)]) == fam["tfidf"]["results"][:1]
    a = AtireBM25(fam["docs"], fam["atire"]["k1"], fam["atire"]["b"])
    assert repr(a) == fam["atire"]["repr"] and isinstance(a, BM25)
    assert _pairs(a.search_all(fam["queries"], top_k=50)) == fam["atire"]["results"]


def test_tfidf_vs_oracle_larger_and_the_float32_plane(oracle):
    """3,000 documents x 42 queries (repeated, out-of-vocabulary and empty queries): the TF-IDF plane == the oracle's, bit for bit; the
    float32 plane of the same launch == the rounding of the float64 one; BM25's walk is untouched by the new instantiation."""
    from fusion_amd.retrievers.bm25 import BM25, TFIDF
    rng = np.random.default_rng(19)
    vocab = np.array([f"w{i}" for i in range(2000)])
    p = 1.0 / np.arange(1, 2001); p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 120)), p=p)) for _ in range(30_011)]   # five 7,168-document slices
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 12)), p=p)) for _ in range(40)] + ["", "zzz w1 w1"]
    t = TFIDF(docs)
    s64, s32 = t.scores(queries, want_f32=True)
    np.testing.assert_array_equal(s64.cpu().numpy(), oracle.TFIDF(docs).scores(queries))
    assert torch.equal(s32, s64.to(torch.float32))
    np.testing.assert_array_equal(BM25(docs, 2.5, 0.2).scores(queries).cpu().numpy(), oracle.BM25(docs, 2.5, 0.2).scores(queries))


def test_grid_search_matches_the_reference_loop(fam, oracle):
    """BM25.tune over the reference's whole 17 x 11 grid == the rows of its loop (bm25.py:221-233) for every k1 > 0; the k1 = 0 column (NaN
    keys in the reference) == the binary model the posting walk gives, checked against the oracle's lists."""
    from fusion_amd.retrievers.bm25 import BM25
    m = BM25(fam["docs"], k1=0., b=0.)
    rows = m.tune(fam["queries"], fam["gold"], ids=np.array(fam["ids"]))
    assert len(rows) == 187 and (m.k1, m.b) == (0., 0.)                    # the model's own parameters are restored
    assert list(rows[0]) == ["k1", "b", "recall@10", "recall@100", "recall@200", "recall@500", "recall@1000", "r-precision"]
    exp = {(r["k1"], r["b"]): r for r in fam["grid"]["rows"]}
    seen = 0
    for r in rows:
        e = exp.get((r["k1"], r["b"]))
        if e is None:
            assert r["k1"] == 0.0
            continue
        for name in e:
            assert abs(r[name] - e[name]) <= 1e-12, (r["k1"], r["b"], name, r[name], e[name])
        seen += 1
    assert seen == 176
    om = oracle.BM25(fam["docs"], 0.0, 0.3)
    ev = oracle.Metrics(recall_at_k=[10, 100, 200, 500, 1000])
    ranked = [[fam["ids"][x["corpus_id"]] for x in r] for r in om.search_all(fam["queries"], top_k=1000)]
    e0 = ev.compute_all_metrics(fam["gold"], ranked)
    r0 = [r for r in rows if r["k1"] == 0.0 and abs(r["b"] - 0.3) < 1e-9][0]
    for name, v in e0.items():
        assert abs(r0[name] - float(v)) <= 1e-12, name


def test_grid_search_equals_search_all_plus_metrics_on_long_rows():
    """30,011 documents (rows beyond one workgroup: chunk-sort + merge), top_k = 1000 cuts gold documents off: tune == the loop it replaces
    (update_params -> search_all -> Metrics) on the device classes themselves, with duplicate labels and a gold id outside the corpus."""
    from fusion_amd.retrievers.bm25 import BM25
    from fusion_amd.utils.metrics import Metrics
    rng = np.random.default_rng(23)
    vocab = np.array([f"w{i}" for i in range(3000)])
    p = 1.0 / np.arange(1, 3001); p /= p.sum()
    N, Q = 30_011, 7
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 80)), p=p)) for _ in range(N)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(2, 9)), p=p)) for _ in range(Q)]
    ids = np.arange(N) * 2 + 5
    gold = [sorted(int(x) for x in rng.choice(ids, size=int(rng.integers(1, 4)), replace=False)) for _ in range(Q)]
    gold[2] = gold[2] + [gold[2][0], 4]                                    # a duplicate label and an id that is not in the corpus
    m = BM25(docs, 1.0, 1.0)
    k1s, bs = [0.5, 2.5], [0.2, 0.75]
    rows = m.tune(queries, gold, ids=ids, k1_range=k1s, b_range=bs)
    ev = Metrics(recall_at_k=[10, 100, 200, 500, 1000])
    for r, (k1, b) in zip(rows, itertools.product(k1s, bs)):
        m.update_params(k1, b)
        ranked = [[int(ids[x["corpus_id"]]) for x in l] for l in m.search_all(queries, top_k=1000)]
        e = ev.compute_all_metrics(gold, ranked)
        assert (r["k1"], r["b"]) == (k1, b)
        for name, v in e.items():
            assert abs(r[name] - float(v)) <= 1e-12, (k1, b, name)
    pos = m.ranked_positions(queries, top_k=1000, budget_bytes=3 * 20 * 30_016)   # three queries at a time
    assert pos.shape == (Q, 1000) and [int(ids[j]) for j in pos[-1]] == ranked[-1]


def test_driver_end_to_end(tmp_path, oracle):
    """`python src/retrievers/bm25.py` in its three modes on the synthetic corpus: the grid search writes the CSV the reference's loop was
    meant to write (187 rows, %.5f), the evaluation run writes performance_bm25_<dataset>_dev.json == the oracle's BM25 + Metrics on the
    same data, the negatives are the top non-gold predictions, the pickles hold the reference's index types."""
    import pandas as pd
    from fusion_amd.retrievers import bm25 as mod
    base = ["--dataset", "lleqa", "--synthetic", "1500,6"]
    a, _ = mod.build_parser().parse_known_args(base + ["--do_hyperparameter_tuning", "--output_dir", str(tmp_path / "tuning")])
    rows = mod.main(a)
    df = pd.read_csv(tmp_path / "tuning" / "bm25_tuning_results.csv")
    assert len(df) == len(rows) == 187 and list(df.columns)[:3] == ["k1", "b", "recall@10"]
    assert abs(df["recall@100"][40] - round(rows[40]["recall@100"], 5)) < 1e-9
    assert (tmp_path / "tuning" / "bm25_tuning_heatmap.pdf").exists()
    a, _ = mod.build_parser().parse_known_args(base + ["--do_evaluation", "--do_negatives_extraction", "--num_negatives", "4", "--k1", "2.5", "--b", "0.2",
                                                       "--output_dir", str(tmp_path / "testing")])
    perf = mod.main(a)
    corpus, qids, queries, pos = mod.load_data(a)
    docs, ids = list(corpus.values()), list(corpus.keys())
    om = oracle.BM25(docs, 2.5, 0.2)
    ranked = [[ids[x["corpus_id"]] for x in r] for r in om.search_all(queries, top_k=1000)]
    ev = oracle.Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    exp = ev.compute_all_metrics(pos, ranked)
    on_disk = json.load(open(tmp_path / "testing" / "performance_bm25_lleqa_dev.json"))
    assert {k: float(v) for k, v in exp.items()} == on_disk == {k: float(v) for k, v in perf.items()}
    neg = json.load(open(tmp_path / "testing" / "negatives_bm25.json"))
    assert neg == {str(q): [y for y in p if y not in g][:4] for q, g, p in zip(qids, pos, ranked)}
    vocab = pickle.load(open(tmp_path / "testing" / "bm25_vocab_lleqa.pkl", "rb"))
    tf = pickle.load(open(tmp_path / "testing" / "bm25_tf_lleqa.pkl", "rb"))
    dfc = pickle.load(open(tmp_path / "testing" / "bm25_df_lleqa.pkl", "rb"))
    idf = pickle.load(open(tmp_path / "testing" / "bm25_idf_lleqa.pkl", "rb"))
    assert isinstance(vocab, set) and vocab == {w for d in docs for w in d.split()}
    w = docs[3].split()[0]
    assert tf[w][3] == docs[3].split().count(w) and dfc[w] == sum(1 for d in docs if w in d.split())
    assert idf[w] == om.idf[om.vocab[w]] and type(dfc).__name__ == "Counter"


def test_posting_value_table_gives_the_same_bits(oracle):
    """Round 6: BM25 scoring that ADDS tabulated posting terms (fz_bm25_posting_values_f64 + fz_bm25_scores_pv_f64_f32) == the per-posting
    float64 expression (fz_bm25_scores_f64_f32) == the oracle, bit for bit: five 7,168-document slices, repeated / unknown / no query terms,
    negative idf (a term in more than half of the documents), after update_params (a new table per (k1, b)), the float32 plane too."""
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(41)
    vocab = np.array([f"w{i}" for i in range(1500)])
    p = 1.0 / np.arange(1, 1501); p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 120)), p=p)) for _ in range(30_011)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 12)), p=p)) for _ in range(30)] + ["", "zzz w1 w1", "w0 w0 w0"]
    m = BM25(docs, 2.5, 0.2)
    assert float(m.idf_host.min()) < 0.0
    for k1, b in ((2.5, 0.2), (0.9, 0.4), (0.0, 0.7), (8.0, 1.0)):
        m.update_params(k1, b)
        m.USE_POSTING_VALUES = True
        a64, a32 = m.scores(queries, want_f32=True)
        assert m._pval is not None and m._pval.numel() == m.pdoc.numel()
        m.USE_POSTING_VALUES = False
        b64, b32 = m.scores(queries, want_f32=True)
        assert torch.equal(a64.view(torch.int64), b64.view(torch.int64)) and torch.equal(a32.view(torch.int32), b32.view(torch.int32))
        np.testing.assert_array_equal(a64.cpu().numpy(), oracle.BM25(docs, k1, b).scores(queries))


def test_top_k_of_a_corpus_beyond_one_sort_row_is_cut_not_ranked(oracle):
    """90,011 documents (four 28,672-document stretches, the last one short), top-1000 and top-7: the hierarchical cut == the first entries of
    the oracle's full stable ranking, ties (rounded scores, many zeros) by ascending index across stretch borders; queries in budgeted chunks."""
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(43)
    vocab = np.array([f"w{i}" for i in range(600)])
    p = 1.0 / np.arange(3, 603); p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 6)), p=p)) for _ in range(90_011)]      # short documents: few distinct scores, long tie runs
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 4)), p=p)) for _ in range(5)] + ["zzz"]  # (the last one: every score 0.0)
    m = BM25(docs, 1.2, 0.0)
    om = oracle.BM25(docs, 1.2, 0.0)
    exp = om.search_all(queries, top_k=1000)
    got = m.ranked_positions(queries, top_k=1000, budget_bytes=2 * 36 * 90_048)        # two queries at a time
    assert got.shape == (6, 1000)
    for q in range(6):
        assert got[q].tolist() == [x["corpus_id"] for x in exp[q]], q
    assert m.ranked_positions(queries, top_k=7)[3].tolist() == [x["corpus_id"] for x in exp[3][:7]]
