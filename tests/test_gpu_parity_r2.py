"""GPU parity against the round-2 reference fixtures (oracle/gen_golden.py): the HIP scoring / search / SPLADE-pooling
kernels against outputs of the reference's own splade/base.py + splade.py, Aggregator.tune against the reference's
weight-grid loop, analyze_score_distributions against the reference's analysis recipe, unsorted / duplicate-id lists.
Everything goes through the C ABI (fusion_amd.ops / the drop-in classes)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import assert_ranked_close, load_lists, sparse_to_dense

pytestmark = pytest.mark.gpu

COS_TOL = 2e-6     # |cosine error| of fp32 rows; raw dot products: relative to the largest score


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---- scoring: compute_batchwise_similarity (splade/base.py:186-197) --------------------------------------------------
def test_cos_and_dot_match_reference_dpr(ops):
    z = np.load(os.path.join(GOLDEN, "sim_dpr_Q8_N300_d768.npz"))
    Qe, De = dev(z["Qe"]), dev(z["De"])
    c = ops.cos_scores(Qe, De).cpu().numpy()
    assert np.max(np.abs(c - z["cos_sim"])) <= COS_TOL
    assert np.array_equal(c[:, 17], c[:, 3])                       # duplicated document: identical scores
    assert np.max(np.abs(c[:, 40] - c[:, 41])) <= 1e-7             # cosine ignores the scale of a row
    d = ops.dot_scores(Qe, De).cpu().numpy()
    assert np.max(np.abs(d - z["dot_score"])) <= COS_TOL * np.max(np.abs(z["dot_score"]))


def test_cos_and_dot_match_reference_splade(ops):
    z = np.load(os.path.join(GOLDEN, "sim_splade_Q4_N257_V32005.npz"))
    Q, N, V = (int(x) for x in z["shape"])
    Qs, Ds = dev(sparse_to_dense(z, "q", Q, V)), dev(sparse_to_dense(z, "d", N, V))
    assert np.max(np.abs(ops.cos_scores(Qs, Ds).cpu().numpy() - z["cos_sim"])) <= COS_TOL
    assert np.max(np.abs(ops.dot_scores(ops.pad_dim(Qs), ops.pad_dim(Ds)).cpu().numpy() - z["dot_score"])) <= COS_TOL * np.max(np.abs(z["dot_score"]))


# ---- search: chunked mm -> topk -> heap -> sorted (splade/base.py:199-251 == util.semantic_search, hybrid.py:103) ----
@pytest.mark.parametrize("sim", ["cos_sim", "dot_score"])
def test_search_matches_reference(ops, sim):
    from fusion_amd.distributed import ShardedDenseIndex
    from fusion_amd.retrievers.hybrid import _rank_scores
    z = np.load(os.path.join(GOLDEN, "search_Q6_N1000_d64.npz"))
    Qe, De = dev(z["Qe"]), dev(z["De"])
    Q, N = Qe.shape[0], De.shape[0]
    if sim == "cos_sim":
        Qn, Dn = ops.normalize_rows(Qe), ops.normalize_rows(De)
        tol = COS_TOL
    else:
        Qn, Dn = Qe, De
        tol = COS_TOL * float(np.max(np.abs(z[f"scores__{sim}__kN_qc100_dc500000"])))
    S = ops.dot_scores(Qn, Dn)
    # (a) the full ranking hybrid.py:103 asks for (top_k = N): Ranker's score -> rank path
    rs = _rank_scores(S, np.arange(N), None)
    order, sk = rs.order.cpu().numpy(), rs.list_scores().cpu().numpy()
    e_ids, e_sc = z[f"ids__{sim}__kN_qc100_dc500000"], z[f"scores__{sim}__kN_qc100_dc500000"]
    for q in range(Q):
        assert_ranked_close(order[q], sk[q], e_ids[q], e_sc[q], tol)
        assert sorted(order[q].tolist()) == list(range(N))
    for a, b in [(20, 500), (21, 501), (22, 999), (3, 700)]:      # exact duplicates: ties -> ascending document index
        for q in range(Q):
            pa, pb = int(np.flatnonzero(order[q] == a)[0]), int(np.flatnonzero(order[q] == b)[0])
            assert sk[q, pa] == sk[q, pb] and pb == pa + 1
    # (b) top-k of the chunked search: one-shot top-k, and the sharded index's chunked GEMM -> streaming top-k
    for cfg in z["configs"]:
        name, k, _qc, dc = str(cfg).split(":")
        k, dc = int(k), int(dc)
        e_ids, e_sc = z[f"ids__{sim}__{name}"], z[f"scores__{sim}__{name}"]
        g_sc, g_ids = ops.topk_rows(S, min(k, N))
        idx = ShardedDenseIndex(Dn, id_base=0)
        idx.CHUNK = min(dc, N)           # the reference's document chunking
        c_sc, c_ids = idx.search(Qn, k=min(k, N))
        for q in range(Q):
            assert_ranked_close(g_ids[q].cpu().numpy(), g_sc[q].cpu().numpy(), e_ids[q], e_sc[q], tol, truncated=k < N)
            assert_ranked_close(c_ids[q].cpu().numpy(), c_sc[q].cpu().numpy(), e_ids[q], e_sc[q], tol, truncated=k < N)
    if sim == "cos_sim":   # F.normalize's eps clamp: the zero vector scores exactly 0
        assert np.all(S[:, 123].cpu().numpy() == 0.0)


# ---- SPLADE pooling: SPLADE.forward (splade/splade.py:88-99), 'max' (the hybrid path's pooling, hybrid.py:96) --------
def test_splade_max_pool_matches_reference(ops):
    z = np.load(os.path.join(GOLDEN, "splade_pool_B5_L24_V509.npz"))
    logits, lens = z["logits"], z["lens"]
    rows = np.concatenate([logits[b, : lens[b]] for b in range(len(lens))])          # packed: attended tokens only
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    for pad in (0, 3):                                                                 # vector path (V % 4 == 0) and scalar path
        x = np.pad(rows, ((0, 0), (0, pad))) if pad else rows
        got = ops.segment_splade_max(dev(x), cu).cpu().numpy()[:, : rows.shape[1]]
        assert np.max(np.abs(got - z["max"])) <= 5e-7      # log1p: ocml vs torch's vectorised CPU implementation, 2 ulp of <= 2.2
        assert got[1, 7] == 0.0 and np.all(got >= 0)


# ---- N1: the weight-grid loop (hybrid.py:404-426) ---------------------------------------------------------------------
TUNE_FILES = ["tune_seed20_S2_Q4_N257_ties.npz", "tune_seed21_S3_Q4_N257_colbert_first.npz"]
TUNE_NORMS = ["min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent", "none"]


def load_tune(fname):
    z = np.load(os.path.join(GOLDEN, fname), allow_pickle=False)
    systems, lists, Q = load_lists(z)
    labels = [[int(x) for x in str(s).split(",")] for s in z["labels"]]
    distr = {s: z[f"distr_{s}"] for s in systems}
    return z, systems, lists, labels, distr


@pytest.mark.parametrize("fname", TUNE_FILES)
@pytest.mark.parametrize("norm", TUNE_NORMS)
def test_tune_matches_reference_loop(fname, norm):
    """Aggregator.tune == the reference's loop (deepcopy -> fuse -> run_evaluation per lattice vector), every metric of
    every weight vector within 1e-12.  NCE rows with a zero weight are undefined in the reference (-inf * 0 = NaN keys
    handed to sorted()) and excluded."""
    from fusion_amd.retrievers.hybrid import Aggregator, weight_grid
    z, systems, lists, labels, distr = load_tune(fname)
    grid = weight_grid(systems)                                   # np.arange lattice, np.float64 scalars, as hybrid.py:405-409
    assert np.array_equal(np.array([[w[s] for s in systems] for w in grid]), z["weights"])
    names = [str(x) for x in z["metric_names"]]
    got = Aggregator.tune(lists, norm, grid, labels, distr)
    assert all(list(g) == names for g in got)
    G = np.array([[float(g[k]) for k in names] for g in got])
    rows = np.all(z["weights"] != 0.0, axis=1) if norm == "normal-curve-equivalent" else slice(None)
    assert np.max(np.abs(G - z[f"metrics__{norm}"])[rows]) <= 1e-12


def test_tune_with_python_float_weights_is_the_float32_sweep(oracle):
    """A grid of Python floats fuses in float32 under NumPy 2 (weak scalars): tune() must follow -- checked against the
    oracle, which is pinned on both promotions."""
    from fusion_amd.retrievers.hybrid import Aggregator
    z, systems, lists, labels, distr = load_tune(TUNE_FILES[1])
    grid = [{s: float(w) for s, w in zip(systems, row)} for row in z["weights"][::7]]
    for norm in ("min-max", "percentile-rank"):
        got = Aggregator.tune(lists, norm, grid, labels, distr)
        exp = oracle.tune_lists(lists, norm, grid, labels, distr)
        for g, e in zip(got, exp):
            assert list(g) == list(e)
            assert all(abs(float(g[k]) - float(e[k])) <= 1e-12 for k in e)


@pytest.mark.parametrize("norm", ["min-max", "percentile-rank", "z-score"])
def test_fuse_with_float64_weights_matches_reference_order(oracle, norm):
    """Aggregator.fuse with np.float64 weights (float64 products and sums) against the oracle's pinned promotion rules:
    ids identical, scores bit for bit (min-max / percentile) or within the z-score tolerance."""
    from fusion_amd.retrievers.hybrid import Aggregator
    z, systems, lists, labels, distr = load_tune(TUNE_FILES[1])
    for row in z["weights"][[3, 57, 120, 200]]:
        for kinds in ("wide", "mixed"):
            w = {s: (np.float64(x) if (kinds == "wide" or i % 2 == 0) else float(x)) for i, (s, x) in enumerate(zip(systems, row))}
            got = Aggregator.fuse(lists, "nsf", norm, w, distr)
            exp = oracle.fuse_lists(lists, "nsf", norm, w, distr)
            for gq, eq in zip(got, exp):
                g_ids, e_ids = [x["corpus_id"] for x in gq], [x["corpus_id"] for x in eq]
                g_sc, e_sc = [float(x["score"]) for x in gq], [float(x["score"]) for x in eq]
                if norm == "z-score":
                    assert_ranked_close(g_ids, g_sc, e_ids, e_sc, 2e-6)
                else:
                    assert g_ids == e_ids and g_sc == e_sc


# ---- N4: score-distribution analysis (hybrid.py:363-402) --------------------------------------------------------------
@pytest.mark.parametrize("norm", ["none", "min-max", "z-score", "arctan", "percentile-rank"])
def test_analysis_outputs_match_reference(tmp_path, norm):
    import argparse
    import pandas as pd
    from fusion_amd.retrievers.hybrid import Aggregator, analyze_score_distributions
    z = np.load(os.path.join(GOLDEN, "analysis_seed31_S3_Q3_N120.npz"))
    systems, lists, Q = load_lists(z)
    corpus = {int(i): "" for i in z["corpus_ids"]}
    pos_pids = [[int(x) for x in str(s).split(",")] for s in z["pos_pids"]]
    out = str(tmp_path)
    if norm == "percentile-rank":     # the reference reads the 'raw' tables back from disk (hybrid.py:374)
        pd.DataFrame({s: z[f"table__none__1000__{s}"] for s in systems}).to_csv(os.path.join(out, "score_distributions_raw_indomain_10k.csv"), index=False)
    args = argparse.Namespace(normalization=norm, output_dir=out, eval_type="indomain", data_split="test")
    analyze_score_distributions(args, Aggregator._to_device(lists), corpus, pos_pids, table_sizes=(10, 1000))
    tol = {"none": 0.0, "min-max": 0.0, "percentile-rank": 0.0, "z-score": 2e-6, "arctan": 1e-6}[norm]
    # scores_{norm}_{eval}_{split}.csv: per system, every listed (query, document) transformed score, in list order
    df = pd.read_csv(os.path.join(out, f"scores_{norm}_indomain_test.csv"), float_precision="round_trip")
    for s in systems:
        got = df.loc[df["system"] == s, "score"].to_numpy(dtype=np.float64)
        exp = z[f"scores__{norm}__{s}"]
        assert got.shape == exp.shape
        assert np.max(np.abs(np.sort(got) - np.sort(exp))) <= max(tol, 1e-15)      # CSV text round trip of float64 is exact
        if tol == 0.0:
            assert np.array_equal(got, exp)                                         # same order: (query, list position)
    # the quantile tables
    for n_pts, tag in ((10, "0k"), (1000, "1k")):
        t = pd.read_csv(os.path.join(out, f"score_distributions_{norm}_indomain_{tag}.csv"), float_precision="round_trip")
        assert list(t.columns) == systems and len(t) == n_pts + 1
        for s in systems:
            assert np.max(np.abs(t[s].to_numpy() - z[f"table__{norm}__{n_pts}__{s}"])) <= tol + 1e-9
    # labelled scores of the positives and of the random.seed(42) negatives
    ldf = pd.read_csv(os.path.join(out, f"labeled_scores_{norm}_indomain_test.csv"), float_precision="round_trip")
    assert ldf["label"].tolist() == [str(x) for x in z[f"labeled_label__{norm}"]]
    for s in systems:
        assert np.max(np.abs(ldf[s].to_numpy(dtype=np.float64) - z[f"labeled__{norm}__{s}"])) <= tol


# ---- unsorted / duplicate-id host lists (hybrid.py:255-262: statistics over the VALUES) -------------------------------
def test_unsorted_and_duplicate_lists_match_reference():
    from fusion_amd.retrievers.hybrid import Aggregator
    g = json.load(open(os.path.join(GOLDEN, "unsorted_fuse.json")))
    for cname, case in g.items():
        for key, exp in case["out"].items():
            if key in ("rrf", "bcf"):
                got, tol = Aggregator.fuse(case["lists"], key), 0.0
            else:
                got = Aggregator.fuse(case["lists"], "nsf", key, case["weights"], {})
                tol = {"min-max": 0.0, "none": 0.0, "z-score": 2e-6, "arctan": 1e-6}[key]
            assert len(got) == len(exp)
            for gq, eq in zip(got, exp):
                g_ids, e_ids = [x["corpus_id"] for x in gq], [x["corpus_id"] for x in eq]
                g_sc, e_sc = [float(x["score"]) for x in gq], [x["score"] for x in eq]
                if tol == 0.0:
                    assert g_ids == e_ids and g_sc == e_sc, (cname, key)
                else:
                    assert_ranked_close(g_ids, g_sc, e_ids, e_sc, tol)


def test_none_passthrough_keeps_float64_scores(oracle):
    """'none' keeps the systems' Python floats (hybrid.py:280): BM25's float64 scores and host lists whose scores are not
    float32 values must not be rounded on the way."""
    from fusion_amd.retrievers.hybrid import Aggregator
    rng = np.random.default_rng(3)
    n = 200
    ids = rng.permutation(np.arange(1, n + 1))
    mk = lambda v: [{"corpus_id": int(i), "score": float(s)} for i, s in zip(ids, np.sort(v)[::-1])]
    lists = {"a": [mk(rng.gamma(2.0, 1.7, n)) for _ in range(3)], "b": [mk(rng.normal(0, 1, n)) for _ in range(3)]}
    w = {"a": 0.35, "b": 0.65}
    got = Aggregator.fuse(lists, "nsf", "none", w, {})
    exp = oracle.fuse_lists(lists, "nsf", "none", w, {})
    assert got == exp
    got = Aggregator.fuse(lists, "unknown-method")     # raw scores summed (hybrid.py:203-218)
    assert got == oracle.fuse_lists(lists, "unknown-method")


# ---- fp64 row sort: high-word passes + run repair + generic second launch (csrc/sort.hip) -----------------------------
@pytest.mark.parametrize("n", [300, 5000, 27942])
@pytest.mark.parametrize("mode", ["plain", "placed", "gathered"])
def test_sort_f64_equal_high_word_runs(ops, oracle, n, mode):
    """Keys that share their high 32 bits (sign, exponent, 20 mantissa bits) and differ only below: runs of 2..17 go
    through the in-place repair, longer dirty runs through the generic eight-pass launch, runs of EQUAL keys need nothing.
    All must give the oracle's stable order."""
    rng = np.random.default_rng(n)
    rows = 6
    k = rng.gamma(2.0, 3.0, (rows, n))
    ulp20 = 2.0 ** -20
    # row 0: pairs / triples; row 1: runs up to 17; row 2: runs of 18..40 (long, dirty); row 3: ONE run over the whole row;
    # row 4: a long run of equal keys (clean) next to short dirty runs; row 5: negative values, zeros of both signs, inf, nan
    def plant(r, lengths):
        pos = 0
        for L in lengths:
            if pos + L > n:
                break
            base = 1.0 + float(rng.integers(0, 1 << 19)) * ulp20 * 2        # exactly representable, low word zero
            k[r, pos: pos + L] = base + rng.permutation(L) * 2.0 ** -40       # same high word, distinct low words, shuffled
            pos += L + int(rng.integers(0, 3))
    plant(0, rng.integers(2, 4, n // 4))
    plant(1, rng.integers(2, 18, n // 12))
    plant(2, rng.integers(18, 41, n // 40))
    k[3] = 3.0 + rng.permutation(n) * 2.0 ** -45
    k[4, : n // 2] = 0.0
    plant(4, [2] * 10)
    k[4] = np.roll(k[4], n // 3)
    k[5] = -k[5]
    k[5, ::7] = 0.0; k[5, 3::11] = -0.0
    if n > 20:
        k[5, 5] = np.inf; k[5, 6] = -np.inf; k[5, 9] = np.nan; k[5, 10: 14] = k[5, 10] + np.arange(4) * 2.0 ** -50
    kp = ops.alloc_plane(rows, n, torch.float64, "cuda"); kp.copy_(torch.from_numpy(k))
    if mode == "plain":
        order, sk, rank = ops.sort_rows_desc(kp, want_rank=True)
        e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    else:
        init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)       # incoming sequence = a permutation
        lens = np.array([n, n - 1, max(1, n // 2), n, n, max(1, n - 7)], dtype=np.int32)
        e_order, e_sk, e_rank = oracle.sort_rows_desc(k, init_order=init, row_len=lens, want_rank=True)
        if mode == "gathered":
            order, sk, rank = ops.sort_rows_desc(kp, init_order=dev(init), row_len=dev(lens), want_rank=True)
        else:
            inv = np.full((rows, n), -1, dtype=np.int32)
            for r in range(rows):
                inv[r, init[r, : lens[r]]] = np.arange(lens[r], dtype=np.int32)
            order, sk, rank = ops.sort_rows_desc(kp, init_rank=dev(inv), row_len=dev(lens), want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)


# ---- z-score statistics as a by-product of the ranking sort -----------------------------------------------------------
@pytest.mark.parametrize("Q,N", [(3, 1), (4, 2), (5, 300), (3, 27942)])
def test_sort_row_stats_feed_zscore_fusion(ops, oracle, Q, N):
    """sort_rows_desc(stats_out=...) == torch.mean / torch.std of the float32 rows (hybrid.py:261-262), and the z-score fusion
    that uses them (ranked systems: one flat pass) == the oracle's fusion within the z-score tolerance (2e-6)."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    rng = np.random.default_rng(Q * 1000 + N)
    p32 = rng.normal(2.0, 3.0, (Q, N)).astype(np.float32)
    p64 = np.maximum(0.0, rng.gamma(0.5, 4.0, (Q, N)) - 2.0)                      # BM25-like float64 scores, many exact zeros
    if N >= 300:
        p32[1, :] = 0.125                                                           # a constant row: std == 0 -> zeros (hybrid.py:263)
    ids = np.arange(10, 10 + N)
    a = _rank_scores(dev(p32), ids, None)
    from fusion_amd.planes import RankedSystem
    sc64 = ops.alloc_plane(Q, N, torch.float64, "cuda"); sc64.copy_(torch.from_numpy(p64))
    zs = torch.empty((4, Q), device="cuda")
    od, sk, rk = ops.sort_rows_desc(sc64, want_rank=True, stats_out=zs)
    b = RankedSystem(scores=ops.f64_to_f32(sc64), order=od, rank=rk, lens=torch.full((Q,), N, dtype=torch.int32, device="cuda"), ids=ids,
                     sorted_scores=sk, full=True, scores64=sc64, score_sorted=True, stats4=zs)
    for rs, plane_np in ((a, p32), (b, p64.astype(np.float32))):
        e_mean, e_std = oracle.row_stats(plane_np, None, "z-score")
        g_mean, g_std = rs.zstats[0].cpu().numpy(), rs.zstats[1].cpu().numpy()
        if N > 1:
            assert np.max(np.abs(g_mean - e_mean) / np.maximum(1.0, np.abs(e_mean))) <= 2e-7
            assert np.max(np.abs(g_std - e_std) / np.maximum(1e-30, np.abs(e_std)), initial=0.0, where=e_std > 0) <= 2e-7
            assert np.array_equal(g_std == 0, e_std == 0)
        else:
            assert np.all(np.isnan(g_std)) and np.all(np.isnan(e_std))             # torch.std of one element
        np.testing.assert_array_equal(rs.stats4[2].cpu().numpy(), plane_np.min(axis=1))     # min | max: the two ends of the sorted list
        np.testing.assert_array_equal(rs.stats4[3].cpu().numpy(), plane_np.max(axis=1))
    w = {"x": 0.4, "y": 0.6}
    fused = Aggregator.fuse_device({"x": a, "y": b}, "nsf", "z-score", w, {})
    exp = oracle.fuse_nsf([p32, p64.astype(np.float32)], None, [0.4, 0.6], "z-score")
    got = np.empty((Q, N), dtype=np.float32)
    o, s_ = fused.order.cpu().numpy(), fused.scores.cpu().numpy()
    for q in range(Q):
        got[q, o[q]] = s_[q]
    if N > 1:
        assert np.max(np.abs(got - exp)) <= 2e-6
        slow = ops.fuse_nsf([a.scores, b.scores], None, [0.4, 0.6], "z-score").cpu().numpy()   # the reducing row kernel
        assert np.max(np.abs(got - slow)) <= 1e-6
    else:
        assert np.all(np.isnan(got)) and np.all(np.isnan(exp))


# ---- real-checkpoint loaders on the device: the padding-free HIP forward against the plain HF path -----------------------
def test_checkpoint_loaders_packed_path_matches_plain_path(tmp_path):
    """from_pretrained('colbert' / 'monobert') with 64-wide heads takes the padding-free forward (HIP attention / LayerNorm /
    GELU kernels) on the GPU: same token vectors / relevance scores as the plain HF path on the CPU (test_checkpoints_cpu.py
    pins that one on the colbert-ai / CrossEncoder formulas)."""
    from checkpoint_utils import write_colbert_checkpoint, write_monobert_checkpoint
    from fusion_amd import encoders
    from fusion_amd.retrievers.hybrid import Ranker
    queries = ["le juge peut , un bail ?", "la loi", "article de le code civil ( droit ) chat chien"]
    docs = ["le chat est un chien .", "article : le bail , la loi ; le code civil !", "droit"]
    write_colbert_checkpoint(str(tmp_path / "colbert"), heads=2, hidden=128)        # head_dim 64: PackedBertForward applies
    cpu = encoders.from_pretrained(str(tmp_path / "colbert"), "colbert", device="cpu")
    gpu = encoders.from_pretrained(str(tmp_path / "colbert"), "colbert", device="cuda")
    assert gpu._packed_forward(gpu.backbone) is not None
    assert torch.max(torch.abs(gpu.encode_queries(queries).float().cpu() - cpu.encode_queries(queries).float())).item() <= 2e-3
    (tg, og), (tc, oc) = gpu.encode_docs(docs), cpu.encode_docs(docs)
    assert og.cpu().tolist() == oc.tolist()
    assert torch.max(torch.abs(tg.float().cpu() - tc.float())).item() <= 2e-3
    write_monobert_checkpoint(str(tmp_path / "mono"), heads=2, hidden=128)
    pairs = [(q, d) for q in queries for d in docs]
    ce_cpu = encoders.from_pretrained(str(tmp_path / "mono"), "monobert", device="cpu")
    ce_gpu = encoders.from_pretrained(str(tmp_path / "mono"), "monobert", device="cuda")
    assert torch.max(torch.abs(ce_gpu.predict(pairs).cpu() - ce_cpu.predict(pairs))).item() <= 1e-5
    # Ranker.cross_encoder_search builds the loader itself when no model is injected (hybrid.py:151)
    cands = [{10: docs[0], 11: docs[1], 12: docs[2]}] * len(queries)
    got = Ranker.cross_encoder_search(queries, cands, str(tmp_path / "mono"))
    exp = ce_cpu.predict(pairs).view(len(queries), len(docs))
    for q, lst in enumerate(got):
        assert [x["corpus_id"] for x in lst] == [10 + int(i) for i in torch.argsort(exp[q], descending=True, stable=True)]


def test_pipeline_beyond_the_single_workgroup_row_limit(ops, oracle):
    """A 40,000-document corpus (rows longer than the fast sort path and than the register-resident fusion kernel): rank ->
    fuse -> order through the drop-in classes equals the oracle, ranked-list identity for rrf, bit-exact min-max."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    rng = np.random.default_rng(4)
    Q, N = 3, 40000
    a = rng.normal(0, 1, (Q, N)).astype(np.float32)
    b = np.maximum(0, rng.gamma(0.5, 4.0, (Q, N)) - 2).astype(np.float32)
    ids = np.arange(7, 7 + N)
    sa, sb = _rank_scores(dev(a), ids, None), _rank_scores(dev(b), ids, None)
    _, _, ra = oracle.sort_rows_desc(a, want_rank=True); oa, _ = oracle.sort_rows_desc(a)
    _, _, rb = oracle.sort_rows_desc(b, want_rank=True)
    for method, norm in (("rrf", None), ("nsf", "min-max")):
        fused = Aggregator.fuse_device({"a": sa, "b": sb}, method, norm, {"a": 0.3, "b": 0.7}, {})
        f = oracle.fuse_rank([ra, rb], np.full((2, Q), N, dtype=np.int32), "rrf") if method == "rrf" else oracle.fuse_nsf([a, b], None, [0.3, 0.7], "min-max")
        e_order, e_keys = oracle.sort_rows_desc(f, init_order=oa)
        np.testing.assert_array_equal(fused.order.cpu().numpy(), e_order)
        np.testing.assert_array_equal(fused.scores.cpu().numpy(), e_keys)


# ---- config 3 at full size: ColBERT MaxSim over the LLeQA-shaped corpus ---------------------------------------------------
def test_maxsim_full_size_properties(ops):
    """Q = 195, N = 27,942, L_d ~ clip(N(300, 120), 16, 512) with some empty documents, 64 x 128 fp16 unit query tokens:
    every score within [-Lq, Lq], empty documents score 0, and a random sample of (query, document) pairs equals the fp32
    torch evaluation of sum_i max_t <q_i, d_t> on the same fp16 inputs (tolerance 2e-3: fp32 accumulation order only)."""
    rng = np.random.default_rng(0)
    Q, N, Lq = 195, 27942, 64
    lens = np.clip(rng.normal(300, 120, N), 16, 512).astype(np.int64)
    empty = rng.choice(N, size=50, replace=False)
    lens[empty] = 0
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    g = torch.Generator(device="cuda").manual_seed(2)
    Dtok = torch.nn.functional.normalize(torch.randn((int(off[-1]), 128), generator=g, device="cuda"), dim=-1).half()
    Qtok = torch.nn.functional.normalize(torch.randn((Q, Lq, 128), generator=g, device="cuda"), dim=-1).half()
    S = ops.maxsim(Qtok, Dtok, torch.from_numpy(off).cuda(), max_doc_len=512)
    assert S.shape == (Q, N)
    Sh = S.cpu().numpy()
    assert np.isfinite(Sh).all() and np.abs(Sh).max() <= Lq * 1.001
    assert np.all(Sh[:, empty] == 0.0)
    qs, ds = rng.integers(0, Q, 300), rng.integers(0, N, 300)
    for q, d in zip(qs, ds):
        if lens[d] == 0:
            continue
        ref = (Qtok[q].float() @ Dtok[off[d]: off[d + 1]].float().T).max(dim=1).values.sum().item()
        assert abs(Sh[q, d] - ref) <= 2e-3, (q, d, Sh[q, d], ref)
    # permuting the documents permutes the columns (tiles are aligned to document starts: no cross-document leakage)
    perm = rng.permutation(N)[:2000]
    off2 = np.zeros(len(perm) + 1, dtype=np.int64); off2[1:] = np.cumsum(lens[perm])
    D2 = torch.cat([Dtok[off[d]: off[d + 1]] for d in perm])
    S2 = ops.maxsim(Qtok[:16].contiguous(), D2, torch.from_numpy(off2).cuda(), max_doc_len=512).cpu().numpy()
    np.testing.assert_array_equal(S2, Sh[:16][:, perm])


def test_sharded_search_first_chunk_streams_after_an_exact_head(ops, oracle):
    """ShardedDenseIndex with its default 229,376-column chunks: the first chunk is an exact top-k of 28,672 columns + streaming
    pieces; the result equals the oracle's top-k of the full score matrix (scores and ids, ties by ascending id)."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(1)
    N, d, Q, k = 300000, 32, 5, 1000
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Dn[1000:1040] = Dn[7]                                  # exact ties across the head / piece boundaries
    Dn[50000:50040] = Dn[7]
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    idx = ShardedDenseIndex(Dn, id_base=123)
    s, i = idx.local_topk(Qn, k)
    S = ops.dot_scores(Qn, Dn).cpu().numpy()
    es, ei = oracle.topk_rows(S, k, id_base=123)
    np.testing.assert_array_equal(s.cpu().numpy(), es)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)


# ---- partial lists: validity as a bitmap ---------------------------------------------------------------------------------
@pytest.mark.parametrize("Q,N", [(3, 70), (4, 1000), (2, 27942), (2, 40000)])
@pytest.mark.parametrize("norm", ["min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"])
def test_fuse_nsf_validity_bitmap_equals_rank_planes(ops, oracle, Q, N, norm):
    """The nsf fusion reading the partial systems' validity from bitmaps (fz_rank_to_bitmap) == reading their rank planes == the
    oracle (bit for bit between the two device forms; oracle within the norm's tolerance)."""
    rng = np.random.default_rng(Q * 31 + N)
    S = 3
    planes = [rng.normal(s, 1.0 + s, (Q, N)).astype(np.float32) for s in range(S)]
    ranks = []
    for s in range(S):
        _, _, r = oracle.sort_rows_desc(planes[s], want_rank=True)
        if s >= 1:
            keep = int(N * (0.6 if s == 1 else 0.25)) or 1
            r = np.where(r < keep, r, -1).astype(np.int32)
        ranks.append(r)
    P = 101
    distr = [np.quantile(p, np.linspace(0, 1, P)).astype(np.float32) for p in planes]
    w = [0.2, 0.5, 0.3]
    tabled = norm in ("percentile-rank", "normal-curve-equivalent")
    dp = [ops.alloc_plane(Q, N, torch.float32, "cuda") for _ in range(S)]
    for t, p in zip(dp, planes):
        t.copy_(torch.from_numpy(p))
    dr = []
    for r in ranks:
        t = ops.alloc_plane(Q, N, torch.int32, "cuda"); t.copy_(torch.from_numpy(r)); dr.append(t)
    dd = [dev(d) for d in distr] if tabled else None
    a = ops.fuse_nsf(dp, [None, dr[1], dr[2]], w, norm, dd).cpu().numpy()
    bits = [None, ops.rank_to_bitmap(dr[1]), ops.rank_to_bitmap(dr[2])]
    for b, r in zip(bits[1:], ranks[1:]):
        got = np.unpackbits(b.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :N].astype(bool)
        assert np.array_equal(got, r >= 0)
    b = ops.fuse_nsf(dp, [None, dr[1], dr[2]], w, norm, dd, valid_bits=bits).cpu().numpy()
    np.testing.assert_array_equal(a, b)
    if N <= 32768 or norm not in ("min-max", "z-score"):    # (the two-pass form for longer rows takes its statistics over the rank planes)
        c = ops.fuse_nsf(dp, None, w, norm, dd, valid_bits=bits).cpu().numpy()     # no rank planes at all
        np.testing.assert_array_equal(a, c)
    e = oracle.fuse_nsf(planes, [None, ranks[1], ranks[2]], w, norm, distr if tabled else None)
    tol = {"min-max": 0.0, "percentile-rank": 0.0, "z-score": 2e-6, "arctan": 1e-6, "normal-curve-equivalent": 1e-4}[norm]
    fin = np.isfinite(e)
    assert np.array_equal(np.isfinite(a), fin) and np.max(np.abs(a[fin] - e[fin]), initial=0.0) <= tol


# ---- C1 behind the C ABI: fz_topk_allgather over a real RCCL communicator (one rank: all a one-GPU box can host) ----------
def test_topk_allgather_c_abi_with_a_one_rank_rccl_communicator(ops, oracle):
    import ctypes as C
    import glob
    from fusion_amd import _lib
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so", "/opt/rocm/lib/librccl.so"]
    rccl = None
    for c in cands:
        try:
            rccl = C.CDLL(c, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None, "no RCCL library on this box"

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        rng = np.random.default_rng(9)
        Q, k = 7, 50
        sc = -np.sort(-rng.normal(0, 1, (Q, k)).astype(np.float32), axis=1)
        ids = np.sort(rng.choice(10**6, (Q, k), replace=False), axis=1).astype(np.int64)
        ls, li = dev(sc), dev(ids)
        os_, oi = torch.empty_like(ls), torch.empty_like(li)
        L = _lib.lib()
        wsb = int(L.fz_topk_allgather_workspace_bytes(1, Q, k))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.fz_topk_allgather(C.c_void_p(ls.data_ptr()), C.c_void_p(li.data_ptr()), Q, k, comm, 1, C.c_void_p(os_.data_ptr()),
                                 C.c_void_p(oi.data_ptr()), C.c_void_p(ws.data_ptr()), wsb, st)
        assert rc == 0, L.fz_strerror(rc)
        torch.cuda.synchronize()
        es, ei = oracle.topk_merge(sc[None], ids[None])
        np.testing.assert_array_equal(os_.cpu().numpy(), es)
        np.testing.assert_array_equal(oi.cpu().numpy(), ei)
        assert L.fz_topk_allgather(None, None, 1, 1, comm, 1, None, None, None, 0, st) == _lib.FZ_ERR_ARG
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


# ---- score GEMM: several tiles per (persistent) workgroup ------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Q,N,d", [(300, 70001, 100), (1024, 40000, 64), (129, 140000, 36)])
def test_dot_scores_tile_stream_across_tiles(ops, Q, N, d):
    """More tiles than resident workgroups: every workgroup walks several tiles in one k-tile stream, the operand loads of a
    tile's first k-tiles are issued while the previous tile is still being multiplied, and the last partial round runs as half
    tiles.  d = 100 / 36 end in a partial k-tile (zero-filled in LDS), d = 36 also in a k-tile that lies entirely past d.
    Against float64 torch on the same operands; 2e-6 is the scoring tolerance of DESIGN section 4."""
    g = torch.Generator(device="cuda").manual_seed(Q + N + d)
    A = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    B = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    S = ops.dot_scores(A, B)
    for r0 in range(0, Q, 128):
        ref = A[r0:r0 + 128].double() @ B.double().t()
        torch.testing.assert_close(S[r0:r0 + 128].double(), ref, rtol=0, atol=2e-6)


# ---- N1: metrics of the sweep on the device -------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Q", [1, 7, 195, 700])
def test_tune_metrics_kernel_equals_host_evaluation(ops, Q):
    """fz_tune_metrics_f64 against metrics_from_gold_ranks (itself checked against the Metrics class and, through tune(), against
    the reference's loop): random gold ranks incl. rank 0, ranks around every cut-off, unlisted golds, padded slots, queries
    with no gold at all and with repeated gold ids (len(ground_truths) > number of distinct golds).  Per-query values are the
    same float64 operations in the same order; the means are exactly accumulated on both sides -> equal to the last bit."""
    from fusion_amd.utils.metrics import MAP_KS, MRR_KS, NDCG_KS, RECALL_KS, gold_rank_tables, metrics_from_gold_ranks
    rng = np.random.default_rng(Q)
    W, N, G = 13, 3000, 8
    pos = rng.permutation(N).astype(np.int32)[None, :].repeat(Q, 0)
    pos[:, ::11] = -1                                            # documents in no list
    gold = np.full((Q, G), -1, dtype=np.int32)
    n_gold = np.zeros(Q, dtype=np.int64)
    for q in range(Q):
        k = int(rng.integers(0, G + 1))
        gold[q, :k] = rng.choice(N, size=k, replace=False)
        n_gold[q] = k + int(rng.integers(0, 3)) * (k > 0)        # repeated ids in the label list
    special = np.array([0, 1, 4, 5, 9, 10, 19, 20, 99, 100, 199, 200, 499, 500, 999, 1000, 1001, 2500])
    ranks = np.where(rng.random((W, Q, G)) < 0.5, rng.choice(special, size=(W, Q, G)), rng.integers(0, N, size=(W, Q, G))).astype(np.int32)
    listed = (gold >= 0) & (np.take_along_axis(pos, np.maximum(gold, 0).astype(np.int64), axis=1) >= 0)
    host_ranks = np.where(listed[None], ranks.astype(np.int64), np.iinfo(np.int64).max)
    exp = metrics_from_gold_ranks(host_ranks, n_gold, np.full(Q, N))
    table, idcg, names = gold_rank_tables(n_gold)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    got = ops.tune_metrics(dev(ranks), dev(gold), ops.as_plane(dev(pos)), dev(n_gold.astype(np.int32)), dev(idcg), dev(table),
                           dict(recall=RECALL_KS, map=MAP_KS, mrr=MRR_KS, ndcg=NDCG_KS)).cpu().numpy()
    assert list(exp[0]) == names
    E = np.array([[e[n] for n in names] for e in exp])
    assert got.shape == E.shape
    assert np.array_equal(got, E), float(np.max(np.abs(got - E)))


# ---- streaming top-k: several chunks per fold ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("k,cap,head", [(100, 2000, 4096), (1000, 7168, 28672), (7, 256, 512)])
def test_topk_stream_equals_oracle_topk_of_the_whole_matrix(ops, oracle, k, cap, head):
    """ops.TopkStream fed with chunks of irregular widths (the windows between folds are the stream's own business) == the oracle's
    top-k of the full score matrix: scores and ids, ties by ascending id -- incl. planted ties inside a chunk, across chunks and
    across folds, a row of identical scores and a NaN."""
    rng = np.random.default_rng(k)
    rows, n = 5, 150_000
    S = np.round(rng.normal(0, 1, (rows, n)), 3).astype(np.float32)        # 3 decimals: ties everywhere
    S[1, :] = 0.25
    S[2, 77_777] = np.nan
    S[3, 60_000:60_050] = S[3].max()
    Sd = ops.as_plane(torch.from_numpy(S).cuda())
    bs, bi = ops.topk_rows(Sd[:, :head], k, id_base=1_000)
    st = ops.TopkStream(bs, bi, seen=head, cap=cap)
    lo = head
    for w in [1, 63, 4096, 5000, 64, 30_001, 17, 50_000, 10**9]:
        hi = min(n, lo + w)
        if hi > lo:
            st.feed(Sd[:, lo:hi], 1_000 + lo)
        lo = hi
    gs, gi, flag = st.result()
    assert int(flag.item()) == 0
    es, ei = oracle.topk_rows(S, k, id_base=1_000)
    np.testing.assert_array_equal(gs.cpu().numpy(), es)
    np.testing.assert_array_equal(gi.cpu().numpy(), ei)


@pytest.mark.gpu
def test_topk_stream_flags_candidate_overflow(ops):
    """Ascending scores: every new document beats the threshold; a window holds more than `cap` of them -> the flag the caller
    checks before trusting the lists (ShardedDenseIndex then redoes the search on the exact path)."""
    rows, n, k = 3, 40_000, 50
    S = ops.as_plane((torch.arange(rows * n, device="cuda", dtype=torch.float32).reshape(rows, n) / 7.0).contiguous())
    bs, bi = ops.topk_rows(S[:, :4096], k)
    st = ops.TopkStream(bs, bi, seen=4096, cap=500, exact_on_overflow=False)
    st.feed(S[:, 4096:], 4096)
    assert int(st.result()[2].item()) == 1
    # default: every overflowed window is redone exactly as it is folded -- the lists are the exact top-k, the flag is clear again
    st = ops.TopkStream(bs, bi, seen=4096, cap=500)
    st.feed(S[:, 4096:], 4096, hold=True)          # the stream keeps views of S until each window is folded: S stays as it is
    rs, ri, flag = st.result()
    es, ei = ops.topk_rows(S, k)
    assert st.windows_redone >= 1 and int(flag.item()) == 0
    assert torch.equal(rs, es) and torch.equal(ri, ei)
    # ONE score buffer reused for every chunk (what a chunked caller does): the stream must not rank overwritten data -- scores fed
    # without hold are never redone, the flag stays set, and the caller's exact fall-back gives the right lists
    st = ops.TopkStream(bs, bi, seen=4096, cap=500)
    buf = ops.alloc_plane(rows, 4096, torch.float32, "cuda")
    for lo in range(4096, n, 4096):
        hi = min(n, lo + 4096)
        buf[:, : hi - lo].copy_(S[:, lo:hi])
        st.feed(buf[:, : hi - lo], lo)
        buf.fill_(float("nan"))                     # the caller's buffer is the caller's again
    rs, ri, flag = st.result()
    assert int(flag.item()) == 1 and st.windows_redone == 0
    assert not torch.isnan(rs).any()


@pytest.mark.gpu
def test_topk_stream_overflow_latch(ops):
    """(a) a HELD window that overflows after a clean unheld one is repaired exactly and the flag is clear again; (b) once an
    UNHELD window has overflowed the search is the caller's to redo: the flag stays set, later held windows are neither read nor redone."""
    rows, n, k, head = 3, 60_000, 50, 4096
    # descending head and middle (nothing beats the thresholds), then an ascending tail that overflows every window
    flat = torch.arange(rows * n, device="cuda", dtype=torch.float32).reshape(rows, n)
    S = -flat.clone()
    S[:, 30_000:] = flat[:, 30_000:]
    S = ops.as_plane(S.contiguous())
    bs, bi = ops.topk_rows(S[:, :head], k)
    # (a) clean unheld window, then held windows that overflow
    st = ops.TopkStream(bs, bi, seen=head, cap=500)
    st.feed(S[:, head:30_000], head)
    st.fold()
    assert not st.unrepairable and st.windows_redone == 0
    st.feed(S[:, 30_000:], 30_000, hold=True)
    rs, ri, flag = st.result()
    es, ei = ops.topk_rows(S, k)
    assert st.windows_redone >= 1 and int(flag.item()) == 0 and not st.unrepairable
    assert torch.equal(rs, es) and torch.equal(ri, ei)
    # (b) the overflow happens in an unheld window: latched; the held windows behind it cost no redo
    st = ops.TopkStream(bs, bi, seen=head, cap=500)
    st.feed(S[:, head:45_000], head)
    st.fold()
    assert st.unrepairable
    st.feed(S[:, 45_000:], 45_000, hold=True)
    _, _, flag = st.result()
    assert int(flag.item()) == 1 and st.windows_redone == 0


# ---- percentile-rank / NCE: the windowed nearest-entry look-up against the plain first-argmin ------------------------------------------
def _nearest_first_argmin(tab, x):
    """hybrid.py:272-275 in NumPy: index of the FIRST minimum of |tab - x| computed in float32 (NaN distances: argmin's rule)."""
    d = np.abs(tab[None, :].astype(np.float32) - x.reshape(-1, 1).astype(np.float32))
    return np.argmin(d, axis=1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mass_points", "huge_offset", "tiny_table", "wide_magnitudes", "exact_hits_and_midpoints", "two_values"])
def test_percentile_rank_window_lookup_edge_cases(ops, case):
    """The table kernel's look-up (bucket guess + window of eight distinct values, exact search when the window cannot prove its
    answer) must give torch's first-argmin index for every score: repeated quantiles (mass points), scores far outside the table
    (all distances round to one float), tables with fewer than eight distinct values, magnitudes where float32 spacing exceeds the
    table's, scores on entries and on midpoints between them (ties -> lower index), +-inf and NaN."""
    rng = np.random.default_rng(len(case))
    P = 1001
    if case == "mass_points":
        tab = np.sort(np.concatenate([np.zeros(400), rng.normal(5, 2, 300).clip(0), np.full(200, 7.5), rng.normal(9, 1, 101)])).astype(np.float32)
        x = np.concatenate([rng.normal(5, 4, 5000), np.zeros(50), np.full(50, 7.5), [1e-30, -1e-30, 7.4999995, 7.5000005]])
    elif case == "huge_offset":
        tab = np.sort(rng.normal(0, 1, P)).astype(np.float32)
        x = np.concatenate([rng.normal(0, 1, 3000), [1e9, -1e9, 3e38, -3e38, 1e6, -1e6, 50.0, -50.0, np.inf, -np.inf, np.nan]])
    elif case == "tiny_table":
        tab = np.array([-1.0, 0.0, 0.0, 2.0, 2.5], dtype=np.float32)
        x = np.concatenate([rng.normal(0.5, 2, 2000), [-1, 0, 2, 2.5, 1.0, -0.5, 2.25]])
    elif case == "wide_magnitudes":
        tab = np.sort(1e6 + rng.integers(0, 4000, P) * 0.0625).astype(np.float32)        # float32 spacing at 1e6 is 0.0625: neighbours collide
        x = np.concatenate([1e6 + rng.uniform(-10, 260, 4000), [0.0, 2e6]])
    elif case == "exact_hits_and_midpoints":
        tab = np.sort(rng.integers(-500, 500, P) * 0.25).astype(np.float32)
        mids = (tab[:-1].astype(np.float64) + tab[1:]) / 2
        x = np.concatenate([tab, mids, rng.uniform(-130, 130, 2000)])
    else:
        tab = np.sort(np.concatenate([np.full(500, -3.0), np.full(501, 4.0)])).astype(np.float32)
        x = np.concatenate([rng.normal(0.5, 3, 3000), [0.5, 0.49999997, 0.50000006]])
    x = x.astype(np.float32)
    Q = 3
    n = len(x)
    S = np.stack([x, x[::-1], np.roll(x, 7)]).astype(np.float32)
    with np.errstate(invalid="ignore"):
        k = np.stack([_nearest_first_argmin(tab, r) for r in S])
    exp = (k.astype(np.float32) / np.float32(len(tab))) * np.float32(1.0) + np.float32(0.0)
    got = ops.fuse_nsf([ops.as_plane(torch.from_numpy(S).cuda())], None, [1.0], "percentile-rank", [torch.from_numpy(tab).cuda()]).cpu().numpy()
    np.testing.assert_array_equal(got, exp)


@pytest.mark.gpu
def test_row_stats_accepts_a_rank_tensor_with_its_own_row_stride(ops, oracle):
    """ops.row_stats(scores plane, rank as a plain contiguous tensor): the odd stride is copied out, as fuse_nsf does."""
    rng = np.random.default_rng(3)
    Q, N = 3, 1001
    p = rng.normal(0, 1, (Q, N)).astype(np.float32)
    r = np.where(rng.random((Q, N)) < 0.6, 1, -1).astype(np.int32)
    mean, std = ops.row_stats(ops.as_plane(torch.from_numpy(p).cuda()), torch.from_numpy(r).cuda(), "z-score")
    for q in range(Q):
        v = p[q][r[q] >= 0].astype(np.float64)
        assert abs(float(mean[q]) - v.mean()) <= 1e-6 and abs(float(std[q]) - v.std(ddof=1)) <= 1e-6


# ---- sharded search: the threshold filter as the GEMM's epilogue ---------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Q,N,d,k", [(5, 120_000, 32, 1000), (130, 60_000, 64, 100), (3, 40_000, 36, 7)])
def test_fused_gemm_filter_search_equals_the_two_pass_search_and_the_oracle(ops, oracle, Q, N, d, k):
    """ShardedDenseIndex.local_topk with FUSED (scores never materialised after the head; candidates in arrival order; ties put in
    id order by the fold) == the same search with the separate filter pass == the oracle's top-k of the full score matrix -- incl.
    duplicated documents (exact score ties inside the head, across the head boundary, across folds, and more than 64 of them
    where the k-th place is NOT at stake).  d = 36 takes the ragged-d GEMM instantiation, Q = 130 a partial query block."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(Q + N)
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    Dn[1000:1010] = Dn[7]
    Dn[30_000:30_040] = Dn[7]
    Dn[N - 70:N] = Dn[123]                       # 71 copies of one document far down most lists
    S = ops.dot_scores(Qn, Dn).cpu().numpy()
    es, ei = oracle.topk_rows(S, k, id_base=10**10)
    for fused in (True, False):
        idx = ShardedDenseIndex(Dn, id_base=10**10)
        idx.FUSED = fused
        idx.CHUNK = 50_000
        s, i = idx.local_topk(Qn, k)
        np.testing.assert_array_equal(s.cpu().numpy(), es)
        np.testing.assert_array_equal(i.cpu().numpy(), ei)


@pytest.mark.gpu
def test_fused_gemm_filter_flags_a_tie_run_it_cannot_order(ops, oracle):
    """More equal scores at the k-th place than the fold looks at (64 past k): the overflow flag, i.e. the exact path, and with
    it still the oracle's answer."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(5)
    N, d, k = 30_000, 32, 50
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = Dn[20_000:20_002].clone()               # the queries ARE documents: their copies score exactly 1.0
    Dn[9_000:9_200] = Dn[20_000]                 # 200 copies behind the head: all tie at the top of query 0's list
    idx = ShardedDenseIndex(Dn, id_base=0)
    st = ops.TopkStream(*ops.topk_rows(ops.dot_scores(Qn, Dn[:8192]), k), seen=8192, cap=7168, exact_on_overflow=False)
    st.feed_gemm(Qn, Dn[8192:], 8192)
    assert int(st.result()[2].item()) == 1
    s, i = idx.local_topk(Qn, k)
    assert idx.last_overflow >= 1
    es, ei = oracle.topk_rows(ops.dot_scores(Qn, Dn).cpu().numpy(), k)
    np.testing.assert_array_equal(s.cpu().numpy(), es)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
