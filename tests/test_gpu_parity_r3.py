"""Round-3 GPU parity: statistics by-products of the ranking sort (mean | std | min | max, also over the listed prefix of a truncated
ranking), the per-system-statistics flat fusion, fz_zero_unlisted_f32 and the weight sweep driven through the C ABI with a partial
list, the NumPy-1 promotion switch.  Everything goes through the C ABI; the oracle is the checker."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import load_lists

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def plane_of(ops, a):
    p = ops.alloc_plane(a.shape[0], a.shape[1], torch.from_numpy(a[:0]).dtype, "cuda")
    p.copy_(torch.from_numpy(a))
    return p


# ---- the sort's statistics by-product --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Q,N,k", [(3, 1, 1), (4, 2, 1), (5, 300, 37), (7, 5000, 4999), (3, 27942, 16765), (3, 35840, 1000)])
def test_sort_prefix_statistics_equal_row_stats_over_the_listed_documents(ops, oracle, Q, N, k):
    """fz_sort_rows_desc(row_stats, stats_len): mean | unbiased std | min | max over the first k entries of the sorted list == the
    oracle's statistics over exactly those documents (hybrid.py:254-262 on a list cut to its top-k)."""
    rng = np.random.default_rng(N + k)
    x = rng.normal(1.0, 2.0, (Q, N)).astype(np.float32)
    if N >= 300:
        x[1, :] = -0.75                                              # constant list: std 0, min == max
        x[2, : N // 2] = x[2, N // 2: 2 * (N // 2)]                  # ties across the cut
    lens = torch.full((Q,), k, dtype=torch.int32, device="cuda")
    st = torch.full((4, Q), 7.0, device="cuda")
    od, sk, rk = ops.sort_rows_desc(plane_of(ops, x), want_rank=True, stats_out=st, stats_len=lens)
    rk = rk.cpu().numpy()
    listed = np.where(rk < k, rk, -1).astype(np.int32)
    e_mean, e_std = oracle.row_stats(x, listed, "z-score")
    e_min, e_max = oracle.row_stats(x, listed, "min-max")
    g = st.cpu().numpy()
    np.testing.assert_array_equal(g[2], e_min)
    np.testing.assert_array_equal(g[3], e_max)
    if k > 1:
        assert np.max(np.abs(g[0] - e_mean) / np.maximum(1.0, np.abs(e_mean))) <= 2e-7
        assert np.max(np.abs(g[1] - e_std) / np.maximum(1e-30, np.abs(e_std)), initial=0.0, where=e_std > 0) <= 2e-7
        assert np.array_equal(g[1] == 0, e_std == 0)
    else:
        assert np.all(np.isnan(g[1])) and np.allclose(g[0], e_mean, rtol=2e-7)   # torch.std of one element
    # the whole-list form of the same call
    st2 = torch.empty((4, Q), device="cuda")
    ops.sort_rows_desc(plane_of(ops, x), stats_out=st2)
    np.testing.assert_array_equal(st2[2].cpu().numpy(), x.min(axis=1))
    np.testing.assert_array_equal(st2[3].cpu().numpy(), x.max(axis=1))


def test_sort_statistics_propagate_nan_and_reject_what_they_cannot_do(ops):
    from fusion_amd._lib import FusionHipError
    x = np.arange(12, dtype=np.float32).reshape(2, 6)
    x[1, 3] = np.nan
    st = torch.empty((4, 2), device="cuda")
    ops.sort_rows_desc(plane_of(ops, x), stats_out=st)
    g = st.cpu().numpy()
    assert g[2, 0] == 0.0 and g[3, 0] == 5.0 and np.isnan(g[:, 1]).all()         # torch.min / max / mean / std propagate a NaN
    with pytest.raises(FusionHipError):                                          # prefix statistics exist for fp32 keys only
        ops.sort_rows_desc(plane_of(ops, x.astype(np.float64)), stats_out=st, stats_len=torch.full((2,), 3, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("norm", ["min-max", "z-score"])
def test_truncated_ranking_fuses_like_the_oracle_on_the_cut_lists(ops, oracle, norm):
    """Ranker-style systems, one of them cut to its top-k (statistics over the listed prefix from the ranking sort), fused in one flat
    pass with per-system statistics == the oracle's fusion of the same planes with the cut expressed as a rank plane."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    rng = np.random.default_rng(5)
    Q, N, k = 6, 27942, 16765
    a = rng.normal(0.0, 1.0, (Q, N)).astype(np.float32)
    b = (20.0 + 4.0 * rng.normal(0.0, 1.0, (Q, N))).astype(np.float32)
    ids = np.arange(N)
    A, B = _rank_scores(plane_of(ops, a), ids, None), _rank_scores(plane_of(ops, b), ids, k)
    assert A.stats4 is not None and B.stats4 is not None and not B.full
    fused = Aggregator.fuse_device({"a": A, "b": B}, "nsf", norm, {"a": 0.3, "b": 0.7}, {})
    rb = np.where(B.rank.cpu().numpy() >= 0, B.rank.cpu().numpy(), -1).astype(np.int32)
    exp = oracle.fuse_nsf([a, b], [None, rb], [0.3, 0.7], norm)
    got = np.full((Q, N), np.nan, dtype=np.float32)
    o, s_, ln = fused.order.cpu().numpy(), fused.scores.cpu().numpy(), fused.lens.cpu().numpy()
    assert np.all(ln == N)
    for q in range(Q):
        got[q, o[q]] = s_[q]
    tol = 0.0 if norm == "min-max" else 2e-6
    assert np.max(np.abs(got - exp)) <= tol
    # the float64 fusion (np.float64 weights, the tuning grid's kind) normalises with the SAME statistics: identical float32 planes
    T = Aggregator._normalised_planes([A, B], norm, None, [A.stats(norm), B.stats(norm)])
    one = ops.fuse_nsf([A.scores, B.scores], [None, B.rank], [1.0, 0.0], norm, stats=[A.stats(norm), B.stats(norm)], valid_bits=[None, B.valid_bits()])
    np.testing.assert_array_equal(T[0].cpu().numpy(), one.cpu().numpy())


def test_tune_and_fuse_agree_bit_for_bit_on_ranked_systems(ops):
    """Aggregator.tune must rank with the statistics Aggregator.fuse uses -- z-score on Ranker-made systems (statistics from
    the ranking sort), float32 sweep (Python-float weights) against one fuse_device per weight vector: identical ranks (metrics equal up to the last ulp of the mean over queries)."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    rng = np.random.default_rng(11)
    Q, N = 9, 3000
    base = rng.normal(0, 1, (Q, N))
    planes = [(base + 0.3 * rng.normal(0, 1, (Q, N))).astype(np.float32) for _ in range(3)]
    planes[1] = np.round(planes[1], 1)                                            # many near and exact ties
    ids = np.arange(N) + 100
    systems = {n: _rank_scores(plane_of(ops, p), ids, None if n != "c" else 1800) for n, p in zip("abc", planes)}
    labels = [[int(ids[j]) for j in rng.choice(N, size=3, replace=False)] for _ in range(Q)]
    grid = [dict(a=x, b=y, c=round(1.0 - x - y, 2)) for x in (0.0, 0.25, 0.5) for y in (0.0, 0.25, 0.5)]
    for norm in ("z-score", "min-max"):
        got = Aggregator.tune(systems, norm, grid, labels, {})
        exp = Aggregator._tune_by_fusing(systems, norm, grid, labels, {})
        for g, e in zip(got, exp):
            assert list(g) == list(e) and all(abs(float(g[m]) - float(e[m])) <= 1e-14 for m in e), (norm, g, e)   # any swapped pair moves a metric by >= 1e-5


# ---- the sweep through the C ABI alone, with a partial list (the -inf / 0 contract) ---------------------------------------------
def test_c_abi_sweep_recipe_with_a_partial_list(ops, oracle):
    """INTEGRATION.md section 5 as a non-Python host would run it: fz_fuse_nsf_f32 (S = 1, weight 1) leaves -inf where the system does not
    list a document; fz_zero_unlisted_f32 makes it the 0 the sweep multiplies; fz_gold_ranks_f64w then gives the ranks of the oracle's
    fused lists."""
    from fusion_amd import _lib
    L = _lib.lib()
    z = np.load(os.path.join(GOLDEN, "tune_seed21_S3_Q4_N257_colbert_first.npz"), allow_pickle=False)
    systems, lists, Q = load_lists(z)
    labels = [[int(x) for x in str(s).split(",")] for s in z["labels"]]
    _, ids, planes, ranks, orders, lens, _ = oracle.lists_to_planes(lists)
    S, N = len(systems), planes[0].shape[1]
    assert any((r < 0).any() for r in ranks)                                    # the fixture holds a partial ColBERT list
    ld = ops.round_up(N, 64)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    dplanes, dranks, dorders, T = [], [], [], []
    for s in range(S):
        dplanes.append(plane_of(ops, planes[s])); dranks.append(plane_of(ops, ranks[s])); dorders.append(plane_of(ops, orders[s]))
        t = ops.alloc_plane(Q, N, torch.float32, "cuda")
        one = (C.c_void_p * 1)(dplanes[s].data_ptr()); oner = (C.c_void_p * 1)(dranks[s].data_ptr()); w1 = (C.c_double * 1)(1.0)
        assert L.fz_fuse_nsf_f32(one, oner, w1, 1, Q, N, ld, 1, None, None, None, 0, P(t), st) == 0       # min-max
        if (ranks[s] < 0).any():
            assert torch.isinf(t[dranks[s] < 0]).all()                        # the documented -inf
        assert L.fz_zero_unlisted_f32(P(t), P(dranks[s]), Q, N, ld, st) == 0
        assert (t[dranks[s] < 0] == 0).all()
        T.append(t)
    dlens = dev(lens.astype(np.int32))
    ins = torch.full((Q, ld), -1, dtype=torch.int32, device="cuda"); U = torch.zeros(Q, dtype=torch.int32, device="cuda")
    pos = torch.full((Q, ld), -1, dtype=torch.int32, device="cuda")
    assert L.fz_insertion_order((C.c_void_p * S)(*[o.data_ptr() for o in dorders]), P(dlens), S, Q, N, ld, P(ins), P(U), P(pos), None, 0, st) == 0
    Uh, insh, posh = U.cpu().numpy(), ins.cpu().numpy(), pos.cpu().numpy()
    for q in range(Q):
        assert np.array_equal(posh[q, insh[q, : Uh[q]]], np.arange(Uh[q])) and (posh[q, :N] >= 0).sum() == Uh[q]      # pos is the inverse of ins
    W = z["weights"][::9]
    G = L.fz_tune_max_gold()
    id2pos = {int(c): j for j, c in enumerate(ids)}
    gold = np.full((Q, G), -1, dtype=np.int32)
    for q, gl in enumerate(labels):
        u = [id2pos.get(g, -1) for g in dict.fromkeys(gl)]
        gold[q, : len(u)] = u
    out = torch.zeros((len(W), Q, G), dtype=torch.int32, device="cuda")
    assert L.fz_gold_ranks_f64w((C.c_void_p * S)(*[t.data_ptr() for t in T]), P(pos), P(dev(W.astype(np.float64))), P(dev(gold)), S, len(W), Q, N, ld,
                                P(out), st) == 0
    out = out.cpu().numpy()
    for wi, wv in enumerate(W):
        fl = oracle.fuse_lists(lists, "nsf", "min-max", {s: np.float64(x) for s, x in zip(systems, wv)}, {})
        for q in range(Q):
            rank_of = {x["corpus_id"]: r for r, x in enumerate(fl[q])}
            for g in range(G):
                if gold[q, g] >= 0 and ids[gold[q, g]] in rank_of:
                    assert out[wi, q, g] == rank_of[ids[gold[q, g]]], (wi, q, g)


# ---- NumPy-1 promotion switch -----------------------------------------------------------------------------------------------
def test_numpy1_promotion_switch_fuses_python_float_weights_in_float64(oracle):
    """The reference pins NumPy 1.x, where np.float32 * python_float is float64: with Aggregator.NUMPY1_PROMOTION the equal-weights fusion
    (weights 1/S, Python floats, hybrid.py:448) takes the float64 path -- the very path the np.float64 grid takes, pinned by the tune
    fixtures -- instead of NumPy 2's float32."""
    from fusion_amd.retrievers.hybrid import Aggregator
    z = np.load(os.path.join(GOLDEN, "tune_seed21_S3_Q4_N257_colbert_first.npz"), allow_pickle=False)
    systems, lists, Q = load_lists(z)
    w_py = {s: 1 / len(systems) for s in systems}
    w_np = {s: np.float64(1 / len(systems)) for s in systems}
    for norm in ("min-max", "z-score", "percentile-rank"):
        distr = {s: z[f"distr_{s}"] for s in systems}
        narrow = Aggregator.fuse(lists, "nsf", norm, w_py, distr)
        assert isinstance(narrow[0][0]["score"], np.float32)
        Aggregator.NUMPY1_PROMOTION = True
        try:
            legacy = Aggregator.fuse(lists, "nsf", norm, w_py, distr)
        finally:
            Aggregator.NUMPY1_PROMOTION = False
        wide = Aggregator.fuse(lists, "nsf", norm, w_np, distr)
        assert legacy == wide and isinstance(legacy[0][0]["score"], float)
        exp = oracle.fuse_lists(lists, "nsf", norm, w_np, distr)
        assert [[x["corpus_id"] for x in l] for l in legacy] == [[x["corpus_id"] for x in l] for l in exp]


# ---- SPLADE head with the pooling as the GEMM's epilogue (splade/splade.py:88-99) -----------------------------------------------------
def test_splade_head_epilogue_matches_reference_pooling(ops):
    """The reference's own pooled vectors (fixture from SPLADE.forward) through the fused kernel: with one-hot hidden rows the GEMM
    reproduces the fixture's logits exactly (1 x logit + zeros), so what is checked is the epilogue -- segment cuts, lane-half join,
    bias, log1p o relu, atomicMax -- against the reference's amax(log1p(relu(logits * mask)))."""
    z = np.load(os.path.join(GOLDEN, "splade_pool_B5_L24_V509.npz"))
    logits, lens = z["logits"], z["lens"]
    rows = np.concatenate([logits[b, : lens[b]] for b in range(len(lens))])          # packed: attended tokens only  [T, V]
    T, V = rows.shape
    d = -(-T // 4) * 4
    X = np.zeros((T, d), dtype=np.float32); X[np.arange(T), np.arange(T)] = 1.0
    W = np.zeros((V, d), dtype=np.float32); W[:, :T] = rows.T
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    got = ops.splade_head_max(dev(X), dev(W), torch.zeros(V, device="cuda"), cu).cpu().numpy()
    assert np.max(np.abs(got - z["max"])) <= 5e-7      # log1p: ocml vs torch's vectorised CPU implementation
    assert got[1, 7] == 0.0 and np.all(got >= 0)
    # the bias is added before the transform: shifting every logit of a column by b == giving the column the bias b
    b = np.linspace(-1.0, 1.0, V).astype(np.float32)
    shifted = ops.segment_splade_max(dev(rows + b[None, :]), cu).cpu().numpy()
    got_b = ops.splade_head_max(dev(X), dev(W), dev(b), cu).cpu().numpy()
    assert np.max(np.abs(got_b - shifted)) <= 5e-7


@pytest.mark.parametrize("T,V,d,seed", [(700, 32005, 768, 0), (4100, 2000, 768, 1), (130, 509, 64, 2)])
def test_splade_head_fused_equals_unfused_path(ops, T, V, d, seed):
    """fz_splade_head_max_f32 == decoder GEMM -> [T, V] logits -> fz_segment_splade_max_f32 (the materialising path) on ragged sequences:
    empty ones, one-token ones, sequences that cross the 64-row wave cuts and the 128-row tile cuts, a last partial tile; V = 32,005
    takes the ragged-column tiles.  Both sides accumulate 768 fp32 products in different orders: |diff| <= 2e-6 on values of <= 3."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((T, d), generator=g, device="cuda") * 0.5
    W = torch.randn((V, d), generator=g, device="cuda") * 0.08
    bias = torch.randn((V,), generator=g, device="cuda") * 0.3
    rng = np.random.default_rng(seed)
    cuts = np.sort(rng.choice(np.arange(1, T), size=min(T // 40 + 3, T - 1), replace=False))
    cu_np = np.concatenate([[0, 0, 1], cuts[cuts > 1], [T, T]]).astype(np.int32)      # leading empty + one-token sequence, trailing empty
    cu = torch.from_numpy(cu_np).cuda()
    fused = ops.splade_head_max(x, W, bias, cu)
    logits = torch.nn.functional.linear(x, W, bias)
    unfused = ops.segment_splade_max(logits, cu)
    assert fused.shape == unfused.shape == (len(cu_np) - 1, V)
    assert float((fused - unfused).abs().max()) <= 2e-6
    exact = torch.nn.functional.linear(x.double(), W.double(), bias.double())
    ref = torch.stack([torch.log1p(torch.relu(exact[a:b].max(dim=0).values)) if b > a else torch.zeros(V, dtype=torch.float64, device="cuda")
                       for a, b in zip(cu_np[:-1], cu_np[1:])])
    assert float((fused.double() - ref).abs().max()) <= 5e-6     # 768 fp32 products per logit, logits up to ~15
    assert float(fused[0].abs().max()) == 0.0 and float(fused[-1].abs().max()) == 0.0       # empty sequences pool to 0


def test_splade_encoder_fused_head_equals_materialising_head():
    """SpladeEncoder (random CamemBERT-shaped tiny model): encode_ids_packed with the fused head == with FUSED_HEAD = False == the plain HF
    forward + amax(log1p(relu(logits * mask))) of splade.py:88-99."""
    from fusion_amd import encoders
    cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)   # 64-wide heads: the padding-free forward
    torch.manual_seed(3)
    enc = encoders.SpladeEncoder(encoders._backbone(cfg, mlm=True), encoders.HashTokenizer(cfg["vocab_size"]), "cuda")
    rng = np.random.default_rng(0)
    n, L = 37, 40
    lens = rng.integers(1, L + 1, n)
    ids = rng.integers(7, 500, (n, L))
    ids_d = torch.from_numpy(ids).cuda()
    fused = enc.encode_ids_packed(ids_d, lens)
    enc.FUSED_HEAD = False
    unfused = enc.encode_ids_packed(ids_d, lens)
    mask = torch.from_numpy((np.arange(L)[None, :] < lens[:, None]).astype(np.int64)).cuda()
    plain = enc.encode_ids(torch.where(mask.bool(), ids_d, torch.ones_like(ids_d)), mask)
    assert float((fused - unfused).abs().max()) <= 2e-6
    assert float((fused - plain).abs().max()) <= 5e-6


# ---- top-k form of the fused-list ordering ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("Q,N,k", [(5, 27942, 1000), (3, 1000, 1000), (4, 1500, 1000), (6, 300, 10), (3, 28672, 1)])
def test_select_topk_equals_the_head_of_the_full_sort(ops, oracle, dtype, Q, N, k):
    """ops.select_topk(fused, pos, k) == the first k entries of the stable sort of the row in first-insertion order (hybrid.py:301-306),
    with exact ties across the k-th place, absent documents (pos < 0), NaN, +-0 and fewer listed documents than k."""
    rng = np.random.default_rng(Q * N + k)
    x = rng.normal(0, 1, (Q, N)).astype(dtype)
    if dtype == np.float64:
        x[0] = x[0].astype(np.float32) + 1e-12 * rng.normal(0, 1, N)            # distinct float64 scores that round to equal float32 ones
    x[1, ::7] = x[1, 3]                                                          # a tie run that usually straddles the k-th place
    if N > 50:
        x[2, 5] = np.nan; x[2, 9] = 0.0; x[2, 10] = -0.0
    perm = np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32)      # pos[q, j]: insertion position of column j
    listed = rng.random((Q, N)) < (0.7 if Q > 2 else 1.0)
    listed[0] = True
    pos = np.where(listed, perm, -1).astype(np.int32)
    # make positions dense per row (0 .. U-1) as the insertion order is
    for q in range(Q):
        idx = np.flatnonzero(pos[q] >= 0)
        pos[q, idx[np.argsort(pos[q, idx])]] = np.arange(len(idx), dtype=np.int32)
    U = (pos >= 0).sum(1).astype(np.int32)
    ins = np.full((Q, N), -1, dtype=np.int32)
    for q in range(Q):
        idx = np.flatnonzero(pos[q] >= 0)
        ins[q, pos[q, idx]] = idx
    e_order, e_keys = oracle.sort_rows_desc(x, init_order=ins, row_len=U)
    got = ops.select_topk(plane_of(ops, x), plane_of(ops, pos), k)
    assert got is not None
    cols, sc, lens = (t.cpu().numpy() for t in got)
    kk = min(k, N)
    np.testing.assert_array_equal(lens, np.minimum(U, kk))
    for q in range(Q):
        n = int(lens[q])
        np.testing.assert_array_equal(cols[q, :n], e_order[q, :n])
        np.testing.assert_array_equal(sc[q, :n], e_keys[q, :n])
        assert np.all(cols[q, n:] == -1)


def test_select_topk_reports_a_tie_run_it_cannot_hold(ops):
    x = np.zeros((2, 5000), dtype=np.float32)                                    # 5000 equal scores: every one is a candidate for any k
    assert ops.select_topk(plane_of(ops, x), None, 100) is None


@pytest.mark.parametrize("method,norm", [("rrf", None), ("bcf", None), ("nsf", "min-max"), ("nsf", "z-score"), ("nsf", "none")])
def test_fuse_device_topk_is_the_head_of_the_full_lists(ops, method, norm):
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    rng = np.random.default_rng(3)
    Q, N, k = 7, 27942, 1000
    ids = np.arange(N) + 5
    hidden = rng.normal(0, 1, (Q, N))
    mk = lambda s: (hidden + s * rng.normal(0, 1, (Q, N))).astype(np.float32)
    for cut in (None, int(0.6 * N)):                                             # all lists full | one PLAID-style short list
        systems = {"a": _rank_scores(plane_of(ops, np.maximum(mk(1.0), 0.0)), ids, None), "b": _rank_scores(plane_of(ops, mk(0.5)), ids, None),
                   "c": _rank_scores(plane_of(ops, mk(0.8)), ids, cut)}
        w = {"a": 0.2, "b": 0.5, "c": 0.3}
        full = Aggregator.fuse_device(systems, method, norm, w, {})
        head = Aggregator.fuse_device(systems, method, norm, w, {}, topk=k)
        assert head.order.shape == (Q, k) and head.scores.dtype == full.scores.dtype
        np.testing.assert_array_equal(head.order.cpu().numpy(), full.order.cpu().numpy()[:, :k])
        np.testing.assert_array_equal(head.scores.cpu().numpy(), full.scores.cpu().numpy()[:, :k])
        assert head.predictions(1000) == full.predictions(1000)


# ---- score GEMM: query batches that are not a multiple of 128 rows ---------------------------------------------------------------------------
@pytest.mark.parametrize("d", [768, 100, 36])
def test_dot_scores_do_not_depend_on_the_row_class(ops, d):
    """195 queries run as one 128-row block of MFMA tiles + a 64-row block (2 x 2 waves, one row block each) + a 32-row block (1 x 4 waves).
    Every tile shape accumulates a score as the same chain of fused multiply-adds in the same k order: a query's scores are the same
    bits whichever tile its row falls into (and whatever the batch size)."""
    g = torch.Generator(device="cuda").manual_seed(d)
    N = 5003
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((211, d), generator=g, device="cuda"))
    S195 = ops.dot_scores(Qn[:195].contiguous(), Dn)                 # 128 + 64 + 32 (3 rows used)
    assert torch.equal(S195[192:195], ops.dot_scores(Qn[192:195].contiguous(), Dn))     # the last rows as a batch of their own (Q = 3: a 32-row tile)
    assert torch.equal(S195[128:192], ops.dot_scores(Qn[128:192].contiguous(), Dn))     # the 64-row block on its own
    assert torch.equal(S195[64:192], ops.dot_scores(Qn[64:192].contiguous(), Dn))       # ... and inside a whole 128-row block
    S211 = ops.dot_scores(Qn, Dn)                                    # 128 + 83: 64 + 32
    assert torch.equal(S211[:195], S195)
    # round 5: Q % 128 in 65 .. 68 -- the stragglers ride as ONE 4-row slab (v_mfma_f32_4x4x1_16b_f32) inside the 64-row tail tiles: same bits
    for q in (193, 194, 196):
        assert torch.equal(ops.dot_scores(Qn[:q].contiguous(), Dn), S211[:q]), q
    S67 = ops.dot_scores(Qn[128:195].contiguous(), Dn)               # 67 rows: a 64-row tail + a 3-row slab, no whole block in front
    assert torch.equal(S67, S195[128:195])
    S137 = ops.dot_scores(Qn[:137].contiguous(), Dn)                 # 128 + 32 (9 rows used)
    assert torch.equal(S137, S195[:137])
    S16 = ops.dot_scores(Qn[:16].contiguous(), Dn)                   # a small batch: 32-row tiles only
    S100 = ops.dot_scores(Qn[:100].contiguous(), Dn)                 # 64 + 32 + ... 100 > 96: one padded 128-row block
    assert torch.equal(S100, S195[:100])
    assert torch.equal(S16, S195[:16])
    ref = (Qn[:195].double() @ Dn.double().T)
    assert float((S195.double() - ref).abs().max()) <= 2e-6


@pytest.mark.parametrize("Q", [3, 64, 65, 68, 69, 80, 81, 129, 144, 145, 193, 195, 196, 197, 200, 201, 208, 209, 300, 1024 + 67])
def test_dot_scores_every_tiling_of_the_query_rows(ops, oracle, Q):
    g = torch.Generator(device="cuda").manual_seed(Q)
    N, d = 1283, 64
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    S = ops.dot_scores(Qn, Dn).cpu().numpy()
    exp = oracle.dot_scores(Qn.cpu().numpy(), Dn.cpu().numpy(), fma_chain=True)
    assert S.shape == (Q, N) and np.max(np.abs(S - exp)) <= 2e-6


# ---- BM25: the per-index slice-offset table ---------------------------------------------------------------------------------------------
def test_bm25_slice_offsets_change_nothing_but_the_time(ops, oracle):
    """fz_bm25_scores_f64 with the posting-slice table (fz_bm25_slice_offsets) == without it (binary searches per workgroup) == the oracle,
    bit for bit, on a corpus of more than two document slices with terms that live in one slice only, in none, or in all."""
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(12)
    V, N = 300, 30_000                                            # five slices of 7,168 documents
    p = 1.0 / np.arange(1, V + 1) ** 1.1; p /= p.sum()
    vocab = np.array([f"w{i}" for i in range(V)])
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 30)), p=p)) for _ in range(N)]
    docs[5] += " onlyhere"; docs[20_000] += " onlythere onlythere"
    queries = ["w0 w1 onlyhere", "onlythere w7 w7 nowhere", "w299 w150", "nowhere"]
    ours = BM25(docs, 2.5, 0.2, device="cuda")
    slices = -(-N // int(ops._lib.lib().fz_bm25_slice_docs()))
    assert slices >= 3 and ours.slice_off is not None and tuple(ours.slice_off.shape) == (len(ours.vocab), slices + 1)
    so = ours.slice_off.cpu().numpy(); toff = ours.toff.cpu().numpy()
    assert np.array_equal(so[:, 0], toff[:-1]) and np.array_equal(so[:, -1], toff[1:]) and np.all(np.diff(so, axis=1) >= 0)
    with_table = ours.scores(queries).cpu().numpy()
    ours.slice_off = None
    without = ours.scores(queries).cpu().numpy()
    np.testing.assert_array_equal(with_table, without)
    np.testing.assert_array_equal(with_table, oracle.BM25(docs, 2.5, 0.2).scores(queries))


# ---- mixed-precision encoder forward (ColBERT: colbert-ai's autocast) ------------------------------------------------------------------
def test_f16_encoder_kernels_vs_torch(ops):
    """fz_add_layernorm_x16 / fz_gelu_f16 / fz_attn_varlen_f16 against torch on the same float16 values."""
    g = torch.Generator(device="cuda").manual_seed(3)
    F = torch.nn.functional
    for rows, d in ((1, 128), (37, 768), (5, 2048)):
        x16 = (torch.randn((rows, d), generator=g, device="cuda") * 2).half()
        res = torch.randn((rows, d), generator=g, device="cuda")
        gamma, beta = torch.rand(d, generator=g, device="cuda") + 0.5, torch.randn(d, generator=g, device="cuda")
        out16 = torch.empty((rows, d), dtype=torch.float16, device="cuda")
        out = ops.add_layernorm_x16(x16, res, gamma, beta, 1e-5, out16=out16)
        ref = F.layer_norm(x16.float() + res, (d,), gamma, beta, 1e-5)
        assert (out - ref).abs().max().item() <= 2e-5
        assert torch.equal(out16, out.half())
        assert torch.equal(ops.add_layernorm_x16(x16, None, gamma, beta, 1e-5), ops.add_layernorm(x16.float(), None, gamma, beta, 1e-5))
    for n in (8, 4096, 3072 * 37):
        h = (torch.randn(n, generator=g, device="cuda") * 3).half()
        ref = F.gelu(h.float()).half()
        got = ops.gelu_f16_(h.clone())
        # one float16 rounding of the float32 value: at most one unit in the last place from torch's (erff implementations differ in the last float32 bit)
        assert ((got.float() - ref.float()).abs() <= 2.0 ** -10 * ref.float().abs().clamp_min(2.0 ** -14)).all()
    lengths = np.array([5, 64, 1, 33, 17])
    strips, cu = ops.attn_strips(lengths)
    T = int(cu[-1])
    qkv = torch.randn((T, 3 * 2 * 64), generator=g, device="cuda")
    strips_d = torch.from_numpy(strips).cuda()
    qkv16 = qkv.half()
    ctx = ops.attn_varlen(qkv16.float(), strips_d, 2)           # the float32 kernel on the same (float16-representable) values
    ctx16 = torch.zeros((T, 128), dtype=torch.float16, device="cuda")
    ops.attn_varlen_f16(qkv16, strips_d, 2, ctx16)
    assert torch.equal(ctx16, ctx.half())


def test_colbert_mixed_precision_forward_matches_fp32_and_autocast():
    """ColbertEncoder(amp=True): float16 Linears in the padding-free forward (float32 everywhere else) -- token vectors within float16
    rounding of the float32 forward's, and of the HF module under torch.autocast (what colbert-ai runs)."""
    from fusion_amd import encoders
    cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
    rng = np.random.default_rng(5)
    texts = [" ".join(f"w{rng.integers(0, 40)}" for _ in range(int(k))) for k in rng.integers(1, 90, size=23)]
    tok = encoders.HashTokenizer(cfg["vocab_size"])
    torch.manual_seed(2)
    backbone = encoders._backbone(cfg)
    a = encoders.ColbertEncoder(backbone, tok, "cuda", amp=True)
    b = encoders.ColbertEncoder(backbone, tok, "cuda", amp=False)
    b.linear.load_state_dict(a.linear.state_dict())
    Qa, Qb = a.encode_queries(texts[:7]).float(), b.encode_queries(texts[:7]).float()
    assert a._packed.amp_dtype == torch.float16 and b._packed_forward(b.backbone).amp_dtype is None
    a._packed_forward(a.backbone)
    assert (Qa - Qb).abs().max().item() <= 4e-3
    (Da, Oa), (Db, Ob) = a.encode_docs(texts), b.encode_docs(texts)
    assert torch.equal(Oa, Ob) and (Da.float() - Db.float()).abs().max().item() <= 4e-3
    for i, t in enumerate(texts[:5]):      # the HF module under autocast
        ids1, m1 = tok([t], a.max_doc_length)
        ref = a._tokens(ids1.cuda(), m1.cuda())[0]
        assert (Da[int(Oa[i]): int(Oa[i + 1])].float() - ref).abs().max().item() <= 6e-3


# ---- SPLADE scoring over an inverted index (fz_sparse_dot_f32) ---------------------------------------------------------------------------
@pytest.mark.parametrize("Q,N,V,dn,qn", [(5, 300, 997, 40, 9), (33, 27942, 32005, 200, 40), (3, 30000, 500, 30, 400), (1, 1, 4, 2, 2)])
def test_sparse_dot_equals_the_dense_contraction(ops, Q, N, V, dn, qn):
    """The sparse form of util.cos_sim over SPLADE vectors: same scores as the dense fp32 GEMM of the same (normalised) matrices and as a
    float64 product; bit-identical when repeated; documents without any query term score exactly 0; two document slices at N = 30,000."""
    rng = np.random.default_rng(Q * 7 + N)
    def rows(n, k):
        X = np.zeros((n, -(-V // 4) * 4), dtype=np.float32)
        for i in range(n):
            c = rng.choice(V, size=min(V, max(1, rng.poisson(k))), replace=False)
            X[i, c] = np.log1p(np.maximum(rng.normal(1, 1, c.size), 0.05)).astype(np.float32)
        return X
    D, Qm = rows(N, dn), rows(Q, qn)
    if N > 2: D[1] = 0                                         # an empty document: norm clamp, no postings
    Dn, Qn = ops.normalize_rows(dev(D)), ops.normalize_rows(dev(Qm))
    idx = ops.sparse_index(Dn, V)
    assert idx.nnz == int((D[:, :V] != 0).sum()) and bool((idx.pdoc[1:] >= idx.pdoc[:-1])[(idx.toff[1:-1] - 1).clamp_min(0).unique()].numel() >= 0)
    got = ops.sparse_dot(idx, *ops.sparse_rows(Qn, V))
    assert torch.equal(got, ops.sparse_dot(idx, *ops.sparse_rows(Qn, V)))
    dense = ops.dot_scores(Qn, Dn)
    ref = (Qn.double() @ Dn.double().T)
    assert (got.double() - ref).abs().max().item() <= 2e-6
    assert (got - dense).abs().max().item() <= 2e-6
    never = (ref == 0)
    assert bool((got[never] == 0).all())
    # the drop-in path: dense query vectors in, cosine scores out; queries that are not sparse take the GEMM against the re-densified corpus
    assert torch.equal(ops.sparse_cos_scores(dev(Qm), idx, max_query_density=1.0), got)
    assert torch.equal(idx.to_dense()[:, :V], Dn[:, :V])
    assert (ops.sparse_cos_scores(dev(Qm), idx, max_query_density=0.0) - got).abs().max().item() <= 2e-6


def test_ranker_scores_sparse_splade_through_the_index(ops, tmp_path):
    """Ranker.single_vector_search with a SPLADE encoder whose vectors are sparse: inverted index (also through the on-disk cache) == the
    dense cos_sim path, ranked lists and scores; a dense 'SPLADE' corpus falls back to the GEMM."""
    from fusion_amd.retrievers.hybrid import Ranker
    V = 997
    rng = np.random.default_rng(11)

    class FakeSplade:
        dim = V
        def __init__(self, dense): self.dense = dense
        def encode(self, texts, batch_size=64, query_mode=False):
            out = np.zeros((len(texts), V), dtype=np.float32)
            for i, t in enumerate(texts):
                r = np.random.default_rng(abs(hash(t)) % (1 << 31))
                c = r.choice(V, size=V // 2 if self.dense else (12 if query_mode else 60), replace=False)
                out[i, c] = np.log1p(r.random(c.size) * 3).astype(np.float32)
            return torch.from_numpy(out).cuda()

    corpus = {100 + i: f"doc {i} {rng.integers(0, 1 << 30)}" for i in range(400)}
    queries = [f"query {i}" for i in range(7)]
    for dense in (False, True):
        enc = FakeSplade(dense)
        a = Ranker.single_vector_search(queries, corpus, "splade-fake", encoder=enc, cache_dir=str(tmp_path / f"c{int(dense)}"))
        b = Ranker.single_vector_search(queries, corpus, "splade-fake", encoder=enc, cache_dir=str(tmp_path / f"c{int(dense)}"))     # from the cache
        De, Qe = enc.encode(list(corpus.values())), enc.encode(queries, query_mode=True)
        ref = ops.cos_scores(Qe, De).cpu().numpy()
        ids = np.array(list(corpus.keys()))
        for lists in (a, b):
            for q, lst in enumerate(lists):
                assert len(lst) == len(corpus)
                got = np.array([x["score"] for x in lst]); cid = np.array([x["corpus_id"] for x in lst])
                exp = ref[q][np.searchsorted(ids, cid)]
                assert np.abs(got - exp).max() <= 2e-6 and np.all(np.diff(got) <= 0)
        assert [[x["corpus_id"] for x in l] for l in a] == [[x["corpus_id"] for x in l] for l in b]


def test_mfma_kernels_are_bit_reproducible(ops):
    """Reruns on one input give identical bits (attention, score GEMM at a 7-row-block and a many-tile batch, MaxSim, the fused SPLADE head).
    Round 3: attn_varlen_kernel's tile maximum read the score registers of its MFMA chain too early -- results within tolerance, but different from
    run to run; a reader of an MFMA result that hipcc leaves unpadded behind a branch shows up here (and in tools/check_mfma_hazards.py)."""
    g = torch.Generator(device="cuda").manual_seed(9)
    lens = np.array([442, 548, 389, 487, 516, 212, 127, 250, 242, 33, 1])
    strips, _ = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    qkv = torch.randn((int(lens.sum()), 3 * 12 * 64), generator=g, device="cuda")
    ref = ops.attn_varlen(qkv, sd, 12).clone()
    for _ in range(4):
        assert torch.equal(ops.attn_varlen(qkv, sd, 12), ref)
    q16 = qkv.half()
    o16 = torch.empty((qkv.shape[0], 768), dtype=torch.float16, device="cuda")
    ops.attn_varlen_f16(q16, sd, 12, o16)
    assert torch.equal(o16, ops.attn_varlen(q16.float(), sd, 12).half())
    D = ops.normalize_rows(torch.randn((27942, 768), generator=g, device="cuda"))
    for Q in (195, 1024):
        A = ops.normalize_rows(torch.randn((Q, 768), generator=g, device="cuda"))
        r = ops.dot_scores(A, D).clone()
        for _ in range(3):
            assert torch.equal(ops.dot_scores(A, D), r)
    lens = np.clip(np.random.default_rng(2).normal(300, 120, 1500), 16, 512).astype(np.int64)
    Doff = np.zeros(lens.size + 1, dtype=np.int64); np.cumsum(lens, out=Doff[1:])
    Dtok = torch.nn.functional.normalize(torch.randn((int(Doff[-1]), 128), generator=g, device="cuda"), dim=-1).half()
    Qtok = torch.nn.functional.normalize(torch.randn((195, 64, 128), generator=g, device="cuda"), dim=-1).half()
    r = ops.maxsim(Qtok, Dtok, dev(Doff), max_doc_len=512).clone()
    for _ in range(3):
        assert torch.equal(ops.maxsim(Qtok, Dtok, dev(Doff), max_doc_len=512), r)
    x = torch.randn((3000, 768), generator=g, device="cuda")
    cu = torch.tensor([0, 700, 701, 1800, 3000], dtype=torch.int32, device="cuda")
    W, b = torch.randn((32005, 768), generator=g, device="cuda") * 0.05, torch.randn(32005, generator=g, device="cuda")
    r = ops.splade_head_max(x, W, b, cu).clone()
    for _ in range(2):
        assert torch.equal(ops.splade_head_max(x, W, b, cu), r)


@pytest.mark.parametrize("H,lens", [(2, [5, 64, 1, 33, 17]), (12, [300, 512, 16, 129, 31, 250]), (1, [1]), (3, [16, 32, 48, 15, 47])])
def test_amp_attention_matches_float64_and_the_float32_kernel(ops, H, lens):
    """fz_attn_varlen_f16_amp (float16 MFMAs for q k^T and p v, float32 softmax: autocast's arithmetic) against a float64 softmax(q k^T / 8) v
    of the same float16 inputs and against the float32-arithmetic kernel: within float16 rounding of the probabilities; reruns identical."""
    g = torch.Generator(device="cuda").manual_seed(len(lens) * 10 + H)
    lens = np.array(lens)
    strips, cu = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    T = int(cu[-1])
    qkv16 = (torch.randn((T, 3 * H * 64), generator=g, device="cuda") * 1.5).half()
    out = torch.empty((T, H * 64), dtype=torch.float16, device="cuda")
    ops.attn_varlen_f16(qkv16, sd, H, out, amp=True)
    again = torch.empty_like(out); ops.attn_varlen_f16(qkv16, sd, H, again, amp=True)
    assert torch.equal(out, again)
    exact = torch.empty_like(out); ops.attn_varlen_f16(qkv16, sd, H, exact, amp=False)
    for b, L in enumerate(lens.tolist()):
        blk = qkv16[cu[b]: cu[b] + L].double().view(L, 3, H, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(1, 2) / 8.0, -1) @ v).transpose(0, 1).reshape(L, H * 64)
        scale = max(1.0, ref.abs().max().item())
        assert (out[cu[b]: cu[b] + L].double() - ref).abs().max().item() <= 3e-3 * scale
        assert (out[cu[b]: cu[b] + L].float() - exact[cu[b]: cu[b] + L].float()).abs().max().item() <= 3e-3 * scale
