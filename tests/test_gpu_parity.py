"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs, vs the golden
fixtures made by the reference, and size-independent properties at BASELINE.json's full sizes.
Bar: bit-exact for ranks / orders / fp64 rank fusion / exact-stat fp32 transforms; stated tolerance otherwise."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def plane(ops, a):
    t = ops.alloc_plane(a.shape[0], a.shape[1], torch.from_numpy(a[:0]).dtype, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def keys_with_ties(rng, rows, n, dtype):
    k = rng.normal(0, 1, (rows, n)).astype(dtype)
    if n >= 8:
        for r in range(rows):
            idx = rng.choice(n, size=max(2, n // 3), replace=False)
            k[r, idx] = k[r, idx[0]]
            k[r, idx[1]] = 0.0
            k[r, idx[2]] = -0.0
        k[0, rng.integers(0, n)] = np.inf
        k[0, rng.integers(0, n)] = -np.inf
        if rows > 1:
            k[1, rng.integers(0, n)] = np.nan
    return k


SORT_N32 = [1, 2, 63, 64, 65, 257, 1000, 1024, 1025, 4096, 4097, 8192, 9000, 16384, 20000, 27942, 28672, 28673, 35840]
SORT_N64 = [1, 2, 65, 1000, 1025, 4097, 9000, 16385, 27942, 28672]


@pytest.mark.parametrize("n", SORT_N32)
def test_sort_rows_f32(ops, oracle, n):
    rng = np.random.default_rng(n)
    k = keys_with_ties(rng, 5, n, np.float32)
    order, sk, rank = ops.sort_rows_desc(plane(ops, k), want_rank=True)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)   # -0.0 == 0.0, NaN == NaN positionally


@pytest.mark.parametrize("n", SORT_N64)
def test_sort_rows_f64(ops, oracle, n):
    rng = np.random.default_rng(1000 + n)
    k = keys_with_ties(rng, 4, n, np.float64)
    order, sk, rank = ops.sort_rows_desc(plane(ops, k), want_rank=True)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)


def test_sort_has_no_cpu_path(ops):
    with pytest.raises(TypeError):
        ops.sort_rows_desc(torch.zeros((2, 10)))   # CPU tensor: no CPU path


@pytest.mark.parametrize("n,dtype", [(35841, np.float32), (28673, np.float64), (60000, np.float64), (100003, np.float32)])
@pytest.mark.parametrize("mode", ["plain", "gathered", "placed"])
def test_sort_rows_longer_than_one_workgroup(ops, oracle, n, dtype, mode):
    """Rows beyond the single-workgroup capacity (35,840 fp32 / 28,672 fp64 keys): chunk-sort + cross-chunk ranking, same
    stable order, ties / NaN / signed zeros / infinities included."""
    rng = np.random.default_rng(n)
    rows = 3
    k = keys_with_ties(rng, rows, n, dtype)
    k[2, :] = np.round(k[2, :], 1)                        # heavy ties across chunk boundaries
    kp = plane(ops, k)
    if mode == "plain":
        order, sk, rank = ops.sort_rows_desc(kp, want_rank=True)
        e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    else:
        init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)
        lens = np.array([n, n - 5, n // 2 + 7], dtype=np.int32)
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


@pytest.mark.parametrize("n,dtype", [(5, np.float32), (300, np.float32), (5000, np.float64), (27942, np.float32)])
def test_sort_gathered_sequence(ops, oracle, n, dtype):
    """init_order + row_len: the fused-list ordering call (ties keep first-insertion order)."""
    rng = np.random.default_rng(n)
    rows = 4
    k = np.round(rng.normal(0, 1, (rows, n)), 1).astype(dtype)   # many ties
    init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)
    lens = np.array([n, max(1, n // 2), 1, max(1, n - 1)], dtype=np.int32)
    for r in range(rows):
        init[r, lens[r]:] = -1
    order, sk, _ = ops.sort_rows_desc(plane(ops, k), init_order=plane(ops, init), row_len=dev(lens))
    e_order, e_sk = oracle.sort_rows_desc(k, init_order=init, row_len=lens)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)


@pytest.mark.parametrize("n,dtype", [(1, np.float32), (300, np.float32), (5000, np.float64), (27942, np.float32), (27942, np.float64)])
def test_sort_placed_sequence(ops, oracle, n, dtype):
    """init_rank form (coalesced, placed through LDS) == init_order form == oracle."""
    rng = np.random.default_rng(n + 1)
    rows = 4
    k = np.round(rng.normal(0, 1, (rows, n)), 1).astype(dtype)
    init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)
    lens = np.array([n, max(1, n // 2), 1, max(1, n - 1)], dtype=np.int32)
    irank = np.full((rows, n), -1, dtype=np.int32)
    for r in range(rows):
        init[r, lens[r]:] = -1
        irank[r, init[r, :lens[r]]] = np.arange(lens[r], dtype=np.int32)
    order, sk, rank = ops.sort_rows_desc(plane(ops, k), init_rank=plane(ops, irank), row_len=dev(lens), want_rank=True)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, init_order=init, row_len=lens, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)


def synth_systems(rng, S, Q, N, partial=True):
    planes, ranks, orders, lens = [], [], [], np.zeros((S, Q), dtype=np.int32)
    for s in range(S):
        if s == 0:
            sc = np.maximum(0, rng.gamma(0.5, 4.0, (Q, N)) - 2).astype(np.float32)
        elif s == 3:
            sc = rng.normal(20, 4, (Q, N)).astype(np.float32)
        else:
            sc = rng.uniform(-0.2, 0.9, (Q, N)).astype(np.float32)
        order = np.argsort(-sc.astype(np.float64), axis=1, kind="stable").astype(np.int32)
        rank = np.full((Q, N), -1, dtype=np.int32)
        for q in range(Q):
            L = N if not (partial and s in (1, 3)) else max(1, int(N * (0.6 if s == 3 else 0.9)))
            lens[s, q] = L
            rank[q, order[q, :L]] = np.arange(L, dtype=np.int32)
            order[q, L:] = -1
        planes.append(sc); ranks.append(rank); orders.append(order)
    return planes, ranks, orders, lens


@pytest.mark.parametrize("S,Q,N", [(2, 3, 1), (2, 4, 257), (4, 5, 1000), (4, 3, 5000), (3, 2, 27942)])
@pytest.mark.parametrize("method", ["rrf", "bcf"])
def test_fuse_rank_exact(ops, oracle, S, Q, N, method):
    rng = np.random.default_rng(S * 1000 + N)
    _, ranks, _, lens = synth_systems(rng, S, Q, N)
    got = ops.fuse_rank([plane(ops, r) for r in ranks], dev(lens), method).cpu().numpy()
    exp = oracle.fuse_rank(ranks, lens, method)
    np.testing.assert_array_equal(got, exp)   # float64, bit for bit


NSF_TOL = {"min-max": 0.0, "percentile-rank": 0.0, "z-score": 1e-6, "arctan": 1e-6, "normal-curve-equivalent": 1e-4}


@pytest.mark.parametrize("S,Q,N", [(1, 2, 1), (2, 4, 257), (4, 3, 1000), (4, 2, 5000), (4, 2, 20000), (4, 2, 27942)])
@pytest.mark.parametrize("norm", list(NSF_TOL))
@pytest.mark.parametrize("partial", [False, True])
def test_fuse_nsf(ops, oracle, S, Q, N, norm, partial):
    rng = np.random.default_rng(S * 77 + N)
    planes, ranks, _, _ = synth_systems(rng, S, Q, N, partial)
    w = rng.dirichlet(np.ones(S))
    distr = None
    if norm in ("percentile-rank", "normal-curve-equivalent"):
        distr = [np.quantile(p.astype(np.float64), np.linspace(0, 1, min(101, N + 2))).astype(np.float32) for p in planes]
    rk = ranks if partial else None
    got = ops.fuse_nsf([plane(ops, p) for p in planes], None if rk is None else [plane(ops, r) for r in rk], w, norm,
                       None if distr is None else [dev(d) for d in distr]).cpu().numpy()
    exp = oracle.fuse_nsf(planes, rk, w, norm, distr)
    tol = NSF_TOL[norm]
    if tol == 0.0:
        np.testing.assert_array_equal(got, exp)
    else:
        fin = np.isfinite(exp)
        np.testing.assert_array_equal(np.isfinite(got), fin)
        np.testing.assert_array_equal(got[~fin], exp[~fin])   # -inf (absent docs / icdf(0)) and NaN in the same places
        assert np.max(np.abs(got[fin] - exp[fin]), initial=0.0) <= tol


def test_fuse_nsf_constant_and_single(ops, oracle):
    const = np.full((2, 300), 3.25, dtype=np.float32)
    for norm in ("min-max", "z-score"):
        got = ops.fuse_nsf([plane(ops, const)], None, [1.0], norm).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.fuse_nsf([const], None, [1.0], norm))
    one = np.array([[2.0]], dtype=np.float32)
    z = ops.fuse_nsf([plane(ops, one)], None, [1.0], "z-score").cpu().numpy()
    assert np.isnan(z[0, 0])   # torch.std of one element (KAT-4)
    assert ops.fuse_nsf([plane(ops, one)], None, [1.0], "min-max").cpu().numpy()[0, 0] == 1.0


@pytest.mark.parametrize("S,Q,N", [(2, 3, 257), (4, 2, 27942)])
def test_fuse_none_exact(ops, oracle, S, Q, N):
    rng = np.random.default_rng(5)
    planes, ranks, _, _ = synth_systems(rng, S, Q, N)
    w = rng.dirichlet(np.ones(S))
    got = ops.fuse_none([plane(ops, p) for p in planes], [plane(ops, r) for r in ranks], w).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.fuse_none(planes, ranks, w))


@pytest.mark.parametrize("S,Q,N", [(2, 3, 5), (4, 4, 1000), (4, 2, 27942)])
def test_insertion_order(ops, oracle, S, Q, N):
    rng = np.random.default_rng(N)
    _, _, orders, lens = synth_systems(rng, S, Q, N)
    ins, U = ops.insertion_order([plane(ops, o) for o in orders], dev(lens), N)
    e_ins, e_U = oracle.insertion_order(orders, lens, N)
    np.testing.assert_array_equal(U.cpu().numpy(), e_U)
    g = ins.cpu().numpy()
    for q in range(Q):
        np.testing.assert_array_equal(g[q, : e_U[q]], e_ins[q, : e_U[q]])


# ---- the reference's own outputs, through the drop-in Aggregator --------------------------------------
FUSE_FILES = sorted(glob.glob(os.path.join(GOLDEN, "fuse_*.npz")))
METHODS = [("rrf", "none"), ("bcf", "none"), ("nsf", "none"), ("nsf", "min-max"), ("nsf", "z-score"), ("nsf", "arctan"),
           ("nsf", "percentile-rank"), ("nsf", "normal-curve-equivalent")]
EXACT = {("rrf", "none"), ("bcf", "none"), ("nsf", "none"), ("nsf", "min-max"), ("nsf", "percentile-rank")}
TOL = {("nsf", "z-score"): 2e-6, ("nsf", "arctan"): 1e-6, ("nsf", "normal-curve-equivalent"): 1e-4}


@pytest.mark.parametrize("path", FUSE_FILES, ids=[os.path.basename(p)[:-4] for p in FUSE_FILES])
def test_aggregator_matches_reference_golden(path):
    from fusion_amd.retrievers.hybrid import Aggregator
    z = np.load(path, allow_pickle=False)
    systems = [str(s) for s in z["systems"]]
    ids, sc, ln = z["in_ids"], z["in_scores"], z["in_len"]
    Q = ids.shape[1]
    lists = {s: [[{"corpus_id": int(ids[si, q, r]), "score": float(sc[si, q, r])} for r in range(ln[si, q])] for q in range(Q)]
             for si, s in enumerate(systems)}
    weights = {s: float(w) for s, w in zip(systems, z["weights"])}
    distr = {s: z[f"distr_{s}"] for s in systems}
    for method, norm in METHODS:
        got = Aggregator.fuse(lists, method=method, normalization=norm, linear_weights=weights, percentile_distributions=distr)
        key = f"{method}__{norm}"
        e_ids, e_sc, e_len = z[f"out_ids__{key}"], z[f"out_scores__{key}"], z[f"out_len__{key}"]
        assert len(got) == Q
        for q in range(Q):
            n = int(e_len[q])
            assert len(got[q]) == n, (key, q)
            g_ids = np.array([x["corpus_id"] for x in got[q]], dtype=np.int64)
            g_sc = np.array([float(x["score"]) for x in got[q]], dtype=np.float64)
            if (method, norm) in EXACT:
                np.testing.assert_array_equal(g_ids, e_ids[q, :n], err_msg=key)      # ranked-list identity
                np.testing.assert_array_equal(g_sc, e_sc[q, :n], err_msg=key)
            else:
                assert sorted(g_ids.tolist()) == sorted(e_ids[q, :n].tolist())
                exp = {int(i): s for i, s in zip(e_ids[q, :n], e_sc[q, :n])}
                ref = np.array([exp[int(i)] for i in g_ids])
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(g_sc), fin), key
                assert np.max(np.abs(g_sc[fin] - ref[fin]), initial=0.0) <= TOL[(method, norm)], key


def test_aggregator_kat_and_errors():
    from fusion_amd.retrievers.hybrid import Aggregator
    kat = json.load(open(os.path.join(GOLDEN, "kat_fuse.json")))
    L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]
    a = {"s1": [L([(100, 2.0), (200, 1.0)])], "s2": [L([(200, 2.0), (100, 1.0)])]}
    b = {"s2": a["s2"], "s1": a["s1"]}
    assert Aggregator.fuse(a, "rrf") == kat["kat1_s1s2"]          # KAT-1: first-insertion tie-break
    assert Aggregator.fuse(b, "rrf") == kat["kat1_s2s1"]
    three = {"s1": [L([(1, 1.0), (2, .5)])] * 3, "s2": [L([(2, 1.0), (1, .5)])] * 3}
    assert Aggregator.fuse(three, "rrf", return_topk=2) == kat["kat6_topk2"]   # slices queries
    dup = {"s1": [L([(1, 3.0), (2, 2.0), (1, 1.0), (3, 0.5)])], "s2": [L([(3, 1.0), (2, .5)])]}
    assert Aggregator.fuse(dup, "rrf") == kat["kat_dup_rrf"]
    assert Aggregator.fuse(dup, "bcf") == kat["kat_dup_bcf"]
    lists = {"bm25": [L([(10, 7.5), (11, 3.0), (12, 0.0)])], "dpr": [L([(12, .9), (10, .5), (13, .1)])]}
    got = Aggregator.fuse(lists, "nsf", "min-max", {"bm25": .5, "dpr": .5}, {})
    assert [(x["corpus_id"], float(x["score"])) for x in got[0]] == [(x["corpus_id"], x["score"]) for x in kat["kat2_nsf_min-max"][0]]
    with pytest.raises(AssertionError):
        Aggregator.fuse({"a": [L([(1, 1.0)])], "b": [L([(1, 1.0)])] * 2}, "rrf")
    with pytest.raises(KeyError):
        Aggregator.fuse({"a": [L([(1, 1.0)])], "b": [L([(1, 1.0)])]}, "nsf", "min-max", {"a": 1.0}, {})
    with pytest.raises(AttributeError):
        Aggregator.fuse({"a": [L([(1, 1.0)])]}, "nsf", "min-max", {"a": 1.0}, None)
    t = Aggregator.transform_scores({1: 0.2, 2: 4.6, 3: 99.0}, "percentile-rank", percentile_distr=np.linspace(0, 10, 11))
    assert {k: float(v) for k, v in t.items()} == {x["corpus_id"]: x["score"] for x in kat["kat5_percentile"]}


# ---- N1: weight-grid sweep --------------------------------------------------------------------------------
@pytest.mark.parametrize("S,norm,partial", [(2, "min-max", False), (3, "z-score", False), (4, "min-max", True), (4, "z-score", True), (2, "arctan", False)])
def test_tune_equals_fuse_per_weight(ops, S, norm, partial):
    """Aggregator.tune (one counting kernel for the whole grid) == fuse + sort + Metrics per weight vector."""
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator, run_evaluation, weight_grid
    rng = np.random.default_rng(S * 5 + len(norm))
    Q, N = 9, 700
    planes, ranks, orders, lens = synth_systems(rng, S, Q, N, partial)
    for p in planes:                      # plant ties so that the insertion-order tie-break matters
        p[:, ::7] = np.round(p[:, ::7], 1)
    ids = np.arange(5000, 5000 + N)
    names = ["bm25", "dpr", "splade", "colbert"][:S]
    systems = {}
    for n, p, r, o, l in zip(names, planes, ranks, orders, lens):
        # re-rank the (modified) planes on the device so ranks are consistent with the scores
        pl = plane(ops, p)
        od, _, rk = ops.sort_rows_desc(pl, want_rank=True)
        L = torch.from_numpy(l).cuda()
        full = bool((l == N).all())
        if not full:
            keep = torch.arange(N, device="cuda").unsqueeze(0) < L.unsqueeze(1)
            rk = torch.where(rk < L.unsqueeze(1), rk, torch.full_like(rk, -1))
            od = torch.where(keep, od, torch.full_like(od, -1))
        systems[n] = RankedSystem(scores=pl, order=od, rank=rk, lens=L, ids=ids, full=full)
    labels = [rng.choice(ids, size=int(rng.integers(1, 12)), replace=False).tolist() for _ in range(Q)]
    labels[0] = labels[0] + [123456789]          # a gold id that is not in the corpus
    grid = weight_grid(names)
    grid = grid[:: max(1, len(grid) // 25)]       # a spread of ~25 weight vectors incl. zeros
    got = Aggregator.tune(systems, norm, grid, labels, {})
    for w, g in zip(grid, got):
        fused = Aggregator.fuse(systems, "nsf", norm, w, {}, as_device=True)
        exp = run_evaluation(fused.predictions(1000), labels, print2console=False)
        assert list(g) == list(exp)
        for k in exp:
            assert g[k] == pytest.approx(float(exp[k]), rel=0, abs=1e-12), (w, k)


# ---- scoring -------------------------------------------------------------------------------------------
def test_dot_scores_identity_layout(ops):
    """A = I against an ASYMMETRIC B catches a transposed C write / swapped operand map."""
    d = 256
    A = torch.eye(d, device="cuda")[:130].contiguous()          # 130 x 256
    B = (torch.arange(200 * d, device="cuda", dtype=torch.float32).reshape(200, d) % 251) - 97.0
    S = ops.dot_scores(A, B)
    torch.testing.assert_close(S, B[:, :130].t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("Q,N,d", [(1, 1, 4), (3, 5, 8), (195, 300, 768), (130, 1000, 100), (64, 27942, 768), (257, 129, 36)])
def test_cos_scores_vs_oracle(ops, oracle, Q, N, d):
    rng = np.random.default_rng(Q + N + d)
    Qe = rng.normal(0, 1, (Q, d)).astype(np.float32)
    De = rng.normal(0, 1, (N, d)).astype(np.float32)
    Qn = ops.normalize_rows(dev(Qe)); Dn = ops.normalize_rows(dev(De))
    np.testing.assert_allclose(Qn.cpu().numpy()[:, :d], oracle.normalize_rows(Qe), rtol=0, atol=1e-7)
    got = ops.dot_scores(Qn, Dn).cpu().numpy()
    exp = oracle.cos_scores(Qe, De)
    assert got.shape == (Q, N)
    assert np.max(np.abs(got - exp)) <= 2e-6     # contract: 1e-4 (north_star); fp32 MFMA chain is ~1e-7


def test_splade_shaped_scoring(ops, oracle):
    """SPLADE activations (hybrid.py:95-103): V = 32,005 non-negative sparse-ish vectors, scored DENSELY with cos_sim as
    the reference does; 32,005 is not a multiple of 4 -> the binding zero-pads the vocabulary axis."""
    rng = np.random.default_rng(11)
    Q, N, V = 9, 300, 32005
    Qe = np.log1p(np.maximum(0, rng.normal(-1.5, 1.0, (Q, V)))).astype(np.float32)     # amax log1p relu: >= 0, mostly 0
    De = np.log1p(np.maximum(0, rng.normal(-1.0, 1.0, (N, V)))).astype(np.float32)
    got = ops.cos_scores(dev(Qe), dev(De)).cpu().numpy()
    exp = oracle.cos_scores(Qe, De)
    assert got.shape == (Q, N) and np.max(np.abs(got - exp)) <= 2e-6


def test_normalize_zero_row(ops):
    X = torch.zeros((2, 8), device="cuda"); X[1, 0] = 3.0
    Y = ops.normalize_rows(X).cpu().numpy()
    assert np.all(Y[0] == 0) and Y[1, 0] == 1.0       # x / max(||x||, 1e-12)


def ragged_docs(rng, N, dim=128, lo=0, hi=80):
    lens = rng.integers(lo, hi, N)
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    tok = rng.normal(0, 1, (int(off[-1]), dim)).astype(np.float32)
    tok /= np.maximum(np.linalg.norm(tok, axis=1, keepdims=True), 1e-6)
    return tok.astype(np.float16), off


@pytest.mark.parametrize("Q,N,Lq,hi", [(1, 1, 32, 40), (3, 70, 64, 80), (20, 300, 64, 140), (17, 65, 32, 33), (5, 40, 128, 600)])
def test_maxsim_vs_oracle(ops, oracle, Q, N, Lq, hi):
    rng = np.random.default_rng(Q * 31 + N)
    Dtok, Doff = ragged_docs(rng, N, hi=hi)
    if N > 3:
        assert (np.diff(Doff) == 0).any() or True
    Qtok = rng.normal(0, 1, (Q, Lq, 128)).astype(np.float32)
    Qtok /= np.linalg.norm(Qtok, axis=2, keepdims=True)
    Qtok = Qtok.astype(np.float16)
    got = ops.maxsim(dev(Qtok), dev(Dtok), dev(Doff)).cpu().numpy()
    exp = oracle.maxsim(Qtok.astype(np.float32), Dtok.astype(np.float32), Doff)
    assert np.max(np.abs(got - exp)) <= 1e-4 * max(1, Lq / 32)


def test_maxsim_empty_docs(ops, oracle):
    rng = np.random.default_rng(0)
    Dtok, _ = ragged_docs(rng, 1, lo=50, hi=51)
    Doff = np.array([0, 0, 20, 20, 50, 50], dtype=np.int64)     # docs 0, 2, 4 empty
    Qtok = rng.normal(0, 1, (2, 64, 128)).astype(np.float16)
    got = ops.maxsim(dev(Qtok), dev(Dtok), dev(Doff)).cpu().numpy()
    exp = oracle.maxsim(Qtok.astype(np.float32), Dtok.astype(np.float32), Doff)
    assert np.all(got[:, [0, 2, 4]] == 0)
    assert np.max(np.abs(got - exp)) <= 2e-3    # un-normalised tokens: larger magnitudes


# ---- top-k ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,n,k", [(3, 10, 4), (2, 5, 8), (4, 3000, 100), (3, 40000, 1000), (2, 100000, 1000), (2, 300001, 50)])
def test_topk_rows(ops, oracle, rows, n, k):
    rng = np.random.default_rng(n)
    sc = np.round(rng.normal(0, 1, (rows, n)), 3).astype(np.float32)   # ties
    gs, gi = ops.topk_rows(plane(ops, sc), k, id_base=7_000_000_000)
    es, ei = oracle.topk_rows(sc, k, id_base=7_000_000_000)
    np.testing.assert_array_equal(gs.cpu().numpy(), es)
    np.testing.assert_array_equal(gi.cpu().numpy(), ei)


def test_topk_merge(ops, oracle):
    rng = np.random.default_rng(3)
    G, rows, k, nloc = 8, 6, 1000, 5000
    sc = np.round(rng.normal(0, 1, (G, rows, nloc)), 2).astype(np.float32)
    parts = [oracle.topk_rows(sc[g], k, id_base=g * nloc) for g in range(G)]
    in_s = np.stack([p[0] for p in parts]); in_i = np.stack([p[1] for p in parts])
    gs, gi = ops.topk_merge(dev(in_s), dev(in_i))
    es, ei = oracle.topk_merge(in_s, in_i)
    np.testing.assert_array_equal(gs.cpu().numpy(), es)
    np.testing.assert_array_equal(gi.cpu().numpy(), ei)
    # and it equals the top-k of the unsharded matrix
    fs, fi = oracle.topk_rows(np.concatenate(list(sc), axis=1), k)
    np.testing.assert_array_equal(gs.cpu().numpy(), fs)
    np.testing.assert_array_equal(gi.cpu().numpy(), fi)


@pytest.mark.parametrize("streaming", [True, False])
def test_sharded_index_search_equals_oracle(ops, oracle, streaming):
    """Chunked GEMM -> (streaming) top-k over a shard == oracle top-k of the full score matrix (given the device's scores)."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(0)
    N, d, Q, k = 70000, 64, 6, 100
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    idx = ShardedDenseIndex(Dn, id_base=5_000_000_000)
    idx.CHUNK = 16384            # 5 chunks: first exact, then 4 updates
    idx.CAP = 2000
    s, i = idx.local_topk(Qn, k, streaming=streaming)
    S = ops.dot_scores(Qn, Dn).cpu().numpy()
    es, ei = oracle.topk_rows(S, k, id_base=5_000_000_000)
    np.testing.assert_array_equal(s.cpu().numpy(), es)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)
    # candidate-buffer overflow falls back to the exact path (ascending scores: everything beats the threshold)
    S2 = torch.arange(Q * 40000, device="cuda", dtype=torch.float32).reshape(Q, 40000) / 7.0
    rs, ri = ops.topk_rows(ops.as_plane(S2[:, :16384].contiguous()), k)
    ns, ni, flag = ops.topk_update(ops.as_plane(S2[:, 16384:].contiguous()), 16384, rs, ri, cap=500)
    assert int(flag.item()) == 1


# ---- BM25 ----------------------------------------------------------------------------------------------
def test_bm25_matches_reference_golden():
    from fusion_amd.retrievers.bm25 import BM25
    g = json.load(open(os.path.join(GOLDEN, "bm25.json")))
    for (k1, b), exp in zip(g["params"], g["results"]):
        got = BM25(g["docs"], k1=k1, b=b).search_all(g["queries"], top_k=len(g["docs"]))
        for gq, eq in zip(got, exp):
            assert [x["corpus_id"] for x in gq] == [e[0] for e in eq]
            assert [x["score"] for x in gq] == [e[1] for e in eq]     # float64 bit-exact
    k = g["kat8"]
    got = BM25(k["docs"], 2.5, 0.2).search_all(k["queries"], top_k=4)
    assert [[[x["corpus_id"], x["score"]] for x in r] for r in got] == k["results"]


def test_bm25_vs_oracle_larger(oracle):
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(9)
    vocab = np.array([f"w{i}" for i in range(2000)])
    p = 1.0 / np.arange(1, 2001); p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 120)), p=p)) for _ in range(3000)]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(1, 12)), p=p)) for _ in range(40)] + ["", "zzz w1 w1"]
    m = BM25(docs, 2.5, 0.2)
    got = m.scores(queries).cpu().numpy()
    exp = oracle.BM25(docs, 2.5, 0.2).scores(queries)
    np.testing.assert_array_equal(got, exp)


# ---- full-size properties (BASELINE.json sizes: Q=1024, N=27,942) ------------------------------------------
def test_full_size_rrf_pipeline_properties(ops):
    Q, N = 1024, 27942
    g = torch.Generator(device="cuda").manual_seed(0)
    a = ops.alloc_plane(Q, N, torch.float32, "cuda"); a.copy_(torch.rand((Q, N), generator=g, device="cuda"))
    b = ops.alloc_plane(Q, N, torch.float32, "cuda"); b.copy_(torch.randn((Q, N), generator=g, device="cuda").round(decimals=2))
    oa, ka, ra = ops.sort_rows_desc(a, want_rank=True)
    ob, kb, rb = ops.sort_rows_desc(b, want_rank=True)
    for o, k, r, src in ((oa, ka, ra, a), (ob, kb, rb, b)):
        assert bool((k[:, :-1] >= k[:, 1:]).all())                                   # sortedness
        assert bool((torch.sort(o.long(), dim=1).values == torch.arange(N, device="cuda")).all())   # permutation
        assert bool((torch.gather(r.long(), 1, o.long()) == torch.arange(N, device="cuda")).all())  # rank = inverse
        assert bool((torch.gather(src, 1, o.long()) == k).all())                    # keys travel with payloads
    # stability: equal keys keep ascending corpus position
    eq = kb[:, :-1] == kb[:, 1:]
    assert bool((ob[:, :-1][eq] < ob[:, 1:][eq]).all())
    lens = torch.full((2, Q), N, dtype=torch.int32, device="cuda")
    fused = ops.fuse_rank([ra, rb], lens, "rrf")
    ref = 1.0 / (61.0 + ra.double()) + 1.0 / (61.0 + rb.double())                    # same fp64 expression order
    assert bool((fused == ref).all())
    of, kf, _ = ops.sort_rows_desc(fused, init_rank=ra)
    of2, kf2, _ = ops.sort_rows_desc(fused, init_order=oa)
    assert bool((of == of2).all()) and bool((kf == kf2).all())                       # placed == gathered
    assert bool((kf[:, :-1] >= kf[:, 1:]).all())
    assert bool((torch.sort(of.long(), dim=1).values == torch.arange(N, device="cuda")).all())
    # swapping the systems changes nothing but the tie-break (checksum of the fused multiset)
    fused2 = ops.fuse_rank([rb, ra], lens, "rrf")
    assert bool((fused2 == fused).all())                                             # a+b == b+a in IEEE


def test_full_size_nsf_properties(ops):
    Q, N, S = 1024, 27942, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    planes = []
    for s in range(S):
        p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(torch.randn((Q, N), generator=g, device="cuda") * (s + 1) + s)
        planes.append(p)
    w = [0.1, 0.2, 0.3, 0.4]
    f = ops.fuse_nsf(planes, None, w, "min-max")
    # min-max of every system lies in [0,1] -> fused in [0, sum w]; each row attains 0..1 per system
    assert float(f.min()) >= 0.0 and float(f.max()) <= 1.0 + 1e-6
    ref = torch.zeros((Q, N), device="cuda")
    for p, ww in zip(planes, w):
        mn, mx = p.min(1, keepdim=True).values, p.max(1, keepdim=True).values
        ref = ref + ((p - mn) / (mx - mn)) * torch.tensor(ww, dtype=torch.float32, device="cuda")
    assert bool((f == ref).all())                                                    # same unfused fp32 ops
    # idempotence: min-max of an already min-max-normalised single system is itself
    one = ops.fuse_nsf([planes[0]], None, [1.0], "min-max")
    one_p = ops.alloc_plane(Q, N, torch.float32, "cuda"); one_p.copy_(one)
    two = ops.fuse_nsf([one_p], None, [1.0], "min-max")
    assert bool((two == one_p).all())
    z = ops.fuse_nsf([planes[1]], None, [1.0], "z-score")
    assert float(z.mean(1).abs().max()) < 1e-4 and float((z.std(1) - 1).abs().max()) < 1e-4


def test_full_size_cos_linearity(ops):
    Q, N, d = 1024, 27942, 768
    g = torch.Generator(device="cuda").manual_seed(2)
    Qe = torch.randn((Q, d), generator=g, device="cuda"); De = torch.randn((N, d), generator=g, device="cuda")
    S = ops.cos_scores(Qe, De)
    assert float(S.abs().max()) <= 1.0 + 1e-5
    ref = torch.nn.functional.normalize(Qe[:64].double()) @ torch.nn.functional.normalize(De.double()).t()
    assert float((S[:64].double() - ref).abs().max()) <= 2e-6
    S2 = ops.cos_scores(Qe * 3.0, De * 0.5)              # cosine is scale-invariant
    assert float((S2 - S).abs().max()) <= 1e-6


# ---- the CLI path end to end (synthetic LLeQA-shaped data, tiny random-init encoders) ---------------------------
def test_cli_main_end_to_end(tmp_path):
    import pandas as pd
    from fusion_amd.retrievers.hybrid import build_parser, main
    out = str(tmp_path)
    base = f"--data_split test --models_domain legal --synthetic 600,6 --output_dir {out}".split()
    a, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --run_splade --run_colbert --fusion rrf --normalization none".split())
    sc = main(a)
    assert set(sc) >= {"recall@500", "map@10", "r-precision"} and 0.0 <= sc["recall@500"] <= 1.0
    a, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --fusion nsf --normalization z-score".split())
    assert 0.0 <= main(a)["recall@1000"] <= 1.0
    # score-distribution analysis (writes the quantile tables) ...
    a, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --fusion nsf --normalization none --analyze_score_distributions".split())
    main(a)
    t = pd.read_csv(os.path.join(out, "score_distributions_none_indomain_1k.csv"))
    # N in [1000, 10000, 100000, len(corpus)=600]: round(N/1e3) names both the first and the last table "1k" (hybrid.py:396)
    assert list(t.columns) == ["bm25", "dpr"] and len(t) == 601 and (np.diff(t["dpr"].values) >= 0).all()
    os.rename(os.path.join(out, "score_distributions_none_indomain_1k.csv"), os.path.join(out, "score_distributions_raw_indomain_28k.csv"))
    # ... which percentile-rank fusion then consumes (hybrid.py:451), and the weight sweep writes the reference's CSV schema
    a, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --fusion nsf --normalization percentile-rank".split())
    assert 0.0 <= main(a)["recall@1000"] <= 1.0
    a, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --fusion nsf --normalization min-max --tune_linear_fusion_weight".split())
    rows = main(a)
    assert os.listdir(os.path.join(out, "corpus_cache"))          # N2: encoded corpora are kept on disk ...
    a2, _ = build_parser().parse_known_args(base + "--run_bm25 --run_dpr --run_monobert --rerank_topk 20 --fusion rrf --normalization none".split())
    sc2 = main(a2)                                                 # ... reused here, plus the monoBERT rerank stage (N3)
    assert 0.0 <= sc2["recall@1000"] <= 1.0
    df = pd.read_csv(os.path.join(out, "nsf_min-max_indomain.csv"))
    assert len(rows) == len(df) == 21 and list(df.columns)[-2:] == ["weight_bm25", "weight_dpr"] and "recall@10" in df.columns


# ---- edge cases ---------------------------------------------------------------------------------------------------
def test_edge_cases_empty_and_degenerate(ops, oracle):
    e = torch.empty((0, 10), device="cuda")
    o, k, r = ops.sort_rows_desc(e, want_rank=True)
    assert o.shape == (0, 10) and r.shape == (0, 10)
    one = torch.tensor([[3.0]], device="cuda")
    o, k, r = ops.sort_rows_desc(one, want_rank=True)
    assert o.tolist() == [[0]] and k.tolist() == [[3.0]] and r.tolist() == [[0]]
    # all-equal, all-NaN, +-inf, denormals: stable order = ascending position; NaN first
    rows = np.zeros((5, 3000), dtype=np.float32)
    rows[1] = np.nan
    rows[2, ::2] = np.inf; rows[2, 1::2] = -np.inf
    rows[3] = np.float32(1e-42) * np.arange(3000)             # subnormals, ascending
    rows[4, :10] = [np.nan, 1.0, -0.0, 0.0, np.inf, -np.inf, 1.0, np.nan, -1e-45, 1e-45]
    o, k, r = ops.sort_rows_desc(plane(ops, rows), want_rank=True)
    eo, ek, er = oracle.sort_rows_desc(rows, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo)
    np.testing.assert_array_equal(r.cpu().numpy(), er)
    assert (o.cpu().numpy()[0] == np.arange(3000)).all()
    # fusion of an empty batch / zero-length rows is a no-op, not an error
    z = ops.fuse_rank([torch.empty((0, 8), dtype=torch.int32, device="cuda")], torch.empty((1, 0), dtype=torch.int32, device="cuda"), "rrf")
    assert z.shape == (0, 8)
    # top-k with k > n pads with (-inf, -1)
    s, i = ops.topk_rows(plane(ops, np.array([[0.5, 2.0, 1.0]], dtype=np.float32)), 5, id_base=10)
    assert s.tolist()[0][:3] == [2.0, 1.0, 0.5] and i.tolist()[0] == [11, 12, 10, -1, -1] and s.tolist()[0][3] == float("-inf")


def test_fuse_rank_single_system_and_no_coverage(ops, oracle):
    """A document listed by no system gets -inf and never appears in the fused list."""
    ranks = np.array([[0, -1, 1, -1]], dtype=np.int32)
    lens = np.array([[2]], dtype=np.int32)
    got = ops.fuse_rank([plane(ops, ranks)], dev(lens), "bcf").cpu().numpy()
    np.testing.assert_array_equal(got, oracle.fuse_rank([ranks], lens, "bcf"))
    assert got[0, 1] == -np.inf and got[0, 0] == (2 - 0 + 1) / 2          # Borda (n - idx + 1)/n, sic (hybrid.py:249)
    order = np.array([[0, 2, -1, -1]], dtype=np.int32)
    ins, U = ops.insertion_order([plane(ops, order)], dev(lens), 4)
    assert U.tolist() == [2] and ins.cpu().numpy()[0, :2].tolist() == [0, 2]


def test_maxsim_full_length_docs_vs_torch(ops):
    """LLeQA-shaped documents (lengths ~ N(300,120) clipped to [16,512]) against an independent fp32 torch reference."""
    rng = np.random.default_rng(5)
    N, Q = 600, 16
    lens = np.clip(rng.normal(300, 120, N), 16, 512).astype(np.int64)
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    g = torch.Generator(device="cuda").manual_seed(5)
    Dtok = torch.nn.functional.normalize(torch.randn((int(off[-1]), 128), generator=g, device="cuda"), dim=-1).half()
    Qtok = torch.nn.functional.normalize(torch.randn((Q, 64, 128), generator=g, device="cuda"), dim=-1).half()
    got = ops.maxsim(Qtok, Dtok, torch.from_numpy(off).cuda(), max_doc_len=512)
    ref = torch.empty((Q, N), device="cuda")
    Qf = Qtok.float()
    for j in range(N):
        d = Dtok[off[j]:off[j + 1]].float()                       # [L, 128]
        ref[:, j] = torch.einsum("qid,ld->qil", Qf, d).max(dim=2).values.sum(dim=1)
    assert float((got - ref).abs().max()) <= 2e-4                   # 64 maxima of unit-vector dot products, fp32 accumulate


def test_recall_at_500_identical_to_oracle_pipeline(ops, oracle):
    """north_star acceptance: recall@500 of the GPU pipeline == recall@500 of the CPU restatement on the same inputs
    (LLeQA-sized corpus, synthetic qrels)."""
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator
    from fusion_amd.utils.metrics import Metrics
    rng = np.random.default_rng(21)
    Q, N = 24, 27942
    ids = np.arange(1, N + 1)
    bm = np.maximum(0, rng.gamma(0.5, 4.0, (Q, N)) - 2).astype(np.float32)        # ~40 % exact zeros: many ties
    dp = rng.uniform(-0.2, 0.9, (Q, N)).astype(np.float32)
    labels = [rng.choice(ids, size=int(rng.integers(1, 6)), replace=False).tolist() for _ in range(Q)]
    systems, o_rank, o_order = {}, [], []
    for name, sc in (("bm25", bm), ("dpr", dp)):
        pl = plane(ops, sc)
        od, _, rk = ops.sort_rows_desc(pl, want_rank=True)
        systems[name] = RankedSystem(scores=pl, order=od, rank=rk, lens=torch.full((Q,), N, dtype=torch.int32, device="cuda"), ids=ids)
        eo, _, er = oracle.sort_rows_desc(sc, want_rank=True)
        o_rank.append(er); o_order.append(eo)
    ev = Metrics(recall_at_k=[10, 500])
    for method, norm, w in (("rrf", None, None), ("bcf", None, None), ("nsf", "min-max", {"bm25": 0.3, "dpr": 0.7})):
        fused = Aggregator.fuse(systems, method, norm, w, {}, as_device=True)
        got = ev.compute_all_metrics(labels, fused.predictions())
        if method == "nsf":
            f = oracle.fuse_nsf([bm, dp], None, [w["bm25"], w["dpr"]], norm)
        else:
            f = oracle.fuse_rank(o_rank, np.full((2, Q), N, dtype=np.int32), method)
        eo, _ = oracle.sort_rows_desc(f, init_order=o_order[0])
        exp = ev.compute_all_metrics(labels, [ids[eo[q]].tolist() for q in range(Q)])
        assert got == exp, (method, got, exp)                                     # identical, not approximately equal
        np.testing.assert_array_equal(fused.order.cpu().numpy(), eo)              # because the ranked lists are identical


# ---- encoder side (packed token rows): fz_attn_varlen_f32 / fz_add_layernorm_f32 / fz_segment_mean_f32 -----------
# floating point: compared with the plain torch fp32 maths of the same op; tolerances are stated per test.
@pytest.mark.parametrize("lengths,H", [([1], 1), ([5, 70, 32, 33, 1], 3), ([64] * 7 + [9, 17], 12), ([130, 512, 31], 2)])
def test_attn_varlen_vs_torch(ops, lengths, H):
    g = torch.Generator(device="cuda").manual_seed(3)
    T = sum(lengths)
    qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda") * 1.5
    strips, cu = ops.attn_strips(lengths)
    out = ops.attn_varlen(qkv, torch.from_numpy(strips).cuda(), H)
    ref = torch.empty_like(out)
    for b, L in enumerate(lengths):
        blk = qkv[cu[b]: cu[b] + L].double().view(L, 3, H, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        p = torch.softmax(q @ k.transpose(1, 2) / 8.0, -1)
        ref[cu[b]: cu[b] + L] = (p @ v).transpose(0, 1).reshape(L, H * 64).float()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())   # fp32 products and sums over <= 512 keys


@pytest.mark.parametrize("rows,d,with_res", [(1, 4, False), (7, 768, True), (1000, 768, False), (5, 1024, True), (3, 2048, True), (9, 100, True)])
def test_add_layernorm_vs_torch(ops, rows, d, with_res):
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((rows, d), generator=g, device="cuda") * 3 + 1
    res = torch.randn((rows, d), generator=g, device="cuda") if with_res else None
    gamma, beta = torch.randn(d, generator=g, device="cuda"), torch.randn(d, generator=g, device="cuda")
    out = ops.add_layernorm(x, res, gamma, beta, 1e-5)
    ref = torch.nn.functional.layer_norm((x + res if with_res else x).double(), (d,), gamma.double(), beta.double(), 1e-5)
    assert (out.double() - ref).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item())


def test_segment_mean_vs_torch(ops):
    g = torch.Generator(device="cuda").manual_seed(5)
    lengths = [3, 0, 64, 1, 200]
    x = torch.randn((sum(lengths), 768), generator=g, device="cuda")
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device="cuda")
    out = ops.segment_mean(x, cu)
    for b, L in enumerate(lengths):
        ref = x[int(cu[b]): int(cu[b + 1])].double().mean(0) if L else torch.zeros(768, dtype=torch.float64, device="cuda")
        assert (out[b].double() - ref).abs().max().item() <= 1e-6


def test_packed_encoder_matches_hf_forward():
    """PackedBertForward (HIP attention / LayerNorm / pooling on packed rows) == the HF module + mean pooling, 5e-5 abs on
    embeddings of magnitude ~1 (fp32 throughout; different summation orders in 2-12 layers)."""
    from fusion_amd import encoders
    cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
    torch.manual_seed(0)
    enc = encoders.DenseEncoder(encoders._backbone(cfg), encoders.HashTokenizer(cfg["vocab_size"]), "cuda")
    rng = np.random.default_rng(0)
    n, Lmax = 37, 64
    lens = rng.integers(2, Lmax + 1, size=n)
    lens[0] = Lmax
    ids = np.full((n, Lmax), cfg["pad_token_id"], dtype=np.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = rng.integers(7, cfg["vocab_size"], size=L)
    I = torch.from_numpy(ids).cuda()
    M = (torch.arange(Lmax, device="cuda")[None, :] < torch.from_numpy(lens).cuda()[:, None]).long()
    a = enc.encode_ids(I, M)
    b = enc.encode_ids_packed(I, lens)
    assert b.shape == a.shape and (a - b).abs().max().item() <= 5e-5
    # the text entry point takes the packed path (token-budgeted sub-batches) and agrees with the HF forward sentence by sentence
    texts = [" ".join(f"w{rng.integers(0, 300)}" for _ in range(int(k))) for k in rng.integers(1, 150, size=23)]
    enc.packed_tokens = 256
    e = enc.encode(texts, batch_size=4)
    for i, t in enumerate(texts):
        ids1, m1 = enc.tokenizer([t], enc.max_doc_length)
        assert (enc.encode_ids(ids1.cuda(), m1.cuda())[0] - e[i]).abs().max().item() <= 5e-5
    # with GEMM tuning on, the packed row count is padded to a multiple of 512 (tuned solutions are keyed by shape): same result
    import torch.cuda.tunable as tn
    real = tn.is_enabled
    try:
        tn.is_enabled = lambda: True          # the padding path only; no tuning is triggered
        c = enc.encode_ids_packed(I, lens)
    finally:
        tn.is_enabled = real
    assert (c - b).abs().max().item() <= 1e-6
    with pytest.raises(ValueError):
        encoders.random_init("dpr", "cuda", size="tiny").encode_ids_packed(I[:, :32] % 500, np.minimum(lens, 32))   # head_dim 16


def test_packed_splade_and_colbert_match_padded_forward(ops):
    """SPLADE and ColBERT encoders on the padding-free forward == the HF module on padded batches, sentence by sentence
    (1e-4 abs: fp32 everywhere, ColBERT token vectors are unit-norm fp16 -> half an fp16 ulp of 1.0 is 5e-4)."""
    from fusion_amd import encoders
    cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
    rng = np.random.default_rng(1)
    texts = [" ".join(f"w{rng.integers(0, 40)}" for _ in range(int(k))) for k in rng.integers(1, 90, size=19)]
    torch.manual_seed(1)
    tok = encoders.HashTokenizer(cfg["vocab_size"])
    sp = encoders.SpladeEncoder(encoders._backbone(cfg, mlm=True), tok, "cuda")
    sp.packed_tokens = 300
    got = sp.encode(texts, batch_size=4, query_mode=False)
    assert sp._packed is not None and got.shape == (len(texts), cfg["vocab_size"])
    for i, t in enumerate(texts):
        ids1, m1 = tok([t], sp.max_doc_length)
        assert (sp.encode_ids(ids1.cuda(), m1.cuda())[0] - got[i]).abs().max().item() <= 1e-4
    punct = (tok._tok("w3"), tok._tok("w7"))
    cb = encoders.ColbertEncoder(encoders._backbone(cfg), tok, "cuda", punct_ids=punct, amp=False)     # float32 end to end (amp: test_gpu_parity_r3)
    cb.packed_tokens = 300
    Dtok, Doff = cb.encode_docs(texts, batch_size=4)
    assert cb._packed is not None and Doff.shape == (len(texts) + 1,) and int(Doff[-1]) == Dtok.shape[0]
    for i, t in enumerate(texts):
        ids1, m1 = tok([t], cb.max_doc_length)
        keep = ~torch.isin(ids1[0], torch.tensor(punct))
        ref = cb._tokens(ids1.cuda(), m1.cuda())[0][keep.cuda()]
        mine = Dtok[int(Doff[i]): int(Doff[i + 1])].float()
        assert mine.shape == ref.shape and (mine - ref).abs().max().item() <= 6e-4
    Qtok = cb.encode_queries(texts[:5])
    for i, t in enumerate(texts[:5]):
        ids1, m1 = tok([t], cb.max_query_length, True)
        ids1 = torch.where(m1.bool(), ids1, torch.full_like(ids1, tok.mask_token_id))
        ref = cb._tokens(ids1.cuda(), torch.ones_like(m1).cuda())[0]
        assert (Qtok[i].float() - ref).abs().max().item() <= 6e-4
    # the packed document tokens feed the MaxSim kernel as they are
    s = ops.maxsim(Qtok, Dtok, Doff, max_doc_len=cb.max_doc_length)
    assert s.shape == (5, len(texts)) and torch.isfinite(s).all()


def test_segment_splade_max_vs_torch(ops):
    g = torch.Generator(device="cuda").manual_seed(6)
    lengths = [3, 0, 64, 1]
    for d in (32005, 512):       # unaligned vocabulary (scalar path) and aligned (vector path)
        x = torch.randn((sum(lengths), d), generator=g, device="cuda") * 3
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lengths)]), dtype=torch.int32, device="cuda")
        out = ops.segment_splade_max(x, cu)
        for b, L in enumerate(lengths):
            ref = torch.log1p(torch.relu(x[int(cu[b]): int(cu[b + 1])])).amax(0) if L else torch.zeros(d, device="cuda")
            assert (out[b] - ref).abs().max().item() <= 1e-6


def test_ops_reject_mismatched_shapes_before_launch(ops):
    """A shape the kernel would index out of bounds with is a ValueError on the host, never a launch."""
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    a, b = ops.alloc_plane(4, 100, torch.int32, "cuda", fill=0), ops.alloc_plane(4, 90, torch.int32, "cuda", fill=0)
    lens = torch.full((2, 4), 100, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.fuse_rank([a, b], lens, "rrf")
    with pytest.raises(ValueError):
        ops.fuse_rank([a, a], lens[:1], "rrf")
    p = ops.alloc_plane(4, 100, torch.float32, "cuda", fill=0.0)
    with pytest.raises(ValueError):
        ops.fuse_nsf([p, p], None, [0.5], "min-max")
    with pytest.raises(ValueError):
        ops.fuse_nsf([p, p], None, [0.5, 0.5], "min-max", out=z(4, 101))
    with pytest.raises(ValueError):
        ops.insertion_order([a, a], lens, 101)
    with pytest.raises(ValueError):
        ops.sort_rows_desc(p, row_len=torch.zeros(3, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        ops.add_layernorm(z(3, 8), None, z(4), z(8), 1e-5)
    with pytest.raises(ValueError):
        ops.topk_merge(z(2, 3, 5), torch.zeros((2, 3, 4), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        ops.dot_scores(z(3, 8), z(5, 8), out=z(3, 6))


def test_full_size_encoder_kernels(ops):
    """The encoder-side kernels at the bench's shape (1024 queries, 12 heads, ~37 k packed rows): attention against a float64
    torch reference on a sample of sequences (1e-5 relative to the largest output), residual+LayerNorm and mean pooling against
    torch on every row."""
    rng = np.random.default_rng(11)
    lens = rng.integers(8, 65, 1024)
    T, H = int(lens.sum()), 12
    g = torch.Generator(device="cuda").manual_seed(11)
    qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda") * 1.2
    strips, cu = ops.attn_strips(lens)
    out = ops.attn_varlen(qkv, torch.from_numpy(strips).cuda(), H)
    assert torch.isfinite(out).all()
    for b in rng.choice(1024, size=40, replace=False).tolist() + [int(np.argmax(lens)), int(np.argmin(lens))]:
        L = int(lens[b])
        blk = qkv[cu[b]: cu[b] + L].double().view(L, 3, H, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(1, 2) / 8.0, -1) @ v).transpose(0, 1).reshape(L, H * 64)
        assert (out[cu[b]: cu[b] + L].double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    x = out
    res = torch.randn((T, 768), generator=g, device="cuda")
    gamma, beta = torch.randn(768, generator=g, device="cuda"), torch.randn(768, generator=g, device="cuda")
    y = ops.add_layernorm(x, res, gamma, beta, 1e-5)
    ref = torch.nn.functional.layer_norm(x + res, (768,), gamma, beta, 1e-5)
    assert (y - ref).abs().max().item() <= 2e-5
    pooled = ops.segment_mean(y, torch.from_numpy(cu).cuda())
    seg = torch.repeat_interleave(torch.arange(1024, device="cuda"), torch.from_numpy(lens).cuda())
    ref_p = torch.zeros((1024, 768), device="cuda", dtype=torch.float64).index_add_(0, seg, y.double()) / torch.from_numpy(lens).cuda().double()[:, None]
    assert (pooled.double() - ref_p).abs().max().item() <= 1e-5


def test_randomised_soak_short():
    """15 s of tests/fuzz_gpu.py (random shapes / lengths / ties / partial lists for every kernel family against the oracle
    or float64 torch); longer runs: python tests/fuzz_gpu.py --seconds 300 --seed N."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_gpu.py"), "--seconds", "15", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK:" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_all_empty_inputs_have_defined_results(ops):
    """Empty tensors carry no data pointer: the all-empty cases are answered before any pointer is needed."""
    cu = torch.zeros(4, dtype=torch.int32, device="cuda")                      # three empty sequences
    x = torch.zeros((0, 768), device="cuda")
    assert not ops.segment_mean(x, cu).any() and ops.segment_mean(x, cu).shape == (3, 768)
    assert not ops.segment_splade_max(x, cu).any()
    Qtok = torch.zeros((2, 32, 128), dtype=torch.float16, device="cuda")
    Doff = torch.zeros(5, dtype=torch.int64, device="cuda")                    # four empty documents
    s = ops.maxsim(Qtok, torch.zeros((0, 128), dtype=torch.float16, device="cuda"), Doff, max_doc_len=1)
    assert s.shape == (2, 4) and not s.any()


def test_embed_layernorm_vs_torch(ops):
    g = torch.Generator(device="cuda").manual_seed(8)
    V, P, d, rows = 500, 70, 768, 333
    word, pos = torch.randn((V, d), generator=g, device="cuda"), torch.randn((P, d), generator=g, device="cuda")
    type0, gamma, beta = (torch.randn(d, generator=g, device="cuda") for _ in range(3))
    ids = torch.randint(0, V, (rows,), generator=g, device="cuda")
    pid = torch.randint(0, P, (rows,), generator=g, device="cuda")
    buf = torch.zeros((rows + 7, d), device="cuda")
    out = ops.embed_layernorm(word, pos, type0, ids, pid, gamma, beta, 1e-5, out=buf)
    ref = torch.nn.functional.layer_norm((word[ids] + type0 + pos[pid]).double(), (d,), gamma.double(), beta.double(), 1e-5)
    assert out is buf and not buf[rows:].any()                     # rows past the batch are left alone
    assert (buf[:rows].double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_gelu_matches_torch(ops):
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((1234, 3072), generator=g, device="cuda") * 3
    ref = torch.nn.functional.gelu(x)
    y = ops.gelu_(x.clone())
    assert (y - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())   # same float expression; erff may differ in the last ulp
    assert ops.gelu_(torch.zeros((0, 8), device="cuda")).numel() == 0


def test_packed_cross_encoder_matches_padded_forward():
    """monoBERT rerank scores on the padding-free forward == the HF sequence classifier on padded batches (1e-4 abs)."""
    from fusion_amd import encoders
    from transformers import CamembertConfig, CamembertForSequenceClassification
    cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
    torch.manual_seed(2)
    tok = encoders.HashTokenizer(cfg["vocab_size"])
    ce = encoders.CrossEncoder(CamembertForSequenceClassification(CamembertConfig(num_labels=1, **cfg)), tok, "cuda")
    rng = np.random.default_rng(2)
    pairs = [(" ".join(f"q{rng.integers(0, 30)}" for _ in range(int(rng.integers(1, 12)))),
              " ".join(f"d{rng.integers(0, 60)}" for _ in range(int(rng.integers(1, 100))))) for _ in range(21)]
    ce.packed_tokens = 300
    got = ce.predict(pairs, batch_size=4)
    assert ce._packed is not None and got.shape == (21,)
    for i, (q, d) in enumerate(pairs):
        ids1, m1 = tok([q + " </s> " + d], ce.max_length)
        with torch.no_grad():
            ref = ce.model(input_ids=ids1.cuda(), attention_mask=m1.cuda()).logits[0, 0]
        assert abs(float(ref) - float(got[i])) <= 1e-4
