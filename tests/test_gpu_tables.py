"""Round 4 GPU parity: percentile-rank / normal-curve-equivalent fusion at the table sizes the reference READS (hybrid.py:412,451:
27,943 quantiles per system; :374: 10,001) -- csrc/tables.hip, one system's table LDS-resident at a time.  Held against the
reference's own outputs (tests/golden/pr28k_*, tune10k_*), against the oracle on fresh inputs, and against the two older kernels of
fz_fuse_nsf_f32 (all tables in LDS / global-memory search) bit for bit; every test pins WHICH kernel took the call.  Everything goes
through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import load_lists, quantile_table

pytestmark = pytest.mark.gpu

PR28K = os.path.join(GOLDEN, "pr28k_seed60_S4_Q1_N27942.npz")
TUNE10K = os.path.join(GOLDEN, "tune10k_seed22_S2_Q4_N257.npz")
NORMS = ["percentile-rank", "normal-curve-equivalent"]


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


def same_bits(a, b):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ---- the reference's own outputs ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("norm,tol", [("percentile-rank", 0.0), ("normal-curve-equivalent", 1e-4)])
def test_aggregator_fuse_matches_reference_at_28k_tables(ops, norm, tol):
    """Aggregator.fuse on one full LLeQA row, S = 4 (ColBERT list cut to 60 %), 27,943-entry tables: the reference's ranked list and
    scores -- percentile-rank bit for bit, NCE within 1e-4 -- through the kernel that swaps the tables through LDS."""
    from fusion_amd.retrievers.hybrid import Aggregator
    z = np.load(PR28K)
    systems, lists, Q = load_lists(z)
    weights = {s: float(w) for s, w in zip(systems, z["weights"])}
    distr = {s: z[f"distr_{s}"] for s in systems}
    ops.last_tables_path = None
    got = Aggregator.fuse(lists, method="nsf", normalization=norm, linear_weights=weights, percentile_distributions=distr)
    assert ops.last_tables_path == "lds-swap"
    e_ids, e_sc, n = z[f"out_ids__nsf__{norm}"][0], z[f"out_scores__nsf__{norm}"][0].astype(np.float64), int(z[f"out_len__nsf__{norm}"][0])
    assert len(got) == 1 and len(got[0]) == n
    g_ids = np.array([x["corpus_id"] for x in got[0]], dtype=np.int64)
    g_sc = np.array([float(x["score"]) for x in got[0]], dtype=np.float64)
    if tol == 0.0:
        np.testing.assert_array_equal(g_ids, e_ids[:n])
        np.testing.assert_array_equal(g_sc, e_sc[:n])
    else:
        assert sorted(g_ids.tolist()) == sorted(e_ids[:n].tolist())
        exp = {int(i): float(s) for i, s in zip(e_ids[:n], e_sc[:n])}
        ref = np.array([exp[int(i)] for i in g_ids])
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(g_sc), fin) and np.array_equal(g_sc[~fin], ref[~fin]) and (~fin).sum() >= 4
        assert np.max(np.abs(g_sc[fin] - ref[fin])) <= tol
        assert np.all(g_sc[:-1] >= g_sc[1:])


@pytest.mark.parametrize("norm", NORMS)
def test_aggregator_tune_matches_reference_loop_at_10k_tables(ops, norm):
    """Aggregator.tune == the reference's weight-grid loop (hybrid.py:404-426) with 10,001-entry tables, every metric <= 1e-12.  tune()
    normalises system by system (S = 1 per call): each call takes the swap kernel with its table loaded once."""
    from fusion_amd.retrievers.hybrid import Aggregator, weight_grid
    z = np.load(TUNE10K, allow_pickle=False)
    systems, lists, Q = load_lists(z)
    labels = [[int(x) for x in str(s).split(",")] for s in z["labels"]]
    distr = {s: z[f"distr_{s}"] for s in systems}
    grid = weight_grid(systems)
    assert np.array_equal(np.array([[w[s] for s in systems] for w in grid]), z["weights"])
    names = [str(x) for x in z["metric_names"]]
    ops.last_tables_path = None
    got = Aggregator.tune(lists, norm, grid, labels, distr)
    assert ops.last_tables_path == "lds-swap"
    G = np.array([[float(g[k]) for k in names] for g in got])
    rows = np.all(z["weights"] != 0.0, axis=1) if norm == "normal-curve-equivalent" else slice(None)
    assert np.max(np.abs(G - z[f"metrics__{norm}"])[rows]) <= 1e-12


# ---- raw ops.fuse_nsf: the swap kernel against the oracle and against fz_fuse_nsf_f32's kernels --------------------------------------
def _tables(rng, S, P, kind="quantile"):
    out = []
    for s in range(S):
        if kind == "quantile":   # quantiles of a distribution shaped like the system's scores: dense where the scores are
            pool = [np.maximum(0.0, rng.gamma(0.5, 4.0, 4 * P) - 2.0), rng.uniform(-0.2, 0.9, 4 * P), rng.normal(20.0, 4.0, 4 * P),
                    np.log1p(np.maximum(rng.normal(0, 1, 4 * P), 0))][s % 4]
            pool = pool[pool != 0.0]
            t = quantile_table(pool, P).astype(np.float32)
        elif kind == "dups":     # long runs of equal quantiles (a discrete-valued system)
            t = np.sort(rng.integers(0, max(2, P // 50), P)).astype(np.float32) * np.float32(0.125)
        elif kind == "const":
            t = np.full(P, 1.5, dtype=np.float32)
        elif kind == "tiny":     # the whole table inside a few ulps: the bucket table degenerates, the search must not
            t = np.sort(np.float32(1.0) + rng.integers(0, 8, P).astype(np.float32) * np.float32(2.0 ** -23))
        elif kind == "huge":     # distances round to plateaus: |score| dwarfs the spacing
            t = np.sort(np.float32(3.0e7) + rng.integers(0, 64, P).astype(np.float32) * np.float32(2.0))
        else:
            raise ValueError(kind)
        out.append(np.ascontiguousarray(t))
    return out


def _scores(rng, tabs, Q, N, specials=True):
    planes = []
    for t in tabs:
        lo, hi = float(t[0]), float(t[-1])
        span = max(hi - lo, 1e-3)
        x = rng.uniform(lo - 0.05 * span, hi + 0.05 * span, (Q, N)).astype(np.float32)
        k = rng.integers(0, len(t), (Q, max(1, N // 7)))
        cols = rng.integers(0, N, (Q, max(1, N // 7)))
        for q in range(Q):
            x[q, cols[q]] = t[k[q]]                                   # scores that ARE table entries
        if specials and N >= 8:
            x[0, 0], x[0, 1], x[0, 2] = np.nan, np.inf, -np.inf
            x[Q - 1, N - 1] = np.float32(lo) - np.float32(1.0)
            x[Q - 1, N - 2] = np.float32(hi) + np.float32(1.0)
            if len(t) > 3:
                x[0, 3] = np.float32((np.float64(t[1]) + np.float64(t[2])) / 2)
        planes.append(x)
    return planes


CASES = [
    # S, Q, N, P, table kind, partial systems
    (4, 3, 27942, 27943, "quantile", (3,)),       # the reference's size, the ColBERT list partial
    (4, 300, 1000, 27943, "quantile", ()),        # more items than workgroups: the persistent loop, a table swap per (item, system)
    (1, 5, 27942, 27943, "quantile", ()),         # tune()'s call: one system, the table loaded once
    (2, 2, 30001, 10001, "quantile", (1,)),       # rows longer than one item (28,672 columns): two chunks per row
    (3, 4, 777, 10001, "dups", (0,)),             # long runs of duplicated quantiles: the FIRST index of the run
    (2, 3, 257, 20000, "const", ()),              # every quantile equal
    (2, 3, 258, 30000, "tiny", ()),               # a table a few ulps wide
    (2, 3, 1023, 12000, "huge", ()),              # rounding plateaus
    (1, 1, 1, 27943, "quantile", ()),
    (4, 2, 513, 38000, "quantile", ()),           # the longest table LDS takes (a smaller bucket table goes with it)
]


@pytest.mark.parametrize("S,Q,N,P,kind,partial", CASES)
@pytest.mark.parametrize("norm", NORMS)
def test_swap_kernel_equals_global_search_and_oracle(ops, oracle, S, Q, N, P, kind, partial, norm):
    rng = np.random.default_rng(S * 1000003 + N * 31 + P)
    tabs = _tables(rng, S, P, kind)
    xs = _scores(rng, tabs, Q, N)
    ranks = [None] * S
    for s in partial:                                                  # a partial list: ~40 % of the documents absent
        r = np.where(rng.random((Q, N)) < 0.6, 1, -1).astype(np.int32)
        ranks[s] = r
    w = [0.15, 0.35, 0.3, 0.2][:S] if S > 1 else [1.0]
    P_d = [dev(t) for t in tabs]
    planes = [plane_of(ops, x) for x in xs]
    rk = None if not partial else [None if r is None else plane_of(ops, r) for r in ranks]
    ops.last_tables_path = None
    got = ops.fuse_nsf(planes, rk, w, norm, P_d)
    assert ops.last_tables_path == "lds-swap"
    old = ops.fuse_nsf(planes, rk, w, norm, P_d, tables=False)         # fz_fuse_nsf_f32: the global-memory search at these sizes
    assert ops.last_tables_path == "row" or N > 28672
    assert same_bits(got, old)
    vb = None if not partial else [None if r is None else ops.rank_to_bitmap(plane_of(ops, r)) for r in ranks]
    if vb is not None:                                                 # validity as bitmaps: the form fuse_device hands over
        assert same_bits(ops.fuse_nsf(planes, None, w, norm, P_d, valid_bits=vb), got)
    if Q * N * P <= 4 * 27942 * 27943:                                 # the oracle's search is the reference's O(N P) scan
        exp = oracle.fuse_nsf(xs, None if not partial else ranks, w, norm, tabs)
        g = got.cpu().numpy()
        if norm == "percentile-rank":
            assert np.array_equal(g.view(np.uint32), exp.view(np.uint32))
        else:
            fin = np.isfinite(exp)
            assert np.array_equal(np.isfinite(g), fin) and np.array_equal(np.isnan(g), np.isnan(exp))
            assert np.array_equal(g[np.isinf(exp)], exp[np.isinf(exp)])
            assert np.max(np.abs(g[fin] - exp[fin]), initial=0.0) <= 1e-4


@pytest.mark.parametrize("norm", NORMS)
def test_prepared_tables_are_reusable_and_small_tables_keep_their_kernel(ops, norm):
    rng = np.random.default_rng(5)
    tabs = _tables(rng, 4, 27943)
    P_d = [dev(t) for t in tabs]
    prep = ops.nsf_tables_prepare(P_d, norm)
    assert prep is not None and prep.matches(P_d, norm)
    outs = []
    for Q in (2, 7):
        planes = [plane_of(ops, x) for x in _scores(rng, tabs, Q, 5000)]
        a = ops.fuse_nsf(planes, None, [0.25] * 4, norm, P_d, tables=prep)
        b = ops.fuse_nsf(planes, None, [0.25] * 4, norm, P_d)
        assert same_bits(a, b)
        outs.append(a)
    # tables that all fit LDS at once stay on the round-3 kernel; tables beyond LDS go to the global-memory search
    small = [dev(t) for t in _tables(rng, 4, 1001)]
    planes = [plane_of(ops, x) for x in _scores(rng, [t.cpu().numpy() for t in small], 3, 999)]
    ops.fuse_nsf(planes, None, [0.25] * 4, norm, small)
    assert ops.last_tables_path == "lds-all"
    big = [dev(t) for t in _tables(rng, 1, 50000)]
    assert ops.nsf_tables_prepare(big, norm) is None
    planes = [plane_of(ops, x) for x in _scores(rng, [big[0].cpu().numpy()], 2, 300)]
    ops.fuse_nsf(planes, None, [1.0], norm, big)
    assert ops.last_tables_path == "row"


def test_tables_c_abi_rejects_what_it_must(ops):
    from fusion_amd import _lib
    L = _lib.lib()
    P = (C.c_int32 * 2)(27943, 27943)
    nb = L.fz_nsf_tables_workspace_bytes(2, P, 4)
    assert nb > 2 * 27943 * 4 and L.fz_nsf_tables_workspace_bytes(2, P, 5) > nb           # NCE carries the value tables
    assert L.fz_nsf_tables_workspace_bytes(2, P, 1) == 0 and L.fz_nsf_tables_workspace_bytes(0, P, 4) == 0
    assert L.fz_nsf_tables_workspace_bytes(2, (C.c_int32 * 2)(27943, 70000), 4) == 0
    t = torch.linspace(0, 1, 27943, device="cuda")
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    d = ops._ptr_array([t, t])
    assert L.fz_nsf_tables_prepare(d, P, 2, 4, ws.data_ptr(), nb - 1, None) == _lib.FZ_ERR_WORKSPACE
    assert L.fz_nsf_tables_prepare(d, P, 2, 3, ws.data_ptr(), nb, None) == _lib.FZ_ERR_ARG
    assert L.fz_nsf_tables_prepare(d, P, 2, 4, ws.data_ptr(), nb, None) == _lib.FZ_OK
    x = ops.alloc_plane(2, 100, torch.float32, "cuda"); x.normal_()
    out = ops.alloc_plane(2, 100, torch.float32, "cuda")
    pl = ops._ptr_array([x, x])
    w = (C.c_double * 2)(0.5, 0.5)
    args = (pl, None, w, 2, 2, 100, ops._ld(x), 4, d, P, None, 0, out.data_ptr())
    assert L.fz_fuse_nsf_tables_f32(*args, None, 0, None) == _lib.FZ_ERR_WORKSPACE
    assert L.fz_fuse_nsf_tables_f32(*args, ws.data_ptr(), nb, None) == _lib.FZ_OK
    assert L.fz_fuse_nsf_tables_f32(*args[:7], 1, *args[8:], ws.data_ptr(), nb, None) == _lib.FZ_ERR_ARG     # min-max has no tables
    torch.cuda.synchronize()
    assert torch.isfinite(out[:, :100]).all()


# ---- configs[3] with the percentile normaliser of run_hybrid.sh:35, at the LLeQA test split's size -----------------------------------
def test_config4_percentile_weight_grid_at_28k_tables(ops):
    """Aggregator.tune over the whole 1771-vector lattice, S = 4, Q = 195, N = 27,942, 27,943-entry tables == fuse_device +
    run_evaluation on sampled vectors, every metric <= 1e-12 (the sweep and the fusion normalise through the same kernel)."""
    from fusion_amd.retrievers.hybrid import Aggregator, run_evaluation, weight_grid
    from test_gpu_fullsize import _lleqa_systems
    Q, N = 195, 27942
    systems, hidden, ids = _lleqa_systems(ops, Q, N, seed=9)
    rng = np.random.default_rng(1)
    distr = {}
    for name, s in systems.items():      # the `_28k` table of hybrid.py:389-397: quantiles of the pooled non-zero scores
        pool = s.scores[:, :N].flatten()
        pool = pool[pool != 0.0].double()
        idx = torch.linspace(0, pool.numel() - 1, N + 1, device="cuda").round().long()
        distr[name] = torch.sort(pool).values[idx].cpu().numpy()
    top = torch.topk(hidden, 1500, dim=1).indices.cpu().numpy()
    labels = []
    for q in range(Q):
        pick = rng.choice(1500, size=int(rng.integers(1, 6)), replace=False)
        pick[0] = int(rng.integers(0, 30))
        labels.append([int(ids[top[q, j]]) for j in dict.fromkeys(pick.tolist())])
    grid = weight_grid(list(systems))
    ops.last_tables_path = None
    got = Aggregator.tune(systems, "percentile-rank", grid, labels, distr)
    assert ops.last_tables_path == "lds-swap" and len(got) == 1771
    worst = 0.0
    for wi in sorted(set(rng.choice(1771, size=10, replace=False).tolist()) | {0, 1770}):
        fused = Aggregator.fuse_device(systems, "nsf", "percentile-rank", grid[wi], distr)
        exp = run_evaluation(fused.predictions(1000), labels, print2console=False)
        worst = max(worst, max(abs(float(got[wi][m]) - float(exp[m])) for m in exp))
    assert worst <= 1e-12, worst
    # and the equal-weights fusion main() runs (hybrid.py:448-455), float32 products: one flat call over all four systems
    eq = {n: 1 / len(systems) for n in systems}
    ops.last_tables_path = None
    f = Aggregator.fuse_device(systems, "nsf", "percentile-rank", eq, distr)
    assert ops.last_tables_path == "lds-swap"
    r = run_evaluation(f.predictions(1000), labels, print2console=False)
    assert 0.0 < r["recall@500"] <= 1.0


# ---- the top-k SELECTION path of fuse_device through the Aggregator (main() takes it from 512 queries up) -------------------
@pytest.mark.parametrize("method,norm", [("rrf", None), ("bcf", None), ("nsf", "none")])
def test_fuse_device_topk_selection_path_equals_the_head_of_the_full_lists(ops, method, norm, monkeypatch):
    """Aggregator.SELECT_MIN_Q = 1 puts a 7-query batch on the path large batches take: select_topk -> FusedResult(order = [Q, k] columns,
    lens clamped) -> predictions(1000).  Order, scores, lens and predictions equal the head of the full lists; a tie run at the k-th
    place that the candidate buffer cannot hold falls back to the full sort."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    monkeypatch.setattr(Aggregator, "SELECT_MIN_Q", 1)
    rng = np.random.default_rng(11)
    Q, N, k = 7, 27942, 1000
    ids = np.arange(N) + 5
    hidden = rng.normal(0, 1, (Q, N))
    mk = lambda s: (hidden + s * rng.normal(0, 1, (Q, N))).astype(np.float32)
    systems = {"a": _rank_scores(plane_of(ops, np.maximum(mk(1.0), 0.0)), ids, None), "b": _rank_scores(plane_of(ops, mk(0.5)), ids, None)}
    w = {"a": 0.4, "b": 0.6}
    full = Aggregator.fuse_device(systems, method, norm, w, {})
    head = Aggregator.fuse_device(systems, method, norm, w, {}, topk=k)
    # round 5: rrf / bcf rows that one workgroup holds are SORTED (rank fusion in the sort's load phase: cheaper than fuse + select + two
    # small sorts) and cut; the float64 'none' sums still take the selection
    assert Aggregator.last_topk_path == ("select" if method == "nsf" else "sort")
    assert Aggregator.last_rank_fused_sort is (method != "nsf")
    assert head.order.shape == (Q, k) and head.scores.dtype == full.scores.dtype == torch.float64
    np.testing.assert_array_equal(head.order.cpu().numpy(), full.order.cpu().numpy()[:, :k])
    np.testing.assert_array_equal(head.scores.cpu().numpy(), full.scores.cpu().numpy()[:, :k])
    np.testing.assert_array_equal(head.lens.cpu().numpy(), np.full(Q, k))
    assert head.predictions(1000) == full.predictions(1000)
    # two systems that rank every document alike below the head: thousands of equal fused scores at the k-th place -> no selection
    flat = np.zeros((Q, N), dtype=np.float32)
    flat[:, :50] = np.arange(50, 0, -1)
    tied = {"a": _rank_scores(plane_of(ops, flat), ids, None), "b": _rank_scores(plane_of(ops, flat), ids, None)}
    if method == "nsf":      # raw sums: 27,892 documents tie at 0.0
        full = Aggregator.fuse_device(tied, method, norm, w, {})
        head = Aggregator.fuse_device(tied, method, norm, w, {}, topk=k)
        assert Aggregator.last_topk_path == "sort"
        np.testing.assert_array_equal(head.order.cpu().numpy(), full.order.cpu().numpy()[:, :k])
        np.testing.assert_array_equal(head.scores.cpu().numpy(), full.scores.cpu().numpy()[:, :k])


# ---- the acceptance metric under the mixed-precision ColBERT encoder ---------------------------------------
@pytest.mark.parametrize("SHARED,moved", [((4, 6, 8, 12, 16), 0), ((2, 3, 4, 6, 8, 12), 1)])
def test_colbert_mixed_precision_leaves_maxsim_and_fused_recall_where_they_were(ops, SHARED, moved):
    """ColbertEncoder(amp=True) -- colbert-ai's autocast, the GPU default -- against amp=False on a task a RANDOM-INIT encoder can do: every
    gold document BEGINS with a prefix of its query's token sequence (same tokens at the same positions: same word + position embeddings,
    the same local context), the prefix lengths graded (SHARED: 4-16 tokens, and a harder 2-12) so that some gold documents are easy and
    some borderline (measured: MaxSim-only recall@10 0.95 and 0.67; prefixes of 12-60 tokens give 1.0 throughout and test nothing).
    Checked: (1) MaxSim ALONE retrieves them (recall@10 >= 0.5 in both precisions: the test is about ColBERT, not about the three synthetic
    planes next to it), (2) MaxSim-only recall@10 / @100 / @500 agree between the precisions -- exactly on the 4-16 grade; on the 2-12 grade
    ONE of the 192 gold documents sits at the rank-10 border and moves (0.6719 / 0.6771), the other depths are equal -- and the top-500
    lists overlap, (3) so does the recall of the nsf min-max fused lists of a config-4-shaped pipeline (north_star's acceptance metric)."""
    from fusion_amd import encoders
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores, run_evaluation
    rng = np.random.default_rng(17)
    N, Q, Lq, Ld = 3000, 96, 64, 96
    enc = encoders.random_init("colbert", device="cuda", size="base", seed=5)
    V = 32000
    doc_ids = rng.integers(7, V - 1, size=(N, Ld))
    lens = rng.integers(70, Ld + 1, N)
    q_ids = rng.integers(7, V - 1, size=(Q, Lq))
    gold_all = rng.permutation(N)[: 2 * Q].reshape(Q, 2)                # two gold documents per query, no document gold twice
    gold = [g.tolist() for g in gold_all]
    for q, gl in enumerate(gold):
        for j, g in enumerate(gl):
            n_shared = int(rng.choice(SHARED))
            doc_ids[g, 2: 2 + n_shared] = q_ids[q, 2: 2 + n_shared]    # (columns 0-1: <s> and the [Q] / [D] marker slot)
    dids, qids = torch.from_numpy(doc_ids).cuda(), torch.from_numpy(q_ids).cuda()
    ids = np.arange(N) + 1
    hidden = torch.zeros((Q, N), device="cuda")
    for q, gl in enumerate(gold):
        hidden[q, gl] = 3.0
    gen = torch.Generator(device="cuda").manual_seed(3)
    noisy = lambda s: hidden + s * torch.randn((Q, N), generator=gen, device="cuda")
    others = {"bm25": torch.clamp(noisy(1.5), min=0.0), "dpr": torch.tanh(0.3 * noisy(1.2)), "splade": torch.log1p(torch.relu(noisy(1.8)))}
    labels = [[int(ids[g]) for g in gl] for gl in gold]

    def plane(t):
        p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(t); return p
    fused, maxsim = {}, {}
    for amp in (True, False):
        enc.amp = amp
        Dtok, Doff = enc.encode_doc_ids(dids, lens)
        Qtok = enc.encode_query_ids(qids)
        S = ops.maxsim(Qtok, Dtok, Doff, max_doc_len=Ld)
        maxsim[amp] = S
        systems = {n: _rank_scores(plane(t), ids, None) for n, t in others.items()}
        systems["colbert"] = _rank_scores(S, ids, None)
        f = Aggregator.fuse_device(systems, "nsf", "min-max", {"bm25": 0.2, "dpr": 0.2, "splade": 0.2, "colbert": 0.4}, {})
        fused[amp] = run_evaluation(f.predictions(1000), labels, print2console=False)
    o16 = torch.argsort(maxsim[True], dim=1, descending=True, stable=True)[:, :500].cpu().numpy()
    o32 = torch.argsort(maxsim[False], dim=1, descending=True, stable=True)[:, :500].cpu().numpy()
    overlap = float(np.mean([len(set(a.tolist()) & set(b.tolist())) / 500 for a, b in zip(o16, o32)]))
    rel = float(((maxsim[True] - maxsim[False]).abs() / maxsim[False].abs().clamp_min(1e-6)).max())
    solo = {amp: run_evaluation(torch.argsort(maxsim[amp], dim=1, descending=True, stable=True)[:, :1000].add(1).cpu().numpy().tolist(), labels,
                                print2console=False) for amp in (True, False)}
    print("ColBERT amp vs fp32: MaxSim rel diff %.1e, top-500 overlap %.4f; MaxSim-only recall@10 %.4f / %.4f, @100 %.4f / %.4f, @500 %.4f / %.4f; "
          "fused recall@10 %.4f / %.4f, @500 %.4f / %.4f" % (rel, overlap, solo[True]["recall@10"], solo[False]["recall@10"], solo[True]["recall@100"],
                                                              solo[False]["recall@100"], solo[True]["recall@500"], solo[False]["recall@500"],
                                                              fused[True]["recall@10"], fused[False]["recall@10"], fused[True]["recall@500"], fused[False]["recall@500"]))
    assert rel <= 2e-3 and overlap >= 0.97
    for amp in (True, False):
        assert solo[amp]["recall@10"] >= 0.5, solo[amp]                  # MaxSim alone does the task, in either precision
        assert solo[amp]["recall@500"] > solo[amp]["recall@10"] or solo[amp]["recall@10"] == 1.0
    for m in ("recall@10", "recall@100", "recall@500"):
        assert abs(solo[True][m] - solo[False][m]) <= moved / (2 * Q) + 1e-12, (m, solo[True][m], solo[False][m])   # the same gold documents (2 Q of them)
        assert abs(fused[True][m] - fused[False][m]) <= 1.0 / Q, (m, fused[True][m], fused[False][m])   # at most one (query, document) of the batch moves
    assert fused[True]["recall@500"] == fused[False]["recall@500"]
    assert solo[True]["recall@10"] < 1.0                                  # the grade is hard enough to tell the precisions apart if they differed


@pytest.mark.parametrize("P", [2898, 27943])
def test_scores_a_rounding_below_the_table_maximum(ops, oracle, P):
    """Two far-apart clusters of quantiles (float32 spacing ~1e-3 at 1e4: long runs of equal entries) and scores that ARE entries of the
    upper cluster, a rounding below the table's maximum: the equi-width bucket of such a score rounds up to the bucket count.  Found by the
    soak in round 4 in the round-3 kernel (all tables in LDS), whose exact search took that for 'at or above the last entry' and skipped the
    entries in between; both table sizes -- both kernels -- against the oracle's first-argmin scan."""
    rng = np.random.default_rng(P)
    tabs = [np.sort(np.concatenate([rng.normal(-1e4, 1e-3, P // 2), rng.normal(1e4, 1e-3, P - P // 2)])).astype(np.float32) for _ in range(2)]
    Q, N = 3, 4000
    xs = []
    for t in tabs:
        x = t[rng.integers(P - 40, P, (Q, N))].copy()                  # entries of the topmost 40: most of them below the maximum
        x[:, ::7] = t[rng.integers(0, P, (Q, len(range(0, N, 7))))]
        xs.append(np.ascontiguousarray(x))
    for norm in NORMS:
        got = ops.fuse_nsf([plane_of(ops, x) for x in xs], None, [0.4, 0.6], norm, [dev(t) for t in tabs]).cpu().numpy()
        assert ops.last_tables_path == ("lds-all" if P < 5000 else "lds-swap")
        exp = oracle.fuse_nsf(xs, None, [0.4, 0.6], norm, tabs)
        if norm == "percentile-rank":
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        else:
            fin = np.isfinite(exp)
            assert np.array_equal(np.isfinite(got), fin) and np.max(np.abs(got[fin] - exp[fin])) <= 1e-4


def test_cli_percentile_rank_with_a_table_of_the_size_the_reference_reads(tmp_path):
    """`python src/retrievers/hybrid.py --fusion nsf --normalization percentile-rank` (one of the three normalisers of every run_hybrid.sh
    combination, run_hybrid.sh:35) reads `score_distributions_raw_<eval>_28k.csv` (hybrid.py:412,451): 27,943 rows per system on LLeQA.  A
    file of that size through main() -- the equal-weights fusion and the weight sweep -- takes the long-table kernel."""
    import pandas as pd
    from fusion_amd import ops
    from fusion_amd.retrievers.hybrid import build_parser, main
    out = str(tmp_path)
    rng = np.random.default_rng(2)
    P = 27943
    pd.DataFrame({"bm25": quantile_table(np.maximum(0.0, rng.gamma(0.5, 4.0, 200_000) - 2.0) + 1e-3, P),
                  "dpr": quantile_table(rng.uniform(-0.2, 0.9, 200_000), P)}).to_csv(os.path.join(out, "score_distributions_raw_indomain_28k.csv"), index=False)
    base = f"--data_split test --models_domain legal --synthetic 600,6 --output_dir {out} --run_bm25 --run_dpr --fusion nsf --normalization percentile-rank".split()
    ops.last_tables_path = None
    a, _ = build_parser().parse_known_args(base)
    sc = main(a)
    assert ops.last_tables_path == "lds-swap" and 0.0 <= sc["recall@1000"] <= 1.0
    ops.last_tables_path = None
    a, _ = build_parser().parse_known_args(base + ["--tune_linear_fusion_weight"])
    rows = main(a)
    df = pd.read_csv(os.path.join(out, "nsf_percentile-rank_indomain.csv"))
    assert ops.last_tables_path == "lds-swap" and len(rows) == len(df) == 21 and list(df.columns)[-2:] == ["weight_bm25", "weight_dpr"]
