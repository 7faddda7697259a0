"""Parity at BASELINE.json's FULL sizes (the cases round 2 only timed):

* configs[4] -- one 1/8 shard of mMARCO-fr (1,105,228 passages x 768, Q = 1024, k = 1000): the chunked score -> top-k -> merge loop the
  reference runs in InformationRetrievalEvaluatorCustom.compute_metrices (src/utils/sentence_transformers.py:314-393).  43 window-cut
  GEMM launches, 4 folds and int64 id arithmetic beyond 2^31 are where size-dependent bugs would live.
* configs[3] -- the whole 1771-vector weight grid of hybrid.py:404-426 at the LLeQA test split's size (S = 4, Q = 195, N = 27,942,
  np.float64 lattice, ColBERT list 40 % absent) against one Aggregator.fuse_device + run_evaluation per sampled weight vector.

Everything goes through the C ABI (fusion_amd.ops / the drop-in classes); the oracle is the checker."""
import numpy as np
import pytest
import torch

from helpers import assert_ranked_close

pytestmark = pytest.mark.gpu

COS_TOL = 2e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


@pytest.fixture(scope="module")
def mmarco_shard(ops):
    """1/8 of mMARCO-fr, generated on the device (3.4 GB; nothing crosses PCIe), with planted exact duplicates."""
    Nl, d, Q = 8841823 // 8, 768, 1024
    g = torch.Generator(device="cuda").manual_seed(20260)
    Dn = torch.empty((Nl, d), dtype=torch.float32, device="cuda")
    for c0 in range(0, Nl, 1 << 19):
        c1 = min(Nl, c0 + (1 << 19))
        Dn[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    # exact ties: copies of one passage inside the head, across the head boundary, across fold windows and shard cuts;
    # query 3 IS passage 777 so that its copies sit at the very top of a list
    for lo in (5_000, 8_190, 200_000, 138_150, 552_610, 1_100_000):
        Dn[lo:lo + 6] = Dn[777]
    Qn[3] = Dn[777]
    yield Dn, Qn
    del Dn, Qn
    torch.cuda.empty_cache()


def test_config5_shard_fused_equals_two_pass_equals_oracle(ops, oracle, mmarco_shard):
    from fusion_amd.distributed import ShardedDenseIndex
    Dn, Qn = mmarco_shard
    Nl, k = Dn.shape[0], 1000
    base = (1 << 31) + 12345                                  # ids beyond int32 from the first passage on
    idx = ShardedDenseIndex(Dn, id_base=base)
    assert idx.FUSED
    s_f, i_f = idx.local_topk(Qn, k)
    assert idx.last_overflow == 0, "the streaming search overflowed a candidate buffer on random data"
    two = ShardedDenseIndex(Dn, id_base=base)
    two.FUSED = False
    s_t, i_t = two.local_topk(Qn, k)
    assert two.last_overflow == 0
    assert torch.equal(s_f, s_t) and torch.equal(i_f, i_t), "fused GEMM-filter search != GEMM + separate filter pass"
    ids = i_f.cpu().numpy()
    assert ids.min() >= base and ids.max() < base + Nl
    sc = s_f.cpu().numpy()
    assert np.all(sc[:, :-1] >= sc[:, 1:])
    ties = sc[:, :-1] == sc[:, 1:]
    assert np.all(ids[:, :-1][ties] < ids[:, 1:][ties]), "ties must be in ascending id order"
    assert ids[3, :7].tolist() == [base + 777] + [base + 5_000 + j for j in range(6)]    # query 3: the passage and its first copies

    # the oracle on sampled queries: (i) its top-k of the device's own scores, bit for bit; (ii) its own fp32 scores -> ranked-close
    sample = [0, 3, 127, 128, 511, 640, 1000, 1023]
    S_dev = ops.dot_scores(Qn[sample].contiguous(), Dn).cpu().numpy()
    es, ei = oracle.topk_rows(S_dev, k, id_base=base)
    np.testing.assert_array_equal(sc[sample], es)
    np.testing.assert_array_equal(ids[sample], ei)
    Dh = Dn.cpu().numpy()
    S_or = oracle.dot_scores(Qn[sample].cpu().numpy(), Dh, fma_chain=True)
    assert np.max(np.abs(S_or - S_dev)) <= COS_TOL
    os_, oi = oracle.topk_rows(S_or, k, id_base=base)
    for r in range(len(sample)):
        assert_ranked_close(ids[sample][r], sc[sample][r], oi[r], os_[r], COS_TOL, truncated=True)
    del Dh, S_or, S_dev

    # the same corpus as 8 shard_bounds pieces, searched one after another and merged (fz_topk_merge): the unsharded answer
    from fusion_amd.distributed import shard_bounds
    parts_s, parts_i = [], []
    for r in range(8):
        lo, hi = shard_bounds(Nl, 8, r)
        sh = ShardedDenseIndex(Dn[lo:hi], id_base=base + lo)
        s, i = sh.local_topk(Qn, k)
        parts_s.append(s); parts_i.append(i)
    ms, mi = ops.topk_merge(torch.stack(parts_s), torch.stack(parts_i))
    assert torch.equal(ms, s_f) and torch.equal(mi, i_f), "8 shards + merge != one shard"


def test_config5_fused_search_survives_a_relevance_sorted_corpus(ops, oracle):
    """A corpus whose best passages come LAST makes every window overflow its candidate buffer: every such window is redone exactly (same
    answer), never a wrong list -- and a single overflowing window in an otherwise random corpus costs that window only."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(9)
    N, d, Q, k = 120_000, 64, 16, 1000
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    q = ops.normalize_rows(torch.randn((1, d), generator=g, device="cuda"))
    order = torch.argsort((Dn @ q.T).flatten())                  # ascending relevance to q
    Dn = Dn[order].contiguous()
    Qn = ops.normalize_rows(q + 0.05 * torch.randn((Q, d), generator=g, device="cuda"))
    idx = ShardedDenseIndex(Dn, id_base=0)
    for fused in (True, False):
        idx.FUSED = fused
        s, i = idx.local_topk(Qn, k)
        # fused: every window behind the head is redone exactly from the GEMM's operands; materialised scores are not held by the
        # stream (one chunk alive at a time), so that path flags the overflow and searches the shard again on the exact path
        assert idx.last_overflow >= (2 if fused else 1)
        es, ei = oracle.topk_rows(ops.dot_scores(Qn, Dn).cpu().numpy(), k)
        np.testing.assert_array_equal(s.cpu().numpy(), es)
        np.testing.assert_array_equal(i.cpu().numpy(), ei)
    # one hot window in a random corpus: 9,000 near-copies of the query direction in one stretch
    Dr = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Dr[60_000:69_000] = ops.normalize_rows(q + 0.2 * torch.randn((9_000, d), generator=g, device="cuda"))
    idx = ShardedDenseIndex(Dr, id_base=7)
    s, i = idx.local_topk(Qn, k)
    assert idx.last_overflow == 1
    es, ei = oracle.topk_rows(ops.dot_scores(Qn, Dr).cpu().numpy(), k, id_base=7)
    np.testing.assert_array_equal(s.cpu().numpy(), es)
    np.testing.assert_array_equal(i.cpu().numpy(), ei)


# ---- configs[3]: the whole weight grid at the LLeQA test split's size ---------------------------------------------------------------
def _lleqa_systems(ops, Q, N, seed):
    """Four ranked systems shaped like SURVEY 8d/C4: bm25-like (>= 0, ~40 % zeros), cosine-like, SPLADE-like, ColBERT-like with the last
    40 % of every ranking absent -- built the way Ranker hands them on (hybrid._rank_scores)."""
    from fusion_amd.retrievers.hybrid import _rank_scores
    g = torch.Generator(device="cuda").manual_seed(seed)
    ids = np.arange(N) + 1

    def plane(t):
        p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(t); return p
    hidden = torch.randn((Q, N), generator=g, device="cuda")          # what the systems agree on
    mk = lambda noise: hidden + noise * torch.randn((Q, N), generator=g, device="cuda")
    bm = torch.clamp(3.0 * mk(1.0) - 1.0, min=0.0)
    dpr = torch.tanh(0.3 * mk(0.8))
    spl = torch.log1p(torch.relu(mk(1.2)))
    col = 20.0 + 4.0 * mk(0.6)
    systems = {"bm25": _rank_scores(plane(bm), ids, None), "dpr": _rank_scores(plane(dpr), ids, None),
               "splade": _rank_scores(plane(spl), ids, None), "colbert": _rank_scores(plane(col), ids, int(0.6 * N))}
    assert not systems["colbert"].full and systems["bm25"].full
    return systems, hidden, ids


@pytest.mark.parametrize("norm", ["min-max", "z-score"])
def test_config4_full_weight_grid_equals_fusing_per_vector(ops, norm):
    from fusion_amd.retrievers.hybrid import Aggregator, run_evaluation, weight_grid
    Q, N = 195, 27942
    systems, hidden, ids = _lleqa_systems(ops, Q, N, seed=4)
    rng = np.random.default_rng(0)
    top = torch.topk(hidden, 1500, dim=1).indices.cpu().numpy()
    labels = []
    for q in range(Q):                                          # 1-5 gold articles per question, most of them retrievable
        n = int(rng.integers(1, 6))
        pick = rng.choice(1500, size=n, replace=False, p=None)
        pick[0] = int(rng.integers(0, 30))
        labels.append([int(ids[top[q, j]]) for j in dict.fromkeys(pick.tolist())])
    labels[7].append(10 ** 9)                                   # an id no system lists: counts in the divisor only
    grid = weight_grid(list(systems))
    assert len(grid) == 1771 and all(isinstance(v, np.float64) for v in grid[5].values())
    got = Aggregator.tune(systems, norm, grid, labels, {})
    assert len(got) == 1771
    sample = sorted(set(rng.choice(1771, size=23, replace=False).tolist()) | {0, 1770})       # incl. (0,0,0,1) and (1,0,0,0)
    worst = 0.0
    for wi in sample:
        fused = Aggregator.fuse_device(systems, "nsf", norm, grid[wi], {})
        exp = run_evaluation(fused.predictions(1000), labels, print2console=False)
        assert list(got[wi]) == list(exp)
        worst = max(worst, max(abs(float(got[wi][m]) - float(exp[m])) for m in exp))
    assert worst <= 1e-12, worst
    r500 = np.array([g["recall@500"] for g in got])
    assert r500.max() > r500.min() and 0.0 < r500.max() <= 1.0  # the sweep discriminates between weight vectors


# ---- the reference-side binding of INTEGRATION.md, executed ----------------------------------------------------------------------------
def test_integration_md_binding_runs_and_matches_the_oracle(oracle):
    """Extract the ctypes stub a maintainer of the reference would paste, run its rrf() on the GPU and hold the
    result against the oracle's rank -> rrf -> stable order (hybrid.py:252,301-306)."""
    from conftest import ROOT
    from helpers import integration_snippet
    ns = {}
    exec(compile(integration_snippet(ROOT), "INTEGRATION.md", "exec"), ns)
    g = torch.Generator(device="cuda").manual_seed(3)
    Q, N = 37, 27968                                              # ld = N: a multiple of 64 floats, as the snippet's contract asks
    planes = [torch.randn((Q, N), generator=g, device="cuda"), torch.clamp(torch.randn((Q, N), generator=g, device="cuda"), min=0.0)]
    order, score = ns["rrf"](planes)
    torch.cuda.synchronize()
    hp = [p.cpu().numpy() for p in planes]
    rk = [oracle.sort_rows_desc(p, want_rank=True)[2] for p in hp]
    o0 = oracle.sort_rows_desc(hp[0])[0]
    f = oracle.fuse_rank(rk, np.full((2, Q), N, dtype=np.int32), "rrf")
    eo, ek = oracle.sort_rows_desc(f, init_order=o0)
    np.testing.assert_array_equal(order.cpu().numpy(), eo)
    np.testing.assert_array_equal(score.cpu().numpy(), ek)
