"""Row sort, bucket ranking (csrc/sort.hip, ABI 17): rows of a ranker's scores are ordered without the digit passes -- the same
permutation, bit for bit, as the digit passes (FZ_SORT_BUCKET_RANK=0), the CPU oracle and torch's stable sort; the path counters
(fz_sort_bucket_rank_rows) pin which rows went which way.  Reference semantics: Python's stable sorted(..., reverse=True)
(bm25.py:104, hybrid.py:306) over util.semantic_search's scores (hybrid.py:103)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


@pytest.fixture(autouse=True)
def _bucket_on():
    os.environ["FZ_SORT_BUCKET_RANK"] = "1"
    yield
    os.environ["FZ_SORT_BUCKET_RANK"] = "1"


def plane(ops, a):
    t = ops.alloc_plane(a.shape[0], a.shape[1], torch.from_numpy(a[:0]).dtype, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scores(kind, rng, rows, n):
    if kind == "cosine":          # DPR: dot products of unit vectors in 768 dimensions
        return rng.normal(0, 768 ** -0.5, (rows, n)).astype(np.float32)
    if kind == "uniform":
        return rng.uniform(-1, 1, (rows, n)).astype(np.float32)
    if kind == "positive":        # MaxSim-like: all positive, two binades
        return (20 + 4 * rng.normal(0, 1, (rows, n))).astype(np.float32)
    if kind == "lognormal":       # a dozen and a half binades, one sign
        return np.exp(rng.normal(0, 1.5, (rows, n))).astype(np.float32)
    if kind == "fused":           # weighted sums of min-max normalised scores
        return (0.3 * rng.uniform(0, 1, (rows, n)) + 0.7 * rng.beta(2, 5, (rows, n))).astype(np.float32)
    raise ValueError(kind)


def both_forms(ops, f):
    """f() under the bucket ranking and under the digit passes; the counters of the first run."""
    ops.sort_bucket_rank_rows(reset=True)
    os.environ["FZ_SORT_BUCKET_RANK"] = "1"
    a = f()
    counts = ops.sort_bucket_rank_rows(reset=True)
    os.environ["FZ_SORT_BUCKET_RANK"] = "0"
    b = f()
    assert ops.sort_bucket_rank_rows(reset=True) == (0, 0, 0)       # switched off: no row even tries
    os.environ["FZ_SORT_BUCKET_RANK"] = "1"
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.equal(x, y) or np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True)
    return a, counts


@pytest.mark.parametrize("kind", ["cosine", "uniform", "positive", "lognormal", "fused"])
@pytest.mark.parametrize("n", [9000, 16384, 20000, 27942, 28672])
def test_bucket_ranking_is_the_digit_passes_permutation(ops, oracle, kind, n):
    rng = np.random.default_rng(n + len(kind))
    rows = 6
    k = scores(kind, rng, rows, n)
    kp = plane(ops, k)
    st = torch.empty((4, rows), dtype=torch.float32, device="cuda")
    (order, sk, rank, stats), counts = both_forms(ops, lambda: ops.sort_rows_desc(kp, want_rank=True, stats_out=st) + (st.clone(),))
    assert counts[0] == rows and counts[2] == 0, counts                # every row was ordered by the bucket ranking
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    t_sk, t_order = torch.sort(kp, dim=1, descending=True, stable=True)
    assert torch.equal(order.long(), t_order) and torch.equal(sk, t_sk)
    np.testing.assert_allclose(stats[0].cpu().numpy(), k.astype(np.float64).mean(1), rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(stats[2].cpu().numpy(), k.min(1))
    np.testing.assert_array_equal(stats[3].cpu().numpy(), k.max(1))


def test_ragged_lengths_specials_and_short_rows(ops, oracle):
    """row_len from 1 to n (rows under 4,096 keys never try), +-inf, one NaN (sorts first), signed zeros."""
    rng = np.random.default_rng(5)
    n = 27942
    lens = np.array([n, 27000, 12345, 4096, 4095, 700, 1, 0, 20000, 9999], dtype=np.int32)
    k = scores("cosine", rng, len(lens), n)
    k[0, 17] = np.inf; k[0, 18] = -np.inf; k[0, 5000] = np.inf
    k[1, 123] = np.nan
    k[2, [5, 50, 500]] = [0.0, -0.0, 0.0]
    (order, sk, rank), counts = both_forms(ops, lambda: ops.sort_rows_desc(plane(ops, k), row_len=dev(lens), want_rank=True))
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, row_len=lens, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    assert counts[0] + counts[2] == int((lens >= 4096).sum()), (counts, lens)
    assert counts[0] >= 4                                              # the plain rows at least


@pytest.mark.parametrize("mode", ["placed", "gathered"])
def test_incoming_sequences(ops, oracle, mode):
    """The fused-list ordering: a permuted incoming sequence (ties keep ITS order), full and cut rows."""
    rng = np.random.default_rng(11)
    rows, n = 5, 27942
    k = scores("fused", rng, rows, n)
    k[:, ::997] = k[:, 0:1]                                             # some ties whose order the incoming sequence decides
    init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)
    lens = np.array([n, n, 20000, 4096, n - 1], dtype=np.int32)
    irank = np.full((rows, n), -1, dtype=np.int32)
    for r in range(rows):
        init[r, lens[r]:] = -1
        irank[r, init[r, :lens[r]]] = np.arange(lens[r], dtype=np.int32)
    kp = plane(ops, k)
    if mode == "placed":
        f = lambda: ops.sort_rows_desc(kp, init_rank=plane(ops, irank), row_len=dev(lens), want_rank=True)
    else:
        f = lambda: ops.sort_rows_desc(kp, init_order=plane(ops, init), row_len=dev(lens), want_rank=True)
    (order, sk, rank), counts = both_forms(ops, f)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, init_order=init, row_len=lens, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    assert counts[0] == rows, counts


def test_rows_the_bucket_ranking_turns_away(ops, oracle):
    """Heavy ties (zeros under the scores, a few hundred distinct values), a crowd at the floor, one giant tie group: the digit
    passes order them -- same result -- and the counters say so."""
    rng = np.random.default_rng(3)
    n = 27942
    k = scores("cosine", rng, 6, n)
    k[0, rng.random(n) < 0.5] = 0.0                                     # SPLADE / BM25: documents that share no term with the query
    k[1] = np.round(k[1] * 4000) / 4000                                 # ~ 600 distinct values
    k[2] = np.exp(rng.normal(0, 1, n)).astype(np.float32) ** 8          # far more than 24 binades
    k[3, 1000:1400] = k[3, 7]                                           # one tie group of 400 in an otherwise smooth row
    k[4, :] = 0.25                                                      # constant row
    (order, sk, rank), counts = both_forms(ops, lambda: ops.sort_rows_desc(plane(ops, k), want_rank=True))
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    assert counts[0] == 1 and counts[2] == 4, counts                    # row 5 is ranked; rows 0-3 handed over; the constant row never tries


def test_neighbours_one_ulp_apart_are_put_back(ops, oracle):
    """Two distinct values in a thinly populated stretch of the row share a bucket value and come out in position order: the
    neighbour check swaps the pair back (counter 1).  Three or more in a row are not a swap: the digit passes finish the row (counter 2)."""
    rng = np.random.default_rng(8)
    n = 27942
    k = scores("cosine", rng, 4, n)
    a = np.float32(1.0e-7)                                              # a stretch of the row nobody else is in; positions the sample skips
    up = lambda x: np.nextafter(np.float32(x), np.float32(1))
    k[0, 100], k[0, 1792 * 11 + 100] = a, up(a)                                   # the larger value LATER in the row: position order is the wrong one
    k[1, 100], k[1, 1792 * 11 + 100] = -a, -up(a)                                 # the same on the negative side, already in the right order
    v = [a]
    for _ in range(4):
        v.append(up(v[-1]))
    k[2, [100, 1792 * 3 + 100, 1792 * 5 + 100, 1792 * 8 + 100, 1792 * 11 + 100]] = v                           # five in a row, reversed: at least three share a bucket value
    (order, sk, rank), counts = both_forms(ops, lambda: ops.sort_rows_desc(plane(ops, k), want_rank=True))
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    assert counts[0] + counts[2] == 4
    assert counts[1] >= 1 and counts[2] >= 1, counts


def test_a_ranking_batch_at_full_size(ops):
    """1024 x 27,942 cosine scores (dpr_rank): every row ordered by the bucket ranking, the output a stable descending permutation."""
    g = torch.Generator(device="cuda").manual_seed(0)
    Q, N = 1024, 27942
    S = ops.alloc_plane(Q, N, torch.float32, "cuda")
    S.copy_(torch.randn((Q, N), generator=g, device="cuda") * 768 ** -0.5)
    ops.sort_bucket_rank_rows(reset=True)
    order, sk, rank = ops.sort_rows_desc(S, want_rank=True)
    ranked, repaired, handed = ops.sort_bucket_rank_rows(reset=True)
    assert ranked + handed == Q and handed <= Q // 100, (ranked, repaired, handed)
    t_sk, t_order = torch.sort(S, dim=1, descending=True, stable=True)
    assert torch.equal(sk, t_sk) and torch.equal(order.long(), t_order)
    assert torch.equal(torch.gather(rank, 1, order.long()), torch.arange(N, device="cuda", dtype=torch.int32).expand(Q, N))


@pytest.mark.parametrize("n", [40000, 70001])
def test_rows_longer_than_one_workgroup_keep_the_digit_passes(ops, oracle, n):
    """Chunk-sort + merge (rows beyond 35,840 fp32 keys): the chunks are 35,840 keys long -- 35 keys per thread, whose LDS leaves no room
    for the ranking's tables -- and take the digit passes; nothing tries."""
    rng = np.random.default_rng(n)
    k = scores("cosine", rng, 3, n)
    (order, sk, rank), counts = both_forms(ops, lambda: ops.sort_rows_desc(plane(ops, k), want_rank=True))
    e_order, e_sk, e_rank = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    assert counts == (0, 0, 0), counts


def test_statistics_of_a_ranking_cut_to_its_top_k(ops):
    """stats_len: mean / std / min / max over the first k entries of the SORTED list, taken from the sorted registers -- after the bucket ranking
    as after the digit passes (hybrid.py:254-262 on a return_topk list)."""
    rng = np.random.default_rng(21)
    rows, n = 6, 27942
    k = scores("cosine", rng, rows, n)
    lens = np.array([1000, 500, 27942, 1, 20000, 4096], dtype=np.int32)
    st = torch.empty((4, rows), dtype=torch.float32, device="cuda")
    (order, sk, rank, stats), counts = both_forms(ops, lambda: ops.sort_rows_desc(plane(ops, k), want_rank=True, stats_out=st, stats_len=dev(lens)) + (st.clone(),))
    assert counts[0] == rows, counts
    srt = -np.sort(-k.astype(np.float64), axis=1)
    for r in range(rows):
        head = srt[r, :lens[r]]
        np.testing.assert_allclose(stats[0, r].item(), head.mean(), rtol=1e-6, atol=1e-7)
        if lens[r] > 1:
            np.testing.assert_allclose(stats[1, r].item(), head.std(ddof=1), rtol=1e-5)
        assert stats[2, r].item() == np.float32(head.min()) and stats[3, r].item() == np.float32(head.max())
