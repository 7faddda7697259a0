"""Row sort, zero compaction (csrc/sort.hip, ZC; fz_sort_rows_desc_lexical, round 6): float64 rows that are mostly exact zeros -- a lexical
ranker's scores (bm25.py:149-156: every document that shares no term with the query scores 0.0) -- are ordered without their zeros taking
part in the digit passes.  Same stable permutation, bit for bit, as the plain instantiation (fz_sort_rows_desc; FZ_SORT_ZERO_COMPACT=0), the
CPU oracle and Python's stable sorted(..., reverse=True) (bm25.py:104); the path counters (fz_sort_zero_compact_rows) pin which rows went
which way."""
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
def _zc_on():
    os.environ["FZ_SORT_ZERO_COMPACT"] = "1"
    yield
    os.environ["FZ_SORT_ZERO_COMPACT"] = "1"


def plane(ops, a):
    t = ops.alloc_plane(a.shape[0], a.shape[1], torch.from_numpy(a[:0]).dtype, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def bm25_like(rng, rows, n, zero_frac, negatives=False, ties=True):
    x = rng.gamma(0.8, 4.0, (rows, n)) + 0.01
    if ties:                                                  # BM25 scores repeat: documents with the same tf and length
        x = np.where(rng.random((rows, n)) < 0.3, np.round(x, 1), x)
    if negatives:                                             # idf <= 0 for terms in more than half of the corpus (bm25.py:147)
        x = np.where(rng.random((rows, n)) < 0.15, -x * 0.1, x)
    x[rng.random((rows, n)) < zero_frac] = 0.0
    return x.astype(np.float64)


def check(ops, oracle, K, row_len=None, stats=False, expect=None):
    """Device (compacting form) == device (whole rows) == oracle: order, sorted keys, rank (+ statistics); expect = (compacted, kept) rows."""
    kp = plane(ops, K)
    rl = None if row_len is None else torch.from_numpy(np.asarray(row_len, dtype=np.int32)).cuda()
    outs = []
    for mode in ("1", "0"):
        os.environ["FZ_SORT_ZERO_COMPACT"] = mode
        ops.sort_zero_compact_rows(reset=True)
        st = torch.full((4, K.shape[0]), 7.0, dtype=torch.float32, device="cuda") if stats else None
        o, k, r = ops.sort_rows_desc(kp, row_len=rl, want_rank=True, stats_out=st, lexical=True)
        counts = ops.sort_zero_compact_rows(reset=True)
        if mode == "1" and expect is not None:
            assert counts == expect, counts
        if mode == "0":
            assert counts == (0, 0)
        outs.append((o.cpu().numpy(), k.cpu().numpy(), r.cpu().numpy(), None if st is None else st.cpu().numpy()))
    eo, ek, er = oracle.sort_rows_desc(K, row_len=None if row_len is None else np.asarray(row_len, dtype=np.int32), want_rank=True)
    for o, k, r, st in outs:
        np.testing.assert_array_equal(o, eo)
        np.testing.assert_array_equal(k, ek)          # (values: the device writes a sorted -0.0 as +0.0 in both forms; NaN == NaN here)
        np.testing.assert_array_equal(r, er)
    np.testing.assert_array_equal(outs[0][1].view(np.uint64), outs[1][1].view(np.uint64))   # the two device forms: bit for bit
    if stats:
        np.testing.assert_array_equal(outs[0][3].view(np.uint32), outs[1][3].view(np.uint32))
        n = K.shape[1] if row_len is None else None
        for q in range(K.shape[0]):
            m = K.shape[1] if row_len is None else int(row_len[q])
            v = K[q, :m].astype(np.float32)
            if m and not np.isnan(v).any():
                assert outs[0][3][2, q] == v.min() and outs[0][3][3, q] == v.max(), q
    return outs[0]


@pytest.mark.parametrize("n", [4096, 8193, 12000, 16384, 16385, 20000, 27942, 28672])
@pytest.mark.parametrize("zf", [0.5, 0.62, 0.9])
def test_rows_of_mostly_zeros_equal_the_oracle(ops, oracle, n, zf):
    rng = np.random.default_rng(n * 7 + int(zf * 100))
    K = bm25_like(rng, 6, n, zf, negatives=(n % 2 == 0))
    K[1, ::3] = np.where(K[1, ::3] == 0.0, -0.0, K[1, ::3])            # -0.0 == +0.0: one tie group with the zeros
    eligible = n > 8192                                                  # (1024-thread rows; shorter ones keep the whole-row form)
    check(ops, oracle, K, stats=True, expect=(6, 0) if eligible else (0, 0))


def test_threshold_and_path_counters(ops, oracle):
    """3/8 of the keys: below it the row stays whole (the compaction would cost more than it saves), rows of one call go their own ways."""
    rng = np.random.default_rng(5)
    n = 27942
    K = bm25_like(rng, 5, n, 0.0)
    for q, share in enumerate((0.0, 0.30, 0.36, 0.39, 0.97)):
        idx = rng.permutation(n)[: int(share * n)]
        K[q, idx] = 0.0
    check(ops, oracle, K, stats=True, expect=(2, 3))


def test_every_kind_of_key_around_the_zeros(ops, oracle):
    """NaN (sorts first), +-inf, denormals that share zero's HIGH key word (told apart by the low word: they are not zeros), -0.0, the
    smallest negative, long tie runs among the non-zero keys (the repair phase in the compact layout), rows whose list begins or ends
    with the zeros (min / max statistics come from a compacted zero)."""
    rng = np.random.default_rng(11)
    n = 27942
    K = bm25_like(rng, 8, n, 0.6, negatives=True)
    K[0, 5] = np.nan; K[0, 17000] = np.nan
    K[1, 9] = np.inf; K[1, 10] = -np.inf; K[1, 27941] = np.inf
    K[2, 100:140] = 5e-324 * np.arange(1, 41)                          # denormals: high key word == zero's
    K[2, 200:220] = -5e-324 * np.arange(1, 21)
    K[3] = np.where(K[3] == 0.0, -0.0, K[3])                           # every zero negative
    K[4] = -np.abs(K[4])                                               # nothing above zero: the zeros head the list
    K[5] = np.abs(K[5])                                                # nothing below zero: the zeros end it
    hi = np.float64(3.0)
    K[6, 1000:4000:2] = hi + np.arange(1500) * np.finfo(np.float64).eps * 2   # 1,500 keys under one high key word, ascending: a dirty run
    K[7, :] = 0.0; K[7, 12345] = 1.5; K[7, 3] = -2.5                  # two keys and 27,940 zeros
    out = check(ops, oracle, K, stats=True, expect=(8, 0))
    assert out[3][2, 4] <= 0.0 and out[3][3, 4] == 0.0 and out[3][2, 5] == 0.0


def test_ragged_rows_and_rows_that_are_all_zeros(ops, oracle):
    rng = np.random.default_rng(13)
    n = 27942
    K = bm25_like(rng, 6, n, 0.7)
    K[4, :] = 0.0                                                       # nothing to compact TO: stays whole (and costs no digit pass)
    lens = [n, 20001, 4096, 4095, n, 0]
    check(ops, oracle, K, row_len=lens, stats=True, expect=(3, 1))


def test_outputs_asked_for_one_at_a_time(ops, oracle):
    """order only / rank only / keys only: the zeros' outputs follow the same switches as the sorted entries'."""
    rng = np.random.default_rng(17)
    K = bm25_like(rng, 4, 27942, 0.6)
    kp = plane(ops, K)
    eo, ek, er = oracle.sort_rows_desc(K, want_rank=True)
    o, _, _ = ops.sort_rows_desc(kp, want_keys=False, lexical=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo)
    _, k, _ = ops.sort_rows_desc(kp, want_order=False, lexical=True)            # (no order output: the plain instantiation)
    np.testing.assert_array_equal(k.cpu().numpy(), ek)
    _, _, r = ops.sort_rows_desc(kp, want_order=False, want_keys=False, want_rank=True, lexical=True)
    np.testing.assert_array_equal(r.cpu().numpy(), er)


def test_full_size_bm25_ranking_properties(ops):
    """BASELINE size: 1024 x 27,942 with ~60 % zeros (the bench's BM25 rows): every row compacted; order a permutation, scores non-increasing,
    rank its inverse, the zeros in ascending column order, and the whole thing equal to the whole-row form."""
    Q, N = 1024, 27942
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.distributions.Gamma(0.8, 0.25).sample((Q, N)).cuda().double() + 0.01
    x[torch.rand((Q, N), generator=g, device="cuda") < 0.6] = 0.0
    kp = ops.alloc_plane(Q, N, torch.float64, "cuda"); kp.copy_(x)
    ops.sort_zero_compact_rows(reset=True)
    o, k, r = ops.sort_rows_desc(kp, want_rank=True, lexical=True)
    assert ops.sort_zero_compact_rows(reset=True) == (Q, 0)
    ar = torch.arange(N, device="cuda")
    assert bool((k[:, :-1] >= k[:, 1:]).all())
    assert bool((torch.sort(o.long(), dim=1).values == ar).all())
    assert bool((torch.gather(r.long(), 1, o.long()) == ar).all())
    assert torch.equal(torch.gather(kp, 1, o.long()), k)
    zero_cols = torch.where(k == 0.0, o.long(), torch.full_like(o.long(), -1))
    zc = torch.where(zero_cols >= 0, zero_cols, torch.cummax(zero_cols, dim=1).values)
    assert bool((zc[:, 1:] >= zc[:, :-1]).all())                       # stable: the zeros keep their column order
    o2, k2, r2 = ops.sort_rows_desc(kp, want_rank=True)                # the general entry point: the plain instantiation
    assert ops.sort_zero_compact_rows(reset=True) == (0, 0)
    assert torch.equal(o, o2) and torch.equal(k, k2) and torch.equal(r, r2)


def test_bm25_search_device_takes_the_compacting_sort(ops, oracle):
    """The ranker itself: BM25.search_device on a Zipf corpus (most documents share no term with a short query) -- lists, scores and the
    statistics by-product == the oracle's BM25; the rows were compacted."""
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(29)
    vocab = np.array([f"w{i}" for i in range(20000)])
    p = 1.0 / np.arange(1, 20001) ** 1.2; p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 40)), p=p)) for _ in range(20000)]
    queries = [" ".join(rng.choice(vocab[50:], size=3)) for _ in range(6)]
    m = BM25(docs, 2.5, 0.2)
    ops.sort_zero_compact_rows(reset=True)
    rs = m.search_device(queries)
    done, kept = ops.sort_zero_compact_rows(reset=True)
    assert done >= 4, (done, kept)
    eo, ek = oracle.sort_rows_desc(oracle.BM25(docs, 2.5, 0.2).scores(queries))
    np.testing.assert_array_equal(rs.order.cpu().numpy(), eo)
    np.testing.assert_array_equal(rs.list_scores().cpu().numpy(), ek)


def test_a_compacted_row_the_fast_form_cannot_finish_goes_to_the_generic_launch(ops, oracle):
    """More than 2,048 keys under one high key word, out of order in their low words: the compacted row is flagged after its low words were
    parked in the order output -- the generic eight-pass launch rewrites every output of that row; its neighbours stay compacted."""
    rng = np.random.default_rng(31)
    n = 27942
    K = bm25_like(rng, 3, n, 0.6)
    j = rng.choice(n, size=3000, replace=False)
    K[1, j] = 3.0 + rng.permutation(3000) * np.finfo(np.float64).eps * 2
    check(ops, oracle, K, stats=True, expect=(3, 0))


def test_queries_with_frequent_terms_keep_the_plain_sort(ops):
    """A vocabulary that keeps its stop-word-like terms (the bench step's synthetic index): nearly every document shares a term with every
    query, BM25.search_device expects no zeros and asks for the plain instantiation -- nothing goes through the compacting one."""
    from fusion_amd.retrievers.bm25 import BM25, LEXICAL_MIN_ZERO_SHARE
    rng = np.random.default_rng(37)
    vocab = np.array([f"w{i}" for i in range(3000)])
    p = 1.0 / np.arange(1, 3001) ** 1.05; p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(60, 200)), p=p)) for _ in range(9000)]
    queries = [" ".join(rng.choice(vocab, size=8, p=p)) for _ in range(6)]
    m = BM25(docs, 2.5, 0.2)
    ops.sort_zero_compact_rows(reset=True)
    m.search_device(queries)
    assert m.zero_share_estimate < LEXICAL_MIN_ZERO_SHARE and ops.sort_zero_compact_rows(reset=True) == (0, 0)
