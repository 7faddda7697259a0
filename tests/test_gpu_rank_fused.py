"""Rank fusion as the load phase of the final sort (fz_sort_rank_fused_desc, ABI 18; hybrid.py:248-252,301-306): one kernel == the
two it replaces (fz_fuse_rank_f64 then fz_sort_rows_desc[_placed] on the float64 plane) == the oracle, bit for bit -- order, fused
float64 scores in list order, rank -- for full lists (placed by system 0's rank plane), partial lists (gathered by first-insertion
order) and plain columns; every workgroup configuration of the sort; rrf and bcf; ties across systems (equal sums: KAT-1's A/B swap)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import dev, plane, synth_systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from fusion_amd import ops as o
    return o


def full_systems(rng, S, Q, N):
    """S full rankings with many cross-system ties: system s > 0 is system 0's ranking with blocks reversed, so that pairs of documents
    swap ranks between systems and their rrf sums are EQUAL (the tie-break by first insertion decides)."""
    ranks = []
    base = np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32)
    ranks.append(base)
    for s in range(1, S):
        r = base.copy()
        if s % 2 == 1:
            blk = max(2, int(rng.integers(2, 9)))
            m = (N // blk) * blk
            r[:, :] = np.where(r < m, (r // blk) * blk + (blk - 1 - r % blk), r)
        else:
            r = np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32)
        ranks.append(r.astype(np.int32))
    return ranks, np.full((S, Q), N, dtype=np.int32)


@pytest.mark.parametrize("N", [1, 2, 65, 1000, 1025, 4097, 9000, 16385, 27942, 28672])
@pytest.mark.parametrize("S,method", [(1, "rrf"), (2, "rrf"), (3, "rrf"), (4, "bcf"), (5, "rrf"), (2, "bcf")])
def test_full_lists_placed_by_system0(ops, oracle, N, S, method):
    rng = np.random.default_rng(N * 10 + S)
    Q = 3
    ranks, lens = full_systems(rng, S, Q, N)
    rp = [plane(ops, r) for r in ranks]
    order, sk, rank = ops.sort_rank_fused(rp, dev(lens), method, init_rank=rp[0], want_rank=True, covers_all=True)
    fused = ops.fuse_rank(rp, dev(lens), method)
    o2, s2, r2 = ops.sort_rows_desc(fused, init_rank=rp[0], want_rank=True, covers_all=True)
    np.testing.assert_array_equal(order.cpu().numpy(), o2.cpu().numpy())
    np.testing.assert_array_equal(sk.cpu().numpy(), s2.cpu().numpy())
    np.testing.assert_array_equal(rank.cpu().numpy(), r2.cpu().numpy())
    f = oracle.fuse_rank(ranks, lens, method)
    o0 = np.argsort(ranks[0], axis=1, kind="stable").astype(np.int32)          # system 0's list = the first-insertion order
    e_order, e_sk, e_rank = oracle.sort_rows_desc(f, init_order=o0, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)


@pytest.mark.parametrize("S,Q,N", [(2, 4, 257), (4, 5, 1000), (4, 3, 5000), (3, 2, 20000), (4, 2, 27942)])
@pytest.mark.parametrize("method", ["rrf", "bcf"])
def test_partial_lists_gathered_by_insertion_order(ops, oracle, S, Q, N, method):
    rng = np.random.default_rng(S * 1000 + N)
    _, ranks, orders, lens = synth_systems(rng, S, Q, N)
    rp, op_ = [plane(ops, r) for r in ranks], [plane(ops, o) for o in orders]
    ins, U = ops.insertion_order(op_, dev(lens), N)
    order, sk, rank = ops.sort_rank_fused(rp, dev(lens), method, init_order=ins, row_len=U, want_rank=True)
    fused = ops.fuse_rank(rp, dev(lens), method)
    o2, s2, r2 = ops.sort_rows_desc(fused, init_order=ins, row_len=U, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), o2.cpu().numpy())
    np.testing.assert_array_equal(sk.cpu().numpy(), s2.cpu().numpy())
    np.testing.assert_array_equal(rank.cpu().numpy(), r2.cpu().numpy())
    e_ins, e_U = oracle.insertion_order(orders, lens, N)
    e_order, e_sk = oracle.sort_rows_desc(oracle.fuse_rank(ranks, lens, method), init_order=e_ins, row_len=e_U)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)


@pytest.mark.parametrize("N", [300, 5000, 27942])
def test_plain_columns_and_a_placed_plane_that_is_not_system0(ops, oracle, N):
    """No incoming sequence (ties -> ascending column), and a placed sequence given by ANOTHER plane than ranks[0] (the kernel then
    reads ranks[0] like any other system); documents no system lists come last with -inf."""
    rng = np.random.default_rng(N)
    S, Q = 3, 4
    _, ranks, _, lens = synth_systems(rng, S, Q, N)
    for r in ranks:
        r[:, : N // 7] = np.where(rng.random((Q, N // 7)) < 0.5, -1, r[:, : N // 7])     # holes, some documents in no list at all
    rp = [plane(ops, r) for r in ranks]
    order, sk, rank = ops.sort_rank_fused(rp, dev(lens), "rrf", want_rank=True)
    f = oracle.fuse_rank(ranks, lens, "rrf")
    e_order, e_sk, e_rank = oracle.sort_rows_desc(f, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    perm = np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32)            # position of column j
    order, sk, _ = ops.sort_rank_fused(rp, dev(lens), "rrf", init_rank=plane(ops, perm), covers_all=True)
    e_order, e_sk = oracle.sort_rows_desc(f, init_order=np.argsort(perm, axis=1, kind="stable").astype(np.int32))
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)


def test_kat1_tie_break_and_known_scores(ops):
    """s1 = [A, B], s2 = [B, A], rrf -> [A, B], both 1/61 + 1/62 (equal sums added in system order)."""
    r1 = plane(ops, np.array([[0, 1]], dtype=np.int32)); r2 = plane(ops, np.array([[1, 0]], dtype=np.int32))
    lens = dev(np.full((2, 1), 2, dtype=np.int32))
    order, sk, _ = ops.sort_rank_fused([r1, r2], lens, "rrf", init_rank=r1, covers_all=True)
    assert order.cpu().tolist() == [[0, 1]] and sk.cpu().tolist() == [[0.03252247488101534, 0.03252247488101534]]
    order, sk, _ = ops.sort_rank_fused([r2, r1], lens, "rrf", init_rank=r2, covers_all=True)
    assert order.cpu().tolist() == [[1, 0]]


def test_rows_beyond_one_workgroup_are_refused_loudly(ops):
    from fusion_amd._lib import FZ_ERR_UNSUPPORTED, FusionHipError
    N = 28673
    r = plane(ops, np.arange(N, dtype=np.int32)[None, :].copy())
    with pytest.raises(FusionHipError) as e:
        ops.sort_rank_fused([r, r], dev(np.full((2, 1), N, dtype=np.int32)), "rrf", init_rank=r, covers_all=True)
    assert e.value.status == FZ_ERR_UNSUPPORTED
    with pytest.raises(TypeError):
        ops.sort_rank_fused([torch.zeros((1, 4), dtype=torch.int32)] * 2, torch.zeros((2, 1), dtype=torch.int32), "rrf")   # CPU tensors: no CPU path


@pytest.mark.parametrize("full", [True, False])
@pytest.mark.parametrize("method", ["rrf", "bcf"])
def test_aggregator_takes_the_fused_sort_and_matches_the_two_kernel_path(ops, oracle, full, method):
    """Aggregator.fuse_device: full lists and partial lists both run the one-kernel form (pinned via last_rank_fused_sort); the lists
    equal the two-call form's; a top-k request is the same sort, cut."""
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator
    rng = np.random.default_rng(5)
    S, Q, N = 3, 6, 3000
    planes, ranks, orders, lens = synth_systems(rng, S, Q, N, partial=not full)
    systems = {}
    for s in range(S):
        systems[f"s{s}"] = RankedSystem(scores=plane(ops, planes[s]), order=plane(ops, orders[s]), rank=plane(ops, ranks[s]), lens=dev(lens[s]),
                                        ids=np.arange(N) + 10, full=bool((lens[s] == N).all()))
    got = Aggregator.fuse_device(systems, method)
    assert Aggregator.last_rank_fused_sort is True
    rp = [plane(ops, r) for r in ranks]
    fused = ops.fuse_rank(rp, dev(lens), method)
    if all(s.full for s in systems.values()):
        o2, s2, _ = ops.sort_rows_desc(fused, init_rank=rp[0], covers_all=True)
    else:
        ins, U = ops.insertion_order([plane(ops, o) for o in orders], dev(lens), N)
        o2, s2, _ = ops.sort_rows_desc(fused, init_order=ins, row_len=U)
    np.testing.assert_array_equal(got.order.cpu().numpy(), o2.cpu().numpy())
    np.testing.assert_array_equal(got.scores.cpu().numpy(), s2.cpu().numpy())
    cut = Aggregator.fuse_device(systems, method, topk=100)
    assert Aggregator.last_rank_fused_sort is True and Aggregator.last_topk_path == "sort"
    np.testing.assert_array_equal(cut.order.cpu().numpy(), o2.cpu().numpy()[:, :100])
    np.testing.assert_array_equal(cut.scores.cpu().numpy(), s2.cpu().numpy()[:, :100])


def test_full_size_property_sorted_and_permutation(ops):
    """BASELINE size (Q = 1024, N = 27,942: the bench step's shape; BM25-like + cosine-like rankings): the one-kernel form equals the
    two-kernel form, every row is a permutation, the fused scores are non-increasing and equal 1/(61 + r_b) + 1/(61 + r_d) recomputed from the order."""
    Q, N = 1024, 27942
    g = torch.Generator(device="cuda").manual_seed(3)
    b = torch.clamp(torch.randn((Q, N), generator=g, device="cuda") * 3 - 1, min=0).double()      # ~60 % exact zeros: ties by position
    d = torch.randn((Q, N), generator=g, device="cuda")
    bp = ops.alloc_plane(Q, N, torch.float64, "cuda"); bp.copy_(b)
    dp = ops.alloc_plane(Q, N, torch.float32, "cuda"); dp.copy_(d)
    _, _, r_b = ops.sort_rows_desc(bp, want_keys=False, want_rank=True)
    _, _, r_d = ops.sort_rows_desc(dp, want_keys=False, want_rank=True)
    lens = torch.full((2, Q), N, dtype=torch.int32, device="cuda")
    order, sk, _ = ops.sort_rank_fused([r_b, r_d], lens, "rrf", init_rank=r_b, covers_all=True)
    o2, s2, _ = ops.sort_rows_desc(ops.fuse_rank([r_b, r_d], lens, "rrf"), init_rank=r_b, covers_all=True)
    assert torch.equal(order, o2) and torch.equal(sk, s2)
    assert torch.equal(torch.sort(order.long(), dim=1).values, torch.arange(N, device="cuda").expand(Q, N))
    assert bool((sk[:, :-1] >= sk[:, 1:]).all())
    rb = torch.gather(r_b.long(), 1, order.long()).double(); rd = torch.gather(r_d.long(), 1, order.long()).double()
    assert torch.equal(sk, 1.0 / (61.0 + rb) + 1.0 / (61.0 + rd))


def test_bm25_float32_plane_from_the_scoring_launch(ops):
    """fz_bm25_scores_f64_f32: the float32 plane the normalisations read (hybrid.py:255) comes out of the scoring kernel's accumulators
    and equals the separate conversion pass it replaces (fz_f64_to_f32 on the float64 plane), bit for bit; BM25.search_device uses it."""
    from fusion_amd.retrievers.bm25 import BM25
    rng = np.random.default_rng(0)
    vocab = [f"w{i}" for i in range(400)]
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 60)))) for _ in range(15001)]      # two document slices
    queries = [" ".join(rng.choice(vocab + ["oov"], size=int(rng.integers(1, 9)))) for _ in range(9)]
    bm = BM25(docs, 2.5, 0.2)
    sc64, sc32 = bm.scores(queries, want_f32=True)
    assert sc32.dtype == torch.float32 and sc64.dtype == torch.float64
    assert torch.equal(sc32, ops.f64_to_f32(sc64)) and torch.equal(sc64, bm.scores(queries))
    sys_b = bm.search_device(queries)
    assert torch.equal(sys_b.scores, sc32) and torch.equal(sys_b.scores64, sc64)


def test_division_free_rrf_term_is_the_ieee_quotient_for_every_rank(ops):
    """The fused sort forms 1 / (60 + r + 1) without the division's scale / fixup steps (sort.hip, recip_small_int_f64): equal to the
    IEEE division on the device and to NumPy's, bit for bit, for every rank up to 2^20 (the sort itself sees r < 28,672)."""
    n = 1 << 20
    fast, ieee = ops.rrf_terms(n, True).cpu().numpy(), ops.rrf_terms(n, False).cpu().numpy()
    ref = 1.0 / (np.arange(n, dtype=np.float64) + 61.0)
    np.testing.assert_array_equal(ieee, ref)
    np.testing.assert_array_equal(fast, ref)


@pytest.mark.parametrize("N", [300, 9000, 27942])
def test_placed_sequence_that_does_not_fill_the_row(ops, oracle, N):
    """init_rank with holes + row_len < N (a placed form of a partial insertion order): the kernel may not park anything in the outputs'
    unwritten tails here (it forms the low key words a second time instead); entries beyond the list stay -1 / -inf."""
    rng = np.random.default_rng(N + 1)
    S, Q = 3, 3
    _, ranks, orders, lens = synth_systems(rng, S, Q, N)
    rp = [plane(ops, r) for r in ranks]
    e_ins, e_U = oracle.insertion_order(orders, lens, N)
    inv = np.full((Q, N), -1, dtype=np.int32)
    for q in range(Q):
        inv[q, e_ins[q, : e_U[q]]] = np.arange(e_U[q], dtype=np.int32)
    e_U2 = np.minimum(e_U, np.array([N, N - 7, N // 2 + 1], dtype=np.int32)[:Q]).astype(np.int32)   # cut some rows shorter still
    inv = np.where(inv >= e_U2[:, None], -1, inv).astype(np.int32)
    order, sk, rank = ops.sort_rank_fused(rp, dev(lens), "rrf", init_rank=plane(ops, inv), row_len=dev(e_U2), want_rank=True)
    ins2 = e_ins.copy()
    for q in range(Q):
        ins2[q, e_U2[q]:] = -1
    e_order, e_sk, e_rank = oracle.sort_rows_desc(oracle.fuse_rank(ranks, lens, "rrf"), init_order=ins2, row_len=e_U2, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)


def test_encode_from_strings_with_overlapped_tokenisation_equals_encode_from_ids():
    """DenseEncoder.encode(strings) -- sub-batch i + 1 tokenised (the 32,005-piece synthetic-French BPE) on a host thread
    while sub-batch i runs the padding-free forward -- equals the forward over ids tokenised up front in the caller's thread."""
    pytest.importorskip("tokenizers", reason="the synthetic-French BPE needs the optional `tokenizers` wheel (requirements.txt)")
    from fusion_amd import encoders
    from fusion_amd.synth_text import FrenchLike
    enc = encoders.random_init("dpr", device="cuda", size="base", seed=0, tokenizer="synth-fr")
    enc.packed_tokens = 4096                                       # several sub-batches out of a small input
    rng = np.random.default_rng(9)
    texts = FrenchLike().sentences(rng, 300, 3, 60, question=True)
    got = enc.encode(texts, batch_size=16)
    ids, lens = enc.tokenizer.encode_np(texts, enc.max_doc_length, pad_to_max=False)
    exp = torch.empty_like(got)
    for idx in encoders._id_batches(lens, 4096):
        sel = torch.from_numpy(np.ascontiguousarray(idx))
        exp[sel.cuda()] = enc.encode_ids_packed(torch.from_numpy(ids[idx]).cuda(), lens[idx])
    assert float((got - exp).abs().max()) <= 1e-5 * float(exp.abs().max())
    again = enc.encode(texts, batch_size=16)
    assert torch.equal(got, again)                                  # the worker thread changes nothing run to run


@pytest.mark.parametrize("N", [5000, 9000, 27942])
def test_rows_the_fast_form_hands_to_the_generic_launch(ops, oracle, N):
    """Fused scores with thousands of keys under ONE high key word and distinct low words (bcf with list lengths of 2^31 - 1: every term is
    1 - r / 2^31, a sum of three moves by 2^-31 per rank): the fast form's in-place repair gives such a row up, its fused scores are written out
    as a plain float64 row (fuse_flagged_rows_kernel) and the generic eight-pass launch sorts it -- same lists as the two-call form and the oracle."""
    rng = np.random.default_rng(N)
    Q = 3
    ranks = [np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32) for _ in range(3)]
    lens = np.full((3, Q), 2**31 - 1, dtype=np.int32)
    f = oracle.fuse_rank(ranks, lens, "bcf")
    hi = np.sort(f.view(np.uint64) >> 32, axis=1)
    longest = max(int(np.max(np.diff(np.flatnonzero(np.concatenate([[True], row[1:] != row[:-1], [True]]))))) for row in hi)
    assert longest > 2048, longest                                         # beyond what the fast form repairs in place
    rp = [plane(ops, r) for r in ranks]
    order, sk, rank = ops.sort_rank_fused(rp, dev(lens), "bcf", init_rank=rp[0], want_rank=True, covers_all=True)
    o0 = np.argsort(ranks[0], axis=1, kind="stable").astype(np.int32)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(f, init_order=o0, want_rank=True)
    np.testing.assert_array_equal(order.cpu().numpy(), e_order)
    np.testing.assert_array_equal(sk.cpu().numpy(), e_sk)
    np.testing.assert_array_equal(rank.cpu().numpy(), e_rank)
    o2, s2, _ = ops.sort_rows_desc(ops.fuse_rank(rp, dev(lens), "bcf"), init_rank=rp[0], covers_all=True)
    assert torch.equal(order, o2) and torch.equal(sk, s2)


@pytest.mark.parametrize("N", [700, 27942])
def test_raw_c_abi_with_outputs_left_out(ops, oracle, N):
    """fz_sort_rank_fused_desc through ctypes, as a non-Python host would call it, with outputs a caller may not want: no `order` (the kernel
    then has nowhere to park the low sort words and forms them a second time), no `sorted_scores` (nothing parked, nothing stored), rank only."""
    import ctypes as C
    from fusion_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(N)
    Q, S = 3, 3
    ranks = [np.stack([rng.permutation(N) for _ in range(Q)]).astype(np.int32) for _ in range(S)]
    lens = np.full((S, Q), N, dtype=np.int32)
    rp = [plane(ops, r) for r in ranks]
    ld = rp[0].stride(0) if Q > 1 else rp[0].shape[1]
    ptrs = (C.c_void_p * S)(*[r.data_ptr() for r in rp])
    lens_d = dev(lens)
    nws = L.fz_sort_rank_fused_workspace_bytes(Q, N, ld)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    f = oracle.fuse_rank(ranks, lens, "rrf")
    o0 = np.argsort(ranks[0], axis=1, kind="stable").astype(np.int32)
    e_order, e_sk, e_rank = oracle.sort_rows_desc(f, init_order=o0, want_rank=True)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for want in (("order",), ("scores",), ("rank",), ("scores", "rank"), ("order", "scores", "rank")):
        order = torch.full((Q, ld), -7, dtype=torch.int32, device="cuda") if "order" in want else None
        sk = torch.full((Q, ld), -7.0, dtype=torch.float64, device="cuda") if "scores" in want else None
        rank = torch.full((Q, ld), -7, dtype=torch.int32, device="cuda") if "rank" in want else None
        rc = L.fz_sort_rank_fused_desc(ptrs, p(lens_d), S, 0, None, p(rp[0]), None, Q, N, ld, p(order), p(sk), p(rank), p(ws), nws, st)
        assert rc == 0, (want, rc)
        torch.cuda.synchronize()
        if order is not None: np.testing.assert_array_equal(order[:, :N].cpu().numpy(), e_order)
        if sk is not None: np.testing.assert_array_equal(sk[:, :N].cpu().numpy(), e_sk)
        if rank is not None: np.testing.assert_array_equal(rank[:, :N].cpu().numpy(), e_rank)
    assert L.fz_sort_rank_fused_desc(ptrs, p(lens_d), S, 0, None, p(rp[0]), None, Q, N, ld, None, None, None, p(ws), nws - 1, st) == _lib.FZ_ERR_WORKSPACE
