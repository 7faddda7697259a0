#!/usr/bin/env python3
"""Per-kernel timings at BASELINE.json sizes (one JSON line per kernel). Usage: python tools/bench_kernels.py [names...]"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fusion_amd import ops

HBM, F32, F16 = 8000e9, 157.3e12, 2500e12


def timeit(f, n=10, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def emit(name, ms, work, peak, unit, **kw):
    ach = work / (ms * 1e-3)
    print(json.dumps(dict(kernel=name, ms=round(ms, 4), achieved=round(ach / (1e12 if unit == "TFLOP/s" else 1e9), 2), unit=unit,
                          frac=round(ach / peak, 4), **kw)), flush=True)


def rand_plane(Q, N, g, scale=1.0, shift=0.0):
    p = ops.alloc_plane(Q, N, torch.float32, "cuda")
    p.copy_(torch.randn((Q, N), generator=g, device="cuda") * scale + shift)
    return p


def bench_nsf(Q=1024, N=27942, S=4):
    g = torch.Generator(device="cuda").manual_seed(0)
    planes = [rand_plane(Q, N, g, s + 1, s) for s in range(S)]
    out = ops.alloc_plane(Q, N, torch.float32, "cuda")
    w = [0.25] * S
    for norm in ("min-max", "z-score", "arctan"):
        ms = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, out=out))
        emit(f"fuse_nsf_row_kernel<{norm}> S={S}", ms, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
    so = [ops.sort_rows_desc(p, want_order=True, want_keys=False, want_rank=True) for p in planes]
    orders, ranks = [x[0] for x in so], [x[2] for x in so]
    ms = timeit(lambda: ops.fuse_nsf(planes, None, w, "min-max", out=out, orders=orders))
    emit(f"min-max from list ends + fuse_nsf_elem4_kernel S={S} (ranked systems)", ms, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
    chk = ops.fuse_nsf(planes, None, w, "min-max")
    assert torch.equal(chk, ops.fuse_nsf(planes, None, w, "min-max", orders=orders)), "fast min-max path differs from the reducing kernel"
    ms = timeit(lambda: ops.fuse_nsf(planes, ranks, w, "min-max", out=out))
    emit(f"fuse_nsf_row_kernel<min-max,+validity> S={S}", ms, (2 * S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
    lens = torch.full((S, Q), N, dtype=torch.int32, device="cuda")
    for m in ("rrf", "bcf"):
        ms = timeit(lambda: ops.fuse_rank(ranks, lens, m))
        emit(f"fuse_rank_kernel<{m}> S={S}", ms, (S * 4 + 8) * Q * N, HBM, "GB/s", Q=Q, N=N)
    ms = timeit(lambda: ops.fuse_none(planes, None, w))
    emit(f"fuse_none_kernel S={S}", ms, (S * 4 + 8) * Q * N, HBM, "GB/s", Q=Q, N=N)
    for P in (1001,):   # hybrid.py:391-397: quantile tables of 1001 points
        distr = [torch.quantile(p[:8].flatten()[:1000000].double(), torch.linspace(0, 1, P, device="cuda", dtype=torch.float64)).float().contiguous()
                 for p in planes]
        for norm in ("percentile-rank", "normal-curve-equivalent"):
            ms = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, distr, out=out))
            emit(f"fuse_nsf_table_kernel<{norm}> S={S} P={P}", ms, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
    ms = timeit(lambda: ops.row_stats(planes[0], None, "z-score"))
    emit("row_stats_kernel<z-score>", ms, Q * N * 4, HBM, "GB/s", Q=Q, N=N)


def lleqa_planes(Q, N, g):
    """Four systems shaped like SURVEY 8d/C4: bm25-like (>= 0, ~40 % zeros), cosine-like, SPLADE-like, ColBERT-like."""
    def plane(t):
        p = ops.alloc_plane(Q, N, torch.float32, "cuda"); p.copy_(t); return p
    r = lambda: torch.randn((Q, N), generator=g, device="cuda")
    return [plane(torch.clamp(3.0 * r() - 1.0, min=0.0)), plane(torch.tanh(0.3 * r())), plane(torch.log1p(torch.relu(r()))), plane(20.0 + 4.0 * r())]


def quantile_table(p, P):
    """hybrid.py:389-397: quantiles of a system's pooled non-zero scores, P = n_points + 1 entries."""
    pool = p[:64].flatten()
    pool = torch.sort(pool[pool != 0.0].double()).values
    idx = torch.linspace(0, pool.numel() - 1, P, device="cuda", dtype=torch.float64)
    lo = idx.floor().long(); hi = torch.clamp(lo + 1, max=pool.numel() - 1); f = idx - lo
    return (pool[lo] + (pool[hi] - pool[lo]) * f).float().contiguous()


def bench_tables(N=27942, S=4):
    """percentile-rank / NCE at the table sizes the reference reads (hybrid.py:412,451: 27,943; :374: 10,001): the kernel that
    keeps one system's table in LDS at a time (tables.hip) against fz_fuse_nsf_f32's global-memory search it replaces there."""
    g = torch.Generator(device="cuda").manual_seed(7)
    for Q in (1024, 195):
        planes = lleqa_planes(Q, N, g)
        out = ops.alloc_plane(Q, N, torch.float32, "cuda")
        w = [0.25] * S
        for P in (27943, 10001):
            distr = [quantile_table(p, P) for p in planes]
            for norm in ("percentile-rank", "normal-curve-equivalent"):
                prep = ops.nsf_tables_prepare(distr, norm)
                ms = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, distr, out=out, tables=prep))
                assert ops.last_tables_path == "lds-swap"
                emit(f"fuse_nsf_bigtab_kernel<{norm}> S={S} P={P} (tables prepared once)", ms, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N,
                     search=prep.search_info())
                ms = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, distr, out=out))
                emit(f"fz_nsf_tables_prepare + fuse_nsf_bigtab_kernel<{norm}> S={S} P={P}", ms, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
                if Q == 1024 and P == 27943:
                    ms1 = timeit(lambda: ops.fuse_nsf(planes[:1], None, [1.0], norm, distr[:1], out=out, tables=ops.nsf_tables_prepare(distr[:1], norm)))
                    emit(f"fuse_nsf_bigtab_kernel<{norm}> S=1 P={P} (tune: one system per call)", ms1, 2 * Q * N * 4, HBM, "GB/s", Q=Q, N=N)
                    ref = ops.fuse_nsf(planes, None, w, norm, distr, tables=False)
                    assert torch.equal(ref.view(torch.int32), ops.fuse_nsf(planes, None, w, norm, distr).view(torch.int32)), "swap kernel differs from the global-memory search"
                    ms0 = timeit(lambda: ops.fuse_nsf(planes, None, w, norm, distr, out=out, tables=False), n=3, warm=1)
                    emit(f"fuse_nsf_row_kernel<{norm}> S={S} P={P} (round-3 path: global-memory search)", ms0, (S + 1) * Q * N * 4, HBM, "GB/s", Q=Q, N=N)


def bench_gemm(Q=1024, N=27942, d=768):
    g = torch.Generator(device="cuda").manual_seed(1)
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    Dn = ops.normalize_rows(torch.randn((N, d), generator=g, device="cuda"))
    out = ops.alloc_plane(Q, N, torch.float32, "cuda")
    ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out), n=20)
    emit("dot_scores_kernel", ms, 2.0 * Q * N * d, F32, "TFLOP/s", Q=Q, N=N, d=d)
    for q in (195,):
        Qs = Qn[:q].contiguous()
        ms = timeit(lambda: ops.dot_scores(Qs, Dn), n=20)
        emit("dot_scores_kernel", ms, 2.0 * q * N * d, F32, "TFLOP/s", Q=q, N=N, d=d)
    ms = timeit(lambda: ops.normalize_rows(Dn))
    emit("normalize_rows_kernel", ms, 2 * N * d * 4, HBM, "GB/s", rows=N, d=d)


def bench_splade(Q=1024, N=27942, V=32005):
    """A3: SPLADE vectors scored densely like the reference (hybrid.py:101-103): 1.83 TFLOP at Q=1024."""
    g = torch.Generator(device="cuda").manual_seed(5)
    Vp = ops.round_up(V, 4)
    Dn = torch.zeros((N, Vp), device="cuda")
    for c0 in range(0, N, 4096):
        c1 = min(N, c0 + 4096)
        Dn[c0:c1, :V] = torch.log1p(torch.relu(torch.randn((c1 - c0, V), generator=g, device="cuda") - 1.0))
    Dn = ops.normalize_rows(Dn)
    Qn = torch.zeros((Q, Vp), device="cuda"); Qn[:, :V] = torch.log1p(torch.relu(torch.randn((Q, V), generator=g, device="cuda") - 1.5))
    Qn = ops.normalize_rows(Qn)
    out = ops.alloc_plane(Q, N, torch.float32, "cuda")
    ms = timeit(lambda: ops.dot_scores(Qn, Dn, out=out), n=3, warm=1)
    emit("dot_scores_kernel (SPLADE V=32005)", ms, 2.0 * Q * N * V, F32, "TFLOP/s", Q=Q, N=N, d=V)


def bench_maxsim(N=27942, Qs=(195, 1024)):
    rng = np.random.default_rng(0)
    lens = np.clip(rng.normal(300, 120, N), 16, 512).astype(np.int64)
    off = np.zeros(N + 1, dtype=np.int64); off[1:] = np.cumsum(lens)
    sumL = int(off[-1])
    g = torch.Generator(device="cuda").manual_seed(2)
    Dtok = torch.nn.functional.normalize(torch.randn((sumL, 128), generator=g, device="cuda"), dim=-1).half()
    Doff = torch.from_numpy(off).cuda()
    for Q in Qs:
        Qtok = torch.nn.functional.normalize(torch.randn((Q, 64, 128), generator=g, device="cuda"), dim=-1).half()
        out = ops.alloc_plane(Q, N, torch.float32, "cuda")
        ms = timeit(lambda: ops.maxsim(Qtok, Dtok, Doff, out=out, max_doc_len=512), n=3, warm=1)
        emit("maxsim_kernel", ms, 2.0 * Q * 64 * sumL * 128, F16, "TFLOP/s", Q=Q, N=N, sumL=sumL)


def bench_topk(Q=1024, n=8 * 28672, k=1000):
    g = torch.Generator(device="cuda").manual_seed(3)
    S = ops.alloc_plane(Q, n, torch.float32, "cuda"); S.copy_(torch.rand((Q, n), generator=g, device="cuda"))
    ms = timeit(lambda: ops.topk_rows(S, k), n=5)
    emit("topk_rows (chunk-sort-truncate: the exact fall-back path)", ms, Q * n * 4, HBM, "GB/s", Q=Q, n=n, k=k)

    def stream():   # what the sharded search does with materialised scores: exact head, threshold filter, folds
        st = ops.TopkStream(*ops.topk_rows(S[:, :8192], k), seen=8192)
        st.feed(S[:, 8192:], 8192)
        return st.result()
    ms = timeit(stream, n=5)
    emit("TopkStream: topk_rows head + topk_filter + folds (same rows)", ms, Q * n * 4, HBM, "GB/s", Q=Q, n=n, k=k)


def bench_mmarco(Q=1024, N=8841823 // 8, d=768, k=1000):
    """One 1/8 shard of mMARCO (what each GPU does at G=8), without the all-gather."""
    from fusion_amd.distributed import ShardedDenseIndex
    g = torch.Generator(device="cuda").manual_seed(4)
    Dn = torch.empty((N, d), dtype=torch.float32, device="cuda")
    for c0 in range(0, N, 1 << 19):
        c1 = min(N, c0 + (1 << 19))
        Dn[c0:c1] = ops.normalize_rows(torch.randn((c1 - c0, d), generator=g, device="cuda"))
    Qn = ops.normalize_rows(torch.randn((Q, d), generator=g, device="cuda"))
    idx = ShardedDenseIndex(Dn, 0)
    ms = timeit(lambda: idx.local_topk(Qn, k), n=2, warm=1)
    emit("mmarco shard: GEMM+topk chunks", ms, 2.0 * Q * N * d, F32, "TFLOP/s", Q=Q, N=N, k=k)


ALL = dict(nsf=bench_nsf, tables=bench_tables, gemm=bench_gemm, splade=bench_splade, maxsim=bench_maxsim, topk=bench_topk, mmarco=bench_mmarco)
if __name__ == "__main__":
    for n in (sys.argv[1:] or list(ALL)):
        ALL[n]()
