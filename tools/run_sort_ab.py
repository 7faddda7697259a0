#!/usr/bin/env python3
"""Row sort: bucket ranking against the digit passes (FZ_SORT_BUCKET_RANK=0) on the hot path's three sorts -- same outputs, time of each.

  dpr    float32 cosine scores of random unit vectors (dpr_rank)
  bm25   float64 BM25 scores of bench.py's synthetic index (bm25_rank)
  rrf    float64 RRF scores of the two, placed by first-insertion rank (final_order)
  unif   float32 / float64 uniform keys, tie-heavy and constant-prefix rows (the cases the fallbacks exist for)

Usage: python tools/run_sort_ab.py [Q]
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from fusion_amd import ops


def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def both(tag, f):
    out = {}
    for mode in ("1", "0"):
        os.environ["FZ_SORT_BUCKET_RANK"] = mode
        res = f()
        torch.cuda.synchronize()
        out[mode] = ([None if x is None else x.clone() for x in res], timeit(f))
    same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(out["1"][0], out["0"][0]))
    print(f"{tag:34s} bucket {out['1'][1]:.4f} ms   digits {out['0'][1]:.4f} ms   equal {same}", flush=True)
    os.environ["FZ_SORT_BUCKET_RANK"] = "1"
    return same


def main():
    Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N, d = 27942, 768
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    ok = True
    # dpr
    D = ops.normalize_rows(torch.randn((N, d), generator=g, device=dev))
    Qe = ops.normalize_rows(torch.randn((Q, d), generator=g, device=dev))
    S = ops.dot_scores(Qe, D)
    st = torch.empty((4, Q), dtype=torch.float32, device=dev)
    ok &= both("dpr f32 order+keys+rank+stats", lambda: ops.sort_rows_desc(S, want_rank=True, stats_out=st) + (st,))
    ok &= both("dpr f32 order+rank", lambda: ops.sort_rows_desc(S, want_keys=False, want_rank=True))
    # bm25 (bench.py's synthetic index and query terms)
    rng = np.random.default_rng(7)
    V, lens, tok, doc, p = bench.synth_bm25_index(N, np.random.default_rng(99))
    key = tok.astype(np.int64) * N + doc
    uniq, tf = np.unique(key, return_counts=True)
    pt, pd = uniq // N, uniq % N
    df = np.bincount(pt, minlength=V)
    idf = np.log10((N - df + 0.5) / (df + 0.5))
    toff = np.zeros(V + 1, dtype=np.int64); np.cumsum(df, out=toff[1:])
    qn = rng.integers(4, 16, Q)
    qterms = rng.choice(V, size=int(qn.sum()), p=p).astype(np.int32)
    qoff = np.zeros(Q + 1, dtype=np.int64); np.cumsum(qn, out=qoff[1:])
    t = lambda x: torch.from_numpy(x).to(dev)
    B = ops.bm25_scores(t(toff), t(pd.astype(np.int32)), t(tf.astype(np.int32)), t(idf), t(lens.astype(np.int32)), float(lens.mean()), 2.5, 0.2,
                        t(qoff), t(qterms), Q, N)
    ok &= both("bm25 f64 order+keys+rank+stats", lambda: ops.sort_rows_desc(B, want_rank=True, stats_out=st) + (st,))
    # rrf of the two rankings, placed by first insertion (bm25's list first)
    _, _, r_d = ops.sort_rows_desc(S, want_keys=False, want_rank=True)
    _, _, r_b = ops.sort_rows_desc(B, want_keys=False, want_rank=True)
    F = 1.0 / (60.0 + r_b.double() + 1.0) + 1.0 / (60.0 + r_d.double() + 1.0)
    Fp = ops.alloc_plane(Q, N, torch.float64, dev); Fp.copy_(F)
    ok &= both("rrf f64 placed order+keys", lambda: ops.sort_rows_desc(Fp, init_rank=r_b, covers_all=True))
    # fallbacks
    U = ops.alloc_plane(Q, N, torch.float32, dev); U.copy_(torch.rand((Q, N), generator=g, device=dev) * 2 - 1)
    ok &= both("uniform f32 [-1,1)", lambda: ops.sort_rows_desc(U, want_rank=True))
    Tz = U.clone(); Tz[:, ::2] = 0.0
    ok &= both("half zeros f32", lambda: ops.sort_rows_desc(Tz, want_rank=True))
    Tq = ops.alloc_plane(Q, N, torch.float32, dev); Tq.copy_(torch.round(U * 300) / 300)
    ok &= both("600 distinct values f32", lambda: ops.sort_rows_desc(Tq, want_rank=True))
    Bz = B.clone(); Bz[:, 1::3] = 0.0
    ok &= both("bm25 f64 with a third zeros", lambda: ops.sort_rows_desc(Bz, want_rank=True))
    W = ops.alloc_plane(Q, N, torch.float32, dev); W.copy_(torch.randn((Q, N), generator=g, device=dev) * 1e-3 + 1.0)
    ok &= both("f32 narrow range around 1", lambda: ops.sort_rows_desc(W, want_rank=True))
    X = ops.alloc_plane(Q, N, torch.float32, dev); X.copy_(torch.randn((Q, N), generator=g, device=dev).exp() ** 4)
    ok &= both("f32 heavy tail", lambda: ops.sort_rows_desc(X, want_rank=True))
    rl = torch.randint(4000, N + 1, (Q,), device=dev, dtype=torch.int32)
    ok &= both("dpr f32 ragged row_len", lambda: ops.sort_rows_desc(S, row_len=rl, want_rank=True))
    for n in (5000, 12000, 16384):
        kk = ops.alloc_plane(Q, n, torch.float64, dev); kk.copy_(torch.randn((Q, n), generator=g, device=dev, dtype=torch.float64))
        ok &= both(f"f64 normal {Q}x{n}", lambda: ops.sort_rows_desc(kk, want_rank=True))
    print("ALL EQUAL" if ok else "MISMATCH", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
