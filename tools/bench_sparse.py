"""SPLADE scoring, sparse index against the dense fp32-MFMA GEMM, on SPLADE-like vectors (a few hundred non-zeros of 32,005 per document, a few
dozen per query, Zipf-distributed terms): times, index size and the agreement of the two score planes.  Usage: python tools/bench_sparse.py [Q]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusion_amd import ops
from tools.bench_kernels import timeit


from bench import splade_like


def main():
    Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N, V = 27942, 32005
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    Dn = ops.normalize_rows(splade_like(rng, N, V, 200, dev))
    Qn = ops.normalize_rows(splade_like(rng, Q, V, 40, dev))
    idx = ops.sparse_index(Dn, V)
    qoff, qt, qw = ops.sparse_rows(Qn, V)
    dense = ops.dot_scores(Qn, Dn)
    sparse = ops.sparse_dot(idx, qoff, qt, qw)
    err = (dense - sparse).abs().max().item()
    res = dict(Q=Q, N=N, V=V, doc_density=ops.density(Dn[:, :V]), query_density=ops.density(Qn[:, :V]), index_MB=(idx.nnz * 8 + idx.toff.numel() * 8) / 1e6,
               dense_MB=N * Dn.shape[1] * 4 / 1e6, max_abs_diff=err, reruns_identical=bool(torch.equal(sparse, ops.sparse_dot(idx, qoff, qt, qw))),
               dense_gemm_ms=timeit(lambda: ops.dot_scores(Qn, Dn), n=3, warm=1), sparse_dot_ms=timeit(lambda: ops.sparse_dot(idx, qoff, qt, qw), n=10, warm=2),
               sparse_rows_ms=timeit(lambda: ops.sparse_rows(Qn, V), n=5, warm=1), index_build_ms=timeit(lambda: ops.sparse_index(Dn, V), n=1, warm=1))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
