"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol the header declares (no compute
call without a GPU), host logic of the drop-in module, and the N>1 collective path under gloo (world_size 2)."""
import ctypes as C
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_loads_and_exports_header_symbols():
    from fusion_amd import _lib
    L = _lib.lib()
    assert L.fz_abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "fusion_hip.h")).read()
    declared = set(re.findall(r"\b(fz_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"libfusion_hip.so does not export {name}"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)   # the binding table IS the header
    assert L.fz_strerror(0) == b"ok" and b"invalid" in L.fz_strerror(-1)
    assert L.fz_sort_max_n() == 35840 and L.fz_topk_max_k() >= 1000


def test_abi_argument_validation_without_gpu():
    """Argument errors are reported before any HIP call, so they are checkable on a CPU-only box."""
    from fusion_amd import _lib
    L = _lib.lib()
    assert L.fz_sort_rows_desc(None, 32, None, None, 1, 1, 1, None, None, None, None, None, None, 0, None) == _lib.FZ_ERR_ARG
    assert L.fz_sort_rows_desc(None, 32, None, None, 0, 5, 8, None, None, None, None, None, None, 0, None) == _lib.FZ_OK      # empty batch: nothing to do
    assert L.fz_sort_rows_desc(None, 16, None, None, 1, 1, 1, None, None, None, None, None, None, 0, None) == _lib.FZ_ERR_ARG
    assert L.fz_fuse_rank_f64(None, None, 2, 1, 1, 1, 0, None, None) == _lib.FZ_ERR_ARG
    assert L.fz_dot_scores_f32(None, 4, None, 4, 1, 1, 4, None, 1, None) == _lib.FZ_ERR_ARG
    assert L.fz_topk_workspace_bytes(4, 100000, 1000) > 0 and L.fz_topk_workspace_bytes(4, 1000, 10) <= 256


def test_workspace_planning_is_consistent():
    """Host-side planning code of the C ABI over a grid of shapes (also run under ASan / UBSan, tests/test_sanitizers_cpu.py): workspace
    sizes are finite, monotone in the batch, and every entry point rejects null / negative / inconsistent arguments before any HIP call."""
    from fusion_amd import _lib
    L = _lib.lib()
    ERR = _lib.FZ_ERR_ARG
    for bits in (32, 64):
        prev = 0
        for rows in (1, 7, 195, 1024, 6980):
            for n in (1, 64, 27942, 35840, 35841, 120000, 1105228):
                b = L.fz_sort_workspace_bytes(bits, rows, n)
                assert 0 <= b < (1 << 40)
                if bits == 32 and n <= L.fz_sort_max_n():
                    assert b == 0
            b = L.fz_sort_workspace_bytes(bits, rows, 120000)
            assert b >= prev
            prev = b
    kmax = L.fz_topk_max_k()
    for rows in (1, 195, 1024):
        for n in (10, 8192, 229376, 1105228):
            for k in (1, 10, 1000, kmax):
                assert 0 <= L.fz_topk_workspace_bytes(rows, n, k) < (1 << 40)
        for k, cap in ((10, 64), (1000, 7168), (1024, 7168)):
            assert 0 <= L.fz_topk_update_workspace_bytes(rows, k, cap) < (1 << 40)
            assert 0 <= L.fz_topk_fold_workspace_bytes(rows, k, cap) < (1 << 40)
        for world in (1, 2, 8):
            assert L.fz_topk_allgather_workspace_bytes(world, rows, 1000) >= world * rows * 1000 * 12
        assert 0 <= L.fz_insertion_order_workspace_bytes(rows, 27942) < (1 << 40)
    assert L.fz_sort_max_n_f64() == 28672 and L.fz_tune_max_gold() == 8
    # argument errors, one per family (no HIP call is reached: works without a GPU)
    assert L.fz_normalize_rows_f32(None, 4, 8, 8, None, 8, None) == ERR
    assert L.fz_normalize_rows_f32(None, -1, 8, 8, None, 8, None) == ERR
    assert L.fz_dot_scores_f32(None, 4, None, 4, -1, 1, 4, None, 1, None) == ERR
    assert L.fz_maxsim_f16(None, None, None, 0, 512, 1, 64, 1, 128, None, 1, None) == ERR
    assert L.fz_sort_rows_desc_placed(None, 32, None, None, 1, 1, 1, None, None, None, None, 0, None) == ERR
    # round 5 (ABI 18): rank fusion in the final sort; rows, n, ld -> one flag per row (256-byte rounded) + one float64 row per row
    for rows, n, ld in ((1, 1, 64), (195, 27942, 27968), (1024, 27942, 27968)):
        assert L.fz_sort_rank_fused_workspace_bytes(rows, n, ld) == (rows * 4 + 255) // 256 * 256 + rows * ld * 8
    assert L.fz_sort_rank_fused_workspace_bytes(0, 5, 64) == 0 and L.fz_sort_rank_fused_workspace_bytes(4, 70, 64) == 0
    one = C.c_void_p(64)                                           # (never dereferenced: every call below fails its argument checks)
    assert L.fz_sort_rank_fused_desc(None, None, 0, 0, None, None, None, 1, 1, 1, None, None, None, None, 0, None) == ERR        # S = 0
    assert L.fz_sort_rank_fused_desc(None, None, 9, 0, None, None, None, 1, 1, 1, None, None, None, None, 0, None) == ERR        # S > 8
    assert L.fz_sort_rank_fused_desc(None, None, 2, 7, None, None, None, 1, 1, 1, None, None, None, None, 0, None) == ERR        # method
    assert L.fz_sort_rank_fused_desc(None, None, 2, 0, one, one, None, 1, 1, 1, None, None, None, None, 0, None) == ERR          # both sequences
    assert L.fz_sort_rank_fused_desc(None, None, 2, 0, None, None, None, 1, 4, 2, None, None, None, None, 0, None) == ERR        # ld < n
    assert L.fz_sort_rank_fused_desc(None, None, 2, 0, None, None, None, 1, 1, 1, None, None, None, None, 0, None) == ERR        # no rank planes
    assert L.fz_sort_rank_fused_desc(None, None, 2, 0, None, None, None, 0, 1, 1, None, None, None, None, 0, None) == _lib.FZ_OK  # nothing to do
    assert L.fz_rrf_terms_f64(-1, 1, None, None) == ERR and L.fz_rrf_terms_f64((1 << 20) + 1, 1, None, None) == ERR and L.fz_rrf_terms_f64(4, 1, None, None) == ERR
    assert L.fz_rrf_terms_f64(0, 1, None, None) == _lib.FZ_OK
    assert L.fz_bm25_scores_f64_f32(None, None, None, None, None, None, None, 1.0, 2.5, 0.2, None, None, 1, 1, None, 1, None, 1, None) == ERR
    assert L.fz_bm25_scores_f64_f32(None, None, None, None, None, None, None, 1.0, 2.5, 0.2, None, None, 1, 4, one, 4, one, 2, None) == ERR   # lds32 < N
    assert L.fz_fuse_rank_f64(None, None, 0, 1, 1, 1, 0, None, None) == ERR
    assert L.fz_fuse_rank_f64(None, None, 99, 1, 1, 1, 0, None, None) == ERR
    assert L.fz_row_stats_f32(None, None, 1, 1, 1, 1, None, None, None) == ERR
    assert L.fz_fuse_nsf_f32(None, None, None, 1, 1, 1, 1, 1, None, None, None, 0, None, None) == ERR
    assert L.fz_fuse_nsf_stats_f32(None, None, None, 1, 1, 1, 1, 1, None, None, None, None, None, 0, None, None) == ERR
    assert L.fz_fuse_none_f64(None, None, None, 1, 1, 1, 1, None, None) == ERR
    assert L.fz_fuse_wsum_f64(None, None, None, None, None, 1, 1, 1, 1, None, None) == ERR
    assert L.fz_insertion_order(None, None, 1, 1, 1, 1, None, None, None, None, 0, None) == ERR
    assert L.fz_topk_rows_f32(None, 1, 10, 10, 5, 0, None, None, None, 0, None) == ERR
    assert L.fz_topk_rows_f32(None, 1, 10, 10, kmax + 1, 0, None, None, None, 0, None) in (ERR, _lib.FZ_ERR_UNSUPPORTED)
    assert L.fz_topk_merge(None, None, 2, 1, 10, None, None, None) == ERR
    assert L.fz_topk_allgather(None, None, 1, 10, None, 2, None, None, None, 0, None) == ERR
    assert L.fz_bm25_scores_f64(None, None, None, None, None, None, None, 1.0, 2.5, 0.2, None, None, 1, 1, None, 1, None) == ERR
    assert L.fz_bm25_slice_offsets(None, None, 5, 10, None, None) == ERR and L.fz_bm25_slice_docs() == 3584
    # ABI 19: the posting-value table and the scoring walk over it
    assert L.fz_bm25_posting_values_f64(None, None, None, None, None, 3, 5, 2.5, None, None) == ERR
    assert L.fz_bm25_posting_values_f64(None, None, None, None, None, -1, 5, 2.5, None, None) == ERR
    assert L.fz_bm25_posting_values_f64(None, None, None, None, None, 0, 0, 2.5, None, None) == 0
    assert L.fz_bm25_scores_pv_f64_f32(None, None, None, None, None, None, 1, 1, None, 1, None, 0, None) == ERR
    assert L.fz_bm25_scores_pv_f64_f32(None, None, None, None, None, None, 1, 4, one, 2, None, 0, None) == ERR      # lds < N
    assert L.fz_bm25_scores_pv_f64_f32(None, None, None, None, None, None, 0, 4, None, 4, None, 0, None) == 0
    # ABI 19: the lexical ranking sort
    assert L.fz_sort_rows_desc_lexical(None, None, 1, 4, 2, None, None, None, None, None, 0, None) == ERR             # ld < n
    assert L.fz_sort_rows_desc_lexical(None, None, 1, 20000, 20000, None, None, None, None, None, 0, None) == ERR     # no keys
    assert L.fz_sort_rows_desc_lexical(None, None, 0, 20000, 20000, None, None, None, None, None, 0, None) == 0
    assert L.fz_sort_rows_desc_lexical(one, None, 1, 20000, 20000, one, None, None, None, None, 0, None) == _lib.FZ_ERR_WORKSPACE
    # ABI 19: TF-IDF scoring (bm25.py:108-115) -- null planes, ld < N, empty problems
    assert L.fz_tfidf_scores_f64(None, None, None, None, None, None, None, 1, 1, None, 1, None, 0, None) == ERR
    assert L.fz_tfidf_scores_f64(None, None, None, None, None, None, None, 1, 4, one, 2, None, 0, None) == ERR      # lds < N
    assert L.fz_tfidf_scores_f64(None, None, None, None, None, None, None, 1, 4, one, 4, one, 2, None) == ERR       # lds32 < N
    assert L.fz_tfidf_scores_f64(None, None, None, None, None, None, None, 0, 4, None, 4, None, 0, None) == 0       # Q = 0: nothing to do
    assert L.fz_gold_ranks_f32(None, None, None, None, 2, 1, 1, 1, 1, None, None) == ERR
    assert L.fz_tune_metrics_f64(None, None, None, 1, None, None, None, 1, None, 1, 0, 0, 0, 1, 1, None, None) == ERR
    assert L.fz_attn_varlen_f32(None, 1, None, 1, 12, 64, 0.125, None, 1, None) == ERR
    assert L.fz_add_layernorm_f32(None, 1, None, 1, None, None, 1e-5, 1, 768, None, 1, None) == ERR
    assert L.fz_gelu_f32(None, None, 4, None) == ERR
    assert L.fz_attn_varlen_f16(None, 1, None, 1, 12, 64, 0.125, None, 1, None) == ERR
    assert L.fz_attn_varlen_f16_amp(None, 1, None, 1, 12, 64, 0.125, None, 1, None) == ERR
    assert L.fz_add_layernorm_x16(None, 1, None, 1, None, None, 1e-5, 1, 768, None, 1, None, 0, None) == ERR
    assert L.fz_gelu_f16(None, None, 8, None) == ERR
    assert L.fz_sparse_dot_f32(None, None, None, None, None, None, None, 1, 1, None, 1, None) == ERR and L.fz_sparse_slice_docs() == 7168
    assert L.fz_sparse_slice_offsets(None, None, 5, 10, None, None) == ERR
    assert L.fz_segment_mean_f32(None, 1, None, 1, 768, None, 1, None) == ERR


def test_integration_md_binding_matches_the_abi():
    """INTEGRATION.md's reference-side binding is executable documentation: its argtypes must be the binding table's
    (fusion_amd/_lib.py == include/fusion_hip.h) and every call in it must pass exactly that many arguments -- round 2's
    snippet had rotted to a 13-argument fz_sort_rows_desc."""
    import ast
    from fusion_amd import _lib
    from helpers import integration_snippet
    code = integration_snippet(ROOT)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)             # module level: loads the library, checks the ABI version, sets argtypes
    L = ns["L"]
    bound = [n for n in _lib.EXPORTS if getattr(getattr(L, n), "argtypes", None) is not None]
    assert {"fz_sort_rows_desc", "fz_sort_rank_fused_desc", "fz_sort_rank_fused_workspace_bytes", "fz_sort_workspace_bytes"} <= set(bound)
    for n in bound:
        assert list(getattr(L, n).argtypes) == list(_lib._PROTOS[n][1]), n
    calls = [c for c in ast.walk(ast.parse(code)) if isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute)
             and isinstance(c.func.value, ast.Name) and c.func.value.id == "L" and c.func.attr.startswith("fz_")]
    assert len(calls) >= 4
    for c in calls:
        assert len(c.args) == len(_lib._PROTOS[c.func.attr][1]) and not c.keywords, (c.func.attr, len(c.args))
    assert f"fz_abi_version() == {_lib.ABI_VERSION}" in code


def test_no_cpu_fallback():
    from fusion_amd import ops
    with pytest.raises(TypeError, match="no CPU path"):
        ops.sort_rows_desc(torch.zeros((2, 8)))
    with pytest.raises(TypeError):
        ops.dot_scores(torch.zeros((2, 8)), torch.zeros((3, 8)))
    if not torch.cuda.is_available():
        from fusion_amd.retrievers.hybrid import Aggregator
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            Aggregator.fuse({"a": [[{"corpus_id": 1, "score": 1.0}]]}, "rrf")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under fusion_amd/ (or the CLI shim) may import, load or link it."""
    roots = [os.path.join(ROOT, "fusion_amd"), os.path.join(ROOT, "src")]
    for root in roots:
        for dp, _, fs in os.walk(root):
            for f in fs:
                if not f.endswith((".py", ".hip", ".h")) and f != "Makefile":
                    continue
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libfusion_oracle" not in src and "fusion_oracle.c" not in src and "oracle/" not in src, f


def test_metrics_match_reference_golden():
    from fusion_amd.utils.metrics import Metrics
    g = json.load(open(os.path.join(GOLDEN, "metrics.json")))
    ev = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    for c in g["cases"]:
        got = ev.compute_all_metrics(c["gold"], c["pred"])
        assert list(got) == list(c["scores"])           # same keys, same order (CSV schema, hybrid.py:420-425)
        for k, v in c["scores"].items():
            assert float(got[k]) == pytest.approx(v, rel=0, abs=1e-15), k
    k9 = Metrics([1, 2, 500], [2], [2], [2]).compute_all_metrics([[1, 2], [9]], [[1, 3, 2], [4, 9, 5]])
    assert {k: float(v) for k, v in k9.items()} == g["kat9"]
    assert Metrics([1]).reciprocal_rank([1], [], 10) == 0.0   # reference raises here: guarded


def test_weight_grid_counts():
    from fusion_amd.retrievers.hybrid import weight_grid
    assert [len(weight_grid([f"s{i}" for i in range(S)])) for S in (2, 3, 4)] == [21, 231, 1771]   # hybrid.py:405-409
    g = weight_grid(["bm25", "dpr"])
    assert g[0] == {"bm25": 0.0, "dpr": 1.0} and all(np.isclose(sum(w.values()), 1.0) for w in g)


def test_cli_parser_matches_reference_flags():
    from fusion_amd.retrievers.hybrid import build_parser
    a, unknown = build_parser().parse_known_args(
        "--data_split test --models_domain legal --run_bm25 --run_dpr --fusion nsf --normalization z-score "
        "--tune_linear_fusion_weight --output_dir output/testing --some_unknown_flag 3".split())
    assert a.data_split == "test" and a.run_bm25 and a.run_dpr and not a.run_colbert and a.fusion == "nsf"
    assert a.normalization == "z-score" and a.tune_linear_fusion_weight and unknown == ["--some_unknown_flag", "3"]


def test_run_hybrid_sh_argument_errors():
    import subprocess
    sh = os.path.join(ROOT, "scripts", "run_hybrid.sh")
    r = subprocess.run(["bash", sh, "train", "legal"], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR" in r.stdout
    r = subprocess.run(["bash", sh, "test", "medical"], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR" in r.stdout
    r = subprocess.run(["bash", sh, "test", "legal", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR" in r.stdout
    r = subprocess.run(["bash", sh, "dev", "general", "--tune_linear_fusion_weight"], capture_output=True, text=True, env={**os.environ, "DRY_RUN": "1"})
    assert r.returncode == 0
    lines = [l for l in r.stdout.splitlines() if "hybrid.py" in l]
    assert len(lines) == 11 * (3 + 1 + 1)   # 11 combos x (nsf x 3 normalisers + bcf + rrf): NORMALIZERS reset per combo


def test_shard_bounds():
    from fusion_amd.distributed import shard_bounds
    for n, w in [(8841823, 8), (27942, 3), (5, 8), (0, 2)]:
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(8841823, 8, 0) == (0, 1105228)


def test_encoders_tiny_cpu_shapes():
    from fusion_amd import encoders
    tok = encoders.HashTokenizer(512)
    a, m = tok(["le chat noir", "loi"], 16)
    b, _ = tok(["le chat noir", "loi"], 16)
    assert torch.equal(a, b) and a.shape == (2, 5) and m.sum().item() == 5 + 3
    enc = encoders.random_init("dpr", device="cpu", size="tiny")
    e = enc.encode(["le chat noir dort", "article premier du code civil", "x"], batch_size=2)
    assert e.shape == (3, 64) and torch.isfinite(e).all()
    e2 = enc.encode(["x", "le chat noir dort"], batch_size=8)
    assert torch.allclose(e2[0], e[2], atol=1e-5) and torch.allclose(e2[1], e[0], atol=1e-5)   # order restored after length sort
    sp = encoders.random_init("splade", device="cpu", size="tiny")
    v = sp.encode(["le chat noir"], query_mode=True)
    assert v.shape == (1, 512) and (v >= 0).all()          # log1p(relu(.)) >= 0 (splade.py:94)
    cb = encoders.random_init("colbert", device="cpu", size="tiny")
    q = cb.encode_queries(["le chat"])
    assert q.shape == (1, 64, 128) and q.dtype == torch.float16
    assert torch.allclose(q.float().norm(dim=-1), torch.ones(1, 64), atol=2e-3)
    D, off = cb.encode_docs(["le chat noir dort", "loi", "a b c d e f"])
    assert off.tolist() == [0, 6, 9, 17] and D.shape == (17, 128)


# ---- N > 1: the collective path under gloo, world_size 2, merge done by the oracle ------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fusion_amd.distributed import allgather_rows, allgather_topk, shard_bounds
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    Q, N, k = 5, 4001, 50
    S = np.round(rng.normal(0, 1, (Q, N)), 2).astype(np.float32)       # identical on both ranks; ties on purpose
    lo, hi = shard_bounds(N, world, rank)
    ls, li = oracle.topk_rows(S[:, lo:hi], k, id_base=lo)

    def merge(gs, gi):
        a, b = oracle.topk_merge(gs.numpy(), gi.numpy())
        return torch.from_numpy(a), torch.from_numpy(b)
    gs, gi = allgather_topk(torch.from_numpy(ls), torch.from_numpy(li), merge_fn=merge)
    es, ei = oracle.topk_rows(S, k)
    # query-sharded encoder outputs -> every rank holds all rows, in order (uneven split: 7 rows over 2 ranks)
    E = torch.arange(7 * 3, dtype=torch.float32).view(7, 3)
    elo, ehi = shard_bounds(7, world, rank)
    rows_ok = bool(torch.equal(allgather_rows(E[elo:ehi].clone(), 7), E)) and bool(torch.equal(allgather_rows(E[2 * rank: 2 * rank + 2].clone(), 4), E[:4]))
    q.put((rank, bool(np.array_equal(gs.numpy(), es)) and rows_ok, bool(np.array_equal(gi.numpy(), ei))))
    dist.destroy_process_group()


def test_allgather_topk_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps: p.join(30)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] for r in res), res      # sharded top-k == unsharded top-k, on every rank


def test_bucketed_encode_equals_padded_encode():
    from fusion_amd import encoders
    enc = encoders.random_init("dpr", device="cpu", size="tiny")
    rng = np.random.default_rng(0)
    n, L = 37, 24
    lens = rng.integers(3, L + 1, n)
    ids = rng.integers(7, 500, (n, L))
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = np.where(mask == 1, ids, 1)
    a = enc.encode_ids(torch.from_numpy(ids), torch.from_numpy(mask))
    b = enc.encode_ids_bucketed(torch.from_numpy(ids), torch.from_numpy(mask), lens, n_buckets=5)
    assert torch.allclose(a, b, atol=2e-6)     # padding never attends: trimming it changes nothing but rounding


def test_metrics_from_gold_ranks_equals_metrics_class():
    """The rank-based formulas of the tuning sweep == Metrics on explicit ranked lists."""
    from fusion_amd.utils.metrics import Metrics, metrics_from_gold_ranks
    rng = np.random.default_rng(4)
    Q, N = 23, 1500
    gold = [sorted(rng.choice(N, size=int(rng.integers(1, 7)), replace=False).tolist()) for _ in range(Q)]
    pred = [rng.permutation(N).tolist() for _ in range(Q)]
    ev = Metrics(recall_at_k=[5, 10, 20, 50, 100, 200, 500, 1000], map_at_k=[10, 100], mrr_at_k=[10, 100], ndcg_at_k=[10, 100])
    exp = ev.compute_all_metrics(gold, pred)
    G = max(len(g) for g in gold)
    ranks = np.full((1, Q, G), np.iinfo(np.int64).max, dtype=np.int64)
    for q in range(Q):
        inv = {d: i for i, d in enumerate(pred[q])}
        for i, g in enumerate(gold[q]):
            ranks[0, q, i] = inv[g]
    got = metrics_from_gold_ranks(ranks, np.array([len(g) for g in gold]), np.full(Q, N))[0]
    assert list(got) == list(exp)
    for k in exp:
        assert got[k] == pytest.approx(float(exp[k]), rel=0, abs=1e-14), k


def test_fused_forward_equals_hf_forward():
    from fusion_amd import encoders
    enc = encoders.random_init("dpr", device="cpu", size="tiny")
    rng = np.random.default_rng(1)
    n, L = 41, 30
    lens = rng.integers(2, L + 1, n)
    ids = rng.integers(7, 500, (n, L))
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = np.where(mask == 1, ids, 1)
    a = enc.encode_ids(torch.from_numpy(ids), torch.from_numpy(mask))
    b = enc.encode_ids_fused(torch.from_numpy(ids), lens, n_buckets=4)
    assert torch.allclose(a, b, atol=3e-6), float((a - b).abs().max())


def test_corpus_cache_and_cross_encoder(tmp_path):
    from fusion_amd import encoders
    docs = ["le chat noir dort", "article premier du code civil", "loi"]
    k1 = encoders.corpus_cache_key("ckpt-a", docs, "dpr")
    assert k1 == encoders.corpus_cache_key("ckpt-a", docs, "dpr")
    assert k1 != encoders.corpus_cache_key("ckpt-b", docs, "dpr") != encoders.corpus_cache_key("ckpt-a", docs[:2], "dpr")
    calls = []

    def compute():
        calls.append(1)
        return (torch.arange(12, dtype=torch.float32).reshape(3, 4), torch.tensor([0, 2, 5], dtype=torch.int64))
    a = encoders.cached_tensors(str(tmp_path), k1, ["emb", "off"], compute)
    b = encoders.cached_tensors(str(tmp_path), k1, ["emb", "off"], compute)
    assert len(calls) == 1 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and b[1].dtype == torch.int64
    ce = encoders.random_cross_encoder(device="cpu", size="tiny")
    s = ce.predict([("chat", docs[0]), ("chat", docs[1]), ("loi", docs[2])])
    assert s.shape == (3,) and torch.isfinite(s).all()
    from fusion_amd.retrievers.hybrid import Ranker
    out = Ranker.cross_encoder_search(["chat", "loi"], [{7: docs[0], 8: docs[1]}, [{"corpus_id": 9, "score": 1.0}]], "x", model=ce,
                                      corpus={7: docs[0], 8: docs[1], 9: docs[2]})
    assert sorted(x["corpus_id"] for x in out[0]) == [7, 8] and out[0][0]["score"] >= out[0][1]["score"] and out[1][0]["corpus_id"] == 9
    with pytest.raises(RuntimeError):   # without an injected model the checkpoint is loaded onto the GPU: no GPU here, no CPU fallback
        Ranker.cross_encoder_search(["q"], [{1: "d"}], "maastrichtlawtech/monobert-legal-french")


def test_attn_strip_table_and_token_batches():
    """Host helpers of the padding-free encoder: the strip table covers every query row exactly once, longest sequences first;
    sub-batches are cut by token budget and cover every sentence exactly once."""
    from fusion_amd import encoders, ops
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 200, 57)
    strips, cu = ops.attn_strips(lens)
    assert strips.dtype == np.int32 and strips.shape[1] == 4 and cu.dtype == np.int32 and cu[-1] == lens.sum()
    covered = np.zeros(int(lens.sum()), dtype=np.int64)
    for first, L, q0, _ in strips:
        assert 0 <= q0 < L and q0 % 32 == 0
        covered[first + q0: first + min(q0 + 32, L)] += 1
    assert (covered == 1).all()                                   # every token row is a query of exactly one strip
    assert (np.diff(strips[:, 1]) <= 0).all()                      # longest sequences first
    assert ops.attn_strips(np.zeros(0, dtype=np.int64))[0].shape == (0, 4)

    enc = encoders.random_init("dpr", device="cpu", size="tiny")
    enc.packed_tokens = 64
    texts = [" ".join("w%d" % rng.integers(0, 50) for _ in range(int(k))) for k in rng.integers(1, 40, 33)]
    seen = []
    for idx, ids, ln in encoders._token_batches(enc, texts, enc.max_doc_length, 4):
        assert ids.shape[0] == len(idx) == len(ln) and ids.shape[1] == ln.max()
        seen += idx
    assert sorted(seen) == list(range(len(texts)))
    assert encoders.PackedBertForward.supports(type("C", (), dict(hidden_size=768, num_attention_heads=12, hidden_act="gelu")))
    assert not encoders.PackedBertForward.supports(type("C", (), dict(hidden_size=64, num_attention_heads=4, hidden_act="gelu")))
    assert not encoders.PackedBertForward.supports(type("C", (), dict(hidden_size=768, num_attention_heads=12, hidden_act="gelu_new")))
