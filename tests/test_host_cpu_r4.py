"""Round 4 host-side tests (no GPU): the vectorised packing of the reference's RankedLists against the entry-by-entry route, the
C helper behind it, the rebuilt list-of-dict results, bench.py's self-launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

L = lambda pairs: [{"corpus_id": i, "score": s} for i, s in pairs]


def _same_packing(lists):
    """pack_ranked_lists == _pack_by_dicts up to the id <-> position bijection (any serves: ties follow list order, not position)."""
    from fusion_amd.retrievers.hybrid import _pack_by_dicts, pack_ranked_lists
    ia, Na, pa = pack_ranked_lists(lists)
    ib, Nb, pb = _pack_by_dicts(lists)
    assert Na == Nb and sorted(np.asarray(ia).tolist(), key=repr) == sorted(np.asarray(ib).tolist(), key=repr)
    for n in lists:
        sa, ra, oa, la, sorted_a = pa[n]
        sb, rb, ob, lb, sorted_b = pb[n]
        assert np.array_equal(la, lb) and sorted_a == sorted_b
        for q in range(len(la)):
            m = int(la[q])
            assert np.asarray(ia)[oa[q, :m]].tolist() == np.asarray(ib)[ob[q, :m]].tolist()                # same documents in list order
            assert np.array_equal(sa[q, oa[q, :m]], sb[q, ob[q, :m]], equal_nan=True)                       # same scores
            assert np.array_equal(ra[q, oa[q, :m]], np.arange(m)) and np.all(oa[q, m:] == -1)
            assert (ra[q] >= 0).sum() == m


def test_vectorised_packing_equals_the_entry_by_entry_route():
    rng = np.random.default_rng(3)
    N = 500
    ids = rng.permutation(np.arange(10, 10 * N))[:N]
    mk = lambda keep: [{"corpus_id": int(ids[i]), "score": float(np.float32(v))} for i, v in zip(rng.permutation(N)[:keep], rng.normal(0, 1, keep))]
    _same_packing({"a": [mk(N), mk(N), mk(1)], "b": [mk(300), [], mk(7)]})                        # partial + empty lists
    _same_packing({"s1": [L([(1, 5.0), (2, 4.0), (1, 1.0), (3, 4.5)])], "s2": [L([(3, 1.0), (2, .5), (4, .25)])]})   # duplicate ids (hybrid.py:231)
    _same_packing({"s": [L([(7, float("nan")), (5, 1.0)])]})
    _same_packing({"s": [L([(7, np.float32(0.25)), (9, np.float64(2.0)), (8, 3)])]})              # numpy scalars / ints as scores
    _same_packing({"s": [L([(2 ** 40, 1.0), (3, 0.5)])]})                                          # a sparse id space: searchsorted route
    _same_packing({"s": [L([(2 ** 25, 1.0), (3, 0.5), (77, 0.25)])], "t": [L([(77, 2.0), (2 ** 24 + 1, 1.0)])]})   # a handful of ids over 2^25 values: ADVICE r4 --
    #                                                                                                no 320 MB direct table for them (searchsorted route too)
    _same_packing({"s": [L([("x", 1.0), ("y", 0.5)])], "t": [L([("y", 2.0)])]})                   # string ids: the generic route
    _same_packing({"s": [L([(True, 1.0), (2, 0.5)])]})                                             # bool is not a plain int
    _same_packing({"s": [L([(2 ** 70, 1.0), (2, 0.5)])]})                                          # does not fit int64
    g = json.load(open(os.path.join(GOLDEN, "unsorted_fuse.json")))
    for case in g.values():
        _same_packing(case["lists"])


def test_pyhost_extract_and_build():
    from fusion_amd import _pyhost
    lst = L([(5, 1.5), (2 ** 62, -0.0), (-3, float("inf"))])
    ids, sc = _pyhost.extract(lst)
    assert ids.tolist() == [5, 2 ** 62, -3] and sc.tolist()[0] == 1.5 and np.signbit(sc[1]) and np.isinf(sc[2])
    assert _pyhost.extract(L([("a", 1.0)])) is None and _pyhost.extract(L([(2 ** 64, 1.0)])) is None
    assert _pyhost.extract([{"corpus_id": 1}]) is None and _pyhost.extract([(1, 2.0)]) is None and _pyhost.extract(L([(1, "x")])) is None
    out = _pyhost.build([3, "k", None], [1.0, np.float32(2.0), 7])
    assert out == [{"corpus_id": 3, "score": 1.0}, {"corpus_id": "k", "score": np.float32(2.0)}, {"corpus_id": None, "score": 7}]
    assert list(out[0]) == ["corpus_id", "score"] and type(out[1]["score"]) is np.float32        # key order and score types as the reference's
    with pytest.raises(TypeError):
        _pyhost.build([1, 2], [1.0])


def test_result_lists_keep_the_reference_types():
    from fusion_amd.planes import FusedResult, RankedSystem
    order = torch.tensor([[2, 0, 1, -1], [1, -1, -1, -1]], dtype=torch.int32)
    lens = torch.tensor([3, 1], dtype=torch.int32)
    ids = np.array([10, 20, 30, 40])
    f32 = FusedResult(order=order, scores=torch.tensor([[3., 2., 1., 0.], [5., 0., 0., 0.]]), lens=lens, ids=ids).to_lists()
    assert f32 == [L([(30, 3.0), (10, 2.0), (20, 1.0)]), L([(20, 5.0)])]
    assert type(f32[0][0]["score"]) is np.float32 and type(f32[0][0]["corpus_id"]) is int          # hybrid.py:258: numpy float32 scalars
    f64 = FusedResult(order=order, scores=torch.tensor([[3., 2., 1., 0.], [5., 0., 0., 0.]], dtype=torch.float64), lens=lens, ids=ids).to_lists()
    assert type(f64[0][0]["score"]) is float                                                        # rrf / bcf / 'none': Python floats
    sids = np.array(["a", "b", "c", "d"], dtype=object)
    rs = RankedSystem(scores=torch.tensor([[.1, .2, .3, .4]]), order=torch.tensor([[3, 2, 1, 0]], dtype=torch.int32),
                      rank=torch.tensor([[3, 2, 1, 0]], dtype=torch.int32), lens=torch.tensor([2], dtype=torch.int32), ids=sids, full=False)
    got = rs.to_lists()
    assert [x["corpus_id"] for x in got[0]] == ["d", "c"] and type(got[0][0]["score"]) is float
    assert abs(got[0][0]["score"] - 0.4) < 1e-7


def test_bench_launcher_argv_and_guard():
    sys.path.insert(0, ROOT)
    import bench
    argv = bench.launcher_argv(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, 29511)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=4" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29511"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]                      # the same arguments, after the script
    # --gpus N under a launcher whose WORLD_SIZE disagrees is refused before anything touches a GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_tables_abi_planning_and_validation_without_gpu():
    """The long-table entry points' host side (no launch): workspace sizes, which kernel a call would take, argument checks."""
    import ctypes as C
    from fusion_amd import _lib
    L = _lib.lib()
    i32 = lambda *v: (C.c_int32 * len(v))(*v)
    b4 = L.fz_nsf_tables_workspace_bytes(4, i32(27943, 27943, 27943, 27943), 4)
    b4n = L.fz_nsf_tables_workspace_bytes(4, i32(27943, 27943, 27943, 27943), 5)
    assert 4 * (27943 * 4 + 32768) < b4 < 4 * 160 * 1024 and b4n - b4 >= 4 * 27943 * 4          # NCE adds one value table per system
    assert L.fz_nsf_tables_workspace_bytes(1, i32(38000), 4) > 0 and L.fz_nsf_tables_workspace_bytes(1, i32(40000), 4) == 0   # beyond LDS
    assert L.fz_nsf_tables_workspace_bytes(1, i32(0), 4) == 0 and L.fz_nsf_tables_workspace_bytes(9, i32(5), 4) == 0
    assert L.fz_nsf_tables_header_offset(2, i32(100, 200), 4, 1) > 0 and L.fz_nsf_tables_header_offset(2, i32(100, 200), 4, 2) == C.c_size_t(-1).value
    # which kernel: fake (aligned) plane addresses, no memory behind them -- the function only looks at shapes and alignment
    planes = (C.c_void_p * 4)(0x1000, 0x2000, 0x3000, 0x4000)
    path = lambda S, P, ld=28032, out=0x8000, norm=4: L.fz_nsf_tables_path(planes, None, S, 8, 27942, ld, norm, i32(*P), out)
    assert path(4, [1001] * 4) == 0 and path(4, [27943] * 4) == 1 and path(1, [27943]) == 1 and path(4, [10001] * 4) == 1
    assert path(1, [60000]) == 2 and path(4, [27943] * 4, out=0x8004) == 2 and path(4, [27943] * 4, ld=27943) == 2
    assert path(4, [27943] * 4, norm=1) == _lib.FZ_ERR_ARG and path(4, [27943, 0, 5, 5]) == _lib.FZ_ERR_ARG
    d = (C.c_void_p * 1)(0x1000)
    assert L.fz_nsf_tables_prepare(d, i32(27943), 1, 4, None, 0, None) == _lib.FZ_ERR_WORKSPACE
    assert L.fz_nsf_tables_prepare(d, i32(27943), 1, 2, None, 0, None) == _lib.FZ_ERR_ARG
    assert L.fz_nsf_tables_prepare(d, i32(70000), 1, 4, None, 0, None) == _lib.FZ_ERR_UNSUPPORTED
    w = (C.c_double * 1)(1.0)
    assert L.fz_fuse_nsf_tables_f32(planes, None, w, 1, 8, 27942, 28032, 4, d, i32(27943), None, 0, 0x8000, None, 0, None) == _lib.FZ_ERR_WORKSPACE
    assert L.fz_fuse_nsf_tables_f32(planes, None, w, 1, 0, 27942, 28032, 4, d, i32(27943), None, 0, None, None, 0, None) == _lib.FZ_OK   # empty batch
    assert L.fz_fuse_nsf_tables_f32(planes, None, w, 1, 8, 27942, 28032, 4, d, i32(60000), None, 0, 0x8000, None, 0, None) == _lib.FZ_ERR_UNSUPPORTED
