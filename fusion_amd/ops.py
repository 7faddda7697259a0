"""torch-tensor wrappers over the C ABI (include/fusion_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every op below passes
raw device pointers + the stream to libfusion_hip.so.  No op has a PyTorch/CPU fallback: a CPU tensor
raises, a missing library raises (see _lib.lib()).

"Plane" = 2-D tensor [rows, n] with stride (ld, 1), ld >= n, ld a multiple of 64 elements so every row
starts on a 256-byte boundary (16-byte vector loads).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import NORMS, RANK_METHODS, check

_PAD = 64


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def alloc_plane(rows: int, n: int, dtype, device, fill=None) -> torch.Tensor:
    ld = max(round_up(n, _PAD), _PAD)
    base = torch.empty((max(rows, 1), ld), dtype=dtype, device=device) if fill is None else \
        torch.full((max(rows, 1), ld), fill, dtype=dtype, device=device)
    return base[:rows, :n]


def as_plane(t: torch.Tensor) -> torch.Tensor:
    """Return t if rows are 16-byte aligned with unit inner stride, else copy it into a padded plane."""
    assert t.dim() == 2
    esz = t.element_size()
    row_ok = t.shape[0] <= 1 or (t.stride(0) >= t.shape[1] and (t.stride(0) * esz) % 16 == 0)
    if t.stride(1) == 1 and row_ok and t.data_ptr() % 16 == 0:
        return t
    p = alloc_plane(t.shape[0], t.shape[1], t.dtype, t.device)
    p.copy_(t)
    return p


def _ld(t: torch.Tensor) -> int:
    if t.shape[0] > 1:
        return t.stride(0)
    return max(t.stride(0), t.shape[1]) if t.dim() == 2 else t.shape[-1]


def _dev(t: torch.Tensor, dtype=None, what="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{what}: expected a tensor on the GPU (fusion_amd has no CPU path), got {type(t).__name__}"
                        f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{what}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{what}: innermost dimension must be contiguous")
    return t


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need(cond: bool, msg: str):
    """Host-side shape contract of a kernel launch: a mismatch would be an out-of-bounds access on the device."""
    if not cond:
        raise ValueError(msg)


def _same_shape(ts, what: str):
    shapes = {tuple(t.shape) for t in ts if t is not None}
    _need(len(shapes) <= 1, f"{what}: planes differ in shape: {sorted(shapes)}")


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_RESERVED_WS: dict = {}


def _reserved_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """A workspace that a kernel RESERVES but practically never touches (the fused sort's plane for rows its fast form flags: rows x ld x 8
    bytes, 229 MB at 1024 x 27,942), kept per (device, stream) and grown on demand instead of being taken from the allocator on every call
    (ADVICE r5).  Safe because calls on one stream are ordered and the kernel initialises what it reads (its row flags) itself; another
    stream gets its own buffer."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _RESERVED_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        _RESERVED_WS.pop(key, None)
        ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        _RESERVED_WS[key] = ws
    return ws


def _ptr_array(ts):
    A = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        A[i] = None if t is None else t.data_ptr()
    return A


def _same_ld(*ts):
    lds = {_ld(t) for t in ts if t is not None and t.shape[0] > 1}
    if len(lds) > 1:
        raise ValueError(f"planes passed to one call must share the row stride, got {sorted(lds)}")
    return lds.pop() if lds else max(_ld(t) for t in ts if t is not None)


def harmonise(ts: list):
    """One C-ABI call takes ONE row stride for all its planes: copy the odd ones out (None entries pass through)."""
    real = [t for t in ts if t is not None]
    if not real:
        return ts
    rows, n = real[0].shape
    want = max(round_up(n, _PAD), _PAD)
    lds = {_ld(t) for t in real} if rows > 1 else {want}
    if len(lds) == 1 and all(t.data_ptr() % 16 == 0 and t.stride(1) == 1 for t in real):
        return ts
    out = []
    for t in ts:
        if t is None or (rows > 1 and _ld(t) == want and t.data_ptr() % 16 == 0 and t.stride(1) == 1):
            out.append(t)
        else:
            p = alloc_plane(rows, n, t.dtype, t.device)
            p.copy_(t)
            out.append(p)
    return out


# ---------------------------------------------------------------------------------------
# K1 scoring
# ---------------------------------------------------------------------------------------
def pad_dim(X: torch.Tensor, mult: int = 4) -> torch.Tensor:
    """Zero-pad the embedding dimension to a multiple of `mult` floats and 16-byte row alignment
    (zeros change neither norms nor dot products)."""
    d = X.shape[1]
    dp = round_up(d, mult)
    if dp == d and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0 and X.stride(1) == 1:
        return X
    Y = torch.zeros((X.shape[0], dp), dtype=X.dtype, device=X.device)
    Y[:, :d] = X
    return Y


def normalize_rows(X: torch.Tensor) -> torch.Tensor:
    """F.normalize(x, p=2, dim=-1) -- reference: util.cos_sim / splade/base.py:195-196."""
    _dev(X, torch.float32, "normalize_rows(X)")
    X = pad_dim(X)
    Y = torch.empty_like(X, memory_format=torch.contiguous_format)
    ldx = X.stride(0) if X.shape[0] > 1 else X.shape[1]
    check(_lib.lib().fz_normalize_rows_f32(_ptr(X), X.shape[0], X.shape[1], ldx, _ptr(Y), Y.shape[1], _stream(X)), "fz_normalize_rows_f32")
    return Y


def dot_scores(Qn: torch.Tensor, Dn: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """scores = Qn @ Dn.T on fp32 MFMA -- reference: torch.mm (splade/base.py:197), util.dot_score."""
    _dev(Qn, torch.float32, "dot_scores(Qn)")
    _dev(Dn, torch.float32, "dot_scores(Dn)")
    Qn, Dn = pad_dim(Qn), pad_dim(Dn)
    if Qn.shape[1] != Dn.shape[1]:
        raise ValueError(f"embedding dims differ: {Qn.shape[1]} vs {Dn.shape[1]}")
    Q, N, d = Qn.shape[0], Dn.shape[0], Qn.shape[1]
    if out is None:
        out = alloc_plane(Q, N, torch.float32, Qn.device)
    else:
        _dev(out, torch.float32, "dot_scores(out)")
        _need(tuple(out.shape) == (Q, N), f"dot_scores(out): expected shape {(Q, N)}, got {tuple(out.shape)}")
    check(_lib.lib().fz_dot_scores_f32(_ptr(Qn), Qn.stride(0) if Q > 1 else d, _ptr(Dn), Dn.stride(0) if N > 1 else d, Q, N, d,
                                       _ptr(out), _ld(out), _stream(Qn)), "fz_dot_scores_f32")
    return out


def cos_scores(Qe: torch.Tensor, De: torch.Tensor) -> torch.Tensor:
    """util.cos_sim as called at hybrid.py:103."""
    return dot_scores(normalize_rows(Qe), normalize_rows(De))


# ---------------------------------------------------------------------------------------
# K2 MaxSim
# ---------------------------------------------------------------------------------------
def maxsim(Qtok: torch.Tensor, Dtok: torch.Tensor, Doff: torch.Tensor, out: torch.Tensor | None = None,
           max_doc_len: int | None = None) -> torch.Tensor:
    """Exact ColBERT late interaction. Qtok [Q,Lq,128] f16, Dtok [sumL,128] f16 packed, Doff [N+1] int64.
    max_doc_len: upper bound of the document lengths (default: measured from Doff, one small D2H read)."""
    _dev(Qtok, torch.float16, "maxsim(Qtok)")
    _dev(Dtok, torch.float16, "maxsim(Dtok)")
    _dev(Doff, torch.int64, "maxsim(Doff)")
    Qtok, Dtok, Doff = Qtok.contiguous(), Dtok.contiguous(), Doff.contiguous()
    Q, Lq, dim = Qtok.shape
    N = Doff.numel() - 1
    _need(Dtok.dim() == 2 and Dtok.shape[1] == dim, f"maxsim: Dtok must be [sumL, {dim}]")
    _need(N >= 0, "maxsim: Doff must hold N + 1 offsets")
    if out is None:
        out = alloc_plane(Q, N, torch.float32, Qtok.device)
    else:
        _dev(out, torch.float32, "maxsim(out)")
        _need(tuple(out.shape) == (Q, N), f"maxsim(out): expected shape {(Q, N)}, got {tuple(out.shape)}")
    if max_doc_len is None:
        max_doc_len = max(int((Doff[1:] - Doff[:-1]).max().item()), 1) if N > 0 else 1
    check(_lib.lib().fz_maxsim_f16(_ptr(Qtok), _ptr(Dtok), _ptr(Doff), int(Dtok.shape[0]), int(max_doc_len), Q, Lq, N, dim, _ptr(out),
                                   _ld(out), _stream(Qtok)), "fz_maxsim_f16")
    return out


# ---------------------------------------------------------------------------------------
# K5a / K6 sort
# ---------------------------------------------------------------------------------------
def sort_max_n(dtype=torch.float32) -> int:
    return int(_lib.lib().fz_sort_max_n() if dtype == torch.float32 else _lib.lib().fz_sort_max_n_f64())


def sort_rows_desc(keys: torch.Tensor, init_order: torch.Tensor | None = None, row_len: torch.Tensor | None = None,
                   want_order=True, want_keys=True, want_rank=False, init_rank: torch.Tensor | None = None,
                   stats_out: torch.Tensor | None = None, stats_len: torch.Tensor | None = None, covers_all: bool = False,
                   lexical: bool = False):
    """Stable descending row sort (Python sorted(reverse=True): bm25.py:104, hybrid.py:306).
    Incoming sequence: identity (default), `init_order` (column at each sequence position) or `init_rank`
    (sequence position of each column: a rank plane; read coalesced, the fast form).
    Returns (order|None, sorted_keys|None, rank|None); order/sorted_keys entries beyond row_len are -1 / -inf,
    rank entries of elements outside the sequence are -1.
    stats_out: a contiguous fp32 tensor [4, rows] that receives mean | unbiased std | min | max of each list's float32 values as a
    by-product of the sort (identity / init_order sequences, rows that fit one workgroup); stats_len [rows] int32 restricts the
    statistics to the first stats_len[row] entries of the sorted list (a ranking cut to its top-k; fp32 keys only).
    covers_all: the caller vouches that the incoming sequence holds every column of every row (init_rank is a full ranking's rank
    plane, no row_len): the outputs are then written in full and need no -1 / -inf pre-fill (two plane-sized fill launches).
    lexical: the rows are a lexical ranker's float64 scores (BM25 / TF-IDF: mostly exact zeros) in the identity sequence -> the
    zero-compacting instantiation (fz_sort_rows_desc_lexical); same outputs, bit for bit."""
    _dev(keys, None, "sort_rows_desc(keys)")
    if keys.dtype not in (torch.float32, torch.float64):
        raise TypeError("keys must be float32 or float64")
    keys = as_plane(keys)
    rows, n = keys.shape
    ld = _ld(keys)
    dev = keys.device

    if init_order is not None and init_rank is not None:
        raise ValueError("pass init_order or init_rank, not both")
    partial = (row_len is not None or init_order is not None or init_rank is not None) and not covers_all  # some slots may stay unwritten: pre-fill them

    def mk(dtype, fill):
        shape = (max(rows, 1), ld)
        base = torch.full(shape, fill, dtype=dtype, device=dev) if partial else torch.empty(shape, dtype=dtype, device=dev)
        return base[:rows, :n]
    order = mk(torch.int32, -1) if want_order else None
    sk = mk(keys.dtype, float("-inf")) if want_keys else None
    rank = mk(torch.int32, -1) if want_rank else None
    if init_order is not None:
        _dev(init_order, torch.int32, "init_order")
        _need(tuple(init_order.shape) == (rows, n), f"init_order: expected shape {(rows, n)}, got {tuple(init_order.shape)}")
        if _ld(init_order) != ld and rows > 1:
            t = mk(torch.int32, -1)
            t.copy_(init_order)
            init_order = t
    if row_len is not None:
        _dev(row_len, torch.int32, "row_len")
        row_len = row_len.contiguous()
        _need(row_len.numel() == rows, f"row_len: expected {rows} entries, got {row_len.numel()}")
    bits = 32 if keys.dtype == torch.float32 else 64
    lib = _lib.lib()
    wsb = int(lib.fz_sort_workspace_bytes(bits, rows, n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    if init_rank is not None:
        _need(stats_out is None and stats_len is None, "sort_rows_desc: no statistics by-product for a placed sequence")
        _dev(init_rank, torch.int32, "init_rank")
        _need(tuple(init_rank.shape) == (rows, n), f"init_rank: expected shape {(rows, n)}, got {tuple(init_rank.shape)}")
        if _ld(init_rank) != ld and rows > 1:
            t = mk(torch.int32, -1)
            t.copy_(init_rank)
            init_rank = t
        check(lib.fz_sort_rows_desc_placed(_ptr(keys), bits, _ptr(init_rank), _ptr(row_len), rows, n, ld, _ptr(order), _ptr(sk),
                                           _ptr(rank), _ptr(ws), wsb, _stream(keys)), "fz_sort_rows_desc_placed")
    else:
        if stats_out is not None:
            _dev(stats_out, torch.float32, "sort_rows_desc(stats_out)")
            _need(tuple(stats_out.shape) == (4, rows) and stats_out.is_contiguous(), f"sort_rows_desc(stats_out): need a contiguous [4, {rows}] tensor")
        if stats_len is not None:
            _need(stats_out is not None, "sort_rows_desc(stats_len) needs stats_out")
            _dev(stats_len, torch.int32, "sort_rows_desc(stats_len)")
            _need(stats_len.numel() == rows and stats_len.is_contiguous(), f"sort_rows_desc(stats_len): need {rows} contiguous lengths")
        if lexical and bits == 64 and init_order is None and stats_len is None and order is not None:
            check(lib.fz_sort_rows_desc_lexical(_ptr(keys), _ptr(row_len), rows, n, ld, _ptr(order), _ptr(sk), _ptr(rank), _ptr(stats_out),
                                                _ptr(ws), wsb, _stream(keys)), "fz_sort_rows_desc_lexical")
            return order, sk, rank
        check(lib.fz_sort_rows_desc(_ptr(keys), bits, _ptr(init_order), _ptr(row_len), rows, n, ld, _ptr(order), _ptr(sk),
                                    _ptr(rank), _ptr(stats_out), _ptr(stats_len), _ptr(ws), wsb, _stream(keys)), "fz_sort_rows_desc")
    return order, sk, rank


def sort_bucket_rank_rows(reset: bool = False) -> tuple[int, int, int]:
    """(rows the bucket ranking ordered, of those rows with a swapped pair put back, rows it handed to the digit passes) on the current
    device since the last reset (fz_sort_bucket_rank_rows; synchronises the device: tests and tools)."""
    import ctypes
    c = (ctypes.c_uint64 * 3)()
    check(_lib.lib().fz_sort_bucket_rank_rows(ctypes.cast(c, ctypes.c_void_p), 1 if reset else 0), "fz_sort_bucket_rank_rows")
    return int(c[0]), int(c[1]), int(c[2])


def sort_zero_compact_rows(reset: bool = False) -> tuple[int, int]:
    """(float64 rows whose exact zeros were left out of the ordering phases, eligible rows that had too few zeros) on the current device since
    the last reset (fz_sort_zero_compact_rows; synchronises the device: tests and tools)."""
    import ctypes
    c = (ctypes.c_uint64 * 2)()
    check(_lib.lib().fz_sort_zero_compact_rows(ctypes.cast(c, ctypes.c_void_p), 1 if reset else 0), "fz_sort_zero_compact_rows")
    return int(c[0]), int(c[1])


def select_topk(fused: torch.Tensor, pos: torch.Tensor | None, k: int, cap: int | None = None):
    """First k entries of sort_rows_desc(fused, init_rank=pos) without sorting the rows (fz_select_topk_f + two small row sorts): the
    fused lists main() actually reads (predictions(1000), hybrid.py:537).  fused [Q, N] float32 / float64 plane, pos [Q, N] int32 plane =
    first-insertion position of every column (< 0: in no list; None: the column index).  Returns (cols [Q, k] int32 -- -1 past a row's
    length --, scores [Q, k], lens [Q] int32), or None when a row has more than `cap` candidates (a long tie run at the k-th place) or
    the row is longer than one workgroup holds: the caller then sorts in full."""
    _dev(fused, None, "select_topk(fused)")
    if fused.dtype not in (torch.float32, torch.float64):
        raise TypeError("select_topk(fused): float32 or float64 expected")
    fused = as_plane(fused)
    Q, N = fused.shape
    _need(k > 0, "select_topk: k must be positive")
    if N > 28672 or Q == 0:
        return None
    if pos is not None:
        _dev(pos, torch.int32, "select_topk(pos)")
        _same_shape([fused, pos], "select_topk")
        _need(Q <= 1 or _ld(pos) == _ld(fused), "select_topk: fused and pos must share the row stride")
    cap = int(cap) if cap else round_up(k + 1024, 1024)
    dev = fused.device
    cols = alloc_plane(Q, cap, torch.int32, dev)
    vals = alloc_plane(Q, cap, fused.dtype, dev)
    negp = alloc_plane(Q, cap, torch.float32, dev)
    clen = torch.empty(Q, dtype=torch.int32, device=dev)
    over = torch.zeros(1, dtype=torch.int32, device=dev)
    _need(_ld(cols) == _ld(vals) == _ld(negp), "select_topk: candidate planes must share the row stride")
    rc = _lib.lib().fz_select_topk_f(_ptr(fused), 32 if fused.dtype == torch.float32 else 64, _ptr(pos), Q, N, _ld(fused), int(k), _ld(cols), _ptr(cols),
                                     _ptr(vals), _ptr(negp), _ptr(clen), _ptr(over), _stream(fused))
    check(rc, "fz_select_topk_f")
    by_pos, _, _ = sort_rows_desc(negp, row_len=clen, want_keys=False)                 # candidate slots in first-insertion order
    slots, sk, _ = sort_rows_desc(vals, init_order=by_pos, row_len=clen)              # ... then, stably, by fused score
    if int(over.item()) != 0:
        return None
    kk = min(k, cap)
    sel = slots[:, :kk]
    out_cols = torch.where(sel >= 0, torch.gather(cols, 1, sel.clamp(min=0).long()), torch.full_like(sel, -1))
    return out_cols, sk[:, :kk], torch.clamp(clen, max=kk)


# ---------------------------------------------------------------------------------------
# fusion
# ---------------------------------------------------------------------------------------
def fuse_rank(ranks: list[torch.Tensor], lens: torch.Tensor, method: str) -> torch.Tensor:
    """rrf / bcf in float64 (hybrid.py:248-252,301-304). ranks[s] [Q,N] int32 planes, lens [S,Q] int32."""
    for r in ranks:
        _dev(r, torch.int32, "fuse_rank(ranks)")
    _same_shape(ranks, "fuse_rank")
    ranks = harmonise(list(ranks))
    _dev(lens, torch.int32, "fuse_rank(lens)")
    lens = lens.contiguous()
    Q, N = ranks[0].shape
    _need(tuple(lens.shape) == (len(ranks), Q), f"fuse_rank(lens): expected shape {(len(ranks), Q)}, got {tuple(lens.shape)}")
    ld = _same_ld(*ranks)
    fused = torch.empty((max(Q, 1), ld), dtype=torch.float64, device=ranks[0].device)[:Q, :N]
    check(_lib.lib().fz_fuse_rank_f64(_ptr_array(ranks), _ptr(lens), len(ranks), Q, N, ld, RANK_METHODS[method], _ptr(fused),
                                      _stream(ranks[0])), "fz_fuse_rank_f64")
    return fused


def sort_rank_fused(ranks: list[torch.Tensor], lens: torch.Tensor, method: str, init_order: torch.Tensor | None = None,
                    init_rank: torch.Tensor | None = None, row_len: torch.Tensor | None = None, want_rank: bool = False,
                    covers_all: bool = False):
    """rrf / bcf fusion (hybrid.py:248-252,301-304) and the stable descending sort of the fused scores (hybrid.py:306) in ONE kernel:
    sort_rows_desc(fuse_rank(ranks, lens, method), init_order / init_rank, row_len) without the float64 plane in between
    (fz_sort_rank_fused_desc).  Returns (order, fused scores in list order [float64], rank | None) -- bit-identical to the two calls.
    Rows longer than one workgroup holds raise FusionHipError(FZ_ERR_UNSUPPORTED): callers take the two calls there."""
    for r in ranks:
        _dev(r, torch.int32, "sort_rank_fused(ranks)")
    _same_shape(ranks, "sort_rank_fused")
    first_is_pos = init_rank is not None and init_rank is ranks[0]
    ranks = harmonise(list(ranks))
    if first_is_pos:
        init_rank = ranks[0]
    _dev(lens, torch.int32, "sort_rank_fused(lens)")
    lens = lens.contiguous()
    rows, n = ranks[0].shape
    _need(tuple(lens.shape) == (len(ranks), rows), f"sort_rank_fused(lens): expected shape {(len(ranks), rows)}, got {tuple(lens.shape)}")
    _need(init_order is None or init_rank is None, "sort_rank_fused: pass init_order or init_rank, not both")
    ld = _same_ld(*ranks)
    dev = ranks[0].device
    partial = (row_len is not None or init_order is not None or init_rank is not None) and not covers_all

    def mk(dtype, fill):
        shape = (max(rows, 1), ld)
        base = torch.full(shape, fill, dtype=dtype, device=dev) if partial else torch.empty(shape, dtype=dtype, device=dev)
        return base[:rows, :n]
    order, sk = mk(torch.int32, -1), mk(torch.float64, float("-inf"))
    rank = mk(torch.int32, -1) if want_rank else None
    for name, t in (("init_order", init_order), ("init_rank", init_rank)):
        if t is not None:
            _dev(t, torch.int32, f"sort_rank_fused({name})")
            _need(tuple(t.shape) == (rows, n), f"sort_rank_fused({name}): expected shape {(rows, n)}, got {tuple(t.shape)}")
    if init_order is not None and _ld(init_order) != ld and rows > 1:
        t = mk(torch.int32, -1); t.copy_(init_order); init_order = t
    if init_rank is not None and _ld(init_rank) != ld and rows > 1:
        t = mk(torch.int32, -1); t.copy_(init_rank); init_rank = t
    if row_len is not None:
        _dev(row_len, torch.int32, "sort_rank_fused(row_len)")
        row_len = row_len.contiguous()
        _need(row_len.numel() == rows, f"sort_rank_fused(row_len): expected {rows} entries, got {row_len.numel()}")
    lib = _lib.lib()
    wsb = int(lib.fz_sort_rank_fused_workspace_bytes(rows, n, ld))
    ws = _reserved_workspace(dev, wsb)
    check(lib.fz_sort_rank_fused_desc(_ptr_array(ranks), _ptr(lens), len(ranks), RANK_METHODS[method], _ptr(init_order), _ptr(init_rank),
                                      _ptr(row_len), rows, n, ld, _ptr(order), _ptr(sk), _ptr(rank), _ptr(ws), wsb, _stream(ranks[0])),
          "fz_sort_rank_fused_desc")
    return order, sk, rank


def rrf_terms(count: int, fast: bool, device="cuda") -> torch.Tensor:
    """[count] float64: 1 / (60 + r + 1) as the fused sort forms it (fast) or as fz_fuse_rank_f64 does (IEEE division) -- a diagnostic."""
    out = torch.empty(int(count), dtype=torch.float64, device=device)
    check(_lib.lib().fz_rrf_terms_f64(int(count), 1 if fast else 0, _ptr(out), _stream(out)), "fz_rrf_terms_f64")
    return out


def row_stats(scores: torch.Tensor, rank: torch.Tensor | None, norm: str):
    _dev(scores, torch.float32, "row_stats(scores)")
    rows, N = scores.shape
    a = torch.empty(rows, dtype=torch.float32, device=scores.device)
    b = torch.empty(rows, dtype=torch.float32, device=scores.device)
    if rank is not None:
        _dev(rank, torch.int32, "row_stats(rank)")
        _same_shape([scores, rank], "row_stats")
        scores, rank = harmonise([scores, rank])
        _same_ld(scores, rank)
    check(_lib.lib().fz_row_stats_f32(_ptr(scores), _ptr(rank), rows, N, _ld(scores), NORMS[norm], _ptr(a), _ptr(b), _stream(scores)),
          "fz_row_stats_f32")
    return a, b


def minmax_from_order(scores: torch.Tensor, order: torch.Tensor, lens: torch.Tensor | None, out=None):
    """(min, max) of every ranked list read off its two ends (no reduction): scores / order [Q, N] planes, lens [Q] or None.
    out: optional (mn, mx) fp32 tensors of Q entries to write into."""
    _dev(scores, torch.float32, "minmax_from_order(scores)")
    _dev(order, torch.int32, "minmax_from_order(order)")
    _same_shape([scores, order], "minmax_from_order")
    scores, order = harmonise([scores, order])
    rows, N = scores.shape
    if lens is not None:
        _need(_dev(lens, torch.int32, "minmax_from_order(lens)").numel() == rows and lens.is_contiguous(), f"minmax_from_order: lens must hold {rows} lengths")
    if out is not None:
        mn, mx = out
        _need(mn.numel() == rows and mx.numel() == rows and mn.is_contiguous() and mx.is_contiguous(), f"minmax_from_order(out): need two contiguous tensors of {rows}")
        _dev(mn, torch.float32, "minmax_from_order(out)"); _dev(mx, torch.float32, "minmax_from_order(out)")
    else:
        mn = torch.empty(rows, dtype=torch.float32, device=scores.device)
        mx = torch.empty(rows, dtype=torch.float32, device=scores.device)
    check(_lib.lib().fz_minmax_from_order_f32(_ptr(scores), _ptr(order), _ptr(lens), rows, N, _ld(scores), _ptr(mn), _ptr(mx), _stream(scores)),
          "fz_minmax_from_order_f32")
    return mn, mx


TABLES_PATHS = {0: "lds-all", 1: "lds-swap", 2: "row"}


class PreparedTables:
    """Quantile tables of a percentile-rank / NCE fusion made ready for the kernel that keeps one system's table in LDS at a time
    (fz_nsf_tables_prepare): aligned copies, bucket tables and -- NCE -- the value of every table index, in one workspace tensor.
    Valid for any number of fuse_nsf calls with these tables and this norm (hybrid.py:412,451 read the tables once per process)."""

    def __init__(self, distr: list[torch.Tensor], norm: str, workspace: torch.Tensor):
        self.distr, self.norm, self.workspace = distr, norm, workspace

    def search_info(self) -> list[dict]:
        """Per system: probes per search and entries in the fullest bucket (diagnostics; synchronises)."""
        S = len(self.distr)
        P = (C.c_int32 * S)(*[int(d.numel()) for d in self.distr])
        out = []
        for s in range(S):
            off = int(_lib.lib().fz_nsf_tables_header_offset(S, P, NORMS[self.norm], s))
            h = self.workspace[off:off + 16].cpu().numpy()
            out.append(dict(probes=int(h[8:12].view(np.int32)[0]), fullest_bucket=int(h[12:16].view(np.int32)[0])))
        return out

    def matches(self, distr, norm) -> bool:
        return norm == self.norm and len(distr) == len(self.distr) and all(
            a.data_ptr() == b.data_ptr() and a.numel() == b.numel() for a, b in zip(distr, self.distr))


def nsf_tables_prepare(distr: list[torch.Tensor], norm: str) -> PreparedTables | None:
    """None when a table is too long for LDS (fuse_nsf then searches it in global memory)."""
    _need(norm in ("percentile-rank", "normal-curve-equivalent"), f"nsf_tables_prepare: {norm!r} has no tables")
    distr = [_dev(d, torch.float32, "distr").contiguous() for d in distr]
    S = len(distr)
    _need(all(d.dim() == 1 and d.numel() > 0 for d in distr), "nsf_tables_prepare: tables must be non-empty 1-d tensors")
    P = (C.c_int32 * S)(*[int(d.numel()) for d in distr])
    lib = _lib.lib()
    nbytes = int(lib.fz_nsf_tables_workspace_bytes(S, P, NORMS[norm]))
    if nbytes == 0:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=distr[0].device)
    check(lib.fz_nsf_tables_prepare(_ptr_array(distr), P, S, NORMS[norm], _ptr(ws), nbytes, _stream(distr[0])), "fz_nsf_tables_prepare")
    return PreparedTables(distr, norm, ws)


last_tables_path: str | None = None   # which kernel the last percentile-rank / NCE fuse_nsf call ran (tests pin it)


def fuse_nsf(planes: list[torch.Tensor], ranks: list[torch.Tensor | None] | None, weights, norm: str,
             distr: list[torch.Tensor] | None = None, out: torch.Tensor | None = None,
             orders: list[torch.Tensor] | None = None, lens: torch.Tensor | None = None,
             stats=None,
             valid_bits: list[torch.Tensor | None] | None = None,
             tables: PreparedTables | None | bool = None) -> torch.Tensor:
    """normalise -> weight -> sum in one HBM pass (hybrid.py:212-214,254-280,291,301-304).
    orders (+ lens [S, Q]): the systems' order planes, when they are ranked -- min-max then takes every list's minimum and
    maximum from its two ends and the fusion is one flat streaming pass (same bits as the reducing kernel).
    stats: the row statistics (min / max, or mean / unbiased std), when the caller has them -- the fusion is then one flat streaming
    pass.  Either (a, b) = two contiguous [S*Q] fp32 tensors, or a list of S pairs (a_s, b_s) of [Q] fp32 tensors, one per system:
    each ranked system keeps the statistics its ranking sort produced (sort_rows_desc(stats_out=...)), nothing is concatenated.
    valid_bits[s] (optional): the validity of system s as a bitmap (rank_to_bitmap), read instead of its rank plane.
    tables (optional): nsf_tables_prepare(distr, norm) of these very tables, to prepare them once for many calls; False: stay on
    fz_fuse_nsf_f32 whatever the table size (its all-tables-in-LDS kernel or its global-memory search: what tests compare against)."""
    for p in planes:
        _dev(p, torch.float32, "fuse_nsf(planes)")
    S = len(planes)
    if ranks:
        _need(len(ranks) == S, f"fuse_nsf: {S} planes but {len(ranks)} rank planes")
        for r in ranks:
            if r is not None:
                _dev(r, torch.int32, "fuse_nsf(ranks)")
    _same_shape(list(planes) + (list(ranks) if ranks else []), "fuse_nsf")
    _need(len(weights) == S, f"fuse_nsf: {S} planes but {len(weights)} weights")
    both = harmonise(list(planes) + (list(ranks) if ranks else []))
    planes, ranks = both[:S], (both[S:] if ranks else None)
    Q, N = planes[0].shape
    ld = _same_ld(*planes, *([r for r in ranks if r is not None] if ranks else []))
    dev = planes[0].device
    if out is not None:
        _dev(out, torch.float32, "fuse_nsf(out)")
        _need(tuple(out.shape) == (Q, N) and (Q <= 1 or _ld(out) == ld), f"fuse_nsf(out): expected a [{Q}, {N}] plane with row stride {ld}")
    fused = out if out is not None else torch.empty((max(Q, 1), ld), dtype=torch.float32, device=dev)[:Q, :N]
    w = (C.c_double * S)(*[float(x) for x in weights])
    vb, ldb = None, 0
    if valid_bits is not None and any(b is not None for b in valid_bits):
        _need(len(valid_bits) == S, f"fuse_nsf: {S} planes but {len(valid_bits)} validity bitmaps")
        for b in valid_bits:
            if b is not None:
                _dev(b, torch.int32, "fuse_nsf(valid_bits)")
                _need(b.dim() == 2 and b.shape[0] == Q and b.is_contiguous() and b.shape[1] * 32 >= N, "fuse_nsf(valid_bits): expected contiguous [Q, >= N/32] int32 bitmaps")
        ldbs = {int(b.shape[1]) for b in valid_bits if b is not None}
        _need(len(ldbs) == 1, "fuse_nsf(valid_bits): bitmaps must share their row stride")
        vb, ldb = _ptr_array(valid_bits), ldbs.pop()
    dptr, P = None, None
    if norm in ("percentile-rank", "normal-curve-equivalent"):
        if distr is None or any(d is None for d in distr):
            raise AttributeError("percentile distributions are required for percentile-rank / normal-curve-equivalent")
        _need(len(distr) == S, f"fuse_nsf: {S} planes but {len(distr)} percentile tables")
        distr = [_dev(d, torch.float32, "distr").contiguous() for d in distr]
        dptr = _ptr_array(distr)
        P = (C.c_int32 * S)(*[int(d.numel()) for d in distr])
    lib = _lib.lib()
    if stats is not None and norm in ("min-max", "z-score") and Q > 0 and N > 0 and isinstance(stats, list):
        _need(len(stats) == S, f"fuse_nsf(stats): {S} planes but {len(stats)} statistics pairs")
        for pr in stats:
            for x in pr:
                _dev(x, torch.float32, "fuse_nsf(stats)")
                _need(x.numel() == Q and x.is_contiguous(), f"fuse_nsf(stats): per-system statistics must be contiguous tensors of {Q} entries")
        check(lib.fz_fuse_nsf_pstats_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, NORMS[norm],
                                         dptr, P, _ptr_array([a_ for a_, _ in stats]), _ptr_array([b_ for _, b_ in stats]), vb, ldb, _ptr(fused),
                                         _stream(planes[0])), "fz_fuse_nsf_pstats_f32")
        return fused
    if stats is not None and norm in ("min-max", "z-score") and Q > 0 and N > 0:
        sa, sb = stats
        for x in (sa, sb):
            _dev(x, torch.float32, "fuse_nsf(stats)")
            _need(x.numel() == S * Q and x.is_contiguous(), f"fuse_nsf(stats): need contiguous tensors of {S * Q} entries")
        check(lib.fz_fuse_nsf_stats_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, NORMS[norm],
                                        dptr, P, _ptr(sa), _ptr(sb), vb, ldb, _ptr(fused), _stream(planes[0])), "fz_fuse_nsf_stats_f32")
        return fused
    if norm == "min-max" and orders is not None and Q > 0 and N > 0:
        _need(len(orders) == S, f"fuse_nsf: {S} planes but {len(orders)} order planes")
        if lens is not None:
            _need(tuple(lens.shape) == (S, Q), f"fuse_nsf(lens): expected shape {(S, Q)}, got {tuple(lens.shape)}")
        for o_ in orders:
            _dev(o_, torch.int32, "fuse_nsf(orders)")
        _same_shape(list(planes) + list(orders), "fuse_nsf(orders)")
        allp = harmonise(list(planes) + (list(ranks) if ranks else []) + list(orders))   # one row stride for every plane of the call
        planes, orders = allp[:S], allp[len(allp) - S:]
        ranks = allp[S:2 * S] if ranks else None
        ld = _same_ld(*planes, *orders, *([r for r in ranks if r is not None] if ranks else []))
        if out is None:
            fused = torch.empty((max(Q, 1), ld), dtype=torch.float32, device=dev)[:Q, :N]
        else:
            _need(Q <= 1 or _ld(out) == ld, f"fuse_nsf(out): expected a [{Q}, {N}] plane with row stride {ld}")
        if lens is not None:
            lens = _dev(lens, torch.int32, "fuse_nsf(lens)").contiguous()
        sa = torch.empty(S * Q, dtype=torch.float32, device=dev)
        sb = torch.empty(S * Q, dtype=torch.float32, device=dev)
        check(lib.fz_minmax_from_orders_f32(_ptr_array(planes), _ptr_array(orders), _ptr(lens), S, Q, N, ld, _ptr(sa), _ptr(sb), _stream(planes[0])),
              "fz_minmax_from_orders_f32")
        check(lib.fz_fuse_nsf_stats_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, NORMS[norm],
                                        dptr, P, _ptr(sa), _ptr(sb), vb, ldb, _ptr(fused), _stream(planes[0])), "fz_fuse_nsf_stats_f32")
        return fused
    if dptr is not None and Q > 0 and N > 0:
        # quantile tables: all of them LDS-resident (small tables), one system's at a time (the sizes the reference reads:
        # hybrid.py:412,451 -- 27,943 entries; :374 -- 10,001), or searched in global memory (longer still)
        global last_tables_path
        rp = None if ranks is None else _ptr_array(ranks)
        path = lib.fz_nsf_tables_path(_ptr_array(planes), rp, S, Q, N, ld, NORMS[norm], P, _ptr(fused))
        _need(path in TABLES_PATHS, f"fz_nsf_tables_path: status {path}")
        last_tables_path = TABLES_PATHS[path]
        if tables is False:
            last_tables_path = TABLES_PATHS[0 if path == 0 else 2]
        elif path == 1:
            if tables is None or not tables.matches(distr, norm):
                tables = nsf_tables_prepare(distr, norm)
            check(lib.fz_fuse_nsf_tables_f32(_ptr_array(planes), rp, w, S, Q, N, ld, NORMS[norm], dptr, P, vb, ldb, _ptr(fused),
                                             _ptr(tables.workspace), tables.workspace.numel(), _stream(planes[0])), "fz_fuse_nsf_tables_f32")
            return fused
    rc = lib.fz_fuse_nsf_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, NORMS[norm], dptr, P,
                             vb, ldb, _ptr(fused), _stream(planes[0]))
    if rc == _lib.FZ_ERR_UNSUPPORTED:
        # rows longer than the register-resident kernel holds (N > 32768): statistics pass + elementwise pass
        sa = torch.zeros(S * Q, dtype=torch.float32, device=dev)
        sb = torch.zeros(S * Q, dtype=torch.float32, device=dev)
        if norm in ("min-max", "z-score"):
            for s in range(S):
                _need(vb is None or valid_bits[s] is None or (ranks is not None and ranks[s] is not None),
                      "fuse_nsf: rows longer than 32768 take their statistics over the rank planes: pass ranks next to valid_bits")
                a, b = row_stats(planes[s], None if ranks is None else ranks[s], norm)
                sa[s * Q:(s + 1) * Q] = a
                sb[s * Q:(s + 1) * Q] = b
        rc = lib.fz_fuse_nsf_stats_f32(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, NORMS[norm],
                                       dptr, P, _ptr(sa), _ptr(sb), vb, ldb, _ptr(fused), _stream(planes[0]))
    check(rc, "fz_fuse_nsf_f32")
    return fused


def rank_to_bitmap(rank: torch.Tensor) -> torch.Tensor:
    """Validity bitmap of a rank plane: [Q, ceil(N / 64) * 2] int32, bit (j & 31) of word j >> 5 = (rank[q, j] >= 0).  Built once per
    partial system; the nsf fusion passes then read 1 bit per document instead of the 4-byte rank."""
    _dev(rank, torch.int32, "rank_to_bitmap(rank)")
    rank = as_plane(rank)
    Q, N = rank.shape
    ldb = max(2, (N + 63) // 64 * 2)
    bits = torch.zeros((max(Q, 1), ldb), dtype=torch.int32, device=rank.device)[:Q]
    check(_lib.lib().fz_rank_to_bitmap(_ptr(rank), Q, N, _ld(rank), _ptr(bits), ldb, _stream(rank)), "fz_rank_to_bitmap")
    return bits


def zero_unlisted_(plane: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
    """plane[q, j] = 0 where rank[q, j] < 0, in place: a system adds nothing for a document it does not list (hybrid.py:301-304)."""
    _dev(plane, torch.float32, "zero_unlisted_(plane)"); _dev(rank, torch.int32, "zero_unlisted_(rank)")
    _same_shape([plane, rank], "zero_unlisted_")
    Q, N = plane.shape
    _need(Q <= 1 or _ld(plane) == _ld(rank), "zero_unlisted_: plane and rank must share the row stride")
    check(_lib.lib().fz_zero_unlisted_f32(_ptr(plane), _ptr(rank), Q, N, _ld(plane), _stream(plane)), "fz_zero_unlisted_f32")
    return plane


def fuse_none(planes: list[torch.Tensor], ranks: list[torch.Tensor | None] | None, weights) -> torch.Tensor:
    """'none' / unknown normalisation: float64 passthrough (hybrid.py:280,291,304)."""
    for p in planes:
        _dev(p, torch.float32, "fuse_none(planes)")
    S = len(planes)
    if ranks:
        _need(len(ranks) == S, f"fuse_none: {S} planes but {len(ranks)} rank planes")
        for r in ranks:
            if r is not None:
                _dev(r, torch.int32, "fuse_none(ranks)")
    _same_shape(list(planes) + (list(ranks) if ranks else []), "fuse_none")
    _need(len(weights) == S, f"fuse_none: {S} planes but {len(weights)} weights")
    both = harmonise(list(planes) + (list(ranks) if ranks else []))
    planes, ranks = both[:S], (both[S:] if ranks else None)
    Q, N = planes[0].shape
    ld = _same_ld(*planes, *([r for r in ranks if r is not None] if ranks else []))
    fused = torch.empty((max(Q, 1), ld), dtype=torch.float64, device=planes[0].device)[:Q, :N]
    w = (C.c_double * S)(*[float(x) for x in weights])
    check(_lib.lib().fz_fuse_none_f64(_ptr_array(planes), None if ranks is None else _ptr_array(ranks), w, S, Q, N, ld, _ptr(fused),
                                      _stream(planes[0])), "fz_fuse_none_f64")
    return fused


def is_wide_weight(w) -> bool:
    """NumPy-2 promotion of `np.float32 score * w` (hybrid.py:291): the product is float64 only for an np.float64 weight
    -- which is what the tuning grid holds (np.arange, hybrid.py:405-409); a Python float / int is a weak scalar and
    np.float32 / np.float16 keep float32.  (np.float64 subclasses float, so it is tested by its own type.)"""
    return isinstance(w, (np.float64, np.longdouble))


def fuse_wsum(planes: list[torch.Tensor], ranks: list[torch.Tensor | None] | None, weights, narrow=None) -> torch.Tensor:
    """Weight-and-sum -> float64 plane, with NumPy's scalar promotion (fz_fuse_wsum_f64): planes may be float32 or float64
    (per system); narrow[s] = the weight is weak / float32 (fl32 product; the document's sum stays fl32 until its first
    float64 product).  narrow=None: everything float64 (the 'none' passthrough, hybrid.py:280,291,304)."""
    S = len(planes)
    for p in planes:
        _dev(p, None, "fuse_wsum(planes)")
        if p.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"fuse_wsum(planes): expected float32 or float64, got {p.dtype}")
    if ranks:
        _need(len(ranks) == S, f"fuse_wsum: {S} planes but {len(ranks)} rank planes")
        for r in ranks:
            if r is not None:
                _dev(r, torch.int32, "fuse_wsum(ranks)")
    _same_shape(list(planes) + (list(ranks) if ranks else []), "fuse_wsum")
    _need(len(weights) == S, f"fuse_wsum: {S} planes but {len(weights)} weights")
    both = harmonise(list(planes) + (list(ranks) if ranks else []))
    planes, ranks = both[:S], (both[S:] if ranks else None)
    Q, N = planes[0].shape
    ld = _same_ld(*planes, *([r for r in ranks if r is not None] if ranks else []))
    fused = torch.empty((max(Q, 1), ld), dtype=torch.float64, device=planes[0].device)[:Q, :N]
    w = (C.c_double * S)(*[float(x) for x in weights])
    p64 = (C.c_int32 * S)(*[int(p.dtype == torch.float64) for p in planes])
    nr = (C.c_int32 * S)(*[int(bool(n)) for n in (narrow if narrow is not None else [False] * S)])
    check(_lib.lib().fz_fuse_wsum_f64(_ptr_array(planes), p64, None if ranks is None else _ptr_array(ranks), w, nr, S, Q, N, ld,
                                      _ptr(fused), _stream(planes[0])), "fz_fuse_wsum_f64")
    return fused


def insertion_order(orders: list[torch.Tensor], lens: torch.Tensor, N: int, want_pos: bool = False):
    """First-insertion order of the fused dict (hybrid.py:301-304). Returns (ins_order [Q,N] int32, U [Q] int32), with want_pos also the
    inverse plane pos [Q,N] int32 (first-insertion position of every document, -1 = in no list)."""
    for o in orders:
        _dev(o, torch.int32, "insertion_order(orders)")
    _same_shape(orders, "insertion_order")
    orders = harmonise(list(orders))
    _dev(lens, torch.int32, "insertion_order(lens)")
    lens = lens.contiguous()
    Q = orders[0].shape[0]
    _need(orders[0].shape[1] == N, f"insertion_order: order planes are {orders[0].shape[1]} wide, N = {N}")
    _need(tuple(lens.shape) == (len(orders), Q), f"insertion_order(lens): expected shape {(len(orders), Q)}, got {tuple(lens.shape)}")
    ld = _same_ld(*orders)
    dev = orders[0].device
    ins = torch.full((max(Q, 1), ld), -1, dtype=torch.int32, device=dev)[:Q, :N]
    U = torch.zeros(Q, dtype=torch.int32, device=dev)
    pos = torch.full((max(Q, 1), ld), -1, dtype=torch.int32, device=dev)[:Q, :N] if want_pos else None
    check(_lib.lib().fz_insertion_order(_ptr_array(orders), _ptr(lens), len(orders), Q, N, ld, _ptr(ins), _ptr(U), _ptr(pos), None, 0,
                                        _stream(orders[0])), "fz_insertion_order")
    return (ins, U, pos) if want_pos else (ins, U)


def gold_ranks(T: list[torch.Tensor], pos: torch.Tensor, weights: torch.Tensor, gold: torch.Tensor) -> torch.Tensor:
    """Fused ranks of the gold documents for every weight vector (N1, hybrid.py:404-426).
    T[s] [Q,N] normalised planes, pos [Q,N] int32 insertion positions (-1 absent), weights [W,S] fp32,
    gold [Q,G] int32 corpus positions (-1 pad) -> ranks [W,Q,G] int32 (0 where gold is padding / unlisted: check pos).
    float64 weights select the float64 sweep (np.float64 grid weights: NumPy promotes the products and sums)."""
    for t in T:
        _dev(t, torch.float32, "gold_ranks(T)")
    _dev(pos, torch.int32, "gold_ranks(pos)")
    _dev(weights, None, "gold_ranks(weights)")
    if weights.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"gold_ranks(weights): expected float32 or float64, got {weights.dtype}")
    _dev(gold, torch.int32, "gold_ranks(gold)")
    lib = _lib.lib()
    G = int(lib.fz_tune_max_gold())
    _same_shape(list(T) + [pos], "gold_ranks")
    both = harmonise(list(T) + [pos])
    T, pos = both[:-1], both[-1]
    Q, N = T[0].shape
    W, S = weights.shape
    if S != len(T) or gold.shape != (Q, G):
        raise ValueError(f"weights must be [W,{len(T)}] and gold [Q,{G}]")
    ld = _same_ld(*T, pos)
    out = torch.zeros((W, Q, G), dtype=torch.int32, device=T[0].device)
    fn = lib.fz_gold_ranks_f64w if weights.dtype == torch.float64 else lib.fz_gold_ranks_f32
    check(fn(_ptr_array(T), _ptr(pos), _ptr(weights.contiguous()), _ptr(gold.contiguous()), S, W, Q, N, ld, _ptr(out), _stream(T[0])),
          "fz_gold_ranks")
    return out


def tune_metrics(ranks: torch.Tensor, gold: torch.Tensor, pos: torch.Tensor, n_gold: torch.Tensor, idcg: torch.Tensor,
                 disc: torch.Tensor, cuts: dict[str, list[int]]) -> torch.Tensor:
    """run_evaluation's metrics for every weight vector from the gold ranks of gold_ranks(), on the device (N1).
    ranks [W,Q,G] int32, gold [Q,G] int32, pos [Q,N] int32 plane, n_gold [Q] int32, idcg [Q] float64, disc [top+1] float64,
    cuts = {"recall": [...], "map": [...], "mrr": [...], "ndcg": [...]} -> [W, M] float64, columns in that order + R-precision."""
    _dev(ranks, torch.int32, "tune_metrics(ranks)"); _dev(gold, torch.int32, "tune_metrics(gold)")
    _dev(pos, torch.int32, "tune_metrics(pos)"); _dev(n_gold, torch.int32, "tune_metrics(n_gold)")
    _dev(idcg, torch.float64, "tune_metrics(idcg)"); _dev(disc, torch.float64, "tune_metrics(disc)")
    lib = _lib.lib()
    G = int(lib.fz_tune_max_gold())
    W, Q = ranks.shape[0], ranks.shape[1]
    if ranks.shape != (W, Q, G) or gold.shape != (Q, G) or pos.shape[0] != Q or n_gold.shape != (Q,) or idcg.shape != (Q,):
        raise ValueError("tune_metrics: ranks [W,Q,G], gold [Q,G], pos [Q,N], n_gold [Q], idcg [Q] expected")
    fam = [list(cuts.get(k, [])) for k in ("recall", "map", "mrr", "ndcg")]
    flat = [int(k) for f in fam for k in f]
    if disc.numel() < 1 or any(k < 0 for k in flat):
        raise ValueError("tune_metrics: a discount table and non-negative cut-offs expected")
    pos = as_plane(pos)
    cuts_dev = torch.tensor(flat if flat else [0], dtype=torch.int32, device=ranks.device)
    out = torch.empty((W, len(flat) + 1), dtype=torch.float64, device=ranks.device)
    check(lib.fz_tune_metrics_f64(_ptr(ranks.contiguous()), _ptr(gold.contiguous()), _ptr(pos), pos.stride(0), _ptr(n_gold.contiguous()),
                                  _ptr(idcg.contiguous()), _ptr(disc.contiguous()), int(disc.numel()) - 1, _ptr(cuts_dev), len(fam[0]),
                                  len(fam[1]), len(fam[2]), len(fam[3]), W, Q, _ptr(out), _stream(ranks)), "fz_tune_metrics_f64")
    return out


# ---------------------------------------------------------------------------------------
# top-k
# ---------------------------------------------------------------------------------------
def topk_rows(scores: torch.Tensor, k: int, id_base: int = 0):
    """k best per row by (score desc, id asc) -> (scores [rows,k] f32, ids [rows,k] int64)."""
    _dev(scores, torch.float32, "topk_rows(scores)")
    rows, n = scores.shape
    dev = scores.device
    os_ = torch.empty((rows, k), dtype=torch.float32, device=dev)
    oi = torch.empty((rows, k), dtype=torch.int64, device=dev)
    lib = _lib.lib()
    wsb = int(lib.fz_topk_workspace_bytes(rows, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(lib.fz_topk_rows_f32(_ptr(scores), rows, n, _ld(scores), k, int(id_base), _ptr(os_), _ptr(oi), _ptr(ws), wsb, _stream(scores)),
          "fz_topk_rows_f32")
    return os_, oi


def topk_update(scores: torch.Tensor, id_base: int, run_scores: torch.Tensor, run_ids: torch.Tensor, cap: int = 7168,
                overflow: torch.Tensor | None = None):
    """Merge one more chunk of scores into the running top-k (streaming threshold filter + small sort).
    Returns (new_scores, new_ids, overflow_flag_tensor); the caller checks the flag once at the end of the search."""
    _dev(scores, torch.float32, "topk_update(scores)")
    _dev(run_scores, torch.float32, "topk_update(run_scores)")
    _dev(run_ids, torch.int64, "topk_update(run_ids)")
    rows, n = scores.shape
    k = run_scores.shape[1]
    _need(tuple(run_scores.shape) == (rows, k) and tuple(run_ids.shape) == (rows, k), f"topk_update: running lists must be [{rows}, k]")
    dev = scores.device
    lib = _lib.lib()
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    else:
        _dev(overflow, torch.int32, "topk_update(overflow)")
    ns = torch.empty((rows, k), dtype=torch.float32, device=dev)
    ni = torch.empty((rows, k), dtype=torch.int64, device=dev)
    wsb = int(lib.fz_topk_update_workspace_bytes(rows, k, cap))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(lib.fz_topk_update_f32(_ptr(scores), rows, n, _ld(scores), int(id_base), _ptr(run_scores.contiguous()), _ptr(run_ids.contiguous()), k, cap,
                                 _ptr(ns), _ptr(ni), _ptr(overflow), _ptr(ws), wsb, _stream(scores)), "fz_topk_update_f32")
    return ns, ni, overflow


class TopkStream:
    """Running per-row top-k over a stream of score chunks (documents in ascending id order): only scores above a row's
    threshold (its k-th best when the list was last folded) are kept as candidates (fz_topk_filter_append_f32), and candidates
    are folded into the list (fz_topk_fold_f32: one row sort) only when their EXPECTED number -- k * (documents since the fold) /
    (documents before it), for scores in no particular order -- reaches half the candidate capacity.  The windows between folds
    therefore grow geometrically: 4 folds for a 1.1 M-document shard at k = 1000, cap = 7168 after an 8192-document head, however the scoring is chunked.
    `overflow` (device int32) becomes 1 if a row had more than `cap` candidates in a window.  With `exact_on_overflow` (default) the
    flag is read at every fold (one small device -> host read per window: 4 per 1.1 M-document shard) and an overflowed WINDOW is redone
    exactly -- per-piece top-k (fz_topk_rows_f32) merged into the list as it stood before the window -- so a corpus ordered by relevance
    (every window overflows) costs the exact search once, not a wasted streaming pass plus the exact search of the whole shard;
    `windows_redone` counts them.  That needs what the window was fed with: feed_gemm's pieces are views of Qn / Dn (kept: no copies);
    feed()'s score buffers are kept only on request (hold=True) -- otherwise, and without exact_on_overflow, the flag stays set and
    the caller redoes the search."""

    def __init__(self, run_scores: torch.Tensor, run_ids: torch.Tensor, seen: int, cap: int = 7168, exact_on_overflow: bool = True):
        _dev(run_scores, torch.float32, "TopkStream(run_scores)"); _dev(run_ids, torch.int64, "TopkStream(run_ids)")
        rows, k = run_scores.shape
        _need(tuple(run_ids.shape) == (rows, k) and seen > 0 and cap > 0, "TopkStream: lists [rows, k], seen > 0, cap > 0 expected")
        dev = run_scores.device
        self.rows, self.k, self.cap = rows, k, int(cap)
        self.best_s, self.best_i = run_scores.contiguous(), run_ids.contiguous()
        # thresholds, padded to whole 128-query GEMM blocks with +inf (the fused GEMM reads its tile's 128 rows unconditionally)
        self._tau_pad = torch.full((round_up(max(rows, 1), 128),), float("inf"), dtype=torch.float32, device=dev)
        self.tau = self._tau_pad[:rows]
        self.tau.copy_(self.best_s[:, k - 1])
        self.unordered = False           # candidates appended by the fused GEMM arrive in no particular order
        self.cand_s = torch.empty((rows, cap), dtype=torch.float32, device=dev)
        self.cand_i = torch.empty((rows, cap), dtype=torch.int64, device=dev)
        self.cand_len = torch.zeros(rows, dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seen = int(seen)            # documents folded into (best_s, tau)
        self.pending = 0                 # documents filtered against tau since
        self.exact_on_overflow = bool(exact_on_overflow)
        self.windows_redone = 0
        self._pieces = []                # what the current window was fed with: ("scores", piece, id_base) | ("gemm", Qn, Dpiece, id_base)
        self._unheld = False             # ... and whether some of it was fed without being held (feed(hold=False))
        self.unrepairable = False        # host latch: a window fed WITHOUT hold overflowed -- nothing the stream holds can repair it, the device
                                         #   flag stays set and the caller redoes the search; later windows are neither read nor redone
        wsb = int(_lib.lib().fz_topk_fold_workspace_bytes(rows, k, cap))
        self._ws, self._wsb = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev), wsb

    def _window(self) -> int:
        """documents one threshold may serve: expected candidates k * window / seen <= cap / 2"""
        return max(64, (self.cap // 2) * self.seen // self.k // 64 * 64)

    def feed(self, scores: torch.Tensor, id_base: int, hold: bool = False):
        """scores [rows, n] of documents id_base .. id_base + n - 1.
        hold=False (default): nothing of `scores` is kept -- the caller may overwrite or free the buffer as soon as the call returns; a
        window fed this way cannot be redone, so if it overflows the `overflow` flag STAYS SET and the caller redoes the search
        (ShardedDenseIndex.local_topk does).  hold=True: the stream keeps a view of the piece until the window is folded and redoes an
        overflowed window exactly from it -- only for callers that leave the buffer untouched (and can afford it alive) until then."""
        _dev(scores, torch.float32, "TopkStream.feed(scores)")
        _need(scores.shape[0] == self.rows, "TopkStream.feed: one row per running list")
        lib = _lib.lib()
        n, lo = scores.shape[1], 0
        while lo < n:
            hi = min(n, lo + self._window() - self.pending)
            piece = scores[:, lo:hi]
            check(lib.fz_topk_filter_append_f32(_ptr(piece), self.rows, hi - lo, _ld(scores), int(id_base) + lo, _ptr(self.tau), _ptr(self.cand_s),
                                                _ptr(self.cand_i), _ptr(self.cand_len), self.cap, _ptr(self.overflow), _stream(scores)),
                  "fz_topk_filter_append_f32")
            if hold:
                self._pieces.append(("scores", piece, int(id_base) + lo))
            else:
                self._unheld = True          # this window saw scores the stream does not hold: no exact redo for it
            self.pending += hi - lo
            lo = hi
            if self.pending >= self._window():
                self.fold()

    def feed_gemm(self, Qn: torch.Tensor, Dn: torch.Tensor, id_base: int, mark=None):
        """Score documents id_base .. id_base + len(Dn) - 1 against the queries and keep what beats the thresholds, in ONE kernel:
        the GEMM's epilogue is the filter, the score plane is never written (fz_dot_scores_filter_f32).  Qn [rows, d], Dn [n, d]
        L2-normalised float32, d % 4 == 0.  The documents are cut at the fold windows, one GEMM launch per piece."""
        _dev(Qn, torch.float32, "TopkStream.feed_gemm(Qn)"); _dev(Dn, torch.float32, "TopkStream.feed_gemm(Dn)")
        _need(Qn.shape[0] == self.rows and Qn.shape[1] == Dn.shape[1] and Qn.shape[1] % 4 == 0, "TopkStream.feed_gemm: [rows, d] x [n, d], d % 4 == 0")
        Qn, Dn = Qn.contiguous(), Dn.contiguous()
        lib = _lib.lib()
        n, lo, d = Dn.shape[0], 0, Qn.shape[1]
        self.unordered = True
        while lo < n:
            hi = min(n, lo + self._window() - self.pending)
            piece = Dn[lo:hi]
            check(lib.fz_dot_scores_filter_f32(_ptr(Qn), Qn.stride(0), _ptr(piece), Dn.stride(0), self.rows, hi - lo, d, int(id_base) + lo,
                                               _ptr(self._tau_pad), _ptr(self.cand_s), _ptr(self.cand_i), _ptr(self.cand_len), self.cap,
                                               _ptr(self.overflow), _stream(Qn)), "fz_dot_scores_filter_f32")
            if mark: mark("shard_gemm_filter")
            self._pieces.append(("gemm", Qn, piece, int(id_base) + lo))
            self.pending += hi - lo
            lo = hi
            if self.pending >= self._window():
                self.fold()
                if mark: mark("shard_topk_stream")

    def fold(self):
        if self.pending == 0:
            return
        ns, ni = torch.empty_like(self.best_s), torch.empty_like(self.best_i)
        check(_lib.lib().fz_topk_fold_f32(_ptr(self.best_s), _ptr(self.best_i), self.rows, self.k, _ptr(self.cand_s), _ptr(self.cand_i),
                                          _ptr(self.cand_len), self.cap, 1 if self.unordered else 0, _ptr(ns), _ptr(ni), _ptr(self.tau),
                                          _ptr(self.overflow), _ptr(self._ws), self._wsb, _stream(self.best_s)), "fz_topk_fold_f32")
        # The device flag is read at the fold (one small device -> host read per window) for as long as the search can still come out
        # exact: once an UNHELD window has overflowed the caller redoes the whole search anyway, so later windows are neither read nor
        # redone (with one sticky device flag every later held window ran the full exact redo for nothing).  Every earlier
        # overflow was either repaired (flag cleared below) or latched, so a non-zero flag here is THIS window's.
        if self.exact_on_overflow and not self.unrepairable and int(self.overflow.item()) != 0:
            if self._unheld:
                self.unrepairable = True     # fed without hold: no exact redo for this window; the flag stays set
            else:
                # a candidate list was cut short (or a tie run was too long to order): this window again, exactly, on top of the list as
                # it stood before it -- per piece: scores -> fz_topk_rows_f32 -> merge (ties by ascending id, as everywhere)
                ns, ni = self.best_s, self.best_i
                for kind, *args in self._pieces:
                    if kind == "gemm":
                        Qn_, piece, base = args
                        sc = dot_scores(Qn_, piece)
                    else:
                        sc, base = args
                    if sc.shape[1] == 0:
                        continue
                    ps, pi = topk_rows(sc, self.k, id_base=base)
                    ns, ni = topk_merge(torch.stack([ns, ps]), torch.stack([ni, pi]))
                self.tau.copy_(ns[:, self.k - 1])
                self.cand_len.zero_()
                self.overflow.zero_()
                self.windows_redone += 1
        self._pieces.clear()
        self._unheld = False
        self.best_s, self.best_i = ns, ni
        self.seen += self.pending
        self.pending = 0

    def result(self):
        self.fold()
        return self.best_s, self.best_i, self.overflow


def topk_merge(in_scores: torch.Tensor, in_ids: torch.Tensor):
    """[G,rows,k] per-shard lists -> global top-k [rows,k] (after the RCCL all-gather)."""
    _dev(in_scores, torch.float32, "topk_merge(in_scores)")
    _dev(in_ids, torch.int64, "topk_merge(in_ids)")
    _need(in_scores.dim() == 3 and in_scores.shape == in_ids.shape, "topk_merge: scores and ids must both be [G, rows, k]")
    in_scores, in_ids = in_scores.contiguous(), in_ids.contiguous()
    G, rows, k = in_scores.shape
    os_ = torch.empty((rows, k), dtype=torch.float32, device=in_scores.device)
    oi = torch.empty((rows, k), dtype=torch.int64, device=in_scores.device)
    check(_lib.lib().fz_topk_merge(_ptr(in_scores), _ptr(in_ids), G, rows, k, _ptr(os_), _ptr(oi), _stream(in_scores)), "fz_topk_merge")
    return os_, oi


# ---------------------------------------------------------------------------------------
# BM25
# ---------------------------------------------------------------------------------------
def bm25_doc_norms(doc_len: torch.Tensor, avgdl: float, k1: float, b: float) -> torch.Tensor:
    """k1*(1-b+b*|d|/avgdl) per document (the per-document sub-expression of bm25.py:154), float64."""
    _dev(doc_len, torch.int32, "bm25_doc_norms(doc_len)")
    out = torch.empty(doc_len.numel(), dtype=torch.float64, device=doc_len.device)
    check(_lib.lib().fz_bm25_doc_norms_f64(_ptr(doc_len), doc_len.numel(), float(avgdl), float(k1), float(b), _ptr(out), _stream(doc_len)),
          "fz_bm25_doc_norms_f64")
    return out


def bm25_slice_offsets(toff: torch.Tensor, pdoc: torch.Tensor, N: int) -> torch.Tensor:
    """Per-index table [V, NS + 1] int64 for bm25_scores(slice_off=...): where every term's postings cross the document slices one
    workgroup scores (fz_bm25_slice_offsets; built once per index, like the idf table)."""
    _dev(toff, torch.int64, "bm25_slice_offsets(toff)"); _dev(pdoc, torch.int32, "bm25_slice_offsets(pdoc)")
    _need(toff.is_contiguous() and pdoc.is_contiguous() and toff.numel() >= 1, "bm25_slice_offsets: contiguous toff [V + 1] and pdoc expected")
    V = toff.numel() - 1
    lib = _lib.lib()
    NS = max(1, -(-int(N) // int(lib.fz_bm25_slice_docs())))
    out = torch.empty((V, NS + 1), dtype=torch.int64, device=toff.device)
    check(lib.fz_bm25_slice_offsets(_ptr(toff), _ptr(pdoc), V, int(N), _ptr(out), _stream(toff)), "fz_bm25_slice_offsets")
    return out


def bm25_scores(toff, pdoc, ptf, idf, doc_len, avgdl: float, k1: float, b: float, qoff, qterms, Q: int, N: int,
                doc_norm: torch.Tensor | None = None, slice_off: torch.Tensor | None = None, want_f32: bool = False,
                pval: torch.Tensor | None = None):
    """BM25 scores [Q, N] float64 (bm25.py:149-156).  want_f32: also the float32 rounding of the same scores (the plane the normalisations
    read, hybrid.py:255), written by the same launch -> (float64 plane, float32 plane).  pval: the posting-value table of this index for
    this (k1, b) (bm25_posting_values): the walk then adds tabulated terms instead of dividing per posting -- same bits."""
    dev = idf.device
    if pval is not None:
        for t, dt, what in ((toff, torch.int64, "toff"), (pdoc, torch.int32, "pdoc"), (pval, torch.float64, "pval"), (qoff, torch.int64, "qoff"),
                            (qterms, torch.int32, "qterms")):
            _need(_dev(t, dt, f"bm25_scores({what})").is_contiguous(), f"bm25_scores({what}) must be contiguous")
        _need(pval.numel() == pdoc.numel() and qoff.numel() == Q + 1, "bm25_scores: pval must hold one value per posting and qoff Q + 1 offsets")
        if slice_off is not None:
            NS = max(1, -(-int(N) // int(_lib.lib().fz_bm25_slice_docs())))
            _need(_dev(slice_off, torch.int64, "bm25_scores(slice_off)").is_contiguous() and tuple(slice_off.shape) == (toff.numel() - 1, NS + 1),
                  f"bm25_scores(slice_off): expected a contiguous [{toff.numel() - 1}, {NS + 1}] table (ops.bm25_slice_offsets)")
        out = torch.empty((max(Q, 1), max(round_up(N, _PAD), _PAD)), dtype=torch.float64, device=dev)[:Q, :N]
        out32 = alloc_plane(Q, N, torch.float32, dev) if want_f32 else None
        check(_lib.lib().fz_bm25_scores_pv_f64_f32(_ptr(toff), _ptr(pdoc), _ptr(pval), _ptr(slice_off), _ptr(qoff), _ptr(qterms), Q, N, _ptr(out),
                                                   _ld(out), _ptr(out32), _ld(out32) if want_f32 else 0, _stream(pval)), "fz_bm25_scores_pv_f64_f32")
        return (out, out32) if want_f32 else out
    for t, dt, what in ((toff, torch.int64, "toff"), (pdoc, torch.int32, "pdoc"), (ptf, torch.int32, "ptf"), (idf, torch.float64, "idf"),
                        (doc_len, torch.int32, "doc_len"), (qoff, torch.int64, "qoff"), (qterms, torch.int32, "qterms")):
        _need(_dev(t, dt, f"bm25_scores({what})").is_contiguous(), f"bm25_scores({what}) must be contiguous")
    _need(toff.numel() == idf.numel() + 1, "bm25_scores: toff must hold V + 1 offsets for idf's V terms")
    _need(pdoc.numel() == ptf.numel(), "bm25_scores: pdoc and ptf differ in length")
    _need(doc_len.numel() == N and qoff.numel() == Q + 1, f"bm25_scores: doc_len must hold {N} lengths and qoff {Q + 1} offsets")
    if doc_norm is not None:
        _need(_dev(doc_norm, torch.float64, "bm25_scores(doc_norm)").numel() == N and doc_norm.is_contiguous(), f"bm25_scores: doc_norm must hold {N} values")
    if slice_off is not None:
        NS = max(1, -(-int(N) // int(_lib.lib().fz_bm25_slice_docs())))
        _need(_dev(slice_off, torch.int64, "bm25_scores(slice_off)").is_contiguous() and tuple(slice_off.shape) == (idf.numel(), NS + 1),
              f"bm25_scores(slice_off): expected a contiguous [{idf.numel()}, {NS + 1}] table (ops.bm25_slice_offsets)")
    out = torch.empty((max(Q, 1), max(round_up(N, _PAD), _PAD)), dtype=torch.float64, device=dev)[:Q, :N]
    out32 = alloc_plane(Q, N, torch.float32, dev) if want_f32 else None
    check(_lib.lib().fz_bm25_scores_f64_f32(_ptr(toff), _ptr(pdoc), _ptr(ptf), _ptr(idf), _ptr(doc_len), _ptr(doc_norm), _ptr(slice_off), float(avgdl),
                                            float(k1), float(b), _ptr(qoff), _ptr(qterms), Q, N, _ptr(out), _ld(out), _ptr(out32),
                                            _ld(out32) if want_f32 else 0, _stream(idf)), "fz_bm25_scores_f64_f32")
    return (out, out32) if want_f32 else out


def bm25_posting_values(toff, pdoc, ptf, idf, doc_norm, k1: float) -> torch.Tensor:
    """[nnz] float64: every posting's BM25 term idf * (tf (k1 + 1)) / (tf + doc_norm[d]) (bm25.py:154) for the (k1, b) doc_norm was made
    with -- per index, like the idf table (fz_bm25_posting_values_f64); bm25_scores(pval=...) then only adds."""
    for t, dt, what in ((toff, torch.int64, "toff"), (pdoc, torch.int32, "pdoc"), (ptf, torch.int32, "ptf"), (idf, torch.float64, "idf"),
                        (doc_norm, torch.float64, "doc_norm")):
        _need(_dev(t, dt, f"bm25_posting_values({what})").is_contiguous(), f"bm25_posting_values({what}) must be contiguous")
    _need(toff.numel() == idf.numel() + 1 and pdoc.numel() == ptf.numel(), "bm25_posting_values: toff [V + 1], idf [V], pdoc / ptf [nnz] expected")
    out = torch.empty(pdoc.numel(), dtype=torch.float64, device=idf.device)
    check(_lib.lib().fz_bm25_posting_values_f64(_ptr(toff), _ptr(pdoc), _ptr(ptf), _ptr(idf), _ptr(doc_norm), idf.numel(), pdoc.numel(), float(k1),
                                                _ptr(out), _stream(idf)), "fz_bm25_posting_values_f64")
    return out


def tfidf_scores(toff, pdoc, ptf, idf, qoff, qterms, Q: int, N: int, slice_off: torch.Tensor | None = None, want_f32: bool = False):
    """TF-IDF scores [Q, N] float64 (TFIDF.score, bm25.py:108-115: sum over the query's terms, in query order, of tf * idf); want_f32: also
    their float32 rounding from the same launch."""
    dev = idf.device
    for t, dt, what in ((toff, torch.int64, "toff"), (pdoc, torch.int32, "pdoc"), (ptf, torch.int32, "ptf"), (idf, torch.float64, "idf"),
                        (qoff, torch.int64, "qoff"), (qterms, torch.int32, "qterms")):
        _need(_dev(t, dt, f"tfidf_scores({what})").is_contiguous(), f"tfidf_scores({what}) must be contiguous")
    _need(toff.numel() == idf.numel() + 1 and pdoc.numel() == ptf.numel() and qoff.numel() == Q + 1,
          "tfidf_scores: toff must hold V + 1 offsets for idf's V terms, pdoc / ptf one entry per posting, qoff Q + 1 offsets")
    if slice_off is not None:
        NS = max(1, -(-int(N) // int(_lib.lib().fz_bm25_slice_docs())))
        _need(_dev(slice_off, torch.int64, "tfidf_scores(slice_off)").is_contiguous() and tuple(slice_off.shape) == (idf.numel(), NS + 1),
              f"tfidf_scores(slice_off): expected a contiguous [{idf.numel()}, {NS + 1}] table (ops.bm25_slice_offsets)")
    out = torch.empty((max(Q, 1), max(round_up(N, _PAD), _PAD)), dtype=torch.float64, device=dev)[:Q, :N]
    out32 = alloc_plane(Q, N, torch.float32, dev) if want_f32 else None
    check(_lib.lib().fz_tfidf_scores_f64(_ptr(toff), _ptr(pdoc), _ptr(ptf), _ptr(idf), _ptr(slice_off), _ptr(qoff), _ptr(qterms), Q, N,
                                         _ptr(out), _ld(out), _ptr(out32), _ld(out32) if want_f32 else 0, _stream(idf)), "fz_tfidf_scores_f64")
    return (out, out32) if want_f32 else out


# ---------------------------------------------------------------------------------------
# A3, sparse form: SPLADE cosine scoring over an inverted index
# ---------------------------------------------------------------------------------------
class SparseIndex:
    """L2-normalised corpus vectors as postings: toff [V + 1] int64, pdoc [nnz] int32 (ascending inside a term), pw [nnz] float32,
    slice_off [V, NS + 1] int64 (fz_sparse_slice_offsets).  N documents over a vocabulary of V terms."""

    def __init__(self, toff, pdoc, pw, N: int, V: int, slice_off=None):
        self.toff, self.pdoc, self.pw, self.N, self.V = toff, pdoc, pw, int(N), int(V)
        self.slice_off = slice_off if slice_off is not None else sparse_slice_offsets(toff, pdoc, self.V, self.N)

    @property
    def nnz(self) -> int:
        return int(self.pdoc.numel())

    def tensors(self):
        return self.toff, self.pdoc, self.pw

    def dense_cached(self) -> torch.Tensor:
        """to_dense(), padded for the GEMM, built once and kept WHILE the index's buffers are the ones it was built from (a rebuilt or
        re-pointed index re-densifies); drop_dense() gives the N x V float32 matrix (3.6 GB at LLeQA size) back."""
        key = (self.toff.data_ptr(), self.pdoc.data_ptr(), self.pw.data_ptr(), self.nnz, self.N, self.V)
        if getattr(self, "_dense_key", None) != key:
            self._dense, self._dense_key = pad_dim(self.to_dense()), key
        return self._dense

    def drop_dense(self) -> None:
        self._dense, self._dense_key = None, None

    def to_dense(self) -> torch.Tensor:
        """The [N, V rounded up to 4] float32 matrix the index was built from (normalised rows)."""
        D = torch.zeros((self.N, round_up(max(self.V, 1), 4)), dtype=torch.float32, device=self.pw.device)
        term = torch.repeat_interleave(torch.arange(self.V, device=self.pw.device), self.toff[1:] - self.toff[:-1])
        D[self.pdoc.long(), term] = self.pw
        return D


def sparse_slice_offsets(toff: torch.Tensor, pdoc: torch.Tensor, V: int, N: int) -> torch.Tensor:
    _need(_dev(toff, torch.int64, "sparse_slice_offsets(toff)").is_contiguous() and toff.numel() == V + 1, f"sparse_slice_offsets: toff must hold {V + 1} offsets")
    _need(_dev(pdoc, torch.int32, "sparse_slice_offsets(pdoc)").is_contiguous(), "sparse_slice_offsets(pdoc) must be contiguous")
    NS = max(1, -(-int(N) // int(_lib.lib().fz_sparse_slice_docs())))
    out = torch.empty((V, NS + 1), dtype=torch.int64, device=toff.device)
    check(_lib.lib().fz_sparse_slice_offsets(_ptr(toff), _ptr(pdoc), V, int(N), _ptr(out), _stream(toff)), "fz_sparse_slice_offsets")
    return out


def density(X: torch.Tensor) -> float:
    """Fraction of non-zero entries (one device reduction)."""
    return float(torch.count_nonzero(X).item()) / max(1, X.numel())


def sparse_index(Dn: torch.Tensor, V: int | None = None, rows: int = 4096) -> SparseIndex:
    """Inverted index of the rows of Dn [N, >= V] (already L2-normalised, e.g. ops.normalize_rows): built from row blocks (a block's
    non-zeros, then one stable sort by term -- documents stay ascending inside a term)."""
    _dev(Dn, torch.float32, "sparse_index(Dn)")
    N = Dn.shape[0]
    V = Dn.shape[1] if V is None else int(V)
    docs, terms, vals = [], [], []
    for r0 in range(0, N, rows):
        blk = Dn[r0: r0 + rows, :V]
        nz = blk.nonzero()
        docs.append((nz[:, 0] + r0).int()); terms.append(nz[:, 1]); vals.append(blk[nz[:, 0], nz[:, 1]])
    doc = torch.cat(docs) if docs else torch.zeros(0, dtype=torch.int32, device=Dn.device)
    term = torch.cat(terms) if terms else torch.zeros(0, dtype=torch.int64, device=Dn.device)
    val = torch.cat(vals) if vals else torch.zeros(0, dtype=torch.float32, device=Dn.device)
    order = torch.sort(term, stable=True).indices           # documents were appended ascending: stable keeps them so inside a term
    toff = torch.zeros(V + 1, dtype=torch.int64, device=Dn.device)
    toff[1:] = torch.cumsum(torch.bincount(term, minlength=V), 0)
    return SparseIndex(toff, doc[order].contiguous(), val[order].contiguous(), N, V)


def sparse_rows(Qn: torch.Tensor, V: int | None = None):
    """The non-zero (term, weight) lists of the rows of Qn [Q, >= V]: (qoff [Q + 1] int64, qterms int32, qw float32), terms ascending."""
    _dev(Qn, torch.float32, "sparse_rows(Qn)")
    V = Qn.shape[1] if V is None else int(V)
    blk = Qn[:, :V]
    nz = blk.nonzero()                                        # row-major: by query, then ascending term
    qoff = torch.zeros(Qn.shape[0] + 1, dtype=torch.int64, device=Qn.device)
    qoff[1:] = torch.cumsum(torch.bincount(nz[:, 0], minlength=Qn.shape[0]), 0)
    return qoff, nz[:, 1].int().contiguous(), blk[nz[:, 0], nz[:, 1]].contiguous()


def sparse_dot(index: SparseIndex, qoff: torch.Tensor, qterms: torch.Tensor, qw: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """scores[q][d] = sum over query q's terms of qw * (document d's weight of that term): [Q, N] float32 plane."""
    for t, dt, what in ((qoff, torch.int64, "qoff"), (qterms, torch.int32, "qterms"), (qw, torch.float32, "qw")):
        _need(_dev(t, dt, f"sparse_dot({what})").is_contiguous(), f"sparse_dot({what}) must be contiguous")
    Q = qoff.numel() - 1
    _need(Q >= 0 and qterms.numel() == qw.numel(), "sparse_dot: qoff must hold Q + 1 offsets, qterms and qw one entry per non-zero")
    if out is None:
        out = alloc_plane(Q, index.N, torch.float32, qoff.device)
    else:
        _dev(out, torch.float32, "sparse_dot(out)")
        _need(tuple(out.shape) == (Q, index.N), f"sparse_dot(out): expected shape {(Q, index.N)}, got {tuple(out.shape)}")
    check(_lib.lib().fz_sparse_dot_f32(_ptr(index.toff), _ptr(index.pdoc), _ptr(index.pw), _ptr(index.slice_off), _ptr(qoff), _ptr(qterms), _ptr(qw),
                                       Q, index.N, _ptr(out), _ld(out), _stream(qoff)), "fz_sparse_dot_f32")
    return out


def sparse_cos_scores(Qe: torch.Tensor, index: SparseIndex, max_query_density: float = 0.05) -> torch.Tensor:
    """Cosine scores of dense query vectors Qe [Q, >= V] against a SparseIndex (whose rows were normalised before indexing).  Queries that are
    not sparse themselves (an untrained head activates half the vocabulary: one barrier per term and query) take the dense GEMM against the
    re-densified corpus instead -- same scores either way."""
    Qn = normalize_rows(pad_dim(Qe))
    if density(Qn[:, :index.V]) > max_query_density:
        # re-densified ONCE per index and kept (N x V float32: 3.6 GB at LLeQA size -- a fresh allocation + scatter per query batch would
        # cost more than the GEMM it feeds); a query encoder that is not sparse stays on this path for every batch
        return dot_scores(Qn, index.dense_cached())
    return sparse_dot(index, *sparse_rows(Qn, index.V))


def f64_to_f32(src: torch.Tensor) -> torch.Tensor:
    """Plane-preserving fp64 -> fp32 (what torch.tensor(..., dtype=float32) does to BM25's Python floats, hybrid.py:255)."""
    _dev(src, torch.float64, "f64_to_f32")
    rows, n = src.shape
    ld = _ld(src)
    dst = torch.empty((max(rows, 1), ld), dtype=torch.float32, device=src.device)
    # convert the whole padded buffer when it is one allocation; otherwise row views
    check(_lib.lib().fz_f64_to_f32(_ptr(src), _ptr(dst), rows * ld - (ld - n) if rows > 0 else 0, _stream(src)), "fz_f64_to_f32")
    return dst[:rows, :n]


# ---------------------------------------------------------------------------------------
# encoder side: packed (padding-free) token rows
# ---------------------------------------------------------------------------------------
def attn_strips(lengths, cu_rows=None):
    """HOST helper: the strip table fz_attn_varlen_f32 walks -- one (first row, length, first query, 0) entry per 32 queries
    of every sequence, longest sequences first.  Returns (strips int32 [n_strips, 4] numpy, cu_rows int32 [B+1] numpy)."""
    import numpy as np
    lengths = np.asarray(lengths, dtype=np.int64)
    if cu_rows is None:
        cu_rows = np.zeros(len(lengths) + 1, dtype=np.int64)
        np.cumsum(lengths, out=cu_rows[1:])
    per = (lengths + 31) // 32
    order = np.argsort(-lengths, kind="stable")
    seq = np.repeat(order, per[order])
    first = np.zeros(len(order) + 1, dtype=np.int64)
    np.cumsum(per[order], out=first[1:])
    q0 = (np.arange(len(seq)) - np.repeat(first[:-1], per[order])) * 32
    strips = np.stack([cu_rows[seq], lengths[seq], q0, np.zeros_like(q0)], 1).astype(np.int32)
    return np.ascontiguousarray(strips), cu_rows.astype(np.int32)


def attn_varlen(qkv: torch.Tensor, strips: torch.Tensor, heads: int, scale: float | None = None, out: torch.Tensor | None = None):
    """softmax(q k^T scale) v per sequence and head over packed rows; qkv [T, 3*heads*64] fp32 -> [T, heads*64]."""
    _dev(qkv, torch.float32, "attn_varlen(qkv)")
    _dev(strips, torch.int32, "attn_varlen(strips)")
    T, W = qkv.shape
    if W != 3 * heads * 64:
        raise ValueError(f"attn_varlen: qkv is {W} wide, expected 3*{heads}*64 (head_dim 64 only)")
    if strips.dim() != 2 or strips.shape[1] != 4 or not strips.is_contiguous():
        raise ValueError("attn_varlen: strips must be a contiguous [n_strips, 4] int32 tensor (ops.attn_strips)")
    if out is None:
        out = torch.empty((T, heads * 64), dtype=torch.float32, device=qkv.device)
    else:
        _dev(out, torch.float32, "attn_varlen(out)")
        _need(tuple(out.shape) == (T, heads * 64), f"attn_varlen(out): expected shape {(T, heads * 64)}, got {tuple(out.shape)}")
    if scale is not None and not scale > 0:
        raise ValueError("attn_varlen: scale must be positive")
    check(_lib.lib().fz_attn_varlen_f32(_ptr(qkv), qkv.stride(0) if T > 1 else W, _ptr(strips), strips.shape[0], heads, 64,
                                        float(64 ** -0.5 if scale is None else scale), _ptr(out), out.stride(0) if T > 1 else heads * 64,
                                        _stream(qkv)), "fz_attn_varlen_f32")
    return out


def attn_varlen_f16(qkv16: torch.Tensor, strips: torch.Tensor, heads: int, out: torch.Tensor, amp: bool = False) -> torch.Tensor:
    """attn_varlen on float16 fused-QKV rows with the context rows stored as float16 (`out` [T, heads*64] float16): the attention of the
    mixed-precision forward -- float32 scores, softmax and weighted sum between two float16 Linears."""
    _dev(qkv16, torch.float16, "attn_varlen_f16(qkv16)")
    _dev(strips, torch.int32, "attn_varlen_f16(strips)")
    _dev(out, torch.float16, "attn_varlen_f16(out)")
    T, W = qkv16.shape
    if W != 3 * heads * 64:
        raise ValueError(f"attn_varlen_f16: qkv is {W} wide, expected 3*{heads}*64 (head_dim 64 only)")
    if strips.dim() != 2 or strips.shape[1] != 4 or not strips.is_contiguous():
        raise ValueError("attn_varlen_f16: strips must be a contiguous [n_strips, 4] int32 tensor (ops.attn_strips)")
    _need(tuple(out.shape) == (T, heads * 64), f"attn_varlen_f16(out): expected shape {(T, heads * 64)}, got {tuple(out.shape)}")
    fn = _lib.lib().fz_attn_varlen_f16_amp if amp else _lib.lib().fz_attn_varlen_f16     # amp: float16 matmuls (autocast's arithmetic); else float32 throughout
    check(fn(_ptr(qkv16), qkv16.stride(0) if T > 1 else W, _ptr(strips), strips.shape[0], heads, 64, float(64 ** -0.5),
             _ptr(out), out.stride(0) if T > 1 else heads * 64, _stream(qkv16)), "fz_attn_varlen_f16_amp" if amp else "fz_attn_varlen_f16")
    return out


def add_layernorm_x16(x16: torch.Tensor, res: torch.Tensor | None, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                      out: torch.Tensor | None = None, out16: torch.Tensor | None = None):
    """LayerNorm(x16 + res) with a float16 x (a mixed-precision Linear's output), float32 residual and result; out16 (float16, same
    shape) receives a second, rounded copy of the result.  -> out."""
    _dev(x16, torch.float16, "add_layernorm_x16(x16)")
    rows, d = x16.shape
    if res is not None:
        _dev(res, torch.float32, "add_layernorm_x16(res)")
        if res.shape != x16.shape:
            raise ValueError("add_layernorm_x16: x16 and res differ in shape")
    _need(gamma.numel() == d and beta.numel() == d and gamma.is_contiguous() and beta.is_contiguous(), f"add_layernorm_x16: gamma and beta must hold {d} values")
    if out is None:
        out = torch.empty((rows, d), dtype=torch.float32, device=x16.device)
    else:
        _dev(out, torch.float32, "add_layernorm_x16(out)")
        _need(tuple(out.shape) == (rows, d), f"add_layernorm_x16(out): expected shape {(rows, d)}, got {tuple(out.shape)}")
    if out16 is not None:
        _dev(out16, torch.float16, "add_layernorm_x16(out16)")
        _need(tuple(out16.shape) == (rows, d), f"add_layernorm_x16(out16): expected shape {(rows, d)}, got {tuple(out16.shape)}")
    ldr = 0 if res is None else (res.stride(0) if rows > 1 else d)
    check(_lib.lib().fz_add_layernorm_x16(_ptr(x16), x16.stride(0) if rows > 1 else d, _ptr(res), ldr, _ptr(_dev(gamma, torch.float32, "gamma")),
                                          _ptr(_dev(beta, torch.float32, "beta")), float(eps), rows, d, _ptr(out), out.stride(0) if rows > 1 else d,
                                          _ptr(out16), 0 if out16 is None else (out16.stride(0) if rows > 1 else d), _stream(x16)), "fz_add_layernorm_x16")
    return out


def gelu_f16_(x: torch.Tensor) -> torch.Tensor:
    """In-place erf-GELU of a contiguous float16 tensor (float32 arithmetic, one rounding: torch.nn.functional.gelu on float16)."""
    _dev(x, torch.float16, "gelu_f16_(x)")
    _need(x.is_contiguous() and x.numel() % 8 == 0, "gelu_f16_: contiguous tensor with a multiple of 8 elements")
    check(_lib.lib().fz_gelu_f16(_ptr(x), _ptr(x), x.numel(), _stream(x)), "fz_gelu_f16")
    return x


def add_layernorm(x: torch.Tensor, res: torch.Tensor | None, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor | None = None):
    """LayerNorm(x + res) over the last dimension, fp32, one HBM pass."""
    _dev(x, torch.float32, "add_layernorm(x)")
    rows, d = x.shape
    if res is not None:
        _dev(res, torch.float32, "add_layernorm(res)")
        if res.shape != x.shape:
            raise ValueError("add_layernorm: x and res differ in shape")
    _need(gamma.numel() == d and beta.numel() == d and gamma.is_contiguous() and beta.is_contiguous(), f"add_layernorm: gamma and beta must hold {d} values")
    if out is None:
        out = torch.empty((rows, d), dtype=torch.float32, device=x.device)
    else:
        _dev(out, torch.float32, "add_layernorm(out)")
        _need(tuple(out.shape) == (rows, d), f"add_layernorm(out): expected shape {(rows, d)}, got {tuple(out.shape)}")
    ldr = 0 if res is None else (res.stride(0) if rows > 1 else d)
    check(_lib.lib().fz_add_layernorm_f32(_ptr(x), x.stride(0) if rows > 1 else d, _ptr(res), ldr, _ptr(_dev(gamma, torch.float32, "gamma")),
                                          _ptr(_dev(beta, torch.float32, "beta")), float(eps), rows, d, _ptr(out),
                                          out.stride(0) if rows > 1 else d, _stream(x)), "fz_add_layernorm_f32")
    return out


def gelu_(x: torch.Tensor) -> torch.Tensor:
    """In-place erf-GELU of a contiguous fp32 tensor (torch.nn.functional.gelu's expression)."""
    _dev(x, torch.float32, "gelu_(x)")
    _need(x.is_contiguous() and x.numel() % 4 == 0, "gelu_: contiguous tensor with a multiple of 4 elements")
    check(_lib.lib().fz_gelu_f32(_ptr(x), _ptr(x), x.numel(), _stream(x)), "fz_gelu_f32")
    return x


def embed_layernorm(word: torch.Tensor, pos: torch.Tensor, type0: torch.Tensor, ids: torch.Tensor, pos_ids: torch.Tensor,
                    gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: torch.Tensor | None = None) -> torch.Tensor:
    """LayerNorm(word[ids] + pos[pos_ids] + type0) for packed rows, one pass; out may be a [>= rows, d] buffer (first rows written)."""
    for t_, w_ in ((word, "word"), (pos, "pos"), (type0, "type0"), (gamma, "gamma"), (beta, "beta")):
        _need(_dev(t_, torch.float32, f"embed_layernorm({w_})").is_contiguous(), f"embed_layernorm({w_}) must be contiguous")
    _dev(ids, torch.int64, "embed_layernorm(ids)"); _dev(pos_ids, torch.int64, "embed_layernorm(pos_ids)")
    rows, d = ids.numel(), word.shape[1]
    _need(pos_ids.numel() == rows and ids.is_contiguous() and pos_ids.is_contiguous(), "embed_layernorm: ids and pos_ids must be contiguous and equally long")
    _need(pos.shape[1] == d and type0.numel() == d and gamma.numel() == d and beta.numel() == d, f"embed_layernorm: tables and parameters must be {d} wide")
    if out is None:
        out = torch.empty((rows, d), dtype=torch.float32, device=word.device)
    else:
        _dev(out, torch.float32, "embed_layernorm(out)")
        _need(out.dim() == 2 and out.shape[0] >= rows and out.shape[1] == d, f"embed_layernorm(out): need a [>= {rows}, {d}] buffer")
    check(_lib.lib().fz_embed_layernorm_f32(_ptr(word), _ptr(pos), _ptr(type0), _ptr(ids), _ptr(pos_ids), _ptr(gamma), _ptr(beta), float(eps), rows, d,
                                            _ptr(out), out.stride(0) if out.shape[0] > 1 else d, _stream(word)), "fz_embed_layernorm_f32")
    return out


def _segment_reduce(fn: str, x: torch.Tensor, cu_rows: torch.Tensor) -> torch.Tensor:
    _dev(x, torch.float32, f"{fn}(x)")
    _dev(cu_rows, torch.int32, f"{fn}(cu_rows)")
    B, d = cu_rows.numel() - 1, x.shape[1]
    if x.shape[0] == 0:     # every sequence empty (an empty tensor has no data pointer to hand over): zeros by definition
        return torch.zeros((max(B, 0), d), dtype=torch.float32, device=x.device)
    out = torch.empty((max(B, 0), d), dtype=torch.float32, device=x.device)
    check(getattr(_lib.lib(), fn)(_ptr(x), x.stride(0) if x.shape[0] > 1 else d, _ptr(cu_rows), B, d, _ptr(out), d, _stream(x)), fn)
    return out


def splade_head_max(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, cu_rows: torch.Tensor) -> torch.Tensor:
    """SPLADE vocabulary projection with the pooling as its epilogue (splade/splade.py:88-99): x [T, d] packed hidden rows, weight [V, d],
    bias [V], cu_rows [B+1] int32 -> [B, V] fp32 = max over each sequence's rows of log1p(relu(x @ weight.T + bias)); the [T, V] logits
    are never written (fz_splade_head_max_f32: the fp32-MFMA tile stream of dot_scores with a segment-max / atomicMax epilogue)."""
    _dev(x, torch.float32, "splade_head_max(x)"); _dev(weight, torch.float32, "splade_head_max(weight)")
    _dev(bias, torch.float32, "splade_head_max(bias)"); _dev(cu_rows, torch.int32, "splade_head_max(cu_rows)")
    T, d = x.shape
    V = weight.shape[0]
    B = cu_rows.numel() - 1
    _need(weight.dim() == 2 and weight.shape[1] == d and bias.numel() == V and bias.is_contiguous(), f"splade_head_max: weight [V, {d}] and bias [V] expected")
    _need(cu_rows.is_contiguous() and B >= 0, "splade_head_max: cu_rows must hold B + 1 contiguous offsets")
    x, weight = pad_dim(x), pad_dim(weight)
    ldp = max(round_up(V, _PAD), _PAD)
    pool = torch.zeros((max(B, 1), ldp), dtype=torch.float32, device=x.device)[:B, :V]     # zero = log1p(relu(.)) of an empty sequence
    if T == 0 or B == 0:
        return pool
    check(_lib.lib().fz_splade_head_max_f32(_ptr(x), x.stride(0) if T > 1 else x.shape[1], _ptr(weight), weight.stride(0), _ptr(bias), _ptr(cu_rows), B, T, V,
                                            x.shape[1], _ptr(pool), ldp, _stream(x)), "fz_splade_head_max_f32")
    return pool


def segment_mean(x: torch.Tensor, cu_rows: torch.Tensor) -> torch.Tensor:
    """Mean over each sequence's rows: x [T, d] fp32, cu_rows [B+1] int32 -> [B, d]."""
    return _segment_reduce("fz_segment_mean_f32", x, cu_rows)


def segment_splade_max(x: torch.Tensor, cu_rows: torch.Tensor) -> torch.Tensor:
    """SPLADE-max pooling: log1p(relu(max over each sequence's rows)): x [T, V] fp32 logits -> [B, V]."""
    return _segment_reduce("fz_segment_splade_max_f32", x, cu_rows)
