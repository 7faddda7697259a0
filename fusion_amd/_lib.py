"""ctypes binding of libfusion_hip.so (C ABI: include/fusion_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the ops raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FUSION_AMD_LIB: another build of the same library (the host-sanitized one of `make -C fusion_amd/csrc hostasan`); same ABI check applies
LIB_PATH = os.environ.get("FUSION_AMD_LIB") or os.path.join(_HERE, "libfusion_hip.so")
ABI_VERSION = 19

FZ_OK, FZ_ERR_ARG, FZ_ERR_UNSUPPORTED, FZ_ERR_HIP, FZ_ERR_WORKSPACE = 0, -1, -2, -3, -4
NORMS = {"min-max": 1, "z-score": 2, "arctan": 3, "percentile-rank": 4, "normal-curve-equivalent": 5}
RANK_METHODS = {"rrf": 0, "bcf": 1}


class FusionHipError(RuntimeError):
    def __init__(self, what: str, status: int, detail: str):
        super().__init__(f"{what}: {detail} (fz_status {status})")
        self.status = status


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into libfusion_hip.so (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src_dir, "-j4", "-s"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


_lib = None

_vp, _i, _i64, _sz, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double
_PROTOS = {
    "fz_strerror": (C.c_char_p, [_i]),
    "fz_last_hip_error": (_i, []),
    "fz_abi_version": (_i, []),
    "fz_normalize_rows_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "fz_dot_scores_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "fz_dot_scores_filter_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "fz_maxsim_f16": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "fz_sort_max_n": (_i, []),
    "fz_sort_max_n_f64": (_i, []),
    "fz_sort_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_sort_bucket_rank_rows": (_i, [_vp, _i]),
    "fz_sort_zero_compact_rows": (_i, [_vp, _i]),
    "fz_sort_rows_desc_lexical": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_sort_rows_desc": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_sort_rows_desc_placed": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_sort_rank_fused_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_sort_rank_fused_desc": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_rrf_terms_f64": (_i, [_i, _i, _vp, _vp]),
    "fz_select_topk_f": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fz_fuse_rank_f64": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "fz_row_stats_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "fz_rank_to_bitmap": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "fz_fuse_nsf_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "fz_fuse_nsf_stats_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "fz_fuse_nsf_pstats_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "fz_nsf_tables_workspace_bytes": (_sz, [_i, _vp, _i]),
    "fz_nsf_tables_header_offset": (_sz, [_i, _vp, _i, _i]),
    "fz_nsf_tables_prepare": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "fz_fuse_nsf_tables_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "fz_nsf_tables_path": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "fz_zero_unlisted_f32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "fz_minmax_from_orders_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "fz_minmax_from_order_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "fz_fuse_none_f64": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fz_fuse_wsum_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fz_insertion_order_workspace_bytes": (_sz, [_i, _i]),
    "fz_insertion_order": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_topk_max_k": (_i, []),
    "fz_topk_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_topk_rows_f32": (_i, [_vp, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "fz_topk_update_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_topk_update_f32": (_i, [_vp, _i, _i, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_topk_filter_append_f32": (_i, [_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "fz_topk_fold_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_topk_fold_f32": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fz_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "fz_topk_allgather_workspace_bytes": (_sz, [_i, _i, _i]),
    "fz_topk_allgather": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "fz_bm25_doc_norms_f64": (_i, [_vp, _i, _d, _d, _d, _vp, _vp]),
    "fz_bm25_slice_docs": (_i, []),
    "fz_bm25_slice_offsets": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fz_bm25_scores_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "fz_bm25_scores_f64_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _d, _d, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp]),
    "fz_bm25_posting_values_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _d, _vp, _vp]),
    "fz_bm25_scores_pv_f64_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp]),
    "fz_tfidf_scores_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp]),
    "fz_tune_max_gold": (_i, []),
    "fz_gold_ranks_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "fz_gold_ranks_f64w": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "fz_tune_metrics_f64": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "fz_sparse_slice_docs": (_i, []),
    "fz_sparse_slice_offsets": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fz_sparse_dot_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "fz_attn_varlen_f32": (_i, [_vp, _i, _vp, _i, _i, _i, C.c_float, _vp, _i, _vp]),
    "fz_add_layernorm_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, C.c_float, _i, _i, _vp, _i, _vp]),
    "fz_gelu_f32": (_i, [_vp, _vp, _sz, _vp]),
    "fz_attn_varlen_f16": (_i, [_vp, _i, _vp, _i, _i, _i, C.c_float, _vp, _i, _vp]),
    "fz_attn_varlen_f16_amp": (_i, [_vp, _i, _vp, _i, _i, _i, C.c_float, _vp, _i, _vp]),
    "fz_add_layernorm_x16": (_i, [_vp, _i, _vp, _i, _vp, _vp, C.c_float, _i, _i, _vp, _i, _vp, _i, _vp]),
    "fz_gelu_f16": (_i, [_vp, _vp, _sz, _vp]),
    "fz_embed_layernorm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i, _i, _vp, _i, _vp]),
    "fz_segment_mean_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "fz_segment_splade_max_f32": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "fz_splade_head_max_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "fz_fill_i32": (_i, [_vp, _sz, C.c_int32, _vp]),
    "fz_f64_to_f32": (_i, [_vp, _vp, _sz, _vp]),
}
EXPORTS = tuple(_PROTOS)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). fusion_amd has no CPU or PyTorch fallback for the scoring/fusion path.")
        # torch FIRST: PyTorch-ROCm ships its own libamdhip64, and a process must hold ONE HIP runtime -- loaded before torch, this library
        # binds the system's copy, torch then brings its own, and every device pointer torch hands over is foreign to the runtime the kernels
        # launch through (hipErrorNoDevice on the first launch: seen with build() and smoke() in one process)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            f = getattr(L, name)  # AttributeError if the library does not export what the header declares
            f.restype, f.argtypes = res, args
        v = L.fz_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"libfusion_hip.so ABI {v} != binding ABI {ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def check(status: int, what: str):
    if status != FZ_OK:
        L = lib()
        detail = L.fz_strerror(status).decode()
        if status == FZ_ERR_HIP:
            detail += f" [hipError {L.fz_last_hip_error()}]"
        raise FusionHipError(what, status, detail)
