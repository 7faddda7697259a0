"""ctypes.PyDLL binding of libfusion_pyhost.so (csrc/pyhost.c): the two list-of-dict walks of the reference-typed boundary
(hybrid.py:66-75) against the CPython C API.  Built by `make -C fusion_amd/csrc` next to libfusion_hip.so; like it, required."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfusion_pyhost.so")
_lib = None
KEY_ID, KEY_SCORE = "corpus_id", "score"


def lib() -> C.PyDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.PyDLL(LIB_PATH)   # PyDLL: the GIL stays held across the calls, which work on Python objects
        L.fzh_extract.restype, L.fzh_extract.argtypes = C.c_int, [C.py_object, C.py_object, C.py_object, C.c_void_p, C.c_void_p, C.c_ssize_t]
        L.fzh_build.restype, L.fzh_build.argtypes = C.py_object, [C.py_object, C.py_object, C.py_object, C.py_object]
        _lib = L
    return _lib


def extract(lst: list):
    """-> (ids int64 [n], scores float64 [n]) of a list of {'corpus_id', 'score'} dicts, or None when the ids are not plain ints / the
    list is not of that form (the caller then walks it the reference's own way, which raises what the reference would)."""
    n = len(lst)
    ids = np.empty(n, dtype=np.int64)
    sc = np.empty(n, dtype=np.float64)
    if lib().fzh_extract(lst, KEY_ID, KEY_SCORE, ids.ctypes.data, sc.ctypes.data, n) != 0:
        return None
    return ids, sc


def build(ids: list, scores: list) -> list[dict]:
    """[{'corpus_id': i, 'score': s} ...] from two equally long Python lists."""
    return lib().fzh_build(KEY_ID, KEY_SCORE, ids, scores)
