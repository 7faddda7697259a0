"""Shared comparison helpers for the parity tests."""
import numpy as np


def assert_ranked_close(got_ids, got_sc, exp_ids, exp_sc, tol, truncated=False):
    """Two score-descending ranked lists agree: the score SEQUENCES are within `tol` element by element (order
    statistics move by at most the perturbation of the scores), and the ids agree as sets inside every run of
    expected scores that are closer than 2*tol to their neighbour (ties / near-ties: their relative order is
    implementation-defined in the reference -- unsorted topk + heap order, splade/base.py:229-240).  With
    `truncated` (k < N) the last run may have lost members to the cut and is skipped."""
    got_ids, exp_ids = np.asarray(got_ids), np.asarray(exp_ids)
    got_sc, exp_sc = np.asarray(got_sc, dtype=np.float64), np.asarray(exp_sc, dtype=np.float64)
    assert got_ids.shape == exp_ids.shape and got_sc.shape == exp_sc.shape == got_ids.shape
    assert np.max(np.abs(got_sc - exp_sc), initial=0.0) <= tol
    assert np.all(got_sc[:-1] >= got_sc[1:]), "list not sorted by score descending"
    n = len(exp_sc)
    cut = np.flatnonzero(exp_sc[:-1] - exp_sc[1:] > 2 * tol) + 1
    bounds = np.concatenate([[0], cut, [n]])
    for a, b in zip(bounds[:-1], bounds[1:]):
        if truncated and b == n:
            continue
        if b - a == 1:
            assert got_ids[a] == exp_ids[a], (a, got_ids[a], exp_ids[a])
        else:
            assert sorted(got_ids[a:b].tolist()) == sorted(exp_ids[a:b].tolist()), (a, b)


def sparse_to_dense(z, prefix, rows, V):
    X = np.zeros((rows, V), dtype=np.float32)
    X[z[f"{prefix}_row"], z[f"{prefix}_col"]] = z[f"{prefix}_val"]
    return X


def load_lists(z):
    """tests/golden/*.npz (in_ids / in_scores / in_len) -> the reference's RankedLists per system."""
    systems = [str(s) for s in z["systems"]]
    ids, sc, ln = z["in_ids"], z["in_scores"], z["in_len"]
    Q = ids.shape[1]
    lists = {s: [[{"corpus_id": int(ids[si, q, r]), "score": float(sc[si, q, r])} for r in range(ln[si, q])] for q in range(Q)]
             for si, s in enumerate(systems)}
    return systems, lists, Q


def integration_snippet(root):
    """The reference-side ctypes binding printed in INTEGRATION.md section 2 (the fenced python block that defines rrf()), with the
    library name replaced by this tree's libfusion_hip.so."""
    import os
    import re
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = [b for b in blocks if "def rrf(" in b]
    assert len(code) == 1, "INTEGRATION.md must hold exactly one binding snippet that defines rrf()"
    from fusion_amd import _lib
    return code[0].replace('"libfusion_hip.so"', repr(_lib.LIB_PATH))


def quantile_table(pool, P):
    """np.quantile(pool, np.linspace(0, 1, P)) (linear interpolation) by one sort: numpy's own takes ~40 s for 38,000 quantiles
    (one partition step per quantile)."""
    s = np.sort(np.asarray(pool, dtype=np.float64).ravel())
    pos = np.linspace(0.0, len(s) - 1.0, P)
    lo = np.floor(pos).astype(np.int64)
    hi = np.minimum(lo + 1, len(s) - 1)
    return s[lo] + (s[hi] - s[lo]) * (pos - lo)
