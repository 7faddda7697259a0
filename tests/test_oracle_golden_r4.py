"""Round 4: the CPU oracle pinned on percentile-rank / normal-curve-equivalent at the table sizes the reference READS
(hybrid.py:412,451: the `_28k` table, len(corpus) + 1 = 27,943 quantiles per system; :374: the `_10k` table, 10,001) --
fixtures written by oracle/gen_golden.py (gen_pr28k, gen_tune10k) from the reference's own Aggregator.  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_golden import load_case
from test_oracle_golden_r2 import load_tune, nce_defined_rows

PR28K = os.path.join(GOLDEN, "pr28k_seed60_S4_Q1_N27942.npz")
TUNE10K = os.path.join(GOLDEN, "tune10k_seed22_S2_Q4_N257.npz")


def test_pr28k_fixture_is_the_size_the_reference_reads():
    z = np.load(PR28K)
    systems = [str(s) for s in z["systems"]]
    assert z["in_ids"].shape == (4, 1, 27942)
    assert all(z[f"distr_{s}"].shape == (27943,) for s in systems)          # hybrid.py:396: np.linspace(0, 1, N + 1) with N = len(corpus)
    assert list(z["in_len"][:, 0]) == [27942, 27942, 27942, 16765]            # the ColBERT list is cut to 60 %
    d = z["distr_dpr"]
    assert np.all(np.diff(d) >= 0) and np.all(d[5000:5032] == d[5000])        # ascending, with the planted run of duplicates
    for s in systems:                                                         # planted: scores that ARE table entries, and scores outside the table
        t32, sc = z[f"distr_{s}"].astype(np.float32), z["in_scores"][systems.index(s), 0]
        assert np.isin(sc, t32).sum() >= 18 and sc.max() > t32[-1] and sc.min() < t32[0]


@pytest.mark.parametrize("norm,tol", [("percentile-rank", 0.0), ("normal-curve-equivalent", 1e-4)])
def test_oracle_matches_reference_at_28k_tables(oracle, norm, tol):
    """One full LLeQA row, S = 4, 27,943-entry tables: percentile-rank ranked list and scores bit for bit; NCE <= 1e-4
    (libm vs torch's float32 erfinv), -inf where the percentile rank is 0 (hybrid.py:277)."""
    z, systems, lists, weights, distr, Q = load_case(PR28K)
    got = oracle.fuse_lists(lists, method="nsf", normalization=norm, linear_weights=weights, percentile_distributions=distr)
    e_ids, e_sc, n = z[f"out_ids__nsf__{norm}"][0], z[f"out_scores__nsf__{norm}"][0], int(z[f"out_len__nsf__{norm}"][0])
    assert len(got) == 1 and len(got[0]) == n == 27942
    g_ids = np.array([x["corpus_id"] for x in got[0]], dtype=np.int64)
    g_sc = np.array([float(x["score"]) for x in got[0]], dtype=np.float64)
    if tol == 0.0:
        np.testing.assert_array_equal(g_ids, e_ids[:n])
        np.testing.assert_array_equal(g_sc, e_sc[:n].astype(np.float64))
    else:
        assert sorted(g_ids.tolist()) == sorted(e_ids[:n].tolist())
        exp = {int(i): float(s) for i, s in zip(e_ids[:n], e_sc[:n])}
        ref = np.array([exp[int(i)] for i in g_ids])
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(g_sc), fin) and np.array_equal(g_sc[~fin], ref[~fin])
        assert (~fin).sum() >= 4                                           # the planted below-the-table scores: icdf(0) = -inf
        assert np.max(np.abs(g_sc[fin] - ref[fin])) <= tol
        assert np.all(g_sc[:-1] >= g_sc[1:])


@pytest.mark.parametrize("norm", ["percentile-rank", "normal-curve-equivalent"])
def test_oracle_tune_loop_matches_reference_at_10k_tables(oracle, norm):
    z, systems, lists, labels, combos, distr = load_tune(TUNE10K)
    assert all(len(distr[s]) == 10001 for s in systems) and len(combos) == 21
    names = [str(x) for x in z["metric_names"]]
    got = oracle.tune_lists(lists, norm, combos, labels, distr)
    G = np.array([[float(g[k]) for k in names] for g in got])
    rows = nce_defined_rows(z["weights"]) if norm == "normal-curve-equivalent" else slice(None)
    assert np.max(np.abs(G - z[f"metrics__{norm}"])[rows]) <= 1e-12
