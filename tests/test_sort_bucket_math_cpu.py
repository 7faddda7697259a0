"""The integer arithmetic of the row sort's bucket ranking (fusion_amd/csrc/sort.hip, `bucket_rank`), restated in numpy and checked for
the properties the kernel's correctness argument rests on -- no GPU needed:

  * (fine, rem) is a monotone non-decreasing function of the sort word (so bucket order is key order and the in-bucket rank by
    (rem, position) is a stable order by a coarsening of the word -- which the neighbour check and, failing that, the digit passes finish);
  * fine stays below the 16,384 buckets whatever the sample says;
  * distinct words in a coarse bucket that holds its share of the keys keep distinct (fine, rem) (the mapping is exact there);
  * the gap between the smallest positive and the smallest negative magnitude is cut out and nothing else moves.

Reference semantics of the sort itself: Python's stable sorted(..., reverse=True) (bm25.py:104, hybrid.py:306)."""
import numpy as np
import pytest

FINE, COARSE = 16384, 1024


def desc_key_f32(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.int64)
    nan = (u & 0x7fffffff) > 0x7f800000
    u = np.where(u == 0x80000000, 0, u)
    asc = np.where(u & 0x80000000, (~u) & 0xffffffff, u | 0x80000000)
    return np.where(nan, 0, (~asc) & 0xffffffff).astype(np.int64)


class Mapping:
    """What `bucket_rank` derives from a row's sort words before it touches a key: floor, gap, coarse shift, sub-bucket table."""

    def __init__(self, w, E=28, floor_shift=23):
        w = np.asarray(w, dtype=np.int64)
        self.m = len(w)
        w_lo, w_hi = int(w.min()), int(w.max())
        magp = 0x7fffffff - w_lo if w_lo < 0x80000000 else 0
        magn = w_hi - 0x80000000 if w_hi >= 0x80000000 else 0
        magmax = max(magp, magn)
        floor = 24 << floor_shift
        self.eps = magmax - floor if magmax > floor else 0
        self.zp, self.zn = 0x7fffffff - self.eps, 0x80000000 + self.eps
        self.tmin = int(self._t0(np.array([w_lo]))[0])
        self.range = int(self._t0(np.array([w_hi]))[0]) - self.tmin
        nbits = self.range.bit_length()
        self.s1 = max(0, nbits - 10)
        # the sample: items 0, 7, 14, 21 of every thread (wave-striped slots)
        pos = np.arange(self.m)
        item = (pos % (E * 64)) // 64
        samp = (item % ((E + 3) // 4) == 0) & (item // ((E + 3) // 4) < 4)
        c = (self.t(w) >> self.s1).astype(np.int64)
        cnt = np.bincount(c[samp], minlength=COARSE)
        kf = np.float32(FINE - COARSE - 16) / np.float32(samp.sum())
        self.nsub = 1 + (cnt.astype(np.float32) * kf).astype(np.int64)
        self.base = np.concatenate([[0], np.cumsum(self.nsub)[:-1]])

    def _t0(self, k):
        neg = k >= 0x80000000
        return np.where(neg, np.maximum(k, self.zn) - 2 * self.eps, np.minimum(k, self.zp))

    def t(self, k):
        return (self._t0(np.asarray(k, dtype=np.int64)) - self.tmin).astype(np.int64)

    def fine_rem(self, k):
        tt = self.t(k)
        c = tt >> self.s1
        f24 = (((tt << (32 - self.s1)) & 0xffffffff) >> 8) if self.s1 else np.zeros_like(tt)
        prod = f24 * self.nsub[c]
        return self.base[c] + (prod >> 24), (prod >> 8) & 0xffff


ROWS = {
    "cosine": lambda g, n: g.normal(0, 768 ** -0.5, n),
    "uniform": lambda g, n: g.uniform(-1, 1, n),
    "positive": lambda g, n: 20 + 4 * g.normal(0, 1, n),
    "lognormal": lambda g, n: np.exp(g.normal(0, 1.5, n)),
    "negative": lambda g, n: -np.exp(g.normal(0, 1.0, n)),
    "tiny range": lambda g, n: 1.0 + 1e-4 * g.normal(0, 1, n),
    "with zeros": lambda g, n: np.where(g.random(n) < 0.1, 0.0, g.normal(0, 1, n)),
}


@pytest.mark.parametrize("kind", list(ROWS))
@pytest.mark.parametrize("n", [4096, 16384, 27942])
def test_fine_rem_is_monotone_and_in_range(kind, n):
    g = np.random.default_rng(n + len(kind))
    x = ROWS[kind](g, n).astype(np.float32)
    w = desc_key_f32(x)
    mp = Mapping(w, E=28 if n > 16384 else 16)
    assert mp.range > 0 and 0 <= mp.s1 <= 22
    assert int(mp.nsub.sum()) <= FINE
    ws = np.sort(w)
    fine, rem = mp.fine_rem(ws)
    assert fine.min() >= 0 and fine.max() < FINE
    key = fine * 65536 + rem
    assert np.all(np.diff(key) >= 0), "(fine, rem) must not decrease along the sort words"
    t = mp.t(ws)
    assert t.min() == 0 and t.max() == mp.range and np.all(np.diff(t) >= 0)


def test_the_gap_is_cut_out_and_the_floor_clamps():
    g = np.random.default_rng(1)
    x = g.normal(0, 0.04, 20000).astype(np.float32)
    x[:3] = [1e-30, -1e-30, 0.0]                                      # far below the floor: clamped, one bucket value each side
    w = desc_key_f32(x)
    mp = Mapping(w)
    assert mp.eps > 0 and mp.range < 2 * 25 * (1 << 23)               # at most 2 x 24 binades (+ the top one) of sort words
    tp, tn, tz = mp.t(w[:1])[0], mp.t(w[1:2])[0], mp.t(w[2:3])[0]
    assert tp == tz == mp.zp - mp.tmin and tn == tp + 1               # the two floors are neighbours: the unused exponents are gone
    big = np.abs(x) > np.abs(x).max() * 2.0 ** -20                    # well above the floor: sort words untouched relative to each other
    ws = np.sort(w[big])
    d_w, d_t = np.diff(ws), np.diff(mp.t(ws))
    same_side = (ws[1:] < 0x80000000) == (ws[:-1] < 0x80000000)
    assert np.array_equal(d_w[same_side], d_t[same_side])


def test_exact_where_a_coarse_bucket_holds_its_share():
    """nsub >= 2^(s1 - 16) -> two distinct words of that coarse bucket never share (fine, rem)."""
    g = np.random.default_rng(2)
    x = g.normal(0, 768 ** -0.5, 27942).astype(np.float32)
    w = np.unique(desc_key_f32(x))
    mp = Mapping(desc_key_f32(x))
    c = mp.t(w) >> mp.s1
    dense = mp.nsub[c] >= (1 << max(0, mp.s1 - 16))
    fine, rem = mp.fine_rem(w[dense])
    key = fine * 65536 + rem
    assert len(np.unique(key)) == int(dense.sum())
    # neighbours one pattern apart inside such a bucket, the case the GPU test constructs in a THIN bucket
    w0 = w[dense][len(w[dense]) // 2]
    (f0, r0), (f1, r1) = mp.fine_rem(np.array([w0])), mp.fine_rem(np.array([w0 + 1]))
    assert (f0[0], r0[0]) != (f1[0], r1[0]) or (mp.t(np.array([w0 + 1]))[0] >> mp.s1) != (mp.t(np.array([w0]))[0] >> mp.s1)
