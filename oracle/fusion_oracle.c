/*
 * fusion_oracle.c -- CPU restatement of the reference's encode->score->fuse arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under fusion_amd/ may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
 * there only as the checker / the timed CPU baseline, never as the product path.
 *
 * Parity status: PINNED for fuse (rrf/bcf/nsf x every normalisation), BM25, Metrics,
 * cosine / dot-product scoring, top-k search and SPLADE pooling against golden vectors
 * produced by the reference's own in-tree classes (oracle/gen_golden.py -> tests/golden/;
 * scoring and search through splade/base.py:186-251, the in-tree mirror of the
 * sentence-transformers calls at hybrid.py:103). UNPINNED for MaxSim only: its arithmetic
 * lives in colbert-ai@main, absent from /root/reference and from the image; the published
 * late-interaction formula is restated and checked against analytic known answers and an
 * fp32 torch restatement.
 *
 * Data model (shared with include/fusion_hip.h): the reference's
 *   RankedLists = list[Q] of list[<=N] of {'corpus_id','score'}  (hybrid.py:66-75)
 * becomes, per system s, dense planes indexed by corpus POSITION j in [0,N):
 *   score[s][q][j] fp32, rank[s][q][j] int32 (0-based list position, -1 = absent from
 *   the list), len[s][q] = list length.  "idx" in hybrid.py:249,252 == rank.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FZO_OK 0
#define FZO_ERR_ARG (-1)

enum { NORM_NONE = 0, NORM_MINMAX = 1, NORM_ZSCORE = 2, NORM_ARCTAN = 3, NORM_PERCENTILE = 4, NORM_NCE = 5 };
enum { RANK_RRF = 0, RANK_BCF = 1 };

int fzo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void fzo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------
 * Scoring: cosine similarity (hybrid.py:103 -> sentence_transformers.util.cos_sim;
 * in-tree mirror splade/base.py:186-197: F.normalize(q), F.normalize(d), torch.mm).
 * F.normalize: x / max(||x||_2, 1e-12).  Norm accumulated in double, rounded to fp32.
 * ------------------------------------------------------------------------------------ */
int fzo_normalize_rows_f32(const float* X, int rows, int d, int ldx, float* Y, int ldy) {
    if (!X || !Y || rows < 0 || d <= 0 || ldx < d || ldy < d) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* x = X + (size_t)r * ldx;
        float* y = Y + (size_t)r * ldy;
        double ss = 0.0;
        for (int k = 0; k < d; ++k) ss += (double)x[k] * (double)x[k];
        float nrm = (float)sqrt(ss);
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (int k = 0; k < d; ++k) y[k] = x[k] / nrm;
    }
    return FZO_OK;
}

/* scores[q][j] = sum_k Qn[q][k]*Dn[j][k]  (torch.mm of splade/base.py:197), double accumulate */
int fzo_dot_scores_f32(const float* Qn, const float* Dn, int Q, int N, int d, float* scores, int lds) {
    if (!Qn || !Dn || !scores || Q < 0 || N < 0 || d <= 0 || lds < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        const float* a = Qn + (size_t)q * d;
        for (int j = 0; j < N; ++j) {
            const float* b = Dn + (size_t)j * d;
            double acc = 0.0;
            for (int k = 0; k < d; ++k) acc += (double)a[k] * (double)b[k];
            scores[(size_t)q * lds + j] = (float)acc;
        }
    }
    return FZO_OK;
}

/* fp32-accumulate variant: the SAME arithmetic order as a k-ordered fmaf chain
 * (what v_mfma_f32_32x32x2_f32 computes) -- used as the timed CPU baseline since it
 * vectorises, and as a tighter parity target for K1. */
int fzo_dot_scores_f32_fma(const float* Qn, const float* Dn, int Q, int N, int d, float* scores, int lds) {
    if (!Qn || !Dn || !scores || Q < 0 || N < 0 || d <= 0 || lds < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        const float* a = Qn + (size_t)q * d;
        for (int j = 0; j < N; ++j) {
            const float* b = Dn + (size_t)j * d;
            float acc = 0.0f;
            for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
            scores[(size_t)q * lds + j] = acc;
        }
    }
    return FZO_OK;
}

/* ColBERT late interaction (hybrid.py:108-137 -> colbert-ai; exact form):
 *   s(q,d) = sum_{i<Lq} max_{j<Ld[d]} <Qtok[q][i], Dtok[d][j]>
 * Qtok [Q][Lq][dim] fp16-valued floats; Dtok packed ragged: doc d owns rows
 * [Doff[d], Doff[d+1]) of Dtok [sumL][dim].  Empty doc -> score 0 (sum of nothing).
 * Tokens are passed as fp32 arrays holding fp16-representable values. */
int fzo_maxsim_f32(const float* Qtok, const float* Dtok, const int64_t* Doff, int Q, int Lq, int N, int dim,
                   float* scores, int lds) {
    if (!Qtok || !Dtok || !Doff || !scores || Q < 0 || N < 0 || Lq <= 0 || dim <= 0 || lds < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(dynamic, 16)
    for (int dd = 0; dd < N; ++dd) {
        int64_t b = Doff[dd], e = Doff[dd + 1];
        for (int q = 0; q < Q; ++q) {
            double total = 0.0;
            if (e > b) {
                for (int i = 0; i < Lq; ++i) {
                    const float* qv = Qtok + ((size_t)q * Lq + i) * dim;
                    float best = -INFINITY;
                    for (int64_t j = b; j < e; ++j) {
                        const float* dv = Dtok + (size_t)j * dim;
                        float acc = 0.0f;
                        for (int k = 0; k < dim; ++k) acc += qv[k] * dv[k];
                        if (acc > best) best = acc;
                    }
                    total += (double)best;
                }
            }
            scores[(size_t)q * lds + dd] = (float)total;
        }
    }
    return FZO_OK;
}

/* ------------------------------------------------------------------------------------
 * Ordering.  Python: sorted(items, key=score, reverse=True) is STABLE (bm25.py:104,
 * hybrid.py:306): ties keep the incoming order.  Restated as a total order on
 * (key desc, incoming position asc).  -0.0 == +0.0 as in Python.  NaN (only reachable
 * through z-score of a 1-element list, hybrid.py:261-264) is implementation-defined in
 * Python; here NaN sorts FIRST (documented in DESIGN.md).
 * ------------------------------------------------------------------------------------ */
typedef struct { double key; int32_t pos; int32_t payload; } fzo_item;

static int fzo_cmp_desc(const void* pa, const void* pb) {
    const fzo_item* a = (const fzo_item*)pa;
    const fzo_item* b = (const fzo_item*)pb;
    int an = isnan(a->key), bn = isnan(b->key);
    if (an != bn) return an ? -1 : 1;
    if (!an) {
        if (a->key > b->key) return -1;
        if (a->key < b->key) return 1;
    }
    return (a->pos > b->pos) - (a->pos < b->pos);
}

/* Sort each row descending.  keys: fp32 (key_bits=32) or fp64 (key_bits=64), row stride ld
 * elements.  init_order (nullable, [rows][ld]): incoming sequence = keys gathered through
 * it (element r of the sequence is keys[row][init_order[row][r]], payload = that index);
 * NULL -> identity (ties -> ascending corpus position).  row_len (nullable): elements per
 * row (<= n).  Outputs (each nullable): order[row][r] = payload at output rank r;
 * sorted_keys[row][r]; rank[row][payload] = r (inverse permutation; entries not in the
 * sequence are left untouched -- caller pre-fills with -1). */
int fzo_sort_rows_desc(const void* keys, int key_bits, const int32_t* init_order, const int32_t* row_len,
                       int rows, int n, int ld, int32_t* order, void* sorted_keys, int32_t* rank) {
    if (!keys || (key_bits != 32 && key_bits != 64) || rows < 0 || n < 0 || ld < n) return FZO_ERR_ARG;
    int err = 0;
#pragma omp parallel
    {
        fzo_item* buf = (fzo_item*)malloc(sizeof(fzo_item) * (size_t)(n > 0 ? n : 1));
        if (!buf) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < rows; ++r) {
            if (!buf) continue;
            int m = row_len ? row_len[r] : n;
            if (m < 0) m = 0;
            if (m > n) m = n;
            size_t base = (size_t)r * ld;
            for (int i = 0; i < m; ++i) {
                int32_t src = init_order ? init_order[base + i] : i;
                buf[i].key = key_bits == 32 ? (double)((const float*)keys)[base + src] : ((const double*)keys)[base + src];
                buf[i].pos = i;
                buf[i].payload = src;
            }
            qsort(buf, (size_t)m, sizeof(fzo_item), fzo_cmp_desc);
            for (int i = 0; i < m; ++i) {
                if (order) order[base + i] = buf[i].payload;
                if (sorted_keys) {
                    if (key_bits == 32) ((float*)sorted_keys)[base + i] = (float)buf[i].key;
                    else ((double*)sorted_keys)[base + i] = buf[i].key;
                }
                if (rank) rank[base + buf[i].payload] = i;
            }
        }
        free(buf);
    }
    return err ? FZO_ERR_ARG : FZO_OK;
}

/* ------------------------------------------------------------------------------------
 * Rank-based fusion (hybrid.py:206-211,248-252,301-304):
 *   rrf: 1/(60+idx+1);  bcf: (n-idx+1)/n  [sic: precedence, hybrid.py:249] with n=len(list)
 *   agg = 0.0; agg += contribution, in system (dict insertion) order, Python float64.
 * ranks[s]: [Q][ld] int32, -1 absent.  lens[s*Q+q].  fused[q][j] fp64; docs in no list
 * get -inf (they are never emitted: see fzo_insertion_order).
 * ------------------------------------------------------------------------------------ */
int fzo_fuse_rank_f64(const int32_t* const* ranks, const int32_t* lens, int S, int Q, int N, int ld, int method,
                      double* fused) {
    if (!ranks || !lens || !fused || S <= 0 || Q < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        for (int j = 0; j < N; ++j) {
            double acc = 0.0;
            int present = 0;
            for (int s = 0; s < S; ++s) {
                int32_t idx = ranks[s][(size_t)q * ld + j];
                if (idx < 0) continue;
                present = 1;
                if (method == RANK_RRF) acc += 1.0 / (double)(60 + idx + 1);
                else { double n = (double)lens[s * Q + q]; acc += (n - (double)idx + 1.0) / n; }
            }
            fused[(size_t)q * ld + j] = present ? acc : -INFINITY;
        }
    }
    return FZO_OK;
}

/* ------------------------------------------------------------------------------------
 * Score normalisation (hybrid.py:254-280) -- per (system, query) list, fp32 tensor math.
 * stat_a/stat_b per (s,q): min-max -> (min, max);  z-score -> (mean, unbiased std).
 * mean/std: torch.mean/torch.std (hybrid.py:262) restated with double accumulation,
 * rounded to fp32 (torch's own fp32 summation order is not reproducible; the difference
 * is <= 1 ulp of the statistic, inside the 1e-4 contract).  Two-pass std about the fp32 mean?
 * No: torch computes var about the exact (internal) mean; we use the double mean.
 * ------------------------------------------------------------------------------------ */
int fzo_row_stats_f32(const float* scores, const int32_t* rank /*nullable: validity*/, int rows, int N, int ld, int norm,
                      float* stat_a, float* stat_b) {
    if (!scores || !stat_a || !stat_b || rows < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* x = scores + (size_t)r * ld;
        const int32_t* v = rank ? rank + (size_t)r * ld : NULL;
        if (norm == NORM_MINMAX) {
            float mn = INFINITY, mx = -INFINITY;
            int anynan = 0, cnt = 0;
            for (int j = 0; j < N; ++j) {
                if (v && v[j] < 0) continue;
                ++cnt;
                if (x[j] != x[j]) anynan = 1;   /* torch.min / torch.max propagate a NaN (hybrid.py:255-258) */
                if (x[j] < mn) mn = x[j];
                if (x[j] > mx) mx = x[j];
            }
            if (cnt == 0) { mn = 0.f; mx = 0.f; }   /* an empty list: nothing is transformed with these */
            stat_a[r] = anynan ? NAN : mn; stat_b[r] = anynan ? NAN : mx;
        } else if (norm == NORM_ZSCORE) {
            double sum = 0.0; long cnt = 0;
            for (int j = 0; j < N; ++j) { if (v && v[j] < 0) continue; sum += (double)x[j]; ++cnt; }
            double mean = cnt ? sum / (double)cnt : NAN;
            double ss = 0.0;
            for (int j = 0; j < N; ++j) { if (v && v[j] < 0) continue; double dlt = (double)x[j] - mean; ss += dlt * dlt; }
            double var = cnt > 1 ? ss / (double)(cnt - 1) : NAN;  /* unbiased; n==1 -> NaN (KAT-4) */
            stat_a[r] = (float)mean; stat_b[r] = (float)sqrt(var);
        } else { stat_a[r] = 0.f; stat_b[r] = 0.f; }
    }
    return FZO_OK;
}

static double fzo_erfinv(double y) {
    /* Giles' single-precision polynomial as the seed + two Newton steps on erf() -> ~1e-15 */
    if (y <= -1.0) return -INFINITY;
    if (y >= 1.0) return INFINITY;
    double w = -log((1.0 - y) * (1.0 + y)), p;
    if (w < 5.0) {
        w -= 2.5;
        p = 2.81022636e-08; p = 3.43273939e-07 + p * w; p = -3.5233877e-06 + p * w; p = -4.39150654e-06 + p * w;
        p = 0.00021858087 + p * w; p = -0.00125372503 + p * w; p = -0.00417768164 + p * w; p = 0.246640727 + p * w;
        p = 1.50140941 + p * w;
    } else {
        w = sqrt(w) - 3.0;
        p = -0.000200214257; p = 0.000100950558 + p * w; p = 0.00134934322 + p * w; p = -0.00367342844 + p * w;
        p = 0.00573950773 + p * w; p = -0.0076224613 + p * w; p = 0.00943887047 + p * w; p = 1.00167406 + p * w;
        p = 2.83297682 + p * w;
    }
    double x = p * y;
    for (int it = 0; it < 2; ++it) {
        double e = erf(x) - y;
        x -= e / (1.1283791670955126 * exp(-x * x));
    }
    return x;
}

/* transform one fp32 score exactly as hybrid.py:254-280 does on an fp32 tensor */
static inline float fzo_transform(float s, int norm, float a, float b, const float* distr, int P) {
    switch (norm) {
        case NORM_MINMAX: return (a != b) ? (s - a) / (b - a) : 1.0f;               /* :257 */
        case NORM_ZSCORE: return (b != 0.0f) ? (s - a) / b : 0.0f;                  /* :263 (NaN std != 0 -> NaN) */
        case NORM_ARCTAN: return (float)(2.0 / M_PI) * atanf(0.1f * s);             /* :268 */
        case NORM_PERCENTILE:
        case NORM_NCE: {
            /* :272-275  argmin_k |distr_k - s| (first minimum), / P, all fp32 */
            int best = 0; float bd = INFINITY;
            for (int k = 0; k < P; ++k) {
                float dd = fabsf(distr[k] - s);
                if (dd < bd) { bd = dd; best = k; }
            }
            float pr = (float)best / (float)P;
            if (norm == NORM_PERCENTILE) return pr;
            /* :277  Normal(0,1).icdf(pr/100)*21.06+50 ; icdf(p) = erfinv(2p-1)*sqrt(2) */
            float p = pr / 100.0f;
            float z = (float)(fzo_erfinv((double)(2.0f * p - 1.0f)) * 1.4142135623730951);
            return z * 21.06f + 50.0f;
        }
        default: return s;                                                          /* :280 */
    }
}

/* Normalised weighted-sum fusion (hybrid.py:212-214,291,301-304), NumPy-2 semantics:
 * np.float32 * python-float -> fp32 product (weight rounded to fp32 first);
 * defaultdict(float) 0.0 += np.float32 stays fp32.  So, strictly in system order and
 * WITHOUT fused multiply-add:  acc = 0; acc = fl32(acc + fl32(t_s * fl32(w_s))).
 * planes[s]: [Q][ld] fp32; ranks[s] nullable (NULL => every doc present in system s).
 * distr[s] nullable table of P[s] fp32 entries.  fused[q][j]; docs in no list: -inf.
 * stat_a/stat_b (nullable outputs, [S*Q]). */
int fzo_fuse_nsf_f32(const float* const* planes, const int32_t* const* ranks, const float* w, int S, int Q, int N, int ld,
                     int norm, const float* const* distr, const int32_t* P, float* fused, float* stat_a_out, float* stat_b_out) {
    if (!planes || !w || !fused || S <= 0 || Q < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
    if ((norm == NORM_PERCENTILE || norm == NORM_NCE) && (!distr || !P)) return FZO_ERR_ARG;
    float* sa = (float*)malloc(sizeof(float) * (size_t)S * (Q > 0 ? Q : 1));
    float* sb = (float*)malloc(sizeof(float) * (size_t)S * (Q > 0 ? Q : 1));
    if (!sa || !sb) { free(sa); free(sb); return FZO_ERR_ARG; }
    for (int s = 0; s < S; ++s)
        fzo_row_stats_f32(planes[s], ranks ? ranks[s] : NULL, Q, N, ld, norm, sa + (size_t)s * Q, sb + (size_t)s * Q);
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        for (int j = 0; j < N; ++j) {
            volatile float acc = 0.0f;  /* volatile: forbid contraction/reassociation */
            int present = 0;
            for (int s = 0; s < S; ++s) {
                size_t off = (size_t)q * ld + j;
                if (ranks && ranks[s] && ranks[s][off] < 0) continue;
                present = 1;
                float t = fzo_transform(planes[s][off], norm, sa[s * Q + q], sb[s * Q + q], distr ? distr[s] : NULL, P ? P[s] : 0);
                volatile float prod = t * w[s];
                acc = acc + prod;
            }
            fused[(size_t)q * ld + j] = present ? acc : -INFINITY;
        }
    }
    if (stat_a_out) memcpy(stat_a_out, sa, sizeof(float) * (size_t)S * Q);
    if (stat_b_out) memcpy(stat_b_out, sb, sizeof(float) * (size_t)S * Q);
    free(sa); free(sb);
    return FZO_OK;
}

/* nsf with normalisation 'none' (or any unknown transformation string): transform_scores
 * returns the dict unchanged (hybrid.py:280), so scores stay PYTHON FLOATS and
 * weight_scores/aggregate_scores run in float64: acc = 0.0; acc += score*w (hybrid.py:291,304).
 * planes are fp32 (the device layout); exact whenever the incoming scores are
 * fp32-representable, else within 6e-8 relative of the reference. */
int fzo_fuse_none_f64(const float* const* planes, const int32_t* const* ranks, const double* w, int S, int Q, int N, int ld,
                      double* fused) {
    if (!planes || !w || !fused || S <= 0 || Q < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        for (int j = 0; j < N; ++j) {
            volatile double acc = 0.0;
            int present = 0;
            for (int s = 0; s < S; ++s) {
                size_t off = (size_t)q * ld + j;
                if (ranks && ranks[s] && ranks[s][off] < 0) continue;
                present = 1;
                volatile double prod = (double)planes[s][off] * w[s];
                acc = acc + prod;
            }
            fused[(size_t)q * ld + j] = present ? acc : -INFINITY;
        }
    }
    return FZO_OK;
}

/* General weight-and-sum with the reference's NumPy-2 scalar promotion (hybrid.py:291,304), for the cases the two
 * entries above do not cover:
 *   - a transformed score is np.float32; `score * w` is float32 when w is a Python float (weak scalar) or np.float32,
 *     float64 when w is np.float64 -- which is what the weight grid of the tuning loop produces (np.arange,
 *     hybrid.py:405-409): the tuning loop fuses in float64, the equal-weights path (`1/len(results)`, :448) in float32;
 *   - the per-document accumulator starts as Python 0.0 (weak), so it is float32 until the first float64 product is
 *     added FOR THAT DOCUMENT and float64 from then on;
 *   - 'none' keeps the raw Python-float scores: float64 planes, always wide.
 * planes[s] is float32 (plane_f64[s] == 0) or float64; narrow[s] != 0 <=> w[s] is a weak / float32 weight AND the
 * plane holds float32 values.  fl32(fl64(a) + fl64(b)) == fl32(a + b) for floats (double rounding is innocuous
 * at 53 >= 2*24+2 bits), so the narrow steps are written on doubles. */
int fzo_fuse_wsum_f64(const void* const* planes, const int32_t* plane_f64, const int32_t* const* ranks, const double* w,
                      const int32_t* narrow, int S, int Q, int N, int ld, double* fused) {
    if (!planes || !w || !fused || !narrow || !plane_f64 || S <= 0 || Q < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        for (int j = 0; j < N; ++j) {
            volatile double acc = 0.0;
            int present = 0, wide = 0;
            size_t off = (size_t)q * ld + j;
            for (int s = 0; s < S; ++s) {
                if (ranks && ranks[s] && ranks[s][off] < 0) continue;
                present = 1;
                volatile double prod;
                if (narrow[s] && !plane_f64[s]) {
                    volatile float p32 = ((const float*)planes[s])[off] * (float)w[s];
                    prod = (double)p32;
                } else {
                    double v = plane_f64[s] ? ((const double*)planes[s])[off] : (double)((const float*)planes[s])[off];
                    prod = v * w[s];
                    wide = 1;
                }
                acc = acc + prod;
                if (!wide) { volatile float a32 = (float)acc; acc = (double)a32; }
            }
            fused[off] = present ? acc : -INFINITY;
        }
    }
    return FZO_OK;
}

/* First-insertion order of the fused dict (hybrid.py:301-304: defaultdict filled system by
 * system, each in that system's rank order; Python dicts keep first-insertion order, and
 * the final sorted() is stable, so equal fused scores keep this order -- SURVEY KAT-1).
 * orders[s]: [Q][ld] list of corpus positions in rank order (first lens[s*Q+q] valid).
 * ins_order[q][0..U[q]) = corpus positions in first-insertion order. */
int fzo_insertion_order(const int32_t* const* orders, const int32_t* lens, int S, int Q, int N, int ld, int32_t* ins_order,
                        int32_t* U) {
    if (!orders || !lens || !ins_order || !U || S <= 0 || Q < 0 || N < 0 || ld < N) return FZO_ERR_ARG;
#pragma omp parallel
    {
        uint8_t* seen = (uint8_t*)malloc((size_t)(N > 0 ? N : 1));
#pragma omp for schedule(static)
        for (int q = 0; q < Q; ++q) {
            memset(seen, 0, (size_t)N);
            int u = 0;
            for (int s = 0; s < S; ++s) {
                int m = lens[s * Q + q];
                for (int r = 0; r < m; ++r) {
                    int32_t j = orders[s][(size_t)q * ld + r];
                    if (j < 0 || j >= N || seen[j]) continue;
                    seen[j] = 1;
                    ins_order[(size_t)q * ld + u++] = j;
                }
            }
            U[q] = u;
        }
        free(seen);
    }
    return FZO_OK;
}

/* Per-row top-k (the chunked score -> topk -> heap merge of
 * sentence_transformers.py:346-364, restated as: k best by (score desc, global id asc)). */
int fzo_topk_rows_f32(const float* scores, int rows, int n, int ld, int k, int64_t id_base, float* out_scores,
                      int64_t* out_ids) {
    if (!scores || !out_scores || !out_ids || rows < 0 || n < 0 || ld < n || k < 0) return FZO_ERR_ARG;
    int kk = k < n ? k : n;
#pragma omp parallel
    {
        fzo_item* buf = (fzo_item*)malloc(sizeof(fzo_item) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 1)
        for (int r = 0; r < rows; ++r) {
            for (int i = 0; i < n; ++i) { buf[i].key = scores[(size_t)r * ld + i]; buf[i].pos = i; buf[i].payload = i; }
            qsort(buf, (size_t)n, sizeof(fzo_item), fzo_cmp_desc);
            for (int i = 0; i < k; ++i) {
                if (i < kk) { out_scores[(size_t)r * k + i] = (float)buf[i].key; out_ids[(size_t)r * k + i] = id_base + buf[i].payload; }
                else { out_scores[(size_t)r * k + i] = -INFINITY; out_ids[(size_t)r * k + i] = -1; }
            }
        }
        free(buf);
    }
    return FZO_OK;
}

/* Merge G per-shard top-k lists per row into the global top-k, by (score desc, id asc);
 * entries with id < 0 are padding.  in_scores/in_ids: [G][rows][k]. */
typedef struct { float s; int64_t id; } fzo_cand;
static int fzo_cmp_cand(const void* pa, const void* pb) {
    const fzo_cand* a = (const fzo_cand*)pa; const fzo_cand* b = (const fzo_cand*)pb;
    int ap = a->id < 0, bp = b->id < 0;
    if (ap != bp) return ap ? 1 : -1;
    int an = isnan(a->s), bn = isnan(b->s);
    if (an != bn) return an ? -1 : 1;
    if (!an) { if (a->s > b->s) return -1; if (a->s < b->s) return 1; }
    return (a->id > b->id) - (a->id < b->id);
}
int fzo_topk_merge(const float* in_scores, const int64_t* in_ids, int G, int rows, int k, float* out_scores, int64_t* out_ids) {
    if (!in_scores || !in_ids || !out_scores || !out_ids || G <= 0 || rows < 0 || k < 0) return FZO_ERR_ARG;
#pragma omp parallel
    {
        fzo_cand* buf = (fzo_cand*)malloc(sizeof(fzo_cand) * (size_t)G * (k > 0 ? k : 1));
#pragma omp for schedule(static)
        for (int r = 0; r < rows; ++r) {
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < k; ++i) {
                    size_t o = ((size_t)g * rows + r) * k + i;
                    buf[g * k + i].s = in_scores[o]; buf[g * k + i].id = in_ids[o];
                }
            qsort(buf, (size_t)G * k, sizeof(fzo_cand), fzo_cmp_cand);
            for (int i = 0; i < k; ++i) { out_scores[(size_t)r * k + i] = buf[i].s; out_ids[(size_t)r * k + i] = buf[i].id; }
        }
        free(buf);
    }
    return FZO_OK;
}

/* ------------------------------------------------------------------------------------
 * BM25 (bm25.py:129-156, TFIDF base :33-106).  The index (vocab/tf/df) is built by the
 * Python side of the oracle from whitespace tokens; this is the scoring loop:
 *   score(q,d) = sum_{t in q.split()} idf(t) * tf*(k1+1) / (tf + k1*(1-b+b*|d|/avgdl))
 * in Python float64, term by term in query order (repeated terms count twice), starting
 * from 0.0.  Postings CSR by term: term t owns [toff[t], toff[t+1]) of (pdoc, ptf).
 * qterms: CSR by query of term ids (-1 = OOV -> contributes idf 0 * ... = +0.0).
 * ------------------------------------------------------------------------------------ */
int fzo_bm25_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf,
                        const int32_t* doc_len, double avgdl, double k1, double b, const int64_t* qoff,
                        const int32_t* qterms, int Q, int N, double* scores, int lds) {
    if (!toff || !idf || !doc_len || !qoff || !scores || Q < 0 || N < 0 || lds < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < Q; ++q) {
        double* row = scores + (size_t)q * lds;
        for (int j = 0; j < N; ++j) row[j] = 0.0;
        for (int64_t p = qoff[q]; p < qoff[q + 1]; ++p) {
            int32_t t = qterms[p];
            if (t < 0) continue; /* tf=0, idf=0: adds +0.0 */
            double w = idf[t];
            /* docs without the term: idf * 0 / (0 + k1*(...)) = +/-0.0, no change to the sum */
            for (int64_t e = toff[t]; e < toff[t + 1]; ++e) {
                int32_t dj = pdoc[e];
                double tf = (double)ptf[e];
                row[dj] += w * (tf * (k1 + 1.0)) / (tf + k1 * (1.0 - b + b * (double)doc_len[dj] / avgdl));
            }
        }
    }
    return FZO_OK;
}

/* TFIDF.score (bm25.py:108-115): score += tf * idf, term by term in query order from 0.0 (tf an int, idf a float: one rounding
 * per product).  The idf table is the caller's (TFIDF: log10((N + 1) / (df + 1)), bm25.py:86-88). */
int fzo_tfidf_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const int64_t* qoff,
                         const int32_t* qterms, int Q, int N, double* scores, int lds) {
    if (!toff || !idf || !qoff || !scores || Q < 0 || N < 0 || lds < N) return FZO_ERR_ARG;
#pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < Q; ++q) {
        double* row = scores + (size_t)q * lds;
        for (int j = 0; j < N; ++j) row[j] = 0.0;
        for (int64_t p = qoff[q]; p < qoff[q + 1]; ++p) {
            int32_t t = qterms[p];
            if (t < 0) continue; /* tf = 0, idf = 0: adds 0 */
            double w = idf[t];
            for (int64_t e = toff[t]; e < toff[t + 1]; ++e) row[pdoc[e]] += (double)ptf[e] * w;
        }
    }
    return FZO_OK;
}
