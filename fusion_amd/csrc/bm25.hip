// bm25.hip -- A1: BM25 scoring on device (reference: src/retrievers/bm25.py:149-156, search loop :100-106).
//
//   score(q,d) = sum over query terms, IN QUERY ORDER, of idf*tf*(k1+1) / (tf + k1*(1 - b + b*|d|/avgdl))
// in float64, exactly the reference's expression order (no FMA contraction: -ffp-contract=off), so the
// scores and therefore the ranks are bit-identical to the Python loop.  Documents that lack a term add
// +-0.0 in the reference, i.e. nothing.  The postings of one term touch distinct documents (parallel, no
// atomics); terms are applied one after the other (barrier) to keep the reference's sum order.
#include "common.h"

namespace fz {

struct Bm25Args {
    const int64_t* toff; const int32_t* pdoc; const int32_t* ptf; const double* idf; const int32_t* doc_len;
    const double* doc_norm;   // nullable: k1*(1-b+b*|d|/avgdl) per document (fz_bm25_doc_norms_f64), same bits as inline
    const int64_t* slice_off; // nullable: [V][NS + 1] first posting of term t whose document is >= s * BM25_SLICE (fz_bm25_slice_offsets):
                              //   without it every workgroup finds its posting sub-ranges by ~30 dependent loads per term
    double avgdl, k1, b;
    const int64_t* qoff; const int32_t* qterms;
    int N; double* scores; int lds;
    float* scores32; int lds32;   // nullable: the same scores rounded to float32 (torch.tensor(scores, dtype=float32), hybrid.py:255), written
                                  //   from the same LDS accumulators -- the separate plane-sized conversion pass (fz_f64_to_f32) goes away
    int tfidf;                    // 1: TFIDF.score (bm25.py:108-115): score += tf * idf -- no length norm, no k1 / b (fz_tfidf_scores_f64)
    const double* pval;           // non-null: the posting's whole term idf * (tf (k1 + 1)) / (tf + norm_d), tabulated per index and (k1, b) by
                                  //   fz_bm25_posting_values_f64 -- the walk only adds (fz_bm25_scores_pv_f64_f32)
};

// LDS-resident accumulators AND length norms: one workgroup = (query, slice of BM25_SLICE documents).  The random read-modify-writes of
// the posting walk hit LDS (ds_read_b64 / ds_write_b64) instead of HBM/L2, and so does the per-posting read of the document's length
// norm k1 (1 - b + b |d| / avgdl) -- round 5: as a gather from the [N] table in L2 it was one 64-byte sector per posting, ~110 k of them
// per query (the frequent terms of a Zipf vocabulary list most of the corpus); the slice's norms now come in once per workgroup,
// coalesced.  The slice is written out once, coalesced.  Postings of a term are sorted by document, so the slice's sub-range is found by
// two block-uniform binary searches (or read from the per-index table).
constexpr int BM25_SLICE = 7168;    // 7168 fp64 accumulators + 7168 fp64 norms = 112 KiB of the CU's 160 KiB LDS
constexpr int BM25_GRAIN = 3584;    // granularity of the per-index posting-offset table (fz_bm25_slice_offsets): a workgroup's slice is 1 or 2 of these
constexpr int BM25_PV_GRAINS = 1, BM25_PV_THREADS = 512;   // the table-driven walk's workgroup (see bm25_kernel)

__device__ __forceinline__ int64_t lower_bound_doc(const int32_t* __restrict__ pdoc, int64_t lo, int64_t hi, int doc) {
    while (lo < hi) {   // first e in [lo, hi) with pdoc[e] >= doc  (uniform: scalar loads)
        const int64_t mid = (lo + hi) >> 1;
        if (pdoc[mid] < doc) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int BM25_TERMS = 256;     // query terms whose posting ranges are resolved per batch

// MODE: 0 = BM25's expression per posting; 1 = TFIDF: score += tf * idf (bm25.py:114); 2 = the posting's term comes from a table (pval):
// every posting of the index has ONE value for a given (k1, b) -- idf, tf and the document's length norm are all the index's -- so the
// float64 division (a dozen instructions at half rate: most of this kernel's time) is done once per index, like the idf table, not once
// per (query, posting); the walk adds the same bits in the same order.
enum { BM25_EXPR = 0, BM25_TFIDF = 1, BM25_PVAL = 2 };
template <int MODE>
__global__ __launch_bounds__(1024) void bm25_kernel(Bm25Args a) {
    constexpr bool TFIDF = MODE == BM25_TFIDF, PVAL = MODE == BM25_PVAL;
    // PVAL needs no length norms in LDS: its workgroup takes TWO slices' worth of documents (14,336 accumulators = 112 KiB) -- half the
    // workgroups, and a term's posting sub-range is twice as long against the same per-term barrier
    // PVAL needs no length norms in LDS and little else: 3,584 accumulators (28 KiB) and 512 threads per workgroup, four workgroups per CU
    // -- a posting walk is a chain of (load, LDS add) round trips with a barrier per query term, and independent workgroups fill each
    // other's waits.  Measured per 1024 x 27,942 (bench step): 7,168 documents x 1024 threads 0.165 ms, 14,336 x 1024 0.198, 3,584 x 1024
    // 0.210, 3,584 x 256 0.162, 3,584 x 512 0.146 (the per-posting expression: 0.326).
    constexpr int SL = PVAL ? BM25_PV_GRAINS : 2, BM25_SLICE = fz::BM25_GRAIN * SL;      // table grains per workgroup slice
    extern __shared__ __attribute__((aligned(16))) double acc[];          // [BM25_SLICE] accumulators | [BM25_SLICE] length norms
    double* nrm = acc + BM25_SLICE;
    __shared__ int64_t s_e0[BM25_TERMS], s_e1[BM25_TERMS];
    __shared__ double s_w[BM25_TERMS];
    const int q = blockIdx.y;
    const int d0 = blockIdx.x * BM25_SLICE;
    const int d1 = (d0 + BM25_SLICE < a.N) ? d0 + BM25_SLICE : a.N;
    const int n = d1 - d0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        acc[j] = 0.0;
        if constexpr (!TFIDF && !PVAL)
            nrm[j] = a.doc_norm ? a.doc_norm[d0 + j] : a.k1 * (1.0 - a.b + a.b * (double)a.doc_len[d0 + j] / a.avgdl);   // the sub-expression of bm25.py:154, once per document
    }
    const int64_t p0 = a.qoff[q], p1 = a.qoff[q + 1];
    for (int64_t pb = p0; pb < p1; pb += BM25_TERMS) {
        const int nt = (int)((p1 - pb < BM25_TERMS) ? p1 - pb : BM25_TERMS);
        __syncthreads();   // acc zeroed / previous batch's table no longer read
        // all posting sub-ranges of this slice at once: one thread per term walks its own binary searches
        // (~30 dependent loads each -- done one term after the other they were the whole kernel time)
        if ((int)threadIdx.x < nt) {
            const int t = a.qterms[pb + threadIdx.x];
            int64_t e0 = 0, e1 = 0; double w = 0.0;
            if (t >= 0) {     // out of vocabulary: idf 0, contributes nothing
                if constexpr (!PVAL) w = a.idf[t];
                if (a.slice_off) {   // per-index table: two loads instead of two binary searches
                    const int ns = (a.N + fz::BM25_GRAIN - 1) / fz::BM25_GRAIN;          // grains the table was made for
                    const int s0 = (int)blockIdx.x * SL, s1 = s0 + SL < ns ? s0 + SL : ns;
                    const int64_t* so = a.slice_off + (size_t)t * (ns + 1);
                    e0 = so[s0]; e1 = so[s1];
                } else {
                    e0 = lower_bound_doc(a.pdoc, a.toff[t], a.toff[t + 1], d0);
                    e1 = lower_bound_doc(a.pdoc, e0, a.toff[t + 1], d1);
                }
            }
            s_e0[threadIdx.x] = e0; s_e1[threadIdx.x] = e1; s_w[threadIdx.x] = w;
        }
        __syncthreads();
        for (int k = 0; k < nt; ++k) {          // terms in QUERY ORDER: the reference's addition order
            const int64_t e0 = s_e0[k], e1 = s_e1[k];
            const double w = s_w[k];
            if (e1 <= e0) continue;             // block-uniform
            // the walk is latency-bound (posting -> per-document gather -> accumulate): U postings per lane in flight (U = 8 and a
            // three-stage pipeline across steps and terms -- fetch i + 2, gather i + 1, accumulate i -- were both slower: 0.375 / 0.374
            // vs 0.339 ms per 1024 queries)
            constexpr int U = 4;     // (the table-driven walk: 2 measured the same, 8 7 % slower; an LDS float64 atomic add instead of read + add + write 4 % faster, not taken: its denormal handling is not the VALU's)
            for (int64_t eb = e0; eb < e1; eb += (int64_t)blockDim.x * U) {
                int doc[U]; double tf[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t e = eb + (int64_t)u * blockDim.x + threadIdx.x;
                    const bool ok = e < e1;
                    doc[u] = ok ? a.pdoc[e] : -1;
                    if constexpr (PVAL) tf[u] = ok ? a.pval[e] : 0.0;         // (the posting's whole term)
                    else tf[u] = ok ? (double)a.ptf[e] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (doc[u] >= 0) {
                        double term;
                        if constexpr (PVAL) term = tf[u];
                        else if constexpr (TFIDF) term = tf[u] * w;        // bm25.py:114: score += tf * idf
                        else {
                            const double num = w * (tf[u] * (a.k1 + 1.0));
                            const double den = tf[u] + nrm[doc[u] - d0];
                            term = num / den;
                        }
                        acc[doc[u] - d0] = acc[doc[u] - d0] + term;        // postings of one term hit distinct documents: no race
                    }
                }
            }
            __syncthreads();  // the next term may touch the same documents
        }
    }
    __syncthreads();
    double* __restrict__ row = a.scores + (size_t)q * a.lds + d0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) row[j] = acc[j];
    if (a.scores32) {
        float* __restrict__ row32 = a.scores32 + (size_t)q * a.lds32 + d0;
        for (int j = threadIdx.x; j < n; j += blockDim.x) row32[j] = (float)acc[j];
    }
}

__global__ void bm25_slice_offsets_kernel(const int64_t* __restrict__ toff, const int32_t* __restrict__ pdoc, int V, int NS, int64_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)V * (NS + 1)) return;
    const int t = (int)(i / (NS + 1)), s_ = (int)(i % (NS + 1));
    out[i] = s_ == NS ? toff[t + 1] : lower_bound_doc(pdoc, toff[t], toff[t + 1], s_ * BM25_GRAIN);
}

__global__ void bm25_doc_norms_kernel(const int32_t* __restrict__ doc_len, int N, double avgdl, double k1, double b, double* __restrict__ out) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x)
        out[j] = k1 * (1.0 - b + b * (double)doc_len[j] / avgdl);   // the sub-expression of bm25.py:154, once per document
}

}  // namespace fz

using namespace fz;

static int bm25_launch(const Bm25Args& a, int Q, void* stream);

extern "C" int fz_bm25_doc_norms_f64(const int32_t* doc_len, int N, double avgdl, double k1, double b, double* out, void* stream) {
    if (N < 0) return FZ_ERR_ARG;
    if (N == 0) return FZ_OK;                      // empty tensors carry null pointers
    if (!doc_len || !out) return FZ_ERR_ARG;
    bm25_doc_norms_kernel<<<(N + 255) / 256, 256, 0, as_stream(stream)>>>(doc_len, N, avgdl, k1, b, out);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_bm25_slice_docs(void) { return BM25_GRAIN; }

extern "C" int fz_bm25_slice_offsets(const int64_t* toff, const int32_t* pdoc, int V, int N, int64_t* out, void* stream) {
    if (V < 0 || N < 0) return FZ_ERR_ARG;
    if (V == 0) return FZ_OK;
    if (!toff || !out) return FZ_ERR_ARG;
    const int NS = N > 0 ? (N + BM25_GRAIN - 1) / BM25_GRAIN : 1;
    const long total = (long)V * (NS + 1);
    bm25_slice_offsets_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(toff, pdoc, V, NS, out);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_bm25_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf,
                                  const int32_t* doc_len, const double* doc_norm, const int64_t* slice_off, double avgdl, double k1, double b,
                                  const int64_t* qoff, const int32_t* qterms, int Q, int N, double* scores, int lds, void* stream) {
    return fz_bm25_scores_f64_f32(toff, pdoc, ptf, idf, doc_len, doc_norm, slice_off, avgdl, k1, b, qoff, qterms, Q, N, scores, lds, nullptr, 0, stream);
}

extern "C" int fz_bm25_scores_f64_f32(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf,
                                      const int32_t* doc_len, const double* doc_norm, const int64_t* slice_off, double avgdl, double k1, double b,
                                      const int64_t* qoff, const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32,
                                      int lds32, void* stream) {
    if (Q < 0 || N < 0 || lds < N || (scores32 && lds32 < N)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;           // empty tensors carry null pointers
    if (!toff || !idf || !doc_len || !qoff || !scores) return FZ_ERR_ARG;
    return bm25_launch(Bm25Args{toff, pdoc, ptf, idf, doc_len, doc_norm, slice_off, avgdl, k1, b, qoff, qterms, N, scores, lds, scores32, lds32, 0, nullptr}, Q, stream);
}

// TFIDF.score (bm25.py:108-115): score(q, d) = sum over the query's terms, in query order, of tf(t, d) * idf(t) in float64 -- the same
// posting walk without the length norm (the idf table is the caller's: TFIDF's is log10((N + 1) / (df + 1)), bm25.py:86-88).
extern "C" int fz_tfidf_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const int64_t* slice_off,
                                   const int64_t* qoff, const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32, int lds32,
                                   void* stream) {
    if (Q < 0 || N < 0 || lds < N || (scores32 && lds32 < N)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    if (!toff || !idf || !qoff || !scores) return FZ_ERR_ARG;
    return bm25_launch(Bm25Args{toff, pdoc, ptf, idf, nullptr, nullptr, slice_off, 1.0, 0.0, 0.0, qoff, qterms, N, scores, lds, scores32, lds32, 1, nullptr}, Q, stream);
}

// The posting-value table: pval[e] = idf_t * (tf_e * (k1 + 1)) / (tf_e + norm_d) for posting e of term t on document d, in float64 in the
// reference's expression order (bm25.py:154) -- the very value bm25_kernel<BM25_EXPR> forms per (query, posting).  One thread per posting;
// its term by binary search in toff.
__global__ void bm25_posting_values_kernel(const int64_t* __restrict__ toff, const int32_t* __restrict__ pdoc, const int32_t* __restrict__ ptf,
                                           const double* __restrict__ idf, const double* __restrict__ doc_norm, int V, int64_t nnz, double k1,
                                           double* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = V;                 // last t with toff[t] <= e
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (toff[mid] <= e) lo = mid; else hi = mid; }
        const double w = idf[lo], tf = (double)ptf[e];
        const double num = w * (tf * (k1 + 1.0));
        const double den = tf + doc_norm[pdoc[e]];
        out[e] = num / den;
    }
}

extern "C" int fz_bm25_posting_values_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf, const double* doc_norm,
                                          int V, int64_t nnz, double k1, double* out, void* stream) {
    if (V < 0 || nnz < 0) return FZ_ERR_ARG;
    if (V == 0 || nnz == 0) return FZ_OK;
    if (!toff || !pdoc || !ptf || !idf || !doc_norm || !out) return FZ_ERR_ARG;
    const int64_t blocks = (nnz + 255) / 256;
    bm25_posting_values_kernel<<<(unsigned)(blocks < 65535 * 16 ? blocks : 65535 * 16), 256, 0, as_stream(stream)>>>(toff, pdoc, ptf, idf, doc_norm, V, nnz, k1, out);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// BM25 scores from the posting-value table: the same float64 plane (and float32 rounding) as fz_bm25_scores_f64_f32 with the k1 / b the
// table was made for, bit for bit -- the walk adds the tabulated terms in query order.
extern "C" int fz_bm25_scores_pv_f64_f32(const int64_t* toff, const int32_t* pdoc, const double* pval, const int64_t* slice_off, const int64_t* qoff,
                                         const int32_t* qterms, int Q, int N, double* scores, int lds, float* scores32, int lds32, void* stream) {
    if (Q < 0 || N < 0 || lds < N || (scores32 && lds32 < N)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    if (!toff || !pval || !qoff || !scores) return FZ_ERR_ARG;
    return bm25_launch(Bm25Args{toff, pdoc, nullptr, nullptr, nullptr, nullptr, slice_off, 1.0, 0.0, 0.0, qoff, qterms, N, scores, lds, scores32, lds32, 0, pval}, Q, stream);
}

static int bm25_launch(const Bm25Args& a, int Q, void* stream) {
    const int N = a.N;
    constexpr size_t lds_bytes = 2 * (size_t)BM25_SLICE * sizeof(double);
    static unsigned long long lds_set[3] = {0ull, 0ull, 0ull};
    dim3 grid((unsigned)((N + BM25_SLICE - 1) / BM25_SLICE), (unsigned)Q);
    if (a.pval) {
        constexpr size_t lds_pv = (size_t)BM25_PV_GRAINS * BM25_GRAIN * sizeof(double);   // accumulators only
        grid.x = (unsigned)((N + BM25_PV_GRAINS * BM25_GRAIN - 1) / (BM25_PV_GRAINS * BM25_GRAIN));
        if (int rc = raise_lds_limit((const void*)bm25_kernel<BM25_PVAL>, lds_pv, lds_set[2])) return rc;
        bm25_kernel<BM25_PVAL><<<grid, BM25_PV_THREADS, lds_pv, as_stream(stream)>>>(a);
    } else if (a.tfidf) {
        if (int rc = raise_lds_limit((const void*)bm25_kernel<BM25_TFIDF>, lds_bytes, lds_set[1])) return rc;
        bm25_kernel<BM25_TFIDF><<<grid, 1024, lds_bytes, as_stream(stream)>>>(a);
    } else {
        if (int rc = raise_lds_limit((const void*)bm25_kernel<BM25_EXPR>, lds_bytes, lds_set[0])) return rc;
        bm25_kernel<BM25_EXPR><<<grid, 1024, lds_bytes, as_stream(stream)>>>(a);
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
