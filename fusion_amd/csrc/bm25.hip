// bm25.hip -- A1: BM25 scoring on device (reference: src/retrievers/bm25.py:149-156, search loop :100-106).
//
//   score(q,d) = sum over query terms, IN QUERY ORDER, of idf*tf*(k1+1) / (tf + k1*(1 - b + b*|d|/avgdl))
// in float64, exactly the reference's expression order (no FMA contraction: -ffp-contract=off), so the
// scores and therefore the ranks are bit-identical to the Python loop.  Documents that lack a term add
// +-0.0 in the reference, i.e. nothing.  One workgroup per query: the postings of one term touch distinct
// documents (parallel, no atomics); terms are applied one after the other (barrier) to keep the sum order.
#include "common.h"

namespace fz {

struct Bm25Args {
    const int64_t* toff; const int32_t* pdoc; const int32_t* ptf; const double* idf; const int32_t* doc_len;
    double avgdl, k1, b;
    const int64_t* qoff; const int32_t* qterms;
    int N; double* scores; int lds;
};

__global__ __launch_bounds__(1024) void bm25_kernel(Bm25Args a) {
    const int q = blockIdx.x;
    double* __restrict__ row = a.scores + (size_t)q * a.lds;
    for (int j = threadIdx.x; j < a.N; j += blockDim.x) row[j] = 0.0;
    __syncthreads();
    const int64_t p0 = a.qoff[q], p1 = a.qoff[q + 1];
    for (int64_t p = p0; p < p1; ++p) {
        const int t = a.qterms[p];
        if (t < 0) continue;  // out of vocabulary: idf 0 (block-uniform)
        const double w = a.idf[t];
        const int64_t e0 = a.toff[t], e1 = a.toff[t + 1];
        for (int64_t e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
            const int dj = a.pdoc[e];
            const double tf = (double)a.ptf[e];
            const double num = w * (tf * (a.k1 + 1.0));
            const double den = tf + a.k1 * (1.0 - a.b + a.b * (double)a.doc_len[dj] / a.avgdl);
            row[dj] = row[dj] + num / den;
        }
        __syncthreads();  // next term may touch the same documents: keep the reference's addition order
    }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_bm25_scores_f64(const int64_t* toff, const int32_t* pdoc, const int32_t* ptf, const double* idf,
                                  const int32_t* doc_len, double avgdl, double k1, double b, const int64_t* qoff,
                                  const int32_t* qterms, int Q, int N, double* scores, int lds, void* stream) {
    if (!toff || !idf || !doc_len || !qoff || !scores || Q < 0 || N < 0 || lds < N) return FZ_ERR_ARG;
    if (Q == 0) return FZ_OK;
    Bm25Args a{toff, pdoc, ptf, idf, doc_len, avgdl, k1, b, qoff, qterms, N, scores, lds};
    bm25_kernel<<<Q, 1024, 0, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
