// sparse.hip -- A3, the sparse form of SPLADE's cosine scoring (hybrid.py:95-103: util.semantic_search(..., score_function=util.cos_sim)
// over SPLADE vectors; splade/splade.py:88-99 produces them: log1p(relu(logits)) max-pooled -- a few hundred non-zeros of 32,005).
//
// The reference multiplies the DENSE [Q, 32005] x [32005, N] matrices (3.6 GB of mostly zeros at LLeQA size).  Here the L2-normalised
// corpus vectors are an inverted index -- per vocabulary term the (document, weight) postings, documents ascending -- and a query's score row is
//      score[d] = sum over the query's non-zero terms t, in ascending t, of q_t * w_{t,d}
// i.e. the same products as the dense contraction minus the exact zeros (adding +0.0 to a float32 sum changes nothing: SPLADE weights are
// >= 0), summed in vocabulary order with one rounding per product and per add (the dense MFMA form fuses them): equal within ~1e-7 relative.
// One workgroup = (query, slice of SP_SLICE documents): float32 accumulators in LDS, the postings of
// one term touch distinct documents (no atomics), terms one after the other (barrier): bit-reproducible.  Structure of bm25.hip's kernel.
#include "common.h"

namespace fz {

// 7,168 fp32 accumulators (28 KiB) and 512 threads per workgroup: four workgroups per CU.  A SPLADE query's walk is a few dozen short posting
// lists with a barrier each; independent workgroups fill each other's waits (round 6, measured per 1024 x 27,942: 28,672 documents x 1024
// threads -- the whole corpus in one workgroup -- 0.275 ms, 14,336 x 1024 0.220, 14,336 x 512 0.258, 7,168 x 256 0.246, 3,584 x 256 0.196,
// 7,168 x 512 0.202)
constexpr int SP_SLICE = 7168;
constexpr int SP_THREADS = 512;
constexpr int SP_TERMS = 256;       // query terms whose posting ranges are resolved per batch

struct SparseArgs {
    const int64_t* toff; const int32_t* pdoc; const float* pw;     // index: postings of term t are [toff[t], toff[t+1]), documents ascending
    const int64_t* slice_off;                                      // nullable [V][NS + 1]: first posting of term t with document >= s * SP_SLICE
    const int64_t* qoff; const int32_t* qterms; const float* qw;   // queries: non-zero terms of query q are [qoff[q], qoff[q+1]), ascending
    int N; float* scores; int lds;
};

__device__ __forceinline__ int64_t sp_lower_bound(const int32_t* __restrict__ pdoc, int64_t lo, int64_t hi, int doc) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (pdoc[mid] < doc) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(1024) void sparse_dot_kernel(SparseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sp_acc[];        // [SP_SLICE]
    __shared__ int64_t s_e0[SP_TERMS], s_e1[SP_TERMS];
    __shared__ float s_w[SP_TERMS];
    const int q = blockIdx.y;
    const int d0 = blockIdx.x * SP_SLICE;
    const int d1 = (d0 + SP_SLICE < a.N) ? d0 + SP_SLICE : a.N;
    const int n = d1 - d0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) sp_acc[j] = 0.0f;
    const int64_t p0 = a.qoff[q], p1 = a.qoff[q + 1];
    for (int64_t pb = p0; pb < p1; pb += SP_TERMS) {
        const int nt = (int)((p1 - pb < SP_TERMS) ? p1 - pb : SP_TERMS);
        __syncthreads();   // accumulators zeroed / the previous batch's table no longer read
        if ((int)threadIdx.x < nt) {
            const int t = a.qterms[pb + threadIdx.x];
            int64_t e0, e1;
            if (a.slice_off) {
                const int64_t* so = a.slice_off + (size_t)t * (gridDim.x + 1) + blockIdx.x;
                e0 = so[0]; e1 = so[1];
            } else {
                e0 = sp_lower_bound(a.pdoc, a.toff[t], a.toff[t + 1], d0);
                e1 = sp_lower_bound(a.pdoc, e0, a.toff[t + 1], d1);
            }
            s_e0[threadIdx.x] = e0; s_e1[threadIdx.x] = e1; s_w[threadIdx.x] = a.qw[pb + threadIdx.x];
        }
        __syncthreads();
        for (int k = 0; k < nt; ++k) {          // terms in ascending vocabulary order: the order of the dense contraction
            const int64_t e0 = s_e0[k], e1 = s_e1[k];
            const float w = s_w[k];
            if (e1 <= e0) continue;             // block-uniform
            constexpr int U = 4;
            for (int64_t eb = e0; eb < e1; eb += (int64_t)blockDim.x * U) {
                int doc[U]; float pw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t e = eb + (int64_t)u * blockDim.x + threadIdx.x;
                    const bool ok = e < e1;
                    doc[u] = ok ? a.pdoc[e] : -1;
                    pw[u] = ok ? a.pw[e] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (doc[u] >= 0) sp_acc[doc[u] - d0] = sp_acc[doc[u] - d0] + w * pw[u];   // one term's postings hit distinct documents
            }
            __syncthreads();  // the next term may touch the same documents
        }
    }
    __syncthreads();
    float* __restrict__ row = a.scores + (size_t)q * a.lds + d0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) row[j] = sp_acc[j];
}

__global__ void sparse_slice_offsets_kernel(const int64_t* __restrict__ toff, const int32_t* __restrict__ pdoc, int V, int NS, int64_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)V * (NS + 1)) return;
    const int t = (int)(i / (NS + 1)), s_ = (int)(i % (NS + 1));
    out[i] = s_ == NS ? toff[t + 1] : sp_lower_bound(pdoc, toff[t], toff[t + 1], s_ * SP_SLICE);
}

}  // namespace fz

using namespace fz;

extern "C" int fz_sparse_slice_docs(void) { return SP_SLICE; }

extern "C" int fz_sparse_slice_offsets(const int64_t* toff, const int32_t* pdoc, int V, int N, int64_t* out, void* stream) {
    if (V < 0 || N < 0) return FZ_ERR_ARG;
    if (V == 0) return FZ_OK;
    if (!toff || !out) return FZ_ERR_ARG;
    const int NS = N > 0 ? (N + SP_SLICE - 1) / SP_SLICE : 1;
    const long total = (long)V * (NS + 1);
    sparse_slice_offsets_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(toff, pdoc, V, NS, out);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_sparse_dot_f32(const int64_t* toff, const int32_t* pdoc, const float* pw, const int64_t* slice_off, const int64_t* qoff,
                                 const int32_t* qterms, const float* qw, int Q, int N, float* scores, int lds, void* stream) {
    if (Q < 0 || N < 0 || lds < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;           // empty tensors carry null pointers
    if (!toff || !qoff || !scores) return FZ_ERR_ARG;
    SparseArgs a{toff, pdoc, pw, slice_off, qoff, qterms, qw, N, scores, lds};
    constexpr size_t lds_bytes = (size_t)SP_SLICE * sizeof(float);
    static unsigned long long lds_set = 0ull;
    if (int rc = raise_lds_limit((const void*)sparse_dot_kernel, lds_bytes, lds_set)) return rc;
    dim3 grid((unsigned)((N + SP_SLICE - 1) / SP_SLICE), (unsigned)Q);
    sparse_dot_kernel<<<grid, SP_THREADS, lds_bytes, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
