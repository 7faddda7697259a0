// tune.hip -- N1: the linear-fusion weight sweep of hybrid.py:404-426 without re-fusing or sorting.
//
// Reference: for each of W weight vectors (21 / 231 / 1771 for S = 2 / 3 / 4, hybrid.py:405-409) it deep-copies all
// ranked lists, re-normalises them, fuses, sorts and evaluates (hybrid.py:418-425): ~60 h of Python for LLeQA test.
// Every metric of run_evaluation (metrics.py:40-136) is a function of the fused RANKS of the gold documents only, and
//     rank(g) = #{ j : fused_j > fused_g  or  (fused_j == fused_g and pos_j < pos_g) }
// (pos = first-insertion position: the stable-sort tie-break, hybrid.py:301-306).  So the normalised planes T_s are
// computed ONCE (fz_fuse_nsf_f32 with weight 1) and this kernel only counts.  fused_j is evaluated with exactly the
// arithmetic of the fusion kernel (fl32(t*w), sequential fl32 adds, unfused), so the ranks are identical to those
// of the materialise-and-sort path.
//
// Mapping: grid (column chunks of 4096 / 2048, queries); 256 threads x 16 / 8 columns (float32 / float64 sweep); the chunk's
// normalised scores and positions stay in registers for the whole sweep; per weight vector: S unfused mul+add per column, one
// 64-bit compare + scalar popcount per (column, LISTED gold), one LDS add per wave and one global atomicAdd per workgroup and gold.
#include <type_traits>

#include "common.h"

namespace fz {

constexpr int TUNE_G = 8;        // gold documents per query handled per launch
constexpr int TUNE_WCHUNK = 512;  // weight vectors per pass over the LDS counters and gold keys (16 + 32 KB: three workgroups per CU)
constexpr int TUNE_COLS_F32 = 16, TUNE_COLS_F64 = 8;   // columns per thread (the float64 sweep keeps them as doubles: half as many)

struct TuneArgs {
    const float* T[FZ_MAX_SYSTEMS];   // normalised planes [Q][ld]; entries of docs a system does not list must be 0
    const int32_t* pos;               // [Q][ld] first-insertion position, -1 = doc in no list
    const float* weights;             // [W][S] fp32 (already rounded from the Python floats)   -- narrow sweep
    const double* weights64;          // [W][S] fp64 (np.float64 grid weights)                  -- wide sweep
    const int32_t* gold;              // [Q][TUNE_G] corpus positions, -1 = padding
    int32_t* out;                     // [W][Q][TUNE_G] ranks (zero-initialised by the caller)
    int S, W, Q, N, ld;
};

// Both sweeps, common part.  A query's gold list is short (LLeQA: 1-5 articles) and padded to TUNE_G: the listed gold
// documents are compacted to the front once (block-uniform), and the weight loop is instantiated for every count NG so that the
// per-(column, gold) work -- the bulk of the kernel -- is done for real golds only.  The four waves of a workgroup add their
// counts in LDS; one global atomicAdd per (workgroup, weight vector, gold).
template <int S, bool WIDE>
__global__ __launch_bounds__(256, 2) void gold_ranks_kernel(TuneArgs a) {
    typedef typename std::conditional<WIDE, double, float>::type F;
    constexpr int TUNE_COLS = WIDE ? TUNE_COLS_F64 : TUNE_COLS_F32;
    extern __shared__ __attribute__((aligned(16))) int lds_acc[];   // (16-byte aligned: the uint64 keys behind the counters are read as 8-byte words) [wch][TUNE_G] counts of the workgroup, then [wch][TUNE_G] sort keys of the golds' fused scores
    uint64_t* lds_kg = reinterpret_cast<uint64_t*>(lds_acc + (size_t)(a.W < TUNE_WCHUNK ? a.W : TUNE_WCHUNK) * TUNE_G);
    int w0 = 0, w1 = 0;
    __shared__ int lds_gold[TUNE_G + 1];
    __shared__ float lds_tg[S][TUNE_G];
    __shared__ int lds_pg[TUNE_G];
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const size_t rowoff = (size_t)q * a.ld;
    const int c0 = blockIdx.x * (256 * TUNE_COLS);
    if (threadIdx.x == 0) {   // compact the listed gold documents (padding and golds no system retrieved are never ranked)
        int n = 0;
        for (int g = 0; g < TUNE_G; ++g) {
            const int col = a.gold[q * TUNE_G + g];
            if (col >= 0 && col < a.N && a.pos[rowoff + col] >= 0) lds_gold[1 + n++] = g;
        }
        lds_gold[0] = n;
    }
    __syncthreads();
    const int ng = lds_gold[0];
    if (ng == 0) return;      // block-uniform

    // this thread's columns: strided by 256 inside the chunk (coalesced 4-byte loads); they stay in registers for the whole sweep
    float t[S][TUNE_COLS];
    int pj[TUNE_COLS];
#pragma unroll
    for (int i = 0; i < TUNE_COLS; ++i) {
        const int j = c0 + i * 256 + threadIdx.x;
        const bool in = j < a.N;
        pj[i] = in ? a.pos[rowoff + j] : -1;
#pragma unroll
        for (int s = 0; s < S; ++s) t[s][i] = in ? a.T[s][rowoff + j] : 0.f;
    }
    // the query's listed gold documents, compacted (block-uniform): entries [0, ng), kept in LDS and read (as a broadcast) once per
    // weight vector -- in registers they would be 40 long-lived values per thread for 8 numbers' worth of information
    if ((int)threadIdx.x < TUNE_G) {
        const int k = threadIdx.x;
        const bool ok = k < ng;
        const int col = ok ? a.gold[q * TUNE_G + lds_gold[1 + k]] : 0;
        lds_pg[k] = ok ? a.pos[rowoff + col] : -1;
#pragma unroll
        for (int s = 0; s < S; ++s) lds_tg[s][k] = ok ? a.T[s][rowoff + col] : 0.f;
    }
    __syncthreads();

    // Non-finite normalised scores (a z-score over a constant list, an infinite score times a zero weight) can make a fused score
    // NaN; only then does the sort key need its NaN case.  Block-uniform.
    bool odd = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int i = 0; i < TUNE_COLS; ++i) odd |= !(fabsf(t[s][i]) <= 3.402823466e38f);
#pragma unroll
        for (int k = 0; k < TUNE_G; ++k) odd |= !(fabsf(lds_tg[s][k]) <= 3.402823466e38f);
    }
    const bool special = __syncthreads_or(odd);

    // fused score exactly as the fusion kernel forms it: 0 + fl(t*w) products, sequential unfused adds in system order
    // (hybrid.py:291,304).  Starting from +0 the sum is never -0.0, so the key needs no -0 case either.
    auto fuse = [&](const float (&x)[S], const F (&wv)[S]) __attribute__((always_inline)) -> F {
        F acc = (F)0;
#pragma unroll
        for (int s = 0; s < S; ++s) { const F prod = (F)x[s] * wv[s]; acc = acc + prod; }
        return acc;
    };
    // desc_key_* of common.h (the sort kernel's order: larger score -> smaller key, NaN first) for a value that is neither NaN nor
    // -0.0: positive -> bits ^ 0x7ff..f, negative -> bits.  SPECIAL: the full form.
    auto key_of = [&](F f, auto sp) __attribute__((always_inline)) {
        if constexpr (decltype(sp)::value) {
            if constexpr (WIDE) return desc_key_f64(f); else return desc_key_f32(f);
        } else if constexpr (WIDE) {
            const uint32_t hi = (uint32_t)__double2hiint(f), lo = (uint32_t)__double2loint(f);
            const uint32_t m = ~(uint32_t)((int32_t)hi >> 31);
            return ((uint64_t)(hi ^ (m >> 1)) << 32) | (lo ^ m);
        } else {
            const uint32_t u = __float_as_uint(f);
            return u ^ (~(uint32_t)((int32_t)u >> 31) >> 1);
        }
    };

    // The weight loop is instantiated for every gold count NG, so that the per-(column, gold) work -- the bulk of the kernel --
    // is done for real golds only (a query's gold list is short, LLeQA: 1-5 articles, and padded to TUNE_G).
    auto run = [&](auto ngc, auto sp) __attribute__((always_inline)) {
        constexpr int NG = decltype(ngc)::value;
        F wnext[S];                                      // weights are fetched one vector ahead: their latency is not in the loop
        auto wload = [&](int w) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if constexpr (WIDE) wnext[s] = a.weights64[(size_t)w * S + s];
                else wnext[s] = a.weights[(size_t)w * S + s];
            }
        };
        wload(w0);
        for (int w = w0; w < w1; ++w) {
            F wv[S];
#pragma unroll
            for (int s = 0; s < S; ++s) wv[s] = wnext[s];
            wload(w + 1 < w1 ? w + 1 : w);
            // "j precedes g": narrow -- ONE 64-bit unsigned compare of (key(fused), pos), lexicographic; wide -- the 64-bit key
            // compare, and only if some (column, gold) pair of the wave holds EQUAL keys (one scalar test per weight vector,
            // rare) a second pass for the position tie-break.  The count is exactly the rank the materialise-and-sort path gives.
            uint64_t kg[NG];                                 // the golds' keys of this weight vector: computed once per workgroup (below), scalars here
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint64_t k = lds_kg[(size_t)(w - w0) * TUNE_G + g];
                kg[g] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(k >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            }
            int cnt[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) cnt[g] = 0;
            unsigned long long any_eq = 0ull;
#pragma unroll
            for (int i = 0; i < TUNE_COLS; ++i) {
                float x[S];
#pragma unroll
                for (int s = 0; s < S; ++s) x[s] = t[s][i];
                const F f = fuse(x, wv);
                // docs in no list (pos = -1) precede nothing: OR-ing the sign of pos into the key gives them the largest one
                const uint32_t unl = (uint32_t)(pj[i] >> 31);
                if constexpr (WIDE) {
                    const uint64_t kj = key_of(f, sp) | ((uint64_t)unl << 32 | unl);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        cnt[g] += __popcll(__ballot(kj < kg[g]));
                        any_eq |= __ballot(kj == kg[g]);
                    }
                } else {
                    const uint64_t kj = ((uint64_t)(key_of(f, sp) | unl) << 32) | (uint32_t)pj[i];
#pragma unroll
                    for (int g = 0; g < NG; ++g) cnt[g] += __popcll(__ballot(kj < kg[g]));
                }
            }
            if constexpr (WIDE) {
                if (any_eq) {                            // wave-uniform
#pragma unroll
                    for (int i = 0; i < TUNE_COLS; ++i) {
                        float x[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) x[s] = t[s][i];
                        const uint64_t kj = key_of(fuse(x, wv), sp);
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            cnt[g] += __popcll(__ballot(pj[i] >= 0 && kj == kg[g] && (uint32_t)pj[i] < (uint32_t)lds_pg[g]));
                    }
                }
            }
            // this wave's counts of this weight vector: LDS adds; nothing in the loop waits for another wave
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (lane == 0 && cnt[g] != 0) atomicAdd(&lds_acc[(w - w0) * TUNE_G + g], cnt[g]);
        }
    };
    auto dispatch = [&](auto sp) __attribute__((always_inline)) {
        switch (ng) {         // block-uniform
            case 1: run(std::integral_constant<int, 1>{}, sp); break;
            case 2: run(std::integral_constant<int, 2>{}, sp); break;
            case 3: run(std::integral_constant<int, 3>{}, sp); break;
            case 4: run(std::integral_constant<int, 4>{}, sp); break;
            case 5: run(std::integral_constant<int, 5>{}, sp); break;
            case 6: run(std::integral_constant<int, 6>{}, sp); break;
            case 7: run(std::integral_constant<int, 7>{}, sp); break;
            default: run(std::integral_constant<int, 8>{}, sp); break;
        }
    };
    // The weight vectors go through the LDS counters TUNE_WCHUNK at a time (all 1771 of the 4-system lattice at once): sweep,
    // barrier, one global atomicAdd per (weight vector, gold) of the workgroup, barrier.
    for (w0 = 0; w0 < a.W; w0 += TUNE_WCHUNK) {
        w1 = min(a.W, w0 + TUNE_WCHUNK);
        for (int i = threadIdx.x; i < (w1 - w0) * TUNE_G; i += 256) lds_acc[i] = 0;
        // the golds' fused scores and sort keys for every weight vector of the chunk, ONCE per workgroup (a thread per weight vector)
        // instead of once per wave and weight vector inside the sweep; the full key form: it equals the fast one wherever that is valid
        for (int w = w0 + threadIdx.x; w < w1; w += 256) {
            F wv[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if constexpr (WIDE) wv[s] = a.weights64[(size_t)w * S + s];
                else wv[s] = a.weights[(size_t)w * S + s];
            }
            for (int g = 0; g < ng; ++g) {
                float x[S];
#pragma unroll
                for (int s = 0; s < S; ++s) x[s] = lds_tg[s][g];
                const F f = fuse(x, wv);
                uint64_t k;
                if constexpr (WIDE) k = desc_key_f64(f);
                else k = ((uint64_t)desc_key_f32(f) << 32) | (uint32_t)lds_pg[g];
                lds_kg[(size_t)(w - w0) * TUNE_G + g] = k;
            }
        }
        __syncthreads();
        if (special) dispatch(std::true_type{});
        else dispatch(std::false_type{});
        __syncthreads();
        for (int i = threadIdx.x; i < (w1 - w0) * TUNE_G; i += 256) {
            const int k = i % TUNE_G, c = lds_acc[i];
            if (k < ng && c != 0) atomicAdd(&a.out[((size_t)(w0 + i / TUNE_G) * a.Q + q) * TUNE_G + lds_gold[1 + k]], c);
        }
        __syncthreads();
    }
}

// ---- the metrics of run_evaluation (hybrid.py:24-42, metrics.py:40-136) for every weight vector, from the gold ranks -------------
// One workgroup per weight vector, one thread per query (strided).  Per query, in float64 and in the reference's own
// operation order: recall@k = hits/len(gold); AP@k = sum over hits in rank order of (i+1)/(rank+1), /len(gold); RR@k;
// nDCG@k = (rel[0] + sum_{pos >= 1} rel/log2(pos+1)) / IDCG with the reference's shifted discount (metrics.py:108; the
// discount table and IDCG come from the host, computed with NumPy as the reference does); R-precision.  The mean over the
// queries is accumulated as an unevaluated (hi, lo) double-double pair (error-free TwoSum) and rounded once, like
// statistics.mean's exact rational sum to within one rounding.
constexpr int TM_MAX = 24;       // metric columns per launch

struct MetricsArgs {
    const int32_t* ranks;        // [W][Q][TUNE_G] as written by fz_gold_ranks_*
    const int32_t* gold;         // [Q][TUNE_G] corpus positions, -1 = padding
    const int32_t* pos;          // [Q][ld] first-insertion position, -1 = document in no list (its gold rank is infinite)
    const int32_t* n_gold;       // [Q] len(ground_truths): the reference's divisor
    const double* idcg;          // [Q]
    const double* disc;          // [top + 1]: disc[0] = 1, disc[i] = 1 / log2(i + 1)
    const int32_t* cuts;         // [n_recall + n_map + n_mrr + n_ndcg] cut-offs k, in that order
    double* out;                 // [W][M], M = n_recall + n_map + n_mrr + n_ndcg + 1 (R-precision last)
    int n_recall, n_map, n_mrr, n_ndcg, top;
    int W, Q, ld;
};

struct dd { double hi, lo; };
__device__ __forceinline__ dd dd_add(dd x, dd y) {
    const double s = x.hi + y.hi;
    const double bb = s - x.hi;
    const double e = (x.hi - (s - bb)) + (y.hi - bb);
    const double lo = (x.lo + y.lo) + e;
    dd r; r.hi = s + lo; r.lo = lo - (r.hi - s);
    return r;
}

__global__ __launch_bounds__(256) void tune_metrics_kernel(MetricsArgs a) {
    __shared__ double red[4][TM_MAX][2];
    const int w = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int M = a.n_recall + a.n_map + a.n_mrr + a.n_ndcg + 1;
    constexpr int INF = 0x7fffffff;
    dd acc[TM_MAX];
#pragma unroll
    for (int m = 0; m < TM_MAX; ++m) { acc[m].hi = 0.0; acc[m].lo = 0.0; }
    for (int q = threadIdx.x; q < a.Q; q += 256) {
        // ascending ranks of the listed gold documents; the others never appear in the fused list
        int r[TUNE_G];
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g) {
            const int col = a.gold[q * TUNE_G + g];
            const bool listed = col >= 0 && a.pos[(size_t)q * a.ld + (col >= 0 ? col : 0)] >= 0;
            r[g] = listed ? a.ranks[((size_t)w * a.Q + q) * TUNE_G + g] : INF;
        }
#pragma unroll
        for (int i = 1; i < TUNE_G; ++i)
#pragma unroll
            for (int j = i; j > 0; --j) {
                const int lo = min(r[j - 1], r[j]), hi = max(r[j - 1], r[j]);
                r[j - 1] = lo; r[j] = hi;
            }
        const int ngq = a.n_gold[q];
        const double ng = (double)(ngq > 1 ? ngq : 1);
        const int b1 = a.n_recall, b2 = b1 + a.n_map, b3 = b2 + a.n_mrr, b4 = b3 + a.n_ndcg;   // column ranges of the four families
#pragma unroll
        for (int m = 0; m < TM_MAX; ++m) {
            if (m > b4) continue;                              // uniform
            const int k = m < b4 ? a.cuts[m] : ngq;            // R-precision = recall at len(gold)
            double v;
            if (m < b1 || m == b4) {
                int hits = 0;
#pragma unroll
                for (int g = 0; g < TUNE_G; ++g) hits += r[g] < k;
                v = (double)hits / ng;
            } else if (m < b2) {
                double ap = 0.0;
#pragma unroll
                for (int g = 0; g < TUNE_G; ++g) ap = ap + (r[g] < k ? (double)(g + 1) / ((double)r[g] + 1.0) : 0.0);
                v = ap / ng;
            } else if (m < b3) {
                v = r[0] < k ? 1.0 / ((double)r[0] + 1.0) : 0.0;
            } else {
                int head = 0;
                double tail = 0.0;
#pragma unroll
                for (int g = 0; g < TUNE_G; ++g) {
                    head += r[g] == 0;
                    tail = tail + ((r[g] >= 1 && r[g] < k) ? a.disc[r[g] < a.top ? r[g] : a.top] : 0.0);
                }
                v = ((double)head + tail) / a.idcg[q];
            }
            dd x; x.hi = v; x.lo = 0.0;
            acc[m] = dd_add(acc[m], x);
        }
    }
    // workgroup sum: lanes by shuffles, waves through LDS
#pragma unroll
    for (int mm = 0; mm < TM_MAX; ++mm) {
        if (mm >= M) continue;
        dd v = acc[mm];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dd u; u.hi = __shfl_xor(v.hi, o, 64); u.lo = __shfl_xor(v.lo, o, 64);
            v = dd_add(v, u);
        }
        if (lane == 0) { red[wave][mm][0] = v.hi; red[wave][mm][1] = v.lo; }
    }
    __syncthreads();
    if ((int)threadIdx.x < M) {
        dd v; v.hi = red[0][threadIdx.x][0]; v.lo = red[0][threadIdx.x][1];
        for (int ww = 1; ww < 4; ++ww) { dd u; u.hi = red[ww][threadIdx.x][0]; u.lo = red[ww][threadIdx.x][1]; v = dd_add(v, u); }
        a.out[(size_t)w * M + threadIdx.x] = (v.hi + v.lo) / (double)a.Q;
    }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_tune_max_gold(void) { return TUNE_G; }

extern "C" int fz_gold_ranks_f32(const float* const* T_h, const int32_t* pos, const float* weights, const int32_t* gold, int S, int W,
                                 int Q, int N, int ld, int32_t* out_ranks, void* stream) {
    if (!T_h || S <= 0 || S > FZ_MAX_SYSTEMS || W < 0 || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (W == 0 || Q == 0 || N == 0) return FZ_OK;   // empty tensors carry null pointers
    if (!pos || !weights || !gold || !out_ranks) return FZ_ERR_ARG;
    TuneArgs a{};
    for (int s = 0; s < S; ++s) { if (!T_h[s]) return FZ_ERR_ARG; a.T[s] = T_h[s]; }
    a.pos = pos; a.weights = weights; a.gold = gold; a.out = out_ranks; a.S = S; a.W = W; a.Q = Q; a.N = N; a.ld = ld;
    dim3 grid((unsigned)((N + 256 * TUNE_COLS_F32 - 1) / (256 * TUNE_COLS_F32)), (unsigned)Q);
    hipStream_t st = as_stream(stream);
    const size_t lds_bytes = (size_t)(W < TUNE_WCHUNK ? W : TUNE_WCHUNK) * TUNE_G * (sizeof(int) + sizeof(uint64_t));
    switch (S) {
        case 1: gold_ranks_kernel<1, false><<<grid, 256, lds_bytes, st>>>(a); break;
        case 2: gold_ranks_kernel<2, false><<<grid, 256, lds_bytes, st>>>(a); break;
        case 3: gold_ranks_kernel<3, false><<<grid, 256, lds_bytes, st>>>(a); break;
        case 4: gold_ranks_kernel<4, false><<<grid, 256, lds_bytes, st>>>(a); break;
        default: return FZ_ERR_UNSUPPORTED;   // the reference sweeps at most 4 systems (run_hybrid.sh:21-33)
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_gold_ranks_f64w(const float* const* T_h, const int32_t* pos, const double* weights, const int32_t* gold, int S, int W,
                                  int Q, int N, int ld, int32_t* out_ranks, void* stream) {
    if (!T_h || S <= 0 || S > FZ_MAX_SYSTEMS || W < 0 || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (W == 0 || Q == 0 || N == 0) return FZ_OK;   // empty tensors carry null pointers
    if (!pos || !weights || !gold || !out_ranks) return FZ_ERR_ARG;
    TuneArgs a{};
    for (int s = 0; s < S; ++s) { if (!T_h[s]) return FZ_ERR_ARG; a.T[s] = T_h[s]; }
    a.pos = pos; a.weights64 = weights; a.gold = gold; a.out = out_ranks; a.S = S; a.W = W; a.Q = Q; a.N = N; a.ld = ld;
    dim3 grid((unsigned)((N + 256 * TUNE_COLS_F64 - 1) / (256 * TUNE_COLS_F64)), (unsigned)Q);
    hipStream_t st = as_stream(stream);
    const size_t lds_bytes = (size_t)(W < TUNE_WCHUNK ? W : TUNE_WCHUNK) * TUNE_G * (sizeof(int) + sizeof(uint64_t));
    switch (S) {
        case 1: gold_ranks_kernel<1, true><<<grid, 256, lds_bytes, st>>>(a); break;
        case 2: gold_ranks_kernel<2, true><<<grid, 256, lds_bytes, st>>>(a); break;
        case 3: gold_ranks_kernel<3, true><<<grid, 256, lds_bytes, st>>>(a); break;
        case 4: gold_ranks_kernel<4, true><<<grid, 256, lds_bytes, st>>>(a); break;
        default: return FZ_ERR_UNSUPPORTED;
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_tune_metrics_f64(const int32_t* ranks, const int32_t* gold, const int32_t* pos, int ld, const int32_t* n_gold, const double* idcg,
                                   const double* disc, int top, const int32_t* cuts, int n_recall, int n_map, int n_mrr, int n_ndcg, int W, int Q,
                                   double* out, void* stream) {
    if (W < 0 || Q < 0 || ld < 0 || top < 0 || n_recall < 0 || n_map < 0 || n_mrr < 0 || n_ndcg < 0) return FZ_ERR_ARG;
    if (n_recall + n_map + n_mrr + n_ndcg + 1 > TM_MAX) return FZ_ERR_UNSUPPORTED;
    if (W == 0) return FZ_OK;
    if (Q == 0 || !ranks || !gold || !pos || !n_gold || !idcg || !disc || !out) return FZ_ERR_ARG;   // the mean of no query is undefined
    if (n_recall + n_map + n_mrr + n_ndcg > 0 && !cuts) return FZ_ERR_ARG;
    MetricsArgs a{};
    a.ranks = ranks; a.gold = gold; a.pos = pos; a.ld = ld; a.n_gold = n_gold; a.idcg = idcg; a.disc = disc; a.top = top; a.cuts = cuts;
    a.n_recall = n_recall; a.n_map = n_map; a.n_mrr = n_mrr; a.n_ndcg = n_ndcg; a.W = W; a.Q = Q; a.out = out;
    tune_metrics_kernel<<<(unsigned)W, 256, 0, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
