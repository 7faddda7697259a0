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
// Mapping: grid (column chunks of 4096, queries); 256 threads x 16 columns; the chunk's S x 16 normalised scores and
// 16 positions stay in registers for the whole sweep; per weight vector: S unfused mul+add per column, one 64-bit compare
// + scalar popcount per (column, gold), one atomicAdd per wave and gold.
#include "common.h"

namespace fz {

constexpr int TUNE_G = 8;        // gold documents per query handled per launch
constexpr int TUNE_COLS = 16;    // columns per thread

struct TuneArgs {
    const float* T[FZ_MAX_SYSTEMS];   // normalised planes [Q][ld]; entries of docs a system does not list must be 0
    const int32_t* pos;               // [Q][ld] first-insertion position, -1 = doc in no list
    const float* weights;             // [W][S] fp32 (already rounded from the Python floats)   -- narrow sweep
    const double* weights64;          // [W][S] fp64 (np.float64 grid weights)                  -- wide sweep
    const int32_t* gold;              // [Q][TUNE_G] corpus positions, -1 = padding
    int32_t* out;                     // [W][Q][TUNE_G] ranks (zero-initialised by the caller)
    int S, W, Q, N, ld;
};

template <int S>
__global__ __launch_bounds__(256) void gold_ranks_kernel(TuneArgs a) {
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const size_t rowoff = (size_t)q * a.ld;
    const int c0 = blockIdx.x * (256 * TUNE_COLS);

    // this thread's columns: strided by 256 inside the chunk (coalesced 4-byte loads)
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
    // the query's gold documents (block-uniform)
    float tg[S][TUNE_G];
    int pg[TUNE_G];
#pragma unroll
    for (int g = 0; g < TUNE_G; ++g) {
        const int col = a.gold[q * TUNE_G + g];
        const bool ok = col >= 0 && col < a.N;
        pg[g] = ok ? a.pos[rowoff + col] : -1;   // -1: padding or a gold doc no system retrieved (never ranked)
#pragma unroll
        for (int s = 0; s < S; ++s) tg[s][g] = ok ? a.T[s][rowoff + col] : 0.f;
    }

    // "j precedes g" as ONE 64-bit unsigned compare: (desc_key(fused_j), pos_j) < (desc_key(fused_g), pos_g), lexicographic.
    // desc_key_f32 (common.h) is the sort kernel's order: larger score -> smaller key, -0 == +0, NaN first -- so the count
    // is exactly the rank the materialise-and-sort path would give, ties by first-insertion position included.
    // Docs in no list get the largest key (they precede nothing).  The per-(column, gold) work is one v_cmp + a scalar
    // popcount; the counters live in SGPRs (one atomicAdd per wave and gold at the end, no wave reduction).
    uint32_t lo[TUNE_COLS];
#pragma unroll
    for (int i = 0; i < TUNE_COLS; ++i) lo[i] = (uint32_t)pj[i];
    for (int w = 0; w < a.W; ++w) {
        float wv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) wv[s] = a.weights[w * S + s];
        uint64_t kg[TUNE_G];
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { const float prod = tg[s][g] * wv[s]; acc = acc + prod; }
            kg[g] = ((uint64_t)desc_key_f32(acc) << 32) | (uint32_t)pg[g];
        }
        int cnt[TUNE_G];
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g) cnt[g] = 0;
#pragma unroll
        for (int i = 0; i < TUNE_COLS; ++i) {
            float f = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) { const float prod = t[s][i] * wv[s]; f = f + prod; }   // hybrid.py:291,304 (NumPy 2: fp32)
            const uint64_t kj = pj[i] >= 0 ? (((uint64_t)desc_key_f32(f) << 32) | lo[i]) : ~0ull;
#pragma unroll
            for (int g = 0; g < TUNE_G; ++g) cnt[g] += __popcll(__ballot(kj < kg[g]));
        }
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g)
            if (lane == 0 && cnt[g] != 0 && pg[g] >= 0) atomicAdd(&a.out[((size_t)w * a.Q + q) * TUNE_G + g], cnt[g]);
    }
}

// The same sweep with np.float64 weights -- what the reference's own grid holds (np.arange, hybrid.py:405-409): the
// np.float32 transformed score times an np.float64 weight is a float64 product and the per-document sum is float64
// (NumPy-2 promotion; the pinned NumPy 1.x promotes every nsf product to float64).  fused_j = sum_s fl64(t_s) * w_s in
// system order; adding a 0 * w term for a document a system does not list changes nothing, so the zero-filled planes
// are enough.  "j precedes g" = (desc_key_f64(fused_j), pos_j) < (desc_key_f64(fused_g), pos_g): a 64-bit compare
// plus a 32-bit tie-break.
template <int S>
__global__ __launch_bounds__(256) void gold_ranks_wide_kernel(TuneArgs a) {
    const int q = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const size_t rowoff = (size_t)q * a.ld;
    const int c0 = blockIdx.x * (256 * TUNE_COLS);
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
    float tg[S][TUNE_G];
    int pg[TUNE_G];
#pragma unroll
    for (int g = 0; g < TUNE_G; ++g) {
        const int col = a.gold[q * TUNE_G + g];
        const bool ok = col >= 0 && col < a.N;
        pg[g] = ok ? a.pos[rowoff + col] : -1;
#pragma unroll
        for (int s = 0; s < S; ++s) tg[s][g] = ok ? a.T[s][rowoff + col] : 0.f;
    }
    for (int w = 0; w < a.W; ++w) {
        double wv[S];
#pragma unroll
        for (int s = 0; s < S; ++s) wv[s] = a.weights64[w * S + s];
        uint64_t kg[TUNE_G];
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g) {
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < S; ++s) { const double prod = (double)tg[s][g] * wv[s]; acc = acc + prod; }
            kg[g] = desc_key_f64(acc);
        }
        int cnt[TUNE_G];
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g) cnt[g] = 0;
#pragma unroll
        for (int i = 0; i < TUNE_COLS; ++i) {
            double f = 0.0;
#pragma unroll
            for (int s = 0; s < S; ++s) { const double prod = (double)t[s][i] * wv[s]; f = f + prod; }   // hybrid.py:291,304 with np.float64 weights
            const uint64_t kj = desc_key_f64(f);
            const bool listed = pj[i] >= 0;
#pragma unroll
            for (int g = 0; g < TUNE_G; ++g) {
                const bool before = listed && (kj < kg[g] || (kj == kg[g] && (uint32_t)pj[i] < (uint32_t)pg[g]));
                cnt[g] += __popcll(__ballot(before));
            }
        }
#pragma unroll
        for (int g = 0; g < TUNE_G; ++g)
            if (lane == 0 && cnt[g] != 0 && pg[g] >= 0) atomicAdd(&a.out[((size_t)w * a.Q + q) * TUNE_G + g], cnt[g]);
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
    dim3 grid((unsigned)((N + 256 * TUNE_COLS - 1) / (256 * TUNE_COLS)), (unsigned)Q);
    hipStream_t st = as_stream(stream);
    switch (S) {
        case 1: gold_ranks_kernel<1><<<grid, 256, 0, st>>>(a); break;
        case 2: gold_ranks_kernel<2><<<grid, 256, 0, st>>>(a); break;
        case 3: gold_ranks_kernel<3><<<grid, 256, 0, st>>>(a); break;
        case 4: gold_ranks_kernel<4><<<grid, 256, 0, st>>>(a); break;
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
    dim3 grid((unsigned)((N + 256 * TUNE_COLS - 1) / (256 * TUNE_COLS)), (unsigned)Q);
    hipStream_t st = as_stream(stream);
    switch (S) {
        case 1: gold_ranks_wide_kernel<1><<<grid, 256, 0, st>>>(a); break;
        case 2: gold_ranks_wide_kernel<2><<<grid, 256, 0, st>>>(a); break;
        case 3: gold_ranks_wide_kernel<3><<<grid, 256, 0, st>>>(a); break;
        case 4: gold_ranks_wide_kernel<4><<<grid, 256, 0, st>>>(a); break;
        default: return FZ_ERR_UNSUPPORTED;
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
