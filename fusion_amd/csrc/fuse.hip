// fuse.hip -- Aggregator.fuse arithmetic on dense planes (reference: src/retrievers/hybrid.py:170-307).
//
//   K3  fz_row_stats_f32      per-(system,query) min/max or mean/unbiased-std     (hybrid.py:255-263)
//   K4  fz_fuse_nsf_f32       normalise -> weight -> sum, ONE pass over HBM         (hybrid.py:212-214,291,304)
//       fz_fuse_none_f64      'none' passthrough, float64                          (hybrid.py:280,291,304)
//   K5b fz_fuse_rank_f64      rrf / bcf from rank planes, float64                  (hybrid.py:248-252,304)
//       fz_insertion_order    first-insertion order of the fused dict              (hybrid.py:301-304)
//
// All of these are HBM-bound streaming passes: 16-byte vector accesses, one workgroup per
// query row for the row-statistic kernels (the whole row lives in registers between the
// statistic and the transform, so each plane is read from HBM exactly once), flat grids
// for the purely elementwise ones.  Compiled with -ffp-contract=off: the reference's
// arithmetic is unfused and the oracle checks it bit for bit.
#include "common.h"

namespace fz {

// -------------------------------------------------------------------------------------
// elementwise transform, identical to oracle fzo_transform (hybrid.py:254-280)
// -------------------------------------------------------------------------------------
__device__ __forceinline__ float percentile_rank(float s, const float* __restrict__ distr, int P) {
    // argmin_k |distr_k - s| (first minimum) / P on an ASCENDING table (hybrid.py:272-275).
    // lo = last k with distr_k <= s  (-1 if none)
    int lo = -1, hi = P;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (distr[mid] <= s) lo = mid; else hi = mid;
    }
    int best;
    float bd;
    if (lo < 0) { best = 0; bd = fabsf(distr[0] - s); }
    else {
        best = lo; bd = fabsf(distr[lo] - s);
        if (lo + 1 < P) { float dh = fabsf(distr[lo + 1] - s); if (dh < bd) { best = lo + 1; bd = dh; } }
    }
    // rounded distances are monotone towards the nearest entry: equal ones are contiguous on the left
    while (best > 0 && fabsf(distr[best - 1] - s) == bd) --best;
    // NaN score: every comparison false -> the reference's argmin returns 0 for an all-NaN column
    if (s != s) best = 0;
    return (float)best / (float)P;
}

template <int NORM>
__device__ __forceinline__ float transform(float s, float a, float b, const float* __restrict__ distr, int P) {
    if (NORM == FZ_NORM_MINMAX) return (a != b) ? (s - a) / (b - a) : 1.0f;
    if (NORM == FZ_NORM_ZSCORE) return (b != 0.0f) ? (s - a) / b : 0.0f;
    if (NORM == FZ_NORM_ARCTAN) return (float)(2.0 / M_PI) * atanf(0.1f * s);
    if (NORM == FZ_NORM_PERCENTILE) return percentile_rank(s, distr, P);
    if (NORM == FZ_NORM_NCE) {
        float pr = percentile_rank(s, distr, P);
        float p = pr / 100.0f;
        float y = 2.0f * p - 1.0f;
        float z = (float)(erfinv((double)y) * 1.4142135623730951);
        return z * 21.06f + 50.0f;
    }
    return s;
}

// -------------------------------------------------------------------------------------
// block-level reductions (THREADS threads, THREADS/64 waves)
// -------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red /* 2*THREADS/64 */) {
    constexpr int NW = THREADS / 64;
    mn = wave_reduce_min(mn);
    mx = wave_reduce_max(mx);
    int w = threadIdx.x >> 6;
    __syncthreads();  // red may still be read from a previous use
    if ((threadIdx.x & 63) == 0) { red[w] = mn; red[NW + w] = mx; }
    __syncthreads();
    mn = red[0]; mx = red[NW];
#pragma unroll
    for (int i = 1; i < NW; ++i) { mn = fminf(mn, red[i]); mx = fmaxf(mx, red[NW + i]); }
}
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double* red /* THREADS/64 */) {
    constexpr int NW = THREADS / 64;
    v = wave_reduce_sum(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];  // fixed order: deterministic
    return s;
}

// NaN-propagating min/max as torch.min/torch.max do
__device__ __forceinline__ float tmin(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : fminf(a, b); }
__device__ __forceinline__ float tmax(float a, float b) { return (a != a || b != b) ? __uint_as_float(0x7fc00000u) : fmaxf(a, b); }

// -------------------------------------------------------------------------------------
// K3: row statistics.  One workgroup per row; streaming 16-B loads; fp64 accumulation.
// -------------------------------------------------------------------------------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void row_stats_kernel(const float* __restrict__ scores, const int32_t* __restrict__ rank,
                                                            int N, int ld, int norm, float* __restrict__ stat_a,
                                                            float* __restrict__ stat_b) {
    __shared__ double red_d[THREADS / 64];
    __shared__ float red_f[2 * THREADS / 64];
    const int row = blockIdx.x;
    const float* x = scores + (size_t)row * ld;
    const int32_t* v = rank ? rank + (size_t)row * ld : nullptr;
    if (norm == FZ_NORM_MINMAX) {
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int j = threadIdx.x; j < N; j += THREADS) {
            if (v && v[j] < 0) continue;
            float s = x[j];
            nan |= (s != s);
            mn = fminf(mn, s); mx = fmaxf(mx, s);
        }
        block_minmax<THREADS>(mn, mx, red_f);
        int anynan = __syncthreads_or(nan ? 1 : 0);
        if (threadIdx.x == 0) {
            stat_a[row] = anynan ? __uint_as_float(0x7fc00000u) : mn;
            stat_b[row] = anynan ? __uint_as_float(0x7fc00000u) : mx;
        }
    } else if (norm == FZ_NORM_ZSCORE) {
        double sum = 0.0, cnt = 0.0;
        for (int j = threadIdx.x; j < N; j += THREADS) {
            if (v && v[j] < 0) continue;
            sum += (double)x[j]; cnt += 1.0;
        }
        sum = block_sum<THREADS>(sum, red_d);
        cnt = block_sum<THREADS>(cnt, red_d);
        double mean = cnt > 0.0 ? sum / cnt : (double)NAN;
        double ss = 0.0;
        for (int j = threadIdx.x; j < N; j += THREADS) {
            if (v && v[j] < 0) continue;
            double d = (double)x[j] - mean;
            ss += d * d;
        }
        ss = block_sum<THREADS>(ss, red_d);
        if (threadIdx.x == 0) {
            double var = cnt > 1.0 ? ss / (cnt - 1.0) : (double)NAN;
            stat_a[row] = (float)mean;
            stat_b[row] = (float)sqrt(var);
        }
    } else if (threadIdx.x == 0) { stat_a[row] = 0.f; stat_b[row] = 0.f; }
}

// -------------------------------------------------------------------------------------
// K4: fused normalise -> weight -> sum.  One workgroup (1024 threads) per query; thread t owns
// columns {4*(t + 1024*i) .. +3}.  For each system: the row is loaded ONCE into registers
// (E floats/thread), reduced to its statistic across the workgroup, transformed from the
// registers and accumulated into the fused registers.  HBM traffic = (S+1)*N*4 B per query
// (+ S*N*4 when rank planes carry validity).
// -------------------------------------------------------------------------------------
struct NsfArgs {
    const float* planes[FZ_MAX_SYSTEMS];
    const int32_t* ranks[FZ_MAX_SYSTEMS];
    const float* distr[FZ_MAX_SYSTEMS];
    int P[FZ_MAX_SYSTEMS];
    float w[FZ_MAX_SYSTEMS];
    int S, N, ld;
};

template <int NORM, int E4 /* float4 per thread */, bool VEC>
__global__ __launch_bounds__(1024) void fuse_nsf_row_kernel(NsfArgs a, float* __restrict__ fused) {
    constexpr int T = 1024;
    __shared__ double red_d[T / 64];
    __shared__ float red_f[2 * T / 64];
    const int q = blockIdx.x;
    const int N = a.N;
    const size_t rowoff = (size_t)q * a.ld;

    float acc[E4][4];
    bool present[E4][4];
#pragma unroll
    for (int i = 0; i < E4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[i][c] = 0.0f; present[i][c] = false; }

    for (int s = 0; s < a.S; ++s) {
        const float* __restrict__ x = a.planes[s] + rowoff;
        const int32_t* __restrict__ rk = a.ranks[s] ? a.ranks[s] + rowoff : nullptr;
        float v[E4][4];
        bool ok[E4][4];
#pragma unroll
        for (int i = 0; i < E4; ++i) {
            const int j0 = 4 * (threadIdx.x + T * i);
            if (VEC) {
                if (j0 + 3 < N) {
                    float4 f = *reinterpret_cast<const float4*>(x + j0);
                    v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
                    ok[i][0] = ok[i][1] = ok[i][2] = ok[i][3] = true;
                    if (rk) {
                        int4 r = *reinterpret_cast<const int4*>(rk + j0);
                        ok[i][0] = r.x >= 0; ok[i][1] = r.y >= 0; ok[i][2] = r.z >= 0; ok[i][3] = r.w >= 0;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        bool in = j0 + c < N;
                        v[i][c] = in ? x[j0 + c] : 0.0f;
                        ok[i][c] = in && (!rk || rk[j0 + c] >= 0);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bool in = j0 + c < N;
                    v[i][c] = in ? x[j0 + c] : 0.0f;
                    ok[i][c] = in && (!rk || rk[j0 + c] >= 0);
                }
            }
        }
        float sa = 0.f, sb = 0.f;
        if (NORM == FZ_NORM_MINMAX) {
            float mn = INFINITY, mx = -INFINITY;
            bool nan = false;
#pragma unroll
            for (int i = 0; i < E4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ok[i][c]) { nan |= (v[i][c] != v[i][c]); mn = fminf(mn, v[i][c]); mx = fmaxf(mx, v[i][c]); }
            block_minmax<T>(mn, mx, red_f);
            int anynan = __syncthreads_or(nan ? 1 : 0);
            sa = anynan ? __uint_as_float(0x7fc00000u) : mn;
            sb = anynan ? __uint_as_float(0x7fc00000u) : mx;
        } else if (NORM == FZ_NORM_ZSCORE) {
            double sum = 0.0, cnt = 0.0;
#pragma unroll
            for (int i = 0; i < E4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ok[i][c]) { sum += (double)v[i][c]; cnt += 1.0; }
            sum = block_sum<T>(sum, red_d);
            cnt = block_sum<T>(cnt, red_d);
            double mean = cnt > 0.0 ? sum / cnt : (double)NAN;
            double ss = 0.0;
#pragma unroll
            for (int i = 0; i < E4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ok[i][c]) { double d = (double)v[i][c] - mean; ss += d * d; }
            ss = block_sum<T>(ss, red_d);
            double var = cnt > 1.0 ? ss / (cnt - 1.0) : (double)NAN;
            sa = (float)mean;
            sb = (float)sqrt(var);
        }
        const float w = a.w[s];
        const float* __restrict__ distr = a.distr[s];
        const int P = a.P[s];
#pragma unroll
        for (int i = 0; i < E4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (ok[i][c]) {
                    float t = transform<NORM>(v[i][c], sa, sb, distr, P);
                    float prod = t * w;           // fl32(t * fl32(w))      hybrid.py:291 under NumPy 2
                    acc[i][c] = acc[i][c] + prod; // fl32(acc + prod)        hybrid.py:304
                    present[i][c] = true;
                }
    }
    float* __restrict__ out = fused + rowoff;
#pragma unroll
    for (int i = 0; i < E4; ++i) {
        const int j0 = 4 * (threadIdx.x + T * i);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = present[i][c] ? acc[i][c] : -INFINITY;
        if (VEC && j0 + 3 < N) *reinterpret_cast<float4*>(out + j0) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j0 + c < N) out[j0 + c] = o[c];
        }
    }
}

// general-N path: statistics from a separate pass (stat arrays [S][Q]), then elementwise.
template <int NORM>
__global__ __launch_bounds__(256) void fuse_nsf_elem_kernel(NsfArgs a, const float* __restrict__ stat_a,
                                                            const float* __restrict__ stat_b, int Q, float* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.N; j += gridDim.x * blockDim.x) {
        float acc = 0.0f;
        bool present = false;
        for (int s = 0; s < a.S; ++s) {
            if (a.ranks[s] && a.ranks[s][rowoff + j] < 0) continue;
            float t = transform<NORM>(a.planes[s][rowoff + j], stat_a[s * Q + q], stat_b[s * Q + q], a.distr[s], a.P[s]);
            float prod = t * a.w[s];
            acc = acc + prod;
            present = true;
        }
        fused[rowoff + j] = present ? acc : -INFINITY;
    }
}

// -------------------------------------------------------------------------------------
// 'none' passthrough in float64 and rank fusion in float64: flat elementwise kernels,
// 4 columns per thread (16-B rank/score loads, 32-B stores).
// -------------------------------------------------------------------------------------
struct ElemArgs {
    const float* planes[FZ_MAX_SYSTEMS];
    const int32_t* ranks[FZ_MAX_SYSTEMS];
    double w[FZ_MAX_SYSTEMS];
    int S, N, ld, Q, method;
    const int32_t* lens;
};

template <bool VEC>
__global__ __launch_bounds__(256) void fuse_none_kernel(ElemArgs a, double* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    const int j0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (j0 >= a.N) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    bool present[4] = {false, false, false, false};
    const bool full = VEC && (j0 + 3 < a.N);
    for (int s = 0; s < a.S; ++s) {
        float v[4]; int r[4] = {0, 0, 0, 0};
        if (full) {
            float4 f = *reinterpret_cast<const float4*>(a.planes[s] + rowoff + j0);
            v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            if (a.ranks[s]) { int4 t = *reinterpret_cast<const int4*>(a.ranks[s] + rowoff + j0); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bool in = j0 + c < a.N;
                v[c] = in ? a.planes[s][rowoff + j0 + c] : 0.f;
                r[c] = in ? (a.ranks[s] ? a.ranks[s][rowoff + j0 + c] : 0) : -1;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (r[c] >= 0) { double prod = (double)v[c] * a.w[s]; acc[c] = acc[c] + prod; present[c] = true; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (j0 + c < a.N) fused[rowoff + j0 + c] = present[c] ? acc[c] : -(double)INFINITY;
}

template <bool VEC>
__global__ __launch_bounds__(256) void fuse_rank_kernel(ElemArgs a, double* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    const int j0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (j0 >= a.N) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    bool present[4] = {false, false, false, false};
    const bool full = VEC && (j0 + 3 < a.N);
    for (int s = 0; s < a.S; ++s) {
        int r[4];
        if (full) { int4 t = *reinterpret_cast<const int4*>(a.ranks[s] + rowoff + j0); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = (j0 + c < a.N) ? a.ranks[s][rowoff + j0 + c] : -1;
        }
        const double n = (a.method == FZ_BCF) ? (double)a.lens[s * a.Q + q] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (r[c] >= 0) {
                double contrib;
                if (a.method == FZ_RRF) contrib = 1.0 / (double)(60 + r[c] + 1);   // hybrid.py:252
                else contrib = (n - (double)r[c] + 1.0) / n;                        // hybrid.py:249 (sic)
                acc[c] = acc[c] + contrib;
                present[c] = true;
            }
    }
    if (full) {
        double2* o = reinterpret_cast<double2*>(fused + rowoff + j0);
        o[0] = make_double2(present[0] ? acc[0] : -(double)INFINITY, present[1] ? acc[1] : -(double)INFINITY);
        o[1] = make_double2(present[2] ? acc[2] : -(double)INFINITY, present[3] ? acc[3] : -(double)INFINITY);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (j0 + c < a.N) fused[rowoff + j0 + c] = present[c] ? acc[c] : -(double)INFINITY;
    }
}

// -------------------------------------------------------------------------------------
// first-insertion order: one workgroup per query, `seen` bitmap in LDS (N <= 1,048,576),
// system by system, chunk by chunk, stable compaction by block-wide prefix of "new" flags.
// -------------------------------------------------------------------------------------
struct InsArgs {
    const int32_t* orders[FZ_MAX_SYSTEMS];
    const int32_t* lens;
    int S, Q, N, ld;
};

__global__ __launch_bounds__(1024) void insertion_order_kernel(InsArgs a, int32_t* __restrict__ ins_order, int32_t* __restrict__ U) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    constexpr int T = 1024, NW = T / 64;
    const int words = (a.N + 31) / 32;
    uint32_t* seen = smem;             // [words]
    uint32_t* wtot = smem + words;     // [NW]
    const int q = blockIdx.x;
    const size_t rowoff = (size_t)q * a.ld;
    for (int i = threadIdx.x; i < words; i += T) seen[i] = 0u;
    __syncthreads();
    int base = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int s = 0; s < a.S; ++s) {
        const int len = a.lens[s * a.Q + q];
        const int32_t* __restrict__ ord = a.orders[s] + rowoff;
        for (int r0 = 0; r0 < len; r0 += T) {
            const int r = r0 + threadIdx.x;
            int j = -1;
            bool isnew = false;
            if (r < len) {
                j = ord[r];
                if (j >= 0 && j < a.N) {
                    uint32_t bit = 1u << (j & 31);
                    uint32_t old = atomicOr(&seen[j >> 5], bit);
                    isnew = !(old & bit);
                }
            }
            unsigned long long bal = __ballot(isnew);
            int below = __popcll(bal & ((1ull << lane) - 1ull));
            int wcount = __popcll(bal);
            __syncthreads();
            if (lane == 0) wtot[w] = wcount;
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { int c = wtot[i]; if (i < w) woff += c; tot += c; }
            if (isnew) ins_order[rowoff + base + woff + below] = j;
            base += tot;
        }
    }
    if (threadIdx.x == 0) U[q] = base;
}

}  // namespace fz

using namespace fz;

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" int fz_row_stats_f32(const float* scores, const int32_t* rank, int rows, int N, int ld, int norm, float* stat_a,
                                float* stat_b, void* stream) {
    if (!scores || !stat_a || !stat_b || rows < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    row_stats_kernel<512><<<rows, 512, 0, as_stream(stream)>>>(scores, rank, N, ld, norm, stat_a, stat_b);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

template <int NORM>
static int launch_nsf(const NsfArgs& a, int Q, float* fused, hipStream_t st) {
    const bool vec = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    bool al = vec;
    for (int s = 0; s < a.S; ++s) al = al && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    const int need4 = (a.N + 4095) / 4096;  // float4 per thread at 1024 threads
#define FZ_NSF_CASE(E4)                                                                   \
    if (need4 <= E4) {                                                                    \
        if (al) fuse_nsf_row_kernel<NORM, E4, true><<<Q, 1024, 0, st>>>(a, fused);        \
        else fuse_nsf_row_kernel<NORM, E4, false><<<Q, 1024, 0, st>>>(a, fused);          \
        return 0;                                                                         \
    }
    FZ_NSF_CASE(1) FZ_NSF_CASE(2) FZ_NSF_CASE(4) FZ_NSF_CASE(7) FZ_NSF_CASE(8)
#undef FZ_NSF_CASE
    return 1;  // row too long for the register-resident kernel
}

extern "C" int fz_fuse_nsf_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q,
                               int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h, float* fused,
                               void* stream) {
    if (!planes_h || !w_h || !fused || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (norm == FZ_NORM_NONE) return FZ_ERR_ARG;  // float64 passthrough lives in fz_fuse_none_f64
    if (norm < FZ_NORM_MINMAX || norm > FZ_NORM_NCE) return FZ_ERR_ARG;
    const bool needs_distr = (norm == FZ_NORM_PERCENTILE || norm == FZ_NORM_NCE);
    if (needs_distr && (!distr_h || !P_h)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    NsfArgs a{};
    a.S = S; a.N = N; a.ld = ld;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.distr[s] = needs_distr ? distr_h[s] : nullptr;
        a.P[s] = needs_distr ? P_h[s] : 0;
        if (needs_distr && (!a.distr[s] || a.P[s] <= 0)) return FZ_ERR_ARG;
        a.w[s] = (float)w_h[s];
    }
    hipStream_t st = as_stream(stream);
    int too_long = 1;
    switch (norm) {
        case FZ_NORM_MINMAX: too_long = launch_nsf<FZ_NORM_MINMAX>(a, Q, fused, st); break;
        case FZ_NORM_ZSCORE: too_long = launch_nsf<FZ_NORM_ZSCORE>(a, Q, fused, st); break;
        case FZ_NORM_ARCTAN: too_long = launch_nsf<FZ_NORM_ARCTAN>(a, Q, fused, st); break;
        case FZ_NORM_PERCENTILE: too_long = launch_nsf<FZ_NORM_PERCENTILE>(a, Q, fused, st); break;
        case FZ_NORM_NCE: too_long = launch_nsf<FZ_NORM_NCE>(a, Q, fused, st); break;
    }
    if (too_long) return FZ_ERR_UNSUPPORTED;  // N > 32768: use fz_row_stats_f32 + fz_fuse_nsf_stats_f32
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// general-N two-pass variant (statistics supplied by the caller, e.g. from fz_row_stats_f32)
extern "C" int fz_fuse_nsf_stats_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S,
                                     int Q, int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h,
                                     const float* stat_a, const float* stat_b, float* fused, void* stream) {
    if (!planes_h || !w_h || !fused || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (norm < FZ_NORM_MINMAX || norm > FZ_NORM_NCE) return FZ_ERR_ARG;
    const bool needs_stats = (norm == FZ_NORM_MINMAX || norm == FZ_NORM_ZSCORE);
    const bool needs_distr = (norm == FZ_NORM_PERCENTILE || norm == FZ_NORM_NCE);
    if (needs_stats && (!stat_a || !stat_b)) return FZ_ERR_ARG;
    if (needs_distr && (!distr_h || !P_h)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    NsfArgs a{};
    a.S = S; a.N = N; a.ld = ld;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.distr[s] = needs_distr ? distr_h[s] : nullptr;
        a.P[s] = needs_distr ? P_h[s] : 0;
        a.w[s] = (float)w_h[s];
    }
    dim3 grid((unsigned)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64), (unsigned)Q);
    hipStream_t st = as_stream(stream);
    switch (norm) {
        case FZ_NORM_MINMAX: fuse_nsf_elem_kernel<FZ_NORM_MINMAX><<<grid, 256, 0, st>>>(a, stat_a, stat_b, Q, fused); break;
        case FZ_NORM_ZSCORE: fuse_nsf_elem_kernel<FZ_NORM_ZSCORE><<<grid, 256, 0, st>>>(a, stat_a, stat_b, Q, fused); break;
        case FZ_NORM_ARCTAN: fuse_nsf_elem_kernel<FZ_NORM_ARCTAN><<<grid, 256, 0, st>>>(a, stat_a, stat_b, Q, fused); break;
        case FZ_NORM_PERCENTILE: fuse_nsf_elem_kernel<FZ_NORM_PERCENTILE><<<grid, 256, 0, st>>>(a, stat_a, stat_b, Q, fused); break;
        case FZ_NORM_NCE: fuse_nsf_elem_kernel<FZ_NORM_NCE><<<grid, 256, 0, st>>>(a, stat_a, stat_b, Q, fused); break;
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

static bool elem_aligned(const ElemArgs& a, const void* fused, bool planes) {
    bool al = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    for (int s = 0; s < a.S; ++s) {
        if (planes) al = al && ((uintptr_t)a.planes[s] % 16 == 0);
        al = al && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    }
    return al;
}

extern "C" int fz_fuse_none_f64(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q,
                                int N, int ld, double* fused, void* stream) {
    if (!planes_h || !w_h || !fused || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    ElemArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.w[s] = w_h[s];
    }
    dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)Q);
    if (elem_aligned(a, fused, true)) fuse_none_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(a, fused);
    else fuse_none_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(a, fused);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_fuse_rank_f64(const int32_t* const* ranks_h, const int32_t* lens, int S, int Q, int N, int ld, int method,
                                double* fused, void* stream) {
    if (!ranks_h || !lens || !fused || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (method != FZ_RRF && method != FZ_BCF) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    ElemArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q; a.method = method; a.lens = lens;
    for (int s = 0; s < S; ++s) {
        if (!ranks_h[s]) return FZ_ERR_ARG;
        a.ranks[s] = ranks_h[s];
    }
    dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)Q);
    if (elem_aligned(a, fused, false)) fuse_rank_kernel<true><<<grid, 256, 0, as_stream(stream)>>>(a, fused);
    else fuse_rank_kernel<false><<<grid, 256, 0, as_stream(stream)>>>(a, fused);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" size_t fz_insertion_order_workspace_bytes(int Q, int N) {
    (void)Q; (void)N;
    return 0;  // the seen-bitmap lives in LDS
}

extern "C" int fz_insertion_order(const int32_t* const* orders_h, const int32_t* lens, int S, int Q, int N, int ld,
                                  int32_t* ins_order, int32_t* U, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (!orders_h || !lens || !ins_order || !U || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (N > 1048576) return FZ_ERR_UNSUPPORTED;
    if (Q == 0) return FZ_OK;
    InsArgs a{};
    a.S = S; a.Q = Q; a.N = N; a.ld = ld; a.lens = lens;
    for (int s = 0; s < S; ++s) {
        if (!orders_h[s]) return FZ_ERR_ARG;
        a.orders[s] = orders_h[s];
    }
    size_t lds = ((size_t)(N + 31) / 32 + 16) * 4;
    if (lds > 48 * 1024) {
        FZ_HIP_TRY(hipFuncSetAttribute((const void*)insertion_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    insertion_order_kernel<<<Q, 1024, lds, as_stream(stream)>>>(a, ins_order, U);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
