// fuse.hip -- Aggregator.fuse arithmetic on dense planes (reference: src/retrievers/hybrid.py:170-307).
//
//   K3  fz_row_stats_f32      per-(system,query) min/max or mean/unbiased-std     (hybrid.py:255-263)
//   K4  fz_fuse_nsf_f32       normalise -> weight -> sum, ONE pass over HBM         (hybrid.py:212-214,291,304)
//       fz_fuse_none_f64      'none' passthrough, float64                          (hybrid.py:280,291,304)
//       fz_fuse_wsum_f64      weight + sum with NumPy's scalar promotion (np.float64 grid weights, float64 planes)
//   K5b fz_fuse_rank_f64      rrf / bcf from rank planes, float64                  (hybrid.py:248-252,304)
//       fz_insertion_order    first-insertion order of the fused dict              (hybrid.py:301-304)
//
// All of these are HBM-bound streaming passes: 16-byte vector accesses, one workgroup per
// query row for the row-statistic kernels (the whole row lives in registers between the
// statistic and the transform, so each plane is read from HBM exactly once), flat grids
// for the purely elementwise ones.  Compiled with -ffp-contract=off: the reference's
// arithmetic is unfused and the oracle checks it bit for bit.
#include "common.h"
#include "nsf.h"

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
// K3: row statistics.  One workgroup per row; fp64 accumulation.  Rows of up to 32,768 aligned columns are read ONCE with 16-byte
// loads and kept in registers for the second (centred) pass of the z-score; longer or unaligned rows stream twice (second time
// from cache) with 4-byte loads.
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
    constexpr int E4 = 8;   // float4 per thread held in registers on the one-pass path
    const bool resident = THREADS == 1024 && N <= THREADS * E4 * 4 && ld % 4 == 0 && ((uintptr_t)scores % 16 == 0) && (!rank || (uintptr_t)rank % 16 == 0);
    if (resident && (norm == FZ_NORM_MINMAX || norm == FZ_NORM_ZSCORE)) {   // block-uniform
        float4 r[E4];
        uint32_t ok = 0u;                                       // bit 4*i + e: element e of r[i] is a listed column
#pragma unroll
        for (int i = 0; i < E4; ++i) {
            const int j0 = 4 * (i * THREADS + threadIdx.x);
            r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 < N) {
                r[i] = *reinterpret_cast<const float4*>(x + j0);  // columns [N, ld) of the last float4 are plane padding
                int4 vv = make_int4(0, 0, 0, 0);
                if (v) vv = *reinterpret_cast<const int4*>(v + j0);
                const int vr[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) ok |= (uint32_t)((j0 + e < N) && vr[e] >= 0) << (4 * i + e);
            }
        }
        if (norm == FZ_NORM_MINMAX) {
            float mn = INFINITY, mx = -INFINITY;
            bool nan = false;
#pragma unroll
            for (int i = 0; i < E4; ++i) {
                const float f[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((ok >> (4 * i + e)) & 1u) { nan |= (f[e] != f[e]); mn = fminf(mn, f[e]); mx = fmaxf(mx, f[e]); }
            }
            block_minmax<THREADS>(mn, mx, red_f);
            const int anynan = __syncthreads_or(nan ? 1 : 0);
            if (threadIdx.x == 0) {
                stat_a[row] = anynan ? __uint_as_float(0x7fc00000u) : mn;
                stat_b[row] = anynan ? __uint_as_float(0x7fc00000u) : mx;
            }
        } else {
            double sum = 0.0, cnt = 0.0;
#pragma unroll
            for (int i = 0; i < E4; ++i) {
                const float f[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((ok >> (4 * i + e)) & 1u) { sum += (double)f[e]; cnt += 1.0; }
            }
            sum = block_sum<THREADS>(sum, red_d);
            cnt = block_sum<THREADS>(cnt, red_d);
            const double mean = cnt > 0.0 ? sum / cnt : (double)NAN;
            double ss = 0.0;
#pragma unroll
            for (int i = 0; i < E4; ++i) {
                const float f[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((ok >> (4 * i + e)) & 1u) { const double d = (double)f[e] - mean; ss += d * d; }
            }
            ss = block_sum<THREADS>(ss, red_d);
            if (threadIdx.x == 0) {
                const double var = cnt > 1.0 ? ss / (cnt - 1.0) : (double)NAN;
                stat_a[row] = (float)mean;
                stat_b[row] = (float)sqrt(var);
            }
        }
        return;
    }
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
// Combined block reduction of up to 3 doubles (one barrier pair for all statistics of a row).
template <int THREADS, int NV>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double* red /* NV*THREADS/64 */) {
    constexpr int NW = THREADS / 64;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_reduce_sum(v[k]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[k * NW + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[k * NW + i];  // fixed order: deterministic
        v[k] = s;
    }
}

// Single-barrier block reductions: the partial slots are double-buffered by `parity` (the caller alternates it per
// use), so the write of round r+1 cannot overtake the reads of round r -- one s_barrier per reduction instead of two.
template <int THREADS>
__device__ __forceinline__ void block_minmax_nan(float& mn, float& mx, float& nanflag, float* red /* 2 x 3*THREADS/64 */, int parity) {
    constexpr int NW = THREADS / 64;
    static_assert(NW <= 16, "final combine uses 16 lanes");
    float* r = red + parity * 3 * NW;
    auto fmin_ = [](float x, float y) { return fminf(x, y); };
    auto fmax_ = [](float x, float y) { return fmaxf(x, y); };
    mn = wave_reduce_valu(mn, fmin_); mx = wave_reduce_valu(mx, fmax_); nanflag = wave_reduce_valu(nanflag, fmax_);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { r[w] = mn; r[NW + w] = mx; r[2 * NW + w] = nanflag; }
    lds_barrier();
    // every wave folds the NW partials with its own lanes (lane L takes partial L): 3 LDS reads + 4 DPP steps per
    // thread instead of 3*NW reads and 3*NW min/max
    const int L = lane & 15;
    mn = L < NW ? r[L] : INFINITY; mx = L < NW ? r[NW + L] : -INFINITY; nanflag = L < NW ? r[2 * NW + L] : 0.f;
    mn = row16_reduce_valu(mn, fmin_); mx = row16_reduce_valu(mx, fmax_); nanflag = row16_reduce_valu(nanflag, fmax_);
}
template <int THREADS, int NV>
__device__ __forceinline__ void block_sum_n_nodrain(double (&v)[NV], double* red /* 2 x NV*THREADS/64 */, int parity) {
    constexpr int NW = THREADS / 64;
    double* r = red + parity * NV * NW;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_reduce_sum(v[k]);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) r[k * NW + w] = v[k];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += r[k * NW + i];  // fixed order: deterministic
        v[k] = s;
    }
}

// K4.  One workgroup of T threads per query; thread t owns columns {4*(t + T*i) .. +3}, i < E4.
//   * the accumulator and the CURRENT system's row live in registers (2*4*E4 VGPRs);
//   * DMA = true: the NEXT system's row streams HBM -> LDS by global_load_lds (no VGPRs, asynchronous) while the
//     current one is reduced across the workgroup and transformed; each wave later reads back exactly the 1-KiB
//     pieces it issued itself, so the only synchronisation is that wave's own vmcnt;
//   * every plane is read from HBM once: (S+1)*N*4 bytes per query.
// VALID = rank planes carry validity (partial lists); needs VEC-style planes: ld % 4 == 0, 16-B aligned bases,
// so a float4 starting below N never leaves the padded row.
template <int NORM, int T /* threads */, int E4 /* float4 per thread */, bool VEC, bool VALID, bool DMA>
__global__ __launch_bounds__(T) void fuse_nsf_row_kernel(NsfArgs a, float* __restrict__ fused) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* rowbuf = reinterpret_cast<float*>(smem_raw);                                 // DMA: [T*E4*4] floats
    double* red_d = reinterpret_cast<double*>(smem_raw + (DMA ? (size_t)T * E4 * 16 : 0)); // [2][3*T/64]
    float* red_f = reinterpret_cast<float*>(red_d + 2 * 3 * T / 64);                    // [2][3*T/64]
    const int N = a.N;
    const int lane = threadIdx.x & 63;
    // persistent workgroups (DMA): rows blockIdx.x, blockIdx.x + gridDim.x, ...; the first system of the NEXT row is
    // already streaming into LDS while the current row's last system is transformed and the result stored, so there is
    // no launch / first-load bubble at row boundaries (one 1024-thread workgroup fills a CU: nothing else could hide it)
    int red_parity = 0;
    if (DMA && blockIdx.x < a.Q) {
        const float* __restrict__ x0 = a.planes[0] + (size_t)blockIdx.x * a.ld;
#pragma unroll
        for (int i = 0; i < E4; ++i) {
            const int j0 = 4 * (threadIdx.x + T * i);
            if (j0 < N) __builtin_amdgcn_global_load_lds(x0 + j0, (__attribute__((address_space(3))) void*)(rowbuf + 4 * (threadIdx.x - lane + T * i)), 16, 0, FZ_CPOL_NT);
        }
    }
  for (int q = blockIdx.x; q < a.Q; q += gridDim.x) {
    const size_t rowoff = (size_t)q * a.ld;

    float acc[E4][4];
    uint64_t present = 0ull;   // VALID only: some system lists the column (bit 4*i+c)
#pragma unroll
    for (int i = 0; i < E4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = 0.0f;

    float v[E4][4];
    uint64_t ok = 0ull;        // column inside the row (and listed by the system when VALID)

    auto dma_row = [&](int s, size_t roff) {   // HBM -> LDS, one 1-KiB piece per wave-instruction; lanes past the row end stay off
        const float* __restrict__ x = a.planes[s] + roff;
#pragma unroll
        for (int i = 0; i < E4; ++i) {
            const int j0 = 4 * (threadIdx.x + T * i);
            if (j0 < N) {
                // LDS destination = wave-uniform base + lane*16: pass the address of lane 0's slot
                __builtin_amdgcn_global_load_lds(x + j0, (__attribute__((address_space(3))) void*)(rowbuf + 4 * (threadIdx.x - lane + T * i)), 16, 0, FZ_CPOL_NT);
            }
        }
    };
    auto take_row = [&](int s) {   // current row -> registers (+ validity mask)
        const float* __restrict__ x = a.planes[s] + rowoff;
        const int32_t* __restrict__ rk = (VALID && a.ranks[s] && !a.vbits[s]) ? a.ranks[s] + rowoff : nullptr;
        const bool partial = VALID && (a.ranks[s] || a.vbits[s]);
        ok = 0ull;
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own pieces have landed
#pragma unroll
        for (int i = 0; i < E4; ++i) {
            const int j0 = 4 * (threadIdx.x + T * i);
            uint32_t m = 0u;
            if (VEC) {
                if (j0 < N) {
                    const float4 f = DMA ? *reinterpret_cast<const float4*>(rowbuf + 4 * (threadIdx.x + T * i))
                                         : *reinterpret_cast<const float4*>(x + j0);
                    v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
                    const int rem = N - j0;
                    m = rem >= 4 ? 0xfu : ((1u << rem) - 1u);
                    if (partial) m &= valid_nibble(a, s, q, rowoff, j0);
                } else {
                    v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool in = j0 + c < N;
                    v[i][c] = in ? x[j0 + c] : 0.0f;
                    const bool listed = !partial || (a.vbits[s] ? ((a.vbits[s][(size_t)q * a.ldb + ((j0 + c) >> 5)] >> ((j0 + c) & 31)) & 1u) != 0u : rk[j0 + c] >= 0);
                    if (in && listed) m |= 1u << c;
                }
            }
            ok |= (uint64_t)m << (4 * i);
        }
    };

    for (int s = 0; s < a.S; ++s) {
        take_row(s);
        if (DMA) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // our ds_reads of this row are done before the DMA overwrites the slots
            if (s + 1 < a.S) dma_row(s + 1, rowoff);             // in flight during the reduction + transform below
            else if (q + (int)gridDim.x < a.Q) dma_row(0, (size_t)(q + gridDim.x) * a.ld);   // next row's first system
        }
        float sa = 0.f, sb = 0.f;
        if (NORM == FZ_NORM_MINMAX) {
            float mn = INFINITY, mx = -INFINITY, nanf_ = 0.f;
#pragma unroll
            for (int i = 0; i < E4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {   // branch-free: unlisted / padding columns contribute the identity
                    const bool in = (ok >> (4 * i + c)) & 1ull;
                    const float x = v[i][c];
                    mn = fminf(mn, in ? x : INFINITY);
                    mx = fmaxf(mx, in ? x : -INFINITY);
                    nanf_ = (in && x != x) ? 1.f : nanf_;
                }
            block_minmax_nan<T>(mn, mx, nanf_, red_f, red_parity); red_parity ^= 1;
            sa = nanf_ > 0.f ? __uint_as_float(0x7fc00000u) : mn;   // torch.min/max propagate NaN
            sb = nanf_ > 0.f ? __uint_as_float(0x7fc00000u) : mx;
        } else if (NORM == FZ_NORM_ZSCORE) {
            // one pass per wave, shifted by that wave's own first register value (any finite sample of the row keeps the
            // one-pass sums stable; readfirstlane: no memory access), then the 16 per-wave (n, mean, M2) triples are
            // merged with the parallel-variance update (Chan et al.), in wave order, by every thread identically:
            //   d = x - x0_w;  mean_w = x0_w + sum(d)/n_w;  M2_w = sum(d^2) - sum(d)^2/n_w          (all fp64)
            float x0c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[0][0])));
            if (!(x0c == x0c) || fabsf(x0c) == INFINITY) x0c = 0.f;
            const double x0 = (double)x0c;
            double s1 = 0.0, s2 = 0.0;   // fp64 throughout: fp32 per-thread partials (0.164 -> 0.152 ms) cost the 2e-6 parity on short lists
#pragma unroll
            for (int i = 0; i < E4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool in = (ok >> (4 * i + c)) & 1ull;
                    const double d = in ? (double)v[i][c] - x0 : 0.0;
                    s1 += d; s2 += d * d;
                }
            double cnt = (double)__popcll(ok);
            auto add_ = [](double x, double y) { return x + y; };
            s1 = wave_reduce_valu(s1, add_); s2 = wave_reduce_valu(s2, add_); cnt = wave_reduce_valu(cnt, add_);
            {
                constexpr int NW = T / 64;
                double* r = red_d + red_parity * 3 * NW;
                red_parity ^= 1;
                if (lane == 0) {
                    const int w = threadIdx.x >> 6;
                    r[w] = cnt;
                    r[NW + w] = cnt > 0.0 ? x0 + s1 / cnt : 0.0;
                    r[2 * NW + w] = cnt > 0.0 ? s2 - s1 * s1 / cnt : 0.0;
                }
                lds_barrier();
                // lane L of every wave takes partial L; a 4-step tree towards lane 0 merges (n, mean, M2) pairs with the
                // parallel-variance update; lane 0's result is broadcast, so every thread of every wave uses the same bits
                const int L = lane & 15;
                double n = L < NW ? r[L] : 0.0, mean = L < NW ? r[NW + L] : 0.0, M2 = L < NW ? r[2 * NW + L] : 0.0;
                auto merge_step = [&](double nb, double mb, double m2b) {
                    const double nn = n + nb;
                    if (nb > 0.0) {
                        const double delta = mb - mean, f = nb / nn;
                        mean = (n > 0.0) ? mean + delta * f : mb;
                        M2 = M2 + m2b + delta * delta * (n * f);
                        n = nn;
                    }
                };
                // lane L += lane L + o, o = 8, 4, 2, 1, inside the 16-lane row: DPP row_shl (lanes shifted in from past the row read 0
                // and only reach lanes that lane 0 never depends on)
                merge_step(dpp_f64<0x108>(n), dpp_f64<0x108>(mean), dpp_f64<0x108>(M2));
                merge_step(dpp_f64<0x104>(n), dpp_f64<0x104>(mean), dpp_f64<0x104>(M2));
                merge_step(dpp_f64<0x102>(n), dpp_f64<0x102>(mean), dpp_f64<0x102>(M2));
                merge_step(dpp_f64<0x101>(n), dpp_f64<0x101>(mean), dpp_f64<0x101>(M2));
                auto bcast = [](double x) -> double {
                    const long long b = __double_as_longlong(x);
                    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
                    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
                };
                n = bcast(n); mean = bcast(mean); M2 = bcast(M2);
                const double var = n > 1.0 ? M2 / (n - 1.0) : (double)NAN;
                sa = (float)(n > 0.0 ? mean : (double)NAN);
                sb = (float)sqrt(var < 0.0 ? 0.0 : var);
            }
        }
        const float w = a.w[s];
        const float* __restrict__ distr = a.distr[s];
        const int P = a.P[s];
#pragma unroll
        for (int i = 0; i < E4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                // percentile modes walk a table: keep them predicated; the arithmetic modes are branch-free
                // (!VALID: padding columns compute garbage that is never stored)
                const bool in = !VALID || ((ok >> (4 * i + c)) & 1ull);
                if (NORM == FZ_NORM_PERCENTILE || NORM == FZ_NORM_NCE) {
                    if (in) {
                        const float t = transform<NORM>(v[i][c], sa, sb, distr, P);
                        const float prod = t * w;
                        acc[i][c] = acc[i][c] + prod;
                    }
                } else {
                    const float t = transform<NORM>(v[i][c], sa, sb, distr, P);
                    const float prod = t * w;                       // fl32(t * fl32(w))      hybrid.py:291 under NumPy 2
                    acc[i][c] = in ? acc[i][c] + prod : acc[i][c];  // fl32(acc + prod)        hybrid.py:304
                }
            }
        if (VALID) present |= ok;
    }
    float* __restrict__ out = fused + rowoff;
#pragma unroll
    for (int i = 0; i < E4; ++i) {
        const int j0 = 4 * (threadIdx.x + T * i);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (!VALID || ((present >> (4 * i + c)) & 1ull)) ? acc[i][c] : -INFINITY;
        if (VEC) {
            if (j0 < N) *reinterpret_cast<float4*>(out + j0) = make_float4(o[0], o[1], o[2], o[3]);   // may touch padding columns [N, ld)
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j0 + c < N) out[j0 + c] = o[c];
        }
    }
  }   // persistent row loop
}

// general-N path: statistics from a separate pass (stat arrays [S][Q]), then elementwise.
template <int NORM>
__global__ __launch_bounds__(256) void fuse_nsf_elem_kernel(NsfArgs a, int Q, float* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.N; j += gridDim.x * blockDim.x) {
        float acc = 0.0f;
        bool present = false;
        for (int s = 0; s < a.S; ++s) {
            if (a.vbits[s] ? !((a.vbits[s][(size_t)q * a.ldb + (j >> 5)] >> (j & 31)) & 1u) : (a.ranks[s] && a.ranks[s][rowoff + j] < 0)) continue;
            float t = transform<NORM>(a.planes[s][rowoff + j], a.sa[s] ? a.sa[s][q] : 0.f, a.sb[s] ? a.sb[s][q] : 0.f, a.distr[s], a.P[s]);
            float prod = t * a.w[s];
            acc = acc + prod;
            present = true;
        }
        fused[rowoff + j] = present ? acc : -INFINITY;
    }
}

// The same, 4 columns per thread (16-B loads and stores); needs ld % 4 == 0 and 16-B aligned planes.  With the statistics in
// hand the whole fusion is ONE flat streaming pass -- and for min-max on RANKED systems they are: min and max of a list sorted
// by score are its last and first entries (minmax_from_order_kernel), no reduction over the row at all.
template <int NORM>
__global__ __launch_bounds__(256) void fuse_nsf_elem4_kernel(NsfArgs a, int Q, float* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    const int j0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (j0 >= a.N) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    bool present[4] = {false, false, false, false};
    for (int s = 0; s < a.S; ++s) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.planes[s] + rowoff + j0));   // streamed once
        const float v[4] = {f.x, f.y, f.z, f.w};
        const uint32_t nib = valid_nibble(a, s, q, rowoff, j0);
        const float sa = a.sa[s] ? a.sa[s][q] : 0.f, sb = a.sb[s] ? a.sb[s][q] : 0.f, w = a.w[s];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (((nib >> c) & 1u) && j0 + c < a.N) {
                const float t = transform<NORM>(v[c], sa, sb, a.distr[s], a.P[s]);
                const float prod = t * w;
                acc[c] = acc[c] + prod;
                present[c] = true;
            }
        }
    }
    // columns [N, ld) of the last float4 are padding of the plane: written, never read
    // plain store: the fused plane is read straight back by the ordering sort
    *reinterpret_cast<float4*>(fused + rowoff + j0) = make_float4(present[0] ? acc[0] : -INFINITY, present[1] ? acc[1] : -INFINITY,
                                                                 present[2] ? acc[2] : -INFINITY, present[3] ? acc[3] : -INFINITY);
}

// percentile-rank / normal-curve-equivalent fusion with the quantile tables in LDS.  Neither needs row statistics, so this
// is a flat pass; what costs is the nearest-entry search: ~log2(P) dependent reads per score -- from global memory (the row
// kernel's path) 1.2 ms at P = 1001, and the NCE's double-precision erfinv per score another 6.5 ms.  Here every persistent
// workgroup loads the S tables into LDS once and, for NCE, tabulates the value of every table INDEX next to them (the
// transform depends on the score only through the index of its nearest entry), then streams (row, 1024-column) items.
// Same nearest entry, same float expressions: bit-identical results.
struct TableArgs {
    int off[FZ_MAX_SYSTEMS];        // start of system s's table in the LDS array
    int total;                      // sum of P
    int uoff[FZ_MAX_SYSTEMS + 1];   // start of system s's DISTINCT values: multiples of 4, room for P rounded up to 4 plus 8 pads
    int utotal;
};

// The search itself: binary search costs ~log2(P) + 3 DEPENDENT LDS reads per score (the kernel ran at 0.17 of HBM, bound
// by that latency chain).  An equi-width look-up table over [tab[0], tab[P-1]] with LUT_B buckets per system brackets the
// answer first:  lut[b] = #entries below the bucket's lower edge e_b = tab[0] + b * width.  A score whose bucket comes out
// as b (at most one bucket off after rounding) has its "last entry <= s" in [lut[b-1] - 1, lut[b+2]), usually 2-3 slots;
// the bracket is VERIFIED against the table (two of the reads the search needs anyway) and widened to the full table when
// rounding broke it, so the result is that of the plain search in every case.  first[k] = index of the first entry equal
// to tab[k] replaces the walk to the left over duplicated quantiles (first minimum of |tab - s|, hybrid.py:272-275).
constexpr int LUT_B = 2048;

struct TableSys {
    const float* tab; const uint16_t* lut; const uint16_t* first;
    float lo_v, inv_w; int P; bool use_lut;
};

__device__ __forceinline__ int nearest_entry_lut(const TableSys& y, float s) {
    if (s != s || fabsf(s) == INFINITY) return 0;   // NaN: argmin of an all-NaN column; +-inf: every distance is inf -> first index
    const int P = y.P;
    int lo = -1, hi = P;
    if (y.use_lut) {
        const float t = (s - y.lo_v) * y.inv_w;
        if (t < 0.f) return 0;                       // below the first entry: it is the nearest
        // t >= LUT_B does NOT mean "at or above the last entry": a score a rounding below the table's maximum lands there too, with other
        // entries between it and the maximum (round 4, found by the soak on a table of two far-apart clusters) -- it takes the last bucket's bracket
        const int b = t < (float)LUT_B ? (int)t : LUT_B - 1;
        lo = (int)y.lut[b > 0 ? b - 1 : 0] - 1;
        hi = (int)y.lut[b + 2 < LUT_B ? b + 2 : LUT_B];
        if (lo >= 0 && !(y.tab[lo] <= s)) lo = -1;   // rounding broke the bracket: widen (never seen, kept for exactness)
        if (hi < P && !(y.tab[hi] > s)) hi = P;
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (y.tab[mid] <= s) lo = mid; else hi = mid;
    }
    if (lo < 0) return 0;
    const float dl = fabsf(y.tab[lo] - s);
    if (lo + 1 < P && fabsf(y.tab[lo + 1] - s) < dl) return lo + 1;
    // first minimum of the ROUNDED distances (torch.argmin over |distr - s| in float32): skip the duplicates of tab[lo] in one
    // step, then keep walking while a smaller entry's distance rounds to the same float (|s| >> the table's spacing)
    int k = (int)y.first[lo];
    while (k > 0 && fabsf(y.tab[k - 1] - s) == dl) --k;
    return k;
}

// The common case without a search: the table is stored a second time with its duplicates removed (utab; uval = the transformed
// value of a distinct quantile's FIRST occurrence in the original table, which is the "first minimum" among equal entries), a
// per-system look-up table maps a score's bucket to a GUESS g (the exact answer at the bucket's centre), and the score looks at
// the WINDOW of eight distinct values around the guess -- two aligned 16-byte LDS reads: d_j = |utab[i0 + j] - s| in float32 like
// the reference, j* = first minimum of the window.  j* is the first minimum of the whole table whenever the window shows both
// slopes of the (weakly) V-shaped distance sequence:
//     left :  j* > 0 (then d_0 > d_j*, so the window starts on the descending side and everything before it is >= d_0), or i0 = 0;
//     right:  d_7 > d_j* (the window ends on the ascending side; slots past the table hold +inf), or d_j* = 0.
// Anything else -- a distance plateau reaching a window edge because |s| dwarfs the table's spacing, a guess that is off, NaN /
// inf -- takes the exact search above.  No branch in the common case and four LDS reads (guess, 2 x 16 bytes, value) instead of
// ~11 dependent ones: the kernel is bound by bank-conflicted random LDS reads, so this is what moved it (0.48 -> 0.27 ms at S = 4,
// P = 1001).  The same nearest entry as the plain search in every case.
constexpr int TBL_WIN = 8;

template <bool NCE, int TPB>
__global__ __launch_bounds__(TPB) void fuse_nsf_table_kernel(NsfArgs a, TableArgs t, int Q, float* __restrict__ fused) {
    extern __shared__ __attribute__((aligned(16))) float tabs[];
    float* utab = tabs;                                              // [utotal] distinct quantiles per system, 16-byte aligned, +inf padded
    float* uval = tabs + t.utotal;                                   // [utotal] value of a distinct quantile's FIRST index
    float* orig = tabs + 2 * t.utotal;                               // [total] quantiles as given
    float* val = orig + t.total;                                     // [total] value of every table INDEX: k / P, or its NCE transform
    const int tot2 = (t.total + 1) & ~1;
    uint16_t* first = reinterpret_cast<uint16_t*>(val + t.total);
    uint16_t* uix = first + tot2;                                    // original index -> index of its value in utab
    uint16_t* lut = uix + tot2;
    uint16_t* guess = lut + a.S * (LUT_B + 1);
    __shared__ float sys_lo[FZ_MAX_SYSTEMS], sys_inv[FZ_MAX_SYSTEMS];   // per system: first entry, buckets per unit (0 = no look-up table)
    __shared__ int sys_U[FZ_MAX_SYSTEMS];                               // distinct values
    __shared__ int scan_part[TPB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = 0; s < a.S; ++s)
        for (int k = threadIdx.x; k < a.P[s]; k += TPB) {
            orig[t.off[s] + k] = a.distr[s][k];
            const float pr = (float)k / (float)a.P[s];                   // hybrid.py:275
            if (NCE) {   // transform<FZ_NORM_NCE> as a function of the index
                const float p = pr / 100.0f;
                const float y = 2.0f * p - 1.0f;
                const float z = (float)(erfinv((double)y) * 1.4142135623730951);
                val[t.off[s] + k] = z * 21.06f + 50.0f;
            } else val[t.off[s] + k] = pr;
        }
    __syncthreads();
    auto lower_bound = [&](const float* tab, int P, float x) -> int {   // #entries < x
        int lo = 0, hi = P;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (tab[mid] < x) lo = mid + 1; else hi = mid; }
        return lo;
    };
    for (int s = 0; s < a.S; ++s) {
        const float* tab = orig + t.off[s];
        const int P = a.P[s];
        const float lo_v = tab[0], hi_v = tab[P - 1];
        const float width = (hi_v - lo_v) / (float)LUT_B;
        const bool ok = P >= 2 && width > 0.f && fabsf(width) != INFINITY && lo_v == lo_v && hi_v == hi_v;
        if (threadIdx.x == 0) { sys_lo[s] = lo_v; sys_inv[s] = ok ? (float)LUT_B / (hi_v - lo_v) : 0.f; }
        for (int k = threadIdx.x; k < P; k += TPB) first[t.off[s] + k] = (uint16_t)lower_bound(tab, P, tab[k]);
        if (ok)
            for (int b = threadIdx.x; b <= LUT_B; b += TPB)
                lut[s * (LUT_B + 1) + b] = (uint16_t)(b == LUT_B ? P : lower_bound(tab, P, lo_v + (float)b * width));
        __syncthreads();
        // distinct values: k heads a run iff first[k] == k; its slot = number of heads before it (workgroup scan over contiguous pieces)
        const int per = (P + TPB - 1) / TPB, k0 = threadIdx.x * per, k1 = min(P, k0 + per);
        int heads = 0;
        for (int k = k0; k < k1; ++k) heads += (int)first[t.off[s] + k] == k;
        const int incl = (int)wave_incl_scan_u32((uint32_t)heads, lane);
        if (lane == 63) scan_part[wave] = incl;
        __syncthreads();
        int u = incl - heads;
        for (int w = 0; w < wave; ++w) u += scan_part[w];
        if (threadIdx.x == TPB - 1) sys_U[s] = u + heads;
        for (int k = k0; k < k1; ++k) {
            if ((int)first[t.off[s] + k] == k) { utab[t.uoff[s] + u] = tab[k]; uval[t.uoff[s] + u] = val[t.off[s] + k]; ++u; }
            uix[t.off[s] + k] = (uint16_t)(u - 1);
        }
        __syncthreads();
        for (int j = sys_U[s] + threadIdx.x; j < t.uoff[s + 1] - t.uoff[s]; j += TPB) { utab[t.uoff[s] + j] = INFINITY; uval[t.uoff[s] + j] = 0.f; }
    }
    for (int s = 0; s < a.S; ++s) {   // the guesses: the exact search at every bucket's centre
        const float inv_w = sys_inv[s];
        if (inv_w == 0.f) continue;
        const TableSys y{orig + t.off[s], lut + s * (LUT_B + 1), first + t.off[s], sys_lo[s], inv_w, a.P[s], true};
        const float width = (orig[t.off[s] + a.P[s] - 1] - sys_lo[s]) / (float)LUT_B;
        for (int b = threadIdx.x; b < LUT_B; b += TPB)
            guess[s * LUT_B + b] = uix[t.off[s] + nearest_entry_lut(y, sys_lo[s] + ((float)b + 0.5f) * width)];
    }
    __syncthreads();
    const int chunks = (a.N + 4 * TPB - 1) / (4 * TPB);
    const long long items = (long long)Q * chunks;
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int q = (int)(it / chunks), c = (int)(it - (long long)q * chunks);
        const size_t rowoff = (size_t)q * a.ld;
        const int j0 = 4 * (c * TPB + threadIdx.x);
        if (j0 >= a.N) continue;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        bool present[4] = {false, false, false, false};
        for (int s = 0; s < a.S; ++s) {
            const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.planes[s] + rowoff + j0));   // streamed once
            const float v[4] = {f.x, f.y, f.z, f.w};
            const uint32_t nib = valid_nibble(a, s, q, rowoff, j0);
            const int P = a.P[s];
            const float w = a.w[s];
            const float inv_w = sys_inv[s], lo_v = sys_lo[s];
            const int U = sys_U[s];
            const bool windowed = inv_w != 0.f && U >= 2;            // uniform
            const TableSys y{orig + t.off[s], lut + s * (LUT_B + 1), first + t.off[s], lo_v, inv_w, P, inv_w != 0.f};
            const float* ut = utab + t.uoff[s];
            const float* uv = uval + t.uoff[s];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (((nib >> e) & 1u) && j0 + e < a.N) {
                    const float x = v[e];
                    float tr;
                    bool ok = false;
                    if (windowed && fabsf(x) < INFINITY) {          // (false for NaN)
                        const float tb = fminf(fmaxf((x - lo_v) * inv_w, 0.f), (float)(LUT_B - 1));
                        const int g = (int)guess[s * LUT_B + (int)tb];
                        const int i0 = max(g - 2, 0) & ~3;           // two 16-byte reads: the guess has >= 2 neighbours on either side
                        const float4 w0 = *reinterpret_cast<const float4*>(ut + i0), w1 = *reinterpret_cast<const float4*>(ut + i0 + 4);
                        const float d[TBL_WIN] = {fabsf(w0.x - x), fabsf(w0.y - x), fabsf(w0.z - x), fabsf(w0.w - x),
                                                  fabsf(w1.x - x), fabsf(w1.y - x), fabsf(w1.z - x), fabsf(w1.w - x)};
                        float db = d[0];
                        int jb = 0;
#pragma unroll
                        for (int j = 1; j < TBL_WIN; ++j) { const bool lt = d[j] < db; db = lt ? d[j] : db; jb = lt ? j : jb; }
                        ok = (jb > 0 || i0 == 0) && (d[TBL_WIN - 1] > db || db == 0.f);   // (past the table: +inf entries)
                        tr = uv[i0 + jb];
                    }
                    if (!ok) tr = val[t.off[s] + nearest_entry_lut(y, x)];
                    const float prod = tr * w;
                    acc[e] = acc[e] + prod;
                    present[e] = true;
                }
            }
        }
        *reinterpret_cast<float4*>(fused + rowoff + j0) = make_float4(present[0] ? acc[0] : -INFINITY, present[1] ? acc[1] : -INFINITY,
                                                                     present[2] ? acc[2] : -INFINITY, present[3] ? acc[3] : -INFINITY);
    }
}

// layout of the LDS-resident tables; false when they cannot all live in LDS (tables too long, planes not 16-B aligned)
static bool nsf_tables_plan(const NsfArgs& a, const float* fused, TableArgs& t, size_t& lds) {
    int total = 0;
    bool vec = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    for (int s = 0; s < a.S; ++s) {
        t.off[s] = total;
        total += a.P[s];
        if (a.P[s] > 65535) return false;   // uint16 indices
        vec = vec && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    }
    t.total = total;
    int utotal = 0;
    for (int s = 0; s < a.S; ++s) { t.uoff[s] = utotal; utotal += ((a.P[s] + 3) & ~3) + 8; }
    t.uoff[a.S] = utotal;
    t.utotal = utotal;
    lds = (size_t)(2 * utotal + 2 * total) * 4 + (size_t)((total + 1) & ~1) * 2 * 2 + (size_t)a.S * (2 * LUT_B + 1) * 2 + 16;
    return vec && lds <= 144 * 1024;
}
bool nsf_tables_fit_lds(const NsfArgs& a, const float* fused) {
    TableArgs t{};
    size_t lds = 0;
    return nsf_tables_plan(a, fused, t, lds);
}

// returns 1 when the LDS-table kernel cannot take the call
int launch_nsf_tables(const NsfArgs& a, bool nce, int Q, float* fused, hipStream_t st) {
    TableArgs t{};
    size_t lds = 0;
    if (!nsf_tables_plan(a, fused, t, lds)) return 1;
    // one 1024-thread workgroup per CU: the tables, look-up tables and (NCE) per-index values are built once per workgroup and
    // serve 16 waves
    constexpr int tpb = 1024;
    const long long items = (long long)Q * ((a.N + 4 * tpb - 1) / (4 * tpb));
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
    static unsigned long long set_nce = 0ull, set_pr = 0ull;
    if (nce) {
        if (raise_lds_limit((const void*)fuse_nsf_table_kernel<true, 1024>, lds, set_nce) != FZ_OK) return 1;
        fuse_nsf_table_kernel<true, 1024><<<grid, 1024, lds, st>>>(a, t, Q, fused);
    } else {
        if (raise_lds_limit((const void*)fuse_nsf_table_kernel<false, 1024>, lds, set_pr) != FZ_OK) return 1;
        fuse_nsf_table_kernel<false, 1024><<<grid, 1024, lds, st>>>(a, t, Q, fused);
    }
    return 0;
}

// min / max of every ranked list from its ends: order[row][0] is the best-scored document, order[row][len-1] the worst
// (stable descending sort, NaN first: a NaN anywhere makes both NaN, as torch.min / torch.max propagate it).
__global__ void minmax_from_order_kernel(const float* __restrict__ scores, const int32_t* __restrict__ order, const int32_t* __restrict__ lens,
                                         int rows, int N, int ld, float* __restrict__ mn, float* __restrict__ mx) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int len = lens ? lens[row] : N;
    float lo = 0.f, hi = 0.f;
    if (len > 0) {
        hi = scores[(size_t)row * ld + order[(size_t)row * ld]];
        lo = scores[(size_t)row * ld + order[(size_t)row * ld + len - 1]];
        if (hi != hi) lo = hi;
    }
    mn[row] = lo; mx[row] = hi;
}

// -------------------------------------------------------------------------------------
// 'none' passthrough in float64 and rank fusion in float64: flat elementwise kernels,
// 4 columns per thread (16-B rank/score loads, 32-B stores).
// -------------------------------------------------------------------------------------
struct ElemArgs {
    const void* planes[FZ_MAX_SYSTEMS];   // float32 planes, or float64 where bit s of f64_mask is set (fuse_wsum only)
    const int32_t* ranks[FZ_MAX_SYSTEMS];
    double w[FZ_MAX_SYSTEMS];
    int S, N, ld, Q, method;
    const int32_t* lens;
    unsigned f64_mask;      // plane s holds doubles (BM25 scores, raw Python floats of a host list)
    unsigned narrow_mask;   // weight s is a weak / float32 scalar: fl32 product, and the sum stays fl32 until a wide product arrives
};

// Weight-and-sum with NumPy-2's scalar promotion (hybrid.py:291,304) -> float64 plane.
//   'none' / unknown normalisation: raw Python-float scores, float64 throughout (narrow_mask = 0);
//   np.float64 weights (the tuning grid, hybrid.py:405-409): np.float32 score * np.float64 -> float64 products;
//   the per-document accumulator (Python 0.0, weak) is float32 until ITS first float64 product, float64 after.
// fl32(fl64(a) op fl64(b)) == fl32(a op b) for floats a, b (53 >= 2*24+2: double rounding is innocuous), so the narrow
// steps are computed on doubles and rounded.  MIXED = some plane is float64 or some weight narrow (otherwise the plain
// float32-planes / float64-arithmetic passthrough).
template <bool VEC, bool MIXED>
__global__ __launch_bounds__(256) void fuse_wsum_kernel(ElemArgs a, double* __restrict__ fused) {
    const int q = blockIdx.y;
    const size_t rowoff = (size_t)q * a.ld;
    const int j0 = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (j0 >= a.N) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    bool present[4] = {false, false, false, false};
    bool wide[4] = {false, false, false, false};
    const bool full = VEC && (j0 + 3 < a.N);
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef double d2v __attribute__((ext_vector_type(2)));
    typedef int i4v __attribute__((ext_vector_type(4)));
    for (int s = 0; s < a.S; ++s) {
        double v[4]; int r[4] = {0, 0, 0, 0};
        const bool p64 = MIXED && ((a.f64_mask >> s) & 1u);
        const bool narrow = MIXED && !p64 && ((a.narrow_mask >> s) & 1u);
        if (full) {   // streamed once: non-temporal
            if (p64) {
                const double* x = reinterpret_cast<const double*>(a.planes[s]) + rowoff + j0;
                const d2v lo = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(x));
                const d2v hi = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(x + 2));
                v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
            } else {
                const f4v f = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(a.planes[s]) + rowoff + j0));
                v[0] = (double)f.x; v[1] = (double)f.y; v[2] = (double)f.z; v[3] = (double)f.w;
            }
            if (a.ranks[s]) { const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v*>(a.ranks[s] + rowoff + j0)); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool in = j0 + c < a.N;
                v[c] = !in ? 0.0 : (p64 ? reinterpret_cast<const double*>(a.planes[s])[rowoff + j0 + c]
                                        : (double)reinterpret_cast<const float*>(a.planes[s])[rowoff + j0 + c]);
                r[c] = in ? (a.ranks[s] ? a.ranks[s][rowoff + j0 + c] : 0) : -1;
            }
        }
        const double w = narrow ? (double)(float)a.w[s] : a.w[s];   // weak scalar: np.float32(w)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (r[c] >= 0) {
                double prod = v[c] * w;
                if (narrow) prod = (double)(float)prod; else wide[c] = true;
                acc[c] = acc[c] + prod;
                if (MIXED && !wide[c]) acc[c] = (double)(float)acc[c];
                present[c] = true;
            }
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
        if (full) {   // streamed once: non-temporal
            typedef int i4v __attribute__((ext_vector_type(4)));
            const i4v t = __builtin_nontemporal_load(reinterpret_cast<const i4v*>(a.ranks[s] + rowoff + j0));
            r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
        } else {
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

__global__ __launch_bounds__(1024) void insertion_order_kernel(InsArgs a, int32_t* __restrict__ ins_order, int32_t* __restrict__ U,
                                                              int32_t* __restrict__ pos /* nullable: the inverse, pre-filled with -1 */) {
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
            if (isnew) {
                ins_order[rowoff + base + woff + below] = j;
                if (pos) pos[rowoff + j] = base + woff + below;
            }
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
    if (rows < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;                   // empty tensors carry null pointers
    if (!scores || !stat_a || !stat_b) return FZ_ERR_ARG;
    row_stats_kernel<1024><<<rows, 1024, 0, as_stream(stream)>>>(scores, rank, N, ld, norm, stat_a, stat_b);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

template <int NORM, int TT, int E4, bool VEC, bool VALID, bool DMA>
static int launch_nsf_cfg(const NsfArgs& a, int Q, float* fused, hipStream_t st) {
    constexpr size_t lds = (DMA ? (size_t)TT * E4 * 16 : 0) + 2 * 3 * (TT / 64) * (sizeof(double) + sizeof(float)) + 64;
    static unsigned long long lds_set = 0ull;   // per instantiation
    if (lds > 48 * 1024)
        if (int rc = raise_lds_limit((const void*)fuse_nsf_row_kernel<NORM, TT, E4, VEC, VALID, DMA>, lds, lds_set)) return rc;
    // DMA variants are persistent: one workgroup per CU (256 CUs) walks the rows; the others launch one per row
    const int grid = DMA ? (Q < 256 ? Q : 256) : Q;
    fuse_nsf_row_kernel<NORM, TT, E4, VEC, VALID, DMA><<<grid, TT, lds, st>>>(a, fused);
    return 0;
}

template <int NORM>
static int launch_nsf(const NsfArgs& a, int Q, float* fused, hipStream_t st) {
    bool al = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    bool valid = false;
    for (int s = 0; s < a.S; ++s) {
        al = al && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
        valid = valid || a.ranks[s] || a.vbits[s];
    }
#define FZ_NSF_CASE(TT, E4)                                                                           \
    if (a.N <= TT * E4 * 4) {                                                                         \
        if (al && !valid) return launch_nsf_cfg<NORM, TT, E4, true, false, true>(a, Q, fused, st);    \
        /* partial lists: the persistent + prefetching form pays for z-score / arctan (0.244 -> 0.195 ms at S = 4, one partial \
           system), not for min-max (0.218 -> 0.297), whose ranked form takes the list ends + one flat pass anyway */        \
        if (al && NORM != FZ_NORM_MINMAX) return launch_nsf_cfg<NORM, TT, E4, true, true, true>(a, Q, fused, st);  \
        if (al) return launch_nsf_cfg<NORM, TT, E4, true, true, false>(a, Q, fused, st);              \
        return launch_nsf_cfg<NORM, TT, E4, false, true, false>(a, Q, fused, st);                     \
    }
    FZ_NSF_CASE(256, 1) FZ_NSF_CASE(256, 4) FZ_NSF_CASE(512, 4) FZ_NSF_CASE(1024, 4) FZ_NSF_CASE(1024, 7) FZ_NSF_CASE(1024, 8)
#undef FZ_NSF_CASE
    return 1;  // row too long for the register-resident kernel
}

// one wave per 64 columns: ballot of "rank >= 0"
__global__ __launch_bounds__(256) void rank_to_bitmap_kernel(const int32_t* __restrict__ rank, int N, int ld, uint32_t* __restrict__ bits, int ldb) {
    const int row = blockIdx.y;
    const int j = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (threadIdx.x & 63);
    if ((j & ~63) >= ldb * 32) return;   // wave-uniform
    const bool v = j < N && rank[(size_t)row * ld + j] >= 0;
    const unsigned long long bal = __ballot(v);
    if ((threadIdx.x & 63) == 0) {
        bits[(size_t)row * ldb + (j >> 5)] = (uint32_t)bal;
        if ((j >> 5) + 1 < ldb) bits[(size_t)row * ldb + (j >> 5) + 1] = (uint32_t)(bal >> 32);
    }
}

extern "C" int fz_rank_to_bitmap(const int32_t* rank, int rows, int N, int ld, uint32_t* bits, int ldb, void* stream) {
    if (rows < 0 || N < 0 || ld < N || ldb * 32 < N) return FZ_ERR_ARG;
    if (rows == 0 || N == 0) return FZ_OK;
    if (!rank || !bits) return FZ_ERR_ARG;
    dim3 grid((unsigned)((ldb * 32 + 255) / 256), (unsigned)rows);
    rank_to_bitmap_kernel<<<grid, 256, 0, as_stream(stream)>>>(rank, N, ld, bits, ldb);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

namespace fz {
int nsf_fill_args(NsfArgs& a, const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N, int ld,
                  bool needs_distr, const float* const* distr_h, const int32_t* P_h, const uint32_t* const* valid_bits_h, int ldb) {
    a = NsfArgs{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.vbits[s] = valid_bits_h ? valid_bits_h[s] : nullptr;
        a.distr[s] = needs_distr ? distr_h[s] : nullptr;
        a.P[s] = needs_distr ? P_h[s] : 0;
        if (needs_distr && (!a.distr[s] || a.P[s] <= 0)) return FZ_ERR_ARG;
        a.w[s] = (float)w_h[s];
    }
    a.ldb = ldb;
    if (valid_bits_h && ldb * 32 < N) return FZ_ERR_ARG;
    return FZ_OK;
}
}  // namespace fz

extern "C" int fz_fuse_nsf_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q,
                               int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h,
                               const uint32_t* const* valid_bits_h, int ldb, float* fused, void* stream) {
    if (!planes_h || !w_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (norm == FZ_NORM_NONE) return FZ_ERR_ARG;  // float64 passthrough lives in fz_fuse_none_f64
    if (norm < FZ_NORM_MINMAX || norm > FZ_NORM_NCE) return FZ_ERR_ARG;
    if (!fused && Q != 0 && N != 0) return FZ_ERR_ARG;   // empty tensors carry null pointers
    const bool needs_distr = (norm == FZ_NORM_PERCENTILE || norm == FZ_NORM_NCE);
    if (needs_distr && (!distr_h || !P_h)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    NsfArgs a{};
    if (int rc = nsf_fill_args(a, planes_h, ranks_h, w_h, S, Q, N, ld, needs_distr, distr_h, P_h, valid_bits_h, ldb)) return rc;
    hipStream_t st = as_stream(stream);
    int too_long = 1;
    switch (norm) {
        case FZ_NORM_MINMAX: too_long = launch_nsf<FZ_NORM_MINMAX>(a, Q, fused, st); break;
        case FZ_NORM_ZSCORE: too_long = launch_nsf<FZ_NORM_ZSCORE>(a, Q, fused, st); break;
        case FZ_NORM_ARCTAN: too_long = launch_nsf<FZ_NORM_ARCTAN>(a, Q, fused, st); break;
        case FZ_NORM_PERCENTILE:
            too_long = launch_nsf_tables(a, false, Q, fused, st);
            if (too_long) too_long = launch_nsf<FZ_NORM_PERCENTILE>(a, Q, fused, st);
            break;
        case FZ_NORM_NCE:
            too_long = launch_nsf_tables(a, true, Q, fused, st);
            if (too_long) too_long = launch_nsf<FZ_NORM_NCE>(a, Q, fused, st);
            break;
    }
    if (too_long) return FZ_ERR_UNSUPPORTED;  // N > 32768: use fz_row_stats_f32 + fz_fuse_nsf_stats_f32
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// general-N two-pass variant (statistics supplied by the caller, e.g. from fz_row_stats_f32)
static int fuse_nsf_with_stats(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S,
                               int Q, int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h,
                               const float* const* stat_a_h, const float* const* stat_b_h, const uint32_t* const* valid_bits_h, int ldb,
                               float* fused, void* stream) {
    if (!planes_h || !w_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (norm < FZ_NORM_MINMAX || norm > FZ_NORM_NCE) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;                 // empty tensors carry null pointers
    if (!fused) return FZ_ERR_ARG;
    const bool needs_stats = (norm == FZ_NORM_MINMAX || norm == FZ_NORM_ZSCORE);
    const bool needs_distr = (norm == FZ_NORM_PERCENTILE || norm == FZ_NORM_NCE);
    if (needs_stats && (!stat_a_h || !stat_b_h)) return FZ_ERR_ARG;
    if (needs_distr && (!distr_h || !P_h)) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    NsfArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.vbits[s] = valid_bits_h ? valid_bits_h[s] : nullptr;
        a.distr[s] = needs_distr ? distr_h[s] : nullptr;
        a.P[s] = needs_distr ? P_h[s] : 0;
        a.w[s] = (float)w_h[s];
        if (needs_stats) {
            if (!stat_a_h[s] || !stat_b_h[s]) return FZ_ERR_ARG;
            a.sa[s] = stat_a_h[s]; a.sb[s] = stat_b_h[s];
        }
    }
    a.ldb = ldb;
    if (valid_bits_h && ldb * 32 < N) return FZ_ERR_ARG;
    hipStream_t st = as_stream(stream);
    bool vec = (ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    for (int s = 0; s < S; ++s) vec = vec && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    if (vec) {
        dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)Q);
        switch (norm) {
            case FZ_NORM_MINMAX: fuse_nsf_elem4_kernel<FZ_NORM_MINMAX><<<grid, 256, 0, st>>>(a, Q, fused); break;
            case FZ_NORM_ZSCORE: fuse_nsf_elem4_kernel<FZ_NORM_ZSCORE><<<grid, 256, 0, st>>>(a, Q, fused); break;
            case FZ_NORM_ARCTAN: fuse_nsf_elem4_kernel<FZ_NORM_ARCTAN><<<grid, 256, 0, st>>>(a, Q, fused); break;
            case FZ_NORM_PERCENTILE: fuse_nsf_elem4_kernel<FZ_NORM_PERCENTILE><<<grid, 256, 0, st>>>(a, Q, fused); break;
            case FZ_NORM_NCE: fuse_nsf_elem4_kernel<FZ_NORM_NCE><<<grid, 256, 0, st>>>(a, Q, fused); break;
        }
        FZ_LAUNCH_CHECK();
        return FZ_OK;
    }
    dim3 grid((unsigned)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64), (unsigned)Q);
    switch (norm) {
        case FZ_NORM_MINMAX: fuse_nsf_elem_kernel<FZ_NORM_MINMAX><<<grid, 256, 0, st>>>(a, Q, fused); break;
        case FZ_NORM_ZSCORE: fuse_nsf_elem_kernel<FZ_NORM_ZSCORE><<<grid, 256, 0, st>>>(a, Q, fused); break;
        case FZ_NORM_ARCTAN: fuse_nsf_elem_kernel<FZ_NORM_ARCTAN><<<grid, 256, 0, st>>>(a, Q, fused); break;
        case FZ_NORM_PERCENTILE: fuse_nsf_elem_kernel<FZ_NORM_PERCENTILE><<<grid, 256, 0, st>>>(a, Q, fused); break;
        case FZ_NORM_NCE: fuse_nsf_elem_kernel<FZ_NORM_NCE><<<grid, 256, 0, st>>>(a, Q, fused); break;
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_fuse_nsf_stats_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S,
                                     int Q, int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h,
                                     const float* stat_a, const float* stat_b, const uint32_t* const* valid_bits_h, int ldb,
                                     float* fused, void* stream) {
    const float* pa[FZ_MAX_SYSTEMS] = {};
    const float* pb[FZ_MAX_SYSTEMS] = {};
    if (S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0) return FZ_ERR_ARG;
    for (int s = 0; s < S; ++s) { pa[s] = stat_a ? stat_a + (size_t)s * Q : nullptr; pb[s] = stat_b ? stat_b + (size_t)s * Q : nullptr; }
    return fuse_nsf_with_stats(planes_h, ranks_h, w_h, S, Q, N, ld, norm, distr_h, P_h, stat_a ? pa : nullptr, stat_b ? pb : nullptr, valid_bits_h,
                               ldb, fused, stream);
}

// the same with one statistics pointer PER SYSTEM (host arrays of S device pointers, [Q] floats each): every ranked system keeps the
// statistics its ranking sort produced (fz_sort_rows_desc(row_stats)), so a fusion call concatenates nothing and reduces nothing
extern "C" int fz_fuse_nsf_pstats_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S,
                                      int Q, int N, int ld, int norm, const float* const* distr_h, const int32_t* P_h,
                                      const float* const* stat_a_h, const float* const* stat_b_h, const uint32_t* const* valid_bits_h, int ldb,
                                      float* fused, void* stream) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS) return FZ_ERR_ARG;
    return fuse_nsf_with_stats(planes_h, ranks_h, w_h, S, Q, N, ld, norm, distr_h, P_h, stat_a_h, stat_b_h, valid_bits_h, ldb, fused, stream);
}

// Aggregator.aggregate_scores adds nothing for a document a system does not list (hybrid.py:301-304): plane[q][j] = 0 where rank[q][j] < 0.
// The single-system planes of fz_fuse_nsf_f32 (weight 1) hold -inf there; the weight sweep (fz_gold_ranks_*) wants 0.
__global__ __launch_bounds__(256) void zero_unlisted_kernel(float* __restrict__ plane, const int32_t* __restrict__ rank, int N, int ld) {
    const size_t rowoff = (size_t)blockIdx.y * ld;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x)
        if (rank[rowoff + j] < 0) plane[rowoff + j] = 0.f;
}
extern "C" int fz_zero_unlisted_f32(float* plane, const int32_t* rank, int Q, int N, int ld, void* stream) {
    if (Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    if (!plane || !rank) return FZ_ERR_ARG;
    dim3 grid((unsigned)((N + 255) / 256 < 32 ? (N + 255) / 256 : 32), (unsigned)Q);
    zero_unlisted_kernel<<<grid, 256, 0, as_stream(stream)>>>(plane, rank, N, ld);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// all S systems of a fusion in one launch: thread (s, q)
struct EndsArgs {
    const float* planes[FZ_MAX_SYSTEMS];
    const int32_t* orders[FZ_MAX_SYSTEMS];
};
__global__ void minmax_from_orders_kernel(EndsArgs e, const int32_t* __restrict__ lens, int S, int Q, int N, int ld,
                                          float* __restrict__ mn, float* __restrict__ mx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * Q) return;
    const int s = i / Q, row = i - s * Q;
    const int len = lens ? lens[i] : N;
    float lo = 0.f, hi = 0.f;
    if (len > 0) {
        hi = e.planes[s][(size_t)row * ld + e.orders[s][(size_t)row * ld]];
        lo = e.planes[s][(size_t)row * ld + e.orders[s][(size_t)row * ld + len - 1]];
        if (hi != hi) lo = hi;
    }
    mn[i] = lo; mx[i] = hi;
}

extern "C" int fz_minmax_from_orders_f32(const float* const* planes_h, const int32_t* const* orders_h, const int32_t* lens, int S, int Q,
                                         int N, int ld, float* mn, float* mx, void* stream) {
    if (!planes_h || !orders_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0) return FZ_OK;
    if (!mn || !mx) return FZ_ERR_ARG;
    EndsArgs e{};
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s] || !orders_h[s]) return FZ_ERR_ARG;
        e.planes[s] = planes_h[s]; e.orders[s] = orders_h[s];
    }
    minmax_from_orders_kernel<<<(unsigned)((S * Q + 255) / 256), 256, 0, as_stream(stream)>>>(e, lens, S, Q, N, ld, mn, mx);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_minmax_from_order_f32(const float* scores, const int32_t* order, const int32_t* lens, int rows, int N, int ld,
                                        float* mn, float* mx, void* stream) {
    if (rows < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    if (!scores || !order || !mn || !mx) return FZ_ERR_ARG;
    minmax_from_order_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, as_stream(stream)>>>(scores, order, lens, rows, N, ld, mn, mx);
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

static int launch_wsum(ElemArgs& a, double* fused, void* stream) {
    dim3 grid((unsigned)((a.N + 1023) / 1024), (unsigned)a.Q);
    bool al = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    for (int s = 0; s < a.S; ++s) al = al && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    const bool mixed = a.f64_mask != 0u || a.narrow_mask != 0u;
    hipStream_t st = as_stream(stream);
    if (al && mixed) fuse_wsum_kernel<true, true><<<grid, 256, 0, st>>>(a, fused);
    else if (al) fuse_wsum_kernel<true, false><<<grid, 256, 0, st>>>(a, fused);
    else if (mixed) fuse_wsum_kernel<false, true><<<grid, 256, 0, st>>>(a, fused);
    else fuse_wsum_kernel<false, false><<<grid, 256, 0, st>>>(a, fused);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_fuse_none_f64(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q,
                                int N, int ld, double* fused, void* stream) {
    if (!planes_h || !w_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;                 // empty tensors carry null pointers
    if (!fused) return FZ_ERR_ARG;
    ElemArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.w[s] = w_h[s];
    }
    return launch_wsum(a, fused, stream);
}

extern "C" int fz_fuse_wsum_f64(const void* const* planes_h, const int32_t* plane_is_f64_h, const int32_t* const* ranks_h,
                                const double* w_h, const int32_t* narrow_h, int S, int Q, int N, int ld, double* fused, void* stream) {
    if (!planes_h || !w_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;                 // empty tensors carry null pointers
    if (!fused) return FZ_ERR_ARG;
    ElemArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (!planes_h[s]) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s];
        a.ranks[s] = ranks_h ? ranks_h[s] : nullptr;
        a.w[s] = w_h[s];
        if (plane_is_f64_h && plane_is_f64_h[s]) a.f64_mask |= 1u << s;
        if (narrow_h && narrow_h[s]) a.narrow_mask |= 1u << s;
    }
    return launch_wsum(a, fused, stream);
}

extern "C" int fz_fuse_rank_f64(const int32_t* const* ranks_h, const int32_t* lens, int S, int Q, int N, int ld, int method,
                                double* fused, void* stream) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return (method == FZ_RRF || method == FZ_BCF) ? FZ_OK : FZ_ERR_ARG;   // empty tensors carry null pointers
    if (!ranks_h || !lens || !fused) return FZ_ERR_ARG;
    if (method != FZ_RRF && method != FZ_BCF) return FZ_ERR_ARG;
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
                                  int32_t* ins_order, int32_t* U, int32_t* pos, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (!orders_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (Q != 0 && (!lens || !ins_order || !U)) return FZ_ERR_ARG;   // empty tensors carry null pointers
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
    insertion_order_kernel<<<Q, 1024, lds, as_stream(stream)>>>(a, ins_order, U, pos);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
