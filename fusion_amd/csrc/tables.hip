// tables.hip -- percentile-rank / normal-curve-equivalent fusion against quantile tables of the size the reference READS
// (hybrid.py:412,451: score_distributions_raw_*_28k.csv, P = |corpus| + 1 = 27,943 per system; :374: the _10k table, 10,001).
//
//   fz_nsf_tables_workspace_bytes / fz_nsf_tables_prepare / fz_fuse_nsf_tables_f32 / fz_nsf_tables_path
//
// The arithmetic is Aggregator.transform_scores' (hybrid.py:271-277): per score the FIRST minimum of the float32 distances
// |distr - s| over an ascending table, divided by the table length; NCE = Normal(0,1).icdf(that / 100) * 21.06 + 50.  What
// changes with the table size is where the table can live.  fuse.hip's fuse_nsf_table_kernel keeps ALL S tables (plus a
// de-duplicated copy, value tables and guess tables) in LDS and stops at ~1.4 k entries per system for S = 4; one 27,943-entry
// table alone is 112 KB of the CU's 160.  Here ONE system's table is resident at a time:
//
//   * persistent 1024-thread workgroups (one per CU) walk (query row, 28,672-column chunk) items; the fused accumulators of
//     the item live in registers (7 float4 per thread) across the S systems, so every plane still crosses HBM exactly once
//     and nothing is accumulated through memory;
//   * per (item, system) the system's table and its bucket table are brought into LDS by LDS-DMA (global_load_lds, 16 B per
//     lane, no VGPRs) from an aligned, +inf-padded copy in the workspace -- 144 KB from L2 per swap (the tables are read by
//     every workgroup: L2-resident), ~1 us against ~28 k searches; the row's scores are loaded into registers under it.  With
//     S = 1 (Aggregator.tune normalises system by system) the table is loaded once per workgroup;
//   * the search: an equi-width bucket table over [tab[0], tab[P-1]] with 16,384 buckets (uint16: 32 KB) is built ONCE per call
//     by fz_nsf_tables_prepare with the SAME float expression the scores go through (bucket(x) = (int)clamp((x - lo) * inv_w)),
//     lut[b] = #{k : bucket(tab[k]) < b}.  bucket() is monotone, so for a score in bucket b every entry before lut[b] is
//     smaller and every entry from lut[b+1] on is larger: the bracket [lut[b] - 1, lut[b+1]) holds the answer BY CONSTRUCTION
//     (no verification reads) and is a handful of entries wide where the table is densest (quantiles of a score distribution
//     are dense exactly where the scores are: an equi-width table of 2,048 buckets, fuse.hip's, leaves ~60 entries there).
//     Four scores of a float4 are searched in lockstep (independent LDS reads in flight), then three reads around the result
//     decide between the two neighbours with the reference's float32 distances; equal distances to the left (duplicated
//     quantiles, rounding plateaus) are resolved to the FIRST index by a short walk, then a binary search over the plateau;
//   * percentile-rank's value is (float)k / (float)P in registers; NCE's value depends on the score only through k and costs
//     a double-precision erfinv: tabulated once per call ([P] floats per system in the workspace, the expression of
//     fuse.hip's transform<FZ_NORM_NCE>), it either sits next to the table in LDS (P <= ~16 k) or is swapped in over the
//     table once the item's 28 indices per thread are known.
//
// Same nearest entry, same float expressions as the per-score device function (fuse.hip percentile_rank): bit-identical
// results -- tests/test_gpu_tables.py holds the three kernels against each other and against the reference's own outputs
// (tests/golden/pr28k_*.npz: outputs of the reference itself).
#include "common.h"
#include "nsf.h"

namespace fz {

constexpr int BT_T = 1024;                  // threads per workgroup
constexpr int BT_E4 = 7;                    // float4 per thread: 28,672 columns per item
constexpr int BT_COLS = BT_T * BT_E4 * 4;
constexpr int BT_PIECE = 256;               // floats per LDS-DMA wave instruction (64 lanes x 16 B)
constexpr size_t BT_LDS_BUDGET = 160 * 1024 - 256;   // the CU's LDS minus the kernel's static variables
constexpr size_t BT_HDR_BYTES = 256;

struct BtPlan {
    bool ok;                 // every table (+ bucket table) fits LDS
    int lutb;                // buckets
    int tab_cap;             // floats of LDS for the table: max over systems of Ppad
    int lut_floats;          // bucket table, in floats, a multiple of BT_PIECE
    bool val_in_lds;         // NCE: the value table sits next to the table (no second swap)
    size_t lds_bytes;
    int Ppad[FZ_MAX_SYSTEMS];
    size_t sys_off[FZ_MAX_SYSTEMS + 1];   // byte offset of system s's block in the workspace: hdr | tab | lut | val (NCE)
};

static BtPlan bt_plan(int S, const int32_t* P_h, bool nce) {
    BtPlan p{};
    int cap = 0;
    for (int s = 0; s < S; ++s) {
        if (P_h[s] <= 0 || P_h[s] > 65535) return p;          // uint16 bucket-table entries
        p.Ppad[s] = (P_h[s] + 1 + BT_PIECE - 1) / BT_PIECE * BT_PIECE;   // at least one +inf entry behind the table
        cap = p.Ppad[s] > cap ? p.Ppad[s] : cap;
    }
    p.tab_cap = cap;
    for (int lutb = 16384; lutb >= 2048 && !p.ok; lutb >>= 1) {
        const int lf = ((lutb + 1) * 2 + BT_PIECE * 4 - 1) / (BT_PIECE * 4) * BT_PIECE;
        if ((size_t)(cap + lf) * 4 <= BT_LDS_BUDGET) { p.ok = true; p.lutb = lutb; p.lut_floats = lf; }
    }
    if (!p.ok) return p;
    p.val_in_lds = nce && (size_t)(2 * cap + p.lut_floats) * 4 <= BT_LDS_BUDGET;
    p.lds_bytes = (size_t)((p.val_in_lds ? 2 : 1) * cap + p.lut_floats) * 4;
    size_t off = 0;
    for (int s = 0; s < S; ++s) {
        p.sys_off[s] = off;
        off += BT_HDR_BYTES + (size_t)p.Ppad[s] * 4 + (size_t)p.lut_floats * 4 + (nce ? (size_t)p.Ppad[s] * 4 : 0);
    }
    p.sys_off[S] = off;
    return p;
}

// the bucket of a value: the SAME expression for table entries (prepare) and scores (fusion).  Monotone non-decreasing in x
// for any lo / inv_w >= 0 (subtraction, multiplication by a non-negative constant, clamp and truncation all are; a NaN
// intermediate -- inf * 0 -- lands in bucket 0 together with everything else when inv_w == 0).
__device__ __forceinline__ int bt_bucket(float x, float lo_v, float inv_w, float top) {
    float t = (x - lo_v) * inv_w;
    t = fminf(fmaxf(t, 0.f), top);   // fmaxf(NaN, 0) = 0
    return (int)t;
}

struct BtPrepArgs {
    const float* distr[FZ_MAX_SYSTEMS];
    int P[FZ_MAX_SYSTEMS], Ppad[FZ_MAX_SYSTEMS];
    size_t sys_off[FZ_MAX_SYSTEMS];
    int lutb, lut_floats, nce;
};

// BT_PREP_SLICES workgroups per system: aligned +inf-padded copy of the table, header (lo, inv_w), bucket table, NCE values
constexpr int BT_PREP_SLICES = 32;
__global__ __launch_bounds__(256) void bt_prepare_kernel(BtPrepArgs a, unsigned char* __restrict__ ws) {
    const int s = blockIdx.x;
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, nthr = gridDim.y * blockDim.x;
    const float* __restrict__ tab = a.distr[s];
    const int P = a.P[s], Ppad = a.Ppad[s];
    float* hdr = reinterpret_cast<float*>(ws + a.sys_off[s]);
    float* wtab = reinterpret_cast<float*>(ws + a.sys_off[s] + BT_HDR_BYTES);
    uint16_t* lut = reinterpret_cast<uint16_t*>(wtab + Ppad);
    float* val = wtab + Ppad + a.lut_floats;
    const float lo_v = tab[0], hi_v = tab[P - 1];
    float inv_w = 0.f;
    if (P >= 2) {
        const float d = hi_v - lo_v;
        if (d > 0.f && d < INFINITY) inv_w = (float)a.lutb / d;
        if (!(inv_w < INFINITY)) inv_w = 0.f;   // a denormal range: everything in bucket 0, the search runs over the whole table
    }
    if (tid == 0) { hdr[0] = lo_v; hdr[1] = inv_w; hdr[2] = 0.f; hdr[3] = 0.f; }
    for (int k = tid; k < Ppad; k += nthr) wtab[k] = k < P ? tab[k] : INFINITY;
    const float top = (float)(a.lutb - 1);
    const int lut_n = a.lut_floats * 2;
    for (int b = tid; b < lut_n; b += nthr) {
        int lo = 0, hi = P;   // lower bound: first k whose bucket is >= b
        if (b >= a.lutb) lo = P;
        else
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (bt_bucket(tab[mid], lo_v, inv_w, top) < b) lo = mid + 1; else hi = mid;
            }
        lut[b] = (uint16_t)lo;
    }
    if (a.nce)
        for (int k = tid; k < Ppad; k += nthr) {
            float v = 0.f;
            if (k < P) {   // transform<FZ_NORM_NCE> (fuse.hip) as a function of the index; hybrid.py:275-277
                const float pr = (float)k / (float)P;
                const float p = pr / 100.0f;
                const float y = 2.0f * p - 1.0f;
                const float z = (float)(erfinv((double)y) * 1.4142135623730951);
                v = z * 21.06f + 50.0f;
            }
            val[k] = v;
        }
}

struct BtArgs {
    const float* hdr[FZ_MAX_SYSTEMS];     // {lo, inv_w}
    const float* tab[FZ_MAX_SYSTEMS];     // [Ppad] aligned copy, +inf padded
    const float* lut[FZ_MAX_SYSTEMS];     // [lut_floats] floats = uint16 [lutb + 1 ...]
    const float* val[FZ_MAX_SYSTEMS];     // NCE: [Ppad]
    int Ppad[FZ_MAX_SYSTEMS];
    int lutb, tab_cap, lut_floats, val_in_lds;
};

// first index of the plateau of equal float32 distances that ends at k (|tab[k] - x| == dl, tab ascending, tab[k] <= x):
// the distances fl(x - tab[j]) are non-increasing in j up to k, so "== dl" holds on a suffix of [0, k]
__device__ __noinline__ int bt_plateau_start(const float* tab, float x, float dl, int k) {
    for (int i = 0; i < 4 && k > 0 && fabsf(tab[k - 1] - x) == dl; ++i) --k;
    if (k > 0 && fabsf(tab[k - 1] - x) == dl) {
        int a = -1, b = k - 1;   // tab[b] on the plateau, tab[a] not (or a = -1)
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (fabsf(tab[mid] - x) == dl) b = mid; else a = mid;
        }
        k = b;
    }
    return k;
}

// nearest table entry (first minimum of the float32 distances) of the four scores of a float4; m = which of them are looked up
__device__ __forceinline__ void bt_lookup4(const float* tab, const uint16_t* lut, float lo_v, float inv_w, float top, const float (&x)[4],
                                           uint32_t m, int (&best)[4]) {
    int lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool act = ((m >> e) & 1u) && fabsf(x[e]) < INFINITY;   // NaN: argmin of an all-NaN column; +-inf: every distance is inf -> index 0
        const int b = bt_bucket(x[e], lo_v, inv_w, top);
        const int l = (int)lut[b], h = (int)lut[b + 1];
        lo[e] = act ? l - 1 : -1;
        hi[e] = act ? h : 0;
    }
    while (max(max(hi[0] - lo[0], hi[1] - lo[1]), max(hi[2] - lo[2], hi[3] - lo[3])) > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool go = hi[e] - lo[e] > 1;
            const int mid = go ? (lo[e] + hi[e]) >> 1 : 0;
            const bool le = tab[mid] <= x[e];
            lo[e] = (go && le) ? mid : lo[e];
            hi[e] = (go && !le) ? mid : hi[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int l0 = max(lo[e], 0);
        const float tl = tab[l0], th = tab[l0 + 1], tm = tab[max(l0 - 1, 0)];   // tab[P] = +inf
        const float dl = fabsf(tl - x[e]), dh = fabsf(th - x[e]), dm = fabsf(tm - x[e]);
        int k = lo[e] < 0 ? 0 : (dh < dl ? l0 + 1 : l0);
        if (lo[e] > 0 && !(dh < dl) && dm == dl) k = bt_plateau_start(tab, x[e], dl, l0 - 1);
        best[e] = k;
    }
}

template <bool NCE>
__global__ __launch_bounds__(BT_T) void fuse_nsf_bigtab_kernel(NsfArgs a, BtArgs t, float* __restrict__ fused) {
    extern __shared__ __attribute__((aligned(16))) float bt_lds[];
    float* tab = bt_lds;                                                    // [tab_cap]
    float* lutf = bt_lds + t.tab_cap;                                       // [lut_floats]
    float* valr = bt_lds + t.tab_cap + t.lut_floats;                        // [tab_cap] when val_in_lds
    const uint16_t* lut = reinterpret_cast<const uint16_t*>(lutf);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float top = (float)(t.lutb - 1);
    const int chunks = (a.N + BT_COLS - 1) / BT_COLS;
    const long long items = (long long)a.Q * chunks;
    typedef float f4v __attribute__((ext_vector_type(4)));

    // global -> LDS, one 1-KiB piece per wave instruction (LDS destination = wave-uniform base + lane * 16)
    auto dma = [&](const float* __restrict__ src, float* dst, int floats) {
        for (int p = wave * BT_PIECE; p < floats; p += (BT_T / 64) * BT_PIECE)
            __builtin_amdgcn_global_load_lds(src + p + lane * 4, (__attribute__((address_space(3))) void*)(dst + p), 16, 0, 0);
    };
    int cur = -1;   // the system whose table is in LDS
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        const int q = (int)(it / chunks), c = (int)(it - (long long)q * chunks);
        const size_t rowoff = (size_t)q * a.ld;
        const int col0 = c * BT_COLS + 4 * threadIdx.x;
        float acc[BT_E4][4];
        uint32_t present = 0u;
#pragma unroll
        for (int i = 0; i < BT_E4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int s = 0; s < a.S; ++s) {
            const bool swap = cur != s;
            if (swap) {
                __syncthreads();                       // every wave is done with the table that is there
                dma(t.tab[s], tab, t.Ppad[s]);
                dma(t.lut[s], lutf, t.lut_floats);
                if (NCE && t.val_in_lds) dma(t.val[s], valr, t.Ppad[s]);
                cur = s;
            }
            // the item's scores of this system: loaded under the table's DMA
            f4v v[BT_E4];
            uint32_t ok = 0u;
#pragma unroll
            for (int i = 0; i < BT_E4; ++i) {
                const int j0 = col0 + 4 * BT_T * i;
                v[i] = f4v{0.f, 0.f, 0.f, 0.f};
                if (j0 < a.N) {
                    v[i] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(a.planes[s] + rowoff + j0));   // streamed once
                    const int rem = a.N - j0;
                    const uint32_t mm = (rem >= 4 ? 0xfu : ((1u << rem) - 1u)) & valid_nibble(a, s, q, rowoff, j0);
                    ok |= mm << (4 * i);
                }
            }
            const float lo_v = t.hdr[s][0], inv_w = t.hdr[s][1];
            if (swap) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces have landed ...
                __syncthreads();                                   // ... and everybody else's
            }
            const float w = a.w[s];
            const float Pf = (float)a.P[s];
            int idx[BT_E4][4];
#pragma unroll
            for (int i = 0; i < BT_E4; ++i) {
                const float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
                int best[4];
                bt_lookup4(tab, lut, lo_v, inv_w, top, x, (ok >> (4 * i)) & 0xfu, best);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!NCE) {
                        const float tr = (float)best[e] / Pf;                   // hybrid.py:275
                        const float prod = tr * w;                              // fl32(t * fl32(w))      hybrid.py:291 under NumPy 2
                        acc[i][e] = ((ok >> (4 * i + e)) & 1u) ? acc[i][e] + prod : acc[i][e];
                    } else idx[i][e] = best[e];
                }
            }
            if (NCE) {
                const float* vt = valr;
                if (!t.val_in_lds) {   // the values take the table's place
                    __syncthreads();
                    dma(t.val[s], tab, t.Ppad[s]);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    cur = -1;
                    vt = tab;
                }
#pragma unroll
                for (int i = 0; i < BT_E4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float prod = vt[idx[i][e]] * w;
                        acc[i][e] = ((ok >> (4 * i + e)) & 1u) ? acc[i][e] + prod : acc[i][e];
                    }
            }
            present |= ok;
        }
#pragma unroll
        for (int i = 0; i < BT_E4; ++i) {
            const int j0 = col0 + 4 * BT_T * i;
            if (j0 < a.N) {   // columns [N, ld) of the last float4 are padding of the plane: written, never read
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ((present >> (4 * i + e)) & 1u) ? acc[i][e] : -INFINITY;
                *reinterpret_cast<float4*>(fused + rowoff + j0) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

static bool bt_vec_ok(const NsfArgs& a, const float* fused) {
    bool vec = (a.ld % 4 == 0) && ((uintptr_t)fused % 16 == 0);
    for (int s = 0; s < a.S; ++s) vec = vec && ((uintptr_t)a.planes[s] % 16 == 0) && (!a.ranks[s] || (uintptr_t)a.ranks[s] % 16 == 0);
    return vec;
}

}  // namespace fz

using namespace fz;

extern "C" size_t fz_nsf_tables_workspace_bytes(int S, const int32_t* P_h, int norm) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS || !P_h || (norm != FZ_NORM_PERCENTILE && norm != FZ_NORM_NCE)) return 0;
    const BtPlan p = bt_plan(S, P_h, norm == FZ_NORM_NCE);
    return p.ok ? p.sys_off[S] : 0;
}

extern "C" int fz_nsf_tables_prepare(const float* const* distr_h, const int32_t* P_h, int S, int norm, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS || !distr_h || !P_h || (norm != FZ_NORM_PERCENTILE && norm != FZ_NORM_NCE)) return FZ_ERR_ARG;
    for (int s = 0; s < S; ++s)
        if (!distr_h[s] || P_h[s] <= 0) return FZ_ERR_ARG;
    const BtPlan p = bt_plan(S, P_h, norm == FZ_NORM_NCE);
    if (!p.ok) return FZ_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < p.sys_off[S]) return FZ_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 16 != 0) return FZ_ERR_ARG;
    BtPrepArgs a{};
    for (int s = 0; s < S; ++s) { a.distr[s] = distr_h[s]; a.P[s] = P_h[s]; a.Ppad[s] = p.Ppad[s]; a.sys_off[s] = p.sys_off[s]; }
    a.lutb = p.lutb; a.lut_floats = p.lut_floats; a.nce = norm == FZ_NORM_NCE;
    bt_prepare_kernel<<<dim3((unsigned)S, BT_PREP_SLICES), 256, 0, as_stream(stream)>>>(a, static_cast<unsigned char*>(workspace));
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// which kernel fz_fuse_nsf_tables_f32 runs for these shapes
static int bt_path(const NsfArgs& a, const int32_t* P_h, bool nce, const float* fused, BtPlan& p) {
    if (nsf_tables_fit_lds(a, fused)) return FZ_TABLES_PATH_LDS_ALL;
    p = bt_plan(a.S, P_h, nce);
    if (p.ok && bt_vec_ok(a, fused)) return FZ_TABLES_PATH_LDS_SWAP;
    return FZ_TABLES_PATH_ROW;
}

extern "C" int fz_nsf_tables_path(const float* const* planes_h, const int32_t* const* ranks_h, int S, int Q, int N, int ld, int norm,
                                  const int32_t* P_h, const float* fused) {
    if (!planes_h || S <= 0 || S > FZ_MAX_SYSTEMS || !P_h || (norm != FZ_NORM_PERCENTILE && norm != FZ_NORM_NCE) || ld < N) return FZ_ERR_ARG;
    NsfArgs a{};
    a.S = S; a.N = N; a.ld = ld; a.Q = Q;
    for (int s = 0; s < S; ++s) {
        if (P_h[s] <= 0) return FZ_ERR_ARG;
        a.planes[s] = planes_h[s]; a.ranks[s] = ranks_h ? ranks_h[s] : nullptr; a.P[s] = P_h[s];
    }
    BtPlan p{};
    return bt_path(a, P_h, norm == FZ_NORM_NCE, fused, p);
}

extern "C" int fz_fuse_nsf_tables_f32(const float* const* planes_h, const int32_t* const* ranks_h, const double* w_h, int S, int Q, int N,
                                      int ld, int norm, const float* const* distr_h, const int32_t* P_h, const uint32_t* const* valid_bits_h,
                                      int ldb, float* fused, const void* workspace, size_t workspace_bytes, void* stream) {
    if (!planes_h || !w_h || S <= 0 || S > FZ_MAX_SYSTEMS || Q < 0 || N < 0 || ld < N) return FZ_ERR_ARG;
    if (norm != FZ_NORM_PERCENTILE && norm != FZ_NORM_NCE) return FZ_ERR_ARG;
    if (!distr_h || !P_h) return FZ_ERR_ARG;
    if (!fused && Q != 0 && N != 0) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    const bool nce = norm == FZ_NORM_NCE;
    NsfArgs a{};
    if (int rc = nsf_fill_args(a, planes_h, ranks_h, w_h, S, Q, N, ld, true, distr_h, P_h, valid_bits_h, ldb)) return rc;
    hipStream_t st = as_stream(stream);
    BtPlan p{};
    const int path = bt_path(a, P_h, nce, fused, p);
    if (path == FZ_TABLES_PATH_LDS_ALL) {
        if (launch_nsf_tables(a, nce, Q, fused, st)) return FZ_ERR_HIP;
        FZ_LAUNCH_CHECK();
        return FZ_OK;
    }
    if (path != FZ_TABLES_PATH_LDS_SWAP) return FZ_ERR_UNSUPPORTED;   // fz_fuse_nsf_f32 searches the tables in global memory
    if (!workspace || workspace_bytes < p.sys_off[S]) return FZ_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 16 != 0) return FZ_ERR_ARG;
    BtArgs t{};
    const unsigned char* ws = static_cast<const unsigned char*>(workspace);
    for (int s = 0; s < S; ++s) {
        t.hdr[s] = reinterpret_cast<const float*>(ws + p.sys_off[s]);
        t.tab[s] = reinterpret_cast<const float*>(ws + p.sys_off[s] + BT_HDR_BYTES);
        t.lut[s] = t.tab[s] + p.Ppad[s];
        t.val[s] = nce ? t.lut[s] + p.lut_floats : nullptr;
        t.Ppad[s] = p.Ppad[s];
    }
    t.lutb = p.lutb; t.tab_cap = p.tab_cap; t.lut_floats = p.lut_floats; t.val_in_lds = p.val_in_lds ? 1 : 0;
    const long long items = (long long)Q * ((N + BT_COLS - 1) / BT_COLS);
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
    static unsigned long long set_pr = 0ull, set_nce = 0ull;
    if (nce) {
        if (int rc = raise_lds_limit((const void*)fuse_nsf_bigtab_kernel<true>, p.lds_bytes, set_nce)) return rc;
        fuse_nsf_bigtab_kernel<true><<<grid, BT_T, p.lds_bytes, st>>>(a, t, fused);
    } else {
        if (int rc = raise_lds_limit((const void*)fuse_nsf_bigtab_kernel<false>, p.lds_bytes, set_pr)) return rc;
        fuse_nsf_bigtab_kernel<false><<<grid, BT_T, p.lds_bytes, st>>>(a, t, fused);
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
