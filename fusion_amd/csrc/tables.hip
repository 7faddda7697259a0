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
//   * per (item, system) step the system's table and its bucket table are brought into LDS by LDS-DMA (global_load_lds, 16 B
//     per lane, no VGPRs) from an aligned, sentinel-padded copy in the workspace -- 144 KB from L2 per swap (the tables are read
//     by every workgroup: L2-resident), ~3 us against ~28 k searches (measured: 0.05 of the 0.25 ms).  With S = 1
//     (Aggregator.tune normalises system by system) the table is loaded once per workgroup;
//   * the step's scores are requested from HBM one step AHEAD: a float4's registers take the same float4 of the next step as
//     soon as it has been searched (measured: the stream costs 0.055 ms instead of its 0.09 standalone);
//   * the search: an equi-width bucket table over [tab[0], tab[P-1]] with 16,384 buckets (uint16: 32 KB) is built ONCE per call
//     by fz_nsf_tables_prepare with the SAME float expression the scores go through (bucket(x) = (int)clamp((x - lo) * inv_w)),
//     lut[b] = #{k : bucket(tab[k]) < b}.  bucket() is monotone, so for a score in bucket b every entry before lut[b] is
//     smaller and every entry behind the bucket larger: the answer is found from lut[b] on BY CONSTRUCTION (no verification
//     reads), and the buckets are a handful of entries full where the table is densest (quantiles of a score distribution
//     are dense exactly where the scores are: an equi-width table of 2,048 buckets, fuse.hip's, leaves ~60 entries there;
//     with 16,384 the fullest bucket of an LLeQA-shaped table holds 5-10).  The search itself probes PAIRS of entries
//     (ds_read_b64: two entries for the LDS cycles of one) -- see bt_lookup;
//   * percentile-rank's value is (float)k / (float)P, computed division-free and exactly (bt_quot); NCE's value depends on the
//     score only through k and costs a double-precision erfinv: tabulated once per call ([P] floats per system in the
//     workspace, the expression of fuse.hip's transform<FZ_NORM_NCE>), it either sits next to the table in LDS (P <= ~16 k) or
//     is swapped in over the table once the item's 28 indices per thread are known.
//
// What bounds it (profiles/r04_pmc_tables.json, S = 4, Q = 1024, N = P - 1 = 27,942: 0.25 ms = 2.3 TB/s of the fusion's algorithmic
// bytes): 49 vector instructions and 4.6 LDS instructions per score -- 0.14 ms of vector-ALU issue and 0.12 ms of LDS cycles (62 % of
// them bank conflicts: the reads are random by nature) that overlap only in part; the round-3 path it replaces at these sizes
// (fz_fuse_nsf_f32's global-memory search) takes 2.8 ms, NCE 9.4 ms.
//
// Same nearest entry, same float expressions as the per-score device function (fuse.hip percentile_rank): bit-identical
// results -- tests/test_gpu_tables.py holds the three kernels against each other and against the reference's own outputs
// (tests/golden/pr28k_*.npz: outputs of the reference itself).
#include <type_traits>

#include "common.h"
#include "nsf.h"

namespace fz {

constexpr int BT_T = 1024;                  // threads per workgroup: 16 waves, 128 VGPRs each
constexpr int BT_E4 = 7;                    // float4 per thread: 28,672 columns per item
constexpr int BT_COLS = BT_T * BT_E4 * 4;
constexpr int BT_PIECE = 256;               // floats per LDS-DMA wave instruction (64 lanes x 16 B)
constexpr int BT_LEAD = 4;                  // -inf entries in front of the table (index -1, -2 of the search; keeps the table 16-B aligned)
constexpr int BT_TAIL = 40;                 // +inf entries behind it at least: the unrolled search (<= 4 probes) reaches 15 pairs past its start, unclamped
constexpr int BT_MAX_UNROLLED = 4;
constexpr int BT_R = 1;                    // items per table residency (accumulator sets per thread); 2 needs the 256 VGPRs of a 512-thread workgroup
typedef std::conditional<(BT_E4 <= 8), uint32_t, uint64_t>::type mask_t;   // one bit per column of a thread
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
        p.Ppad[s] = (BT_LEAD + P_h[s] + BT_TAIL + BT_PIECE - 1) / BT_PIECE * BT_PIECE;
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
    const float t = (x - lo_v) * inv_w;
    return (int)__builtin_amdgcn_fmed3f(t, 0.f, top);   // clamp in one instruction; a NaN gives bucket 0 (med3 of a NaN = min3 of the rest; cvt(NaN) = 0 anyway)
}

struct BtPrepArgs {
    const float* distr[FZ_MAX_SYSTEMS];
    int P[FZ_MAX_SYSTEMS], Ppad[FZ_MAX_SYSTEMS];
    size_t sys_off[FZ_MAX_SYSTEMS];
    int lutb, lut_floats, nce;
};

// BT_PREP_SLICES workgroups per system: aligned copy of the table between -inf / +inf sentinels, header (lo, inv_w), bucket
// table, NCE values
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
    if (tid == 0) { hdr[0] = lo_v; hdr[1] = inv_w; }
    for (int k = tid; k < Ppad; k += nthr) wtab[k] = k < BT_LEAD ? -INFINITY : (k - BT_LEAD < P ? tab[k - BT_LEAD] : INFINITY);
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
// second launch (the bucket table is complete): the number of halving probes that cover the fullest bucket
__global__ __launch_bounds__(1024) void bt_prepare_steps_kernel(BtPrepArgs a, unsigned char* __restrict__ ws) {
    const int s = blockIdx.x;
    int* hdr = reinterpret_cast<int*>(ws + a.sys_off[s]);
    const uint16_t* lut = reinterpret_cast<const uint16_t*>(ws + a.sys_off[s] + BT_HDR_BYTES + (size_t)a.Ppad[s] * 4);
    __shared__ int red[16];
    int m = 0;
    for (int b = threadIdx.x; b < a.lutb; b += blockDim.x) m = max(m, (int)lut[b + 1] - (int)lut[b]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = max(m, red[i]);
        // the search counts whole PAIRS of entries <= x from the pair of the bucket's first entry on: at most (m + 1) / 2 of them
        const int pairs = (m + 1) / 2;
        int steps = 0;
        while ((1 << steps) - 1 < pairs) ++steps;
        hdr[2] = steps; hdr[3] = m;
    }
}

struct BtArgs {
    const float* hdr[FZ_MAX_SYSTEMS];     // {lo, inv_w, steps (int), fullest bucket (int)}
    const float* tab[FZ_MAX_SYSTEMS];     // [Ppad] aligned copy: BT_LEAD x -inf, the table, +inf
    const float* lut[FZ_MAX_SYSTEMS];     // [lut_floats] floats = uint16 [lutb + 1 ...]
    const float* val[FZ_MAX_SYSTEMS];     // NCE: [Ppad]
    int Ppad[FZ_MAX_SYSTEMS];
    int lutb, tab_cap, lut_floats, val_in_lds;
    float top;                             // (float)(lutb - 1)
};

typedef __attribute__((address_space(3))) float lds_f32;       // LDS-typed pointers: the searches must compile to ds_read, never to flat loads
typedef __attribute__((address_space(3))) uint16_t lds_u16;

// first index of the plateau of equal float32 distances that ends at k (|tab[k] - x| == dl, tab ascending, tab[k] <= x):
// the distances fl(x - tab[j]) are non-increasing in j up to k, so "== dl" holds on a suffix of [0, k]
__device__ __forceinline__ int bt_plateau_start(const lds_f32* tab, float x, float dl, int k) {
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

// Nearest table entry (first minimum of the float32 distances) of W scores at once.  tab points at the table's entry 0 inside
// LDS (tab[-4 .. -1] = -inf; tab[P ...] = +inf).  A score's bucket starts at entry lut[b]; everything before it is smaller,
// everything behind the bucket larger.  The search runs over aligned PAIRS of entries -- a ds_read_b64 costs the LDS what a
// ds_read_b32 does (2 x 32 lanes on 64 banks against 2 x 32 lanes on 32) and brings two entries: from the pair that holds the
// bucket's first entry on, STEPS halving probes (2^STEPS - 1 >= the pairs the fullest bucket can fill) count the pairs whose
// SECOND entry is <= x; probes that run past the bucket land on larger entries and fail by themselves: no bounds, no branches,
// a compare, a select and an add per probe (the probe's offset is an immediate).  One ds_read2_b64 then brings the four entries
// around x, (C D | A B) with D <= x < B, and the reference's float32 distances decide between the neighbours; a distance equal
// to the one further left (duplicated quantiles, rounding plateaus) takes the exact walk -- one rarely taken branch per group.
// STEPS = 0: tables with fuller buckets (long runs of equal quantiles): `steps` probes in a loop, clamped to the table.
template <int W, int STEPS>
__device__ __forceinline__ void bt_lookup(const lds_f32* tab, const lds_u16* lut, float lo_v, float inv_w, float top, int steps, int last_pair,
                                          const float (&x)[W], int (&best)[W]) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f2v lds_f2;
    const lds_f2* tab2 = (const lds_f2*)tab;
    int p[W];                 // pairs before p are <= x throughout
#pragma unroll
    for (int e = 0; e < W; ++e) p[e] = (int)lut[bt_bucket(x[e], lo_v, inv_w, top)] >> 1;
    if (STEPS > 0) {
#pragma unroll
        for (int st = STEPS - 1; st >= 0; --st)
#pragma unroll
            for (int e = 0; e < W; ++e) p[e] += (tab2[p[e] + (1 << st) - 1].y <= x[e]) ? (1 << st) : 0;
    } else {
        for (int st = steps - 1; st >= 0; --st)
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const int q = min(p[e] + (1 << st) - 1, last_pair);
                p[e] = (tab2[q].y <= x[e]) ? q + 1 : p[e];
            }
#pragma unroll
        for (int e = 0; e < W; ++e) p[e] = min(p[e], last_pair);
    }
    uint32_t need = 0u;
#pragma unroll
    for (int e = 0; e < W; ++e) {   // D <= x < B: the last entry <= x is A (index 2p) or D (2p - 1)
        const f2v cd = tab2[p[e] - 1], ab = tab2[p[e]];
        const bool a_in = ab.x <= x[e];
        const float tl = a_in ? ab.x : cd.y, th = a_in ? ab.y : ab.x, tm = a_in ? cd.y : cd.x;
        const float dl = fabsf(tl - x[e]), dh = fabsf(th - x[e]), dm = fabsf(tm - x[e]);
        const bool right = dh < dl;
        best[e] = 2 * p[e] - (a_in ? 0 : 1) + (right ? 1 : 0);
        need |= (!right && dm == dl) ? (1u << e) : 0u;
    }
    if (need) {   // rare.  Inlined (a call would force everything that lives across it -- the accumulators, the score registers -- into the
                  // callee-saved half of the register file) and rolled: one copy of the walk per group, the lane's e-th score picked by selects
#pragma unroll 1
        for (int e = 0; e < W; ++e) {
            if (!((need >> e) & 1u)) continue;
            float xe = x[0];
            int be = best[0];
#pragma unroll
            for (int j = 1; j < W; ++j) { xe = e == j ? x[j] : xe; be = e == j ? best[j] : be; }
            const int k = bt_plateau_start(tab, xe, fabsf(tab[be] - xe), be - 1);
#pragma unroll
            for (int j = 0; j < W; ++j) best[j] = e == j ? k : best[j];
        }
    }
#pragma unroll
    for (int e = 0; e < W; ++e) best[e] = fabsf(x[e]) < INFINITY ? best[e] : 0;   // NaN: argmin of an all-NaN column; +-inf: every distance is inf -> index 0
}

// (float)k / (float)P (hybrid.py:275) without the division: q = k * fl(1/P) is within an ulp, r = fma(-q, P, k) is the exact
// remainder (it fits 24 bits), and fma(r, fl(1/P), q) rounds k/P * (1 + 2^-46) once -- for k < P < 2^16 the quotient is at least
// 2^-41 (relative) away from every float32 rounding boundary (a boundary is an odd multiple of half an ulp: its numerator has
// 25 bits, k/P's at most 16), so that is the correctly rounded quotient.  4 full-rate instructions for the division's 12.
__device__ __forceinline__ float bt_quot(int k, float Pf, float rP) {
    const float kf = (float)k;
    const float q = kf * rP;
    const float r = __builtin_fmaf(-q, Pf, kf);
    return __builtin_fmaf(r, rP, q);
}

// Persistent workgroups walk (query row, 28,672-column chunk) items x systems as one sequence of steps.  Per step: the
// system's table comes into LDS if another one is there (barrier, LDS-DMA, barrier), the 56 scores per thread are searched ILV
// float4s at a time -- each group's registers then take the NEXT step's scores, which cross HBM under this step's searches --
// their values weighted and added into the item's accumulators; after the item's last system the fused row leaves.
template <bool NCE, int ILV, int R>
__global__ __launch_bounds__(BT_T) void fuse_nsf_bigtab_kernel(NsfArgs a, BtArgs t, float* __restrict__ fused) {
    extern __shared__ __attribute__((aligned(16))) float bt_lds[];
    float* tabr = bt_lds;                                                   // [tab_cap]: BT_LEAD sentinels, the table, +inf
    float* lutf = bt_lds + t.tab_cap;                                       // [lut_floats]
    float* valr = bt_lds + t.tab_cap + t.lut_floats;                        // [tab_cap] when val_in_lds
    const lds_f32* tab = (const lds_f32*)(tabr + BT_LEAD);
    const lds_u16* lut = (const lds_u16*)(lutf);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float top = t.top;
    const int toff = 4 * threadIdx.x;
    const int chunks = (a.N + BT_COLS - 1) / BT_COLS;
    const int items = a.Q * chunks;   // (< 2^31: checked by the launcher)
    typedef float f4v __attribute__((ext_vector_type(4)));
    static_assert(BT_E4 % ILV == 0, "groups of ILV float4");
    static_assert(R == 1 || R == 2, "one or two items per table residency");
    constexpr int W = 4 * ILV;

    // global -> LDS, one 1-KiB piece per wave instruction (LDS destination = wave-uniform base + lane * 16)
    auto dma = [&](const float* __restrict__ src, float* dst, int floats) {
        for (int p = wave * BT_PIECE; p < floats; p += (BT_T / 64) * BT_PIECE)
            __builtin_amdgcn_global_load_lds(src + p + lane * 4, (__attribute__((address_space(3))) void*)(dst + p), 16, 0, 0);
    };
    // The thread's column offset, handed out through an empty asm so that the compiler recomputes the fourteen `offset + 2048 i`
    // where they are used: hoisted out of the step loop they (and what hangs off them) stay live across the whole kernel, the
    // register allocator spills them, and every reload is a `s_waitcnt vmcnt(0)` in front of the requests that should stay in flight.
    auto my_off = [&]() { int o = toff; asm volatile("" : "+v"(o)); return o; };
    // a step's scores: float4s past the row end re-read its last one (no branch, no mask: their columns are masked out of the sums)
    auto load_one = [&](const float* __restrict__ base, int i, int lim) {
        return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(base + min(my_off() + 4 * BT_T * i, lim)));   // streamed once
    };
    auto load_row = [&](f4v (&dst)[BT_E4], const float* __restrict__ base, int lim) {
#pragma unroll
        for (int i = 0; i < BT_E4; ++i) dst[i] = load_one(base, i, lim);
    };
    int cur = -1;   // the system whose table is in LDS
    f4v v[BT_E4];   // the current step's scores; a group's registers are refilled with the next step's as soon as it has been searched
    const float* nxt = a.planes[0];   // the next step's row chunk (wave-uniform) and its last float4
    int nlim = 0;
    // items it = blockIdx.x, + gridDim.x, ...: (row q, chunk c) advanced by hand -- a division per item would be done on the vector ALU,
    // in registers that live across the whole kernel.  A workgroup takes R of its items per ROUND and runs them system by system --
    // (item 0, s), (item 1, s), (item 0, s + 1) ... -- so that a table swap serves R items; R accumulator sets live in registers.
    int q = (int)blockIdx.x / chunks, c = (int)blockIdx.x - q * chunks;
    const int dq = (int)gridDim.x / chunks, dc = (int)gridDim.x - dq * chunks;
    auto advance = [&](int& qq, int& cc) { qq += dq; cc += dc; if (cc >= chunks) { cc -= chunks; ++qq; } };
    auto chunk_lim = [&](int cc) { return (min(a.N - cc * BT_COLS, BT_COLS) - 1) & ~3; };   // the chunk's last float4
    bool first = true;
    if ((int)blockIdx.x < items) load_row(v, a.planes[0] + (size_t)q * a.ld + c * BT_COLS, chunk_lim(c));   // the first step's scores
    for (int it = blockIdx.x; it < items; it += R * gridDim.x) {
        int qr[R], cr[R];
        bool live[R];
        qr[0] = q; cr[0] = c; live[0] = true;
#pragma unroll
        for (int r = 1; r < R; ++r) { qr[r] = qr[r - 1]; cr[r] = cr[r - 1]; advance(qr[r], cr[r]); live[r] = it + r * (int)gridDim.x < items; }
        int qn = qr[R - 1], cn = cr[R - 1];   // the next round's first item
        advance(qn, cn);
        const bool more = it + R * (int)gridDim.x < items;
        float acc[R][BT_E4][4];
        mask_t present[R], tail_ok[R];   // tail_ok: the thread's columns that lie inside the row
#pragma unroll
        for (int r = 0; r < R; ++r) {
            present[r] = 0; tail_ok[r] = 0;
            const int col0 = cr[r] * BT_COLS + my_off();
#pragma unroll
            for (int i = 0; i < BT_E4; ++i) {
                const int rem = a.N - (col0 + 4 * BT_T * i);
                tail_ok[r] |= (mask_t)(rem >= 4 ? 0xfu : (rem > 0 ? (1u << rem) - 1u : 0u)) << (4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][i][e] = 0.f;
            }
        }
        for (int s = 0; s < a.S; ++s) {
            const float lo_v = t.hdr[s][0], inv_w = t.hdr[s][1];
            const int steps = reinterpret_cast<const int*>(t.hdr[s])[2];
            const int last_pair = (t.Ppad[s] - BT_LEAD) / 2 - 1;
            const float w = a.w[s];
            const float Pf = (float)a.P[s], rP = 1.0f / Pf;
            auto step = [&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                const int q = qr[r], c = cr[r];
                const size_t rowoff = (size_t)q * a.ld;
                // which of the item's documents this system lists: wave-uniform bases + clamped per-thread offsets (no branches; what is
                // read past the row end is masked by tail_ok).  Requested BEFORE the table swap: the words arrive under its barrier and DMA
                mask_t ok = tail_ok[r];
                if (a.vbits[s]) {
                    const uint32_t* __restrict__ wb = a.vbits[s] + (size_t)q * a.ldb + ((c * BT_COLS) >> 5);
                    const int wlast = (min(a.N - c * BT_COLS, BT_COLS) - 1) >> 5, sh = toff & 31, wo = my_off() >> 5;
                    mask_t m = 0;
#pragma unroll
                    for (int i = 0; i < BT_E4; ++i) m |= (mask_t)((wb[min(wo + (4 * BT_T / 32) * i, wlast)] >> sh) & 0xfu) << (4 * i);
                    ok &= m;
                } else if (a.ranks[s]) {
                    const int32_t* __restrict__ rk = a.ranks[s] + rowoff + c * BT_COLS;
                    const int lim = chunk_lim(c);
                    mask_t m = 0;
#pragma unroll
                    for (int i = 0; i < BT_E4; ++i) {
                        const int4 rr = *reinterpret_cast<const int4*>(rk + min(my_off() + 4 * BT_T * i, lim));
                        uint32_t nib = (rr.x >= 0 ? 1u : 0u) | (rr.y >= 0 ? 2u : 0u) | (rr.z >= 0 ? 4u : 0u) | (rr.w >= 0 ? 8u : 0u);
                        asm volatile("" : "+v"(nib));   // one rank quad at a time: all of them in flight at once would be the kernel's register peak
                        m |= (mask_t)nib << (4 * i);
                    }
                    ok &= m;
                }
                // Requests to HBM run one step ahead: a group's registers take the same group of the NEXT step right after it has been
                // searched -- except the step's LAST group, whose request would be the youngest when the table swap waits for its LDS-DMA
                // (vmcnt counts in order: waiting for the DMA means waiting for everything older).  That one is issued here, BEHIND the
                // DMA, and the swap waits with vmcnt(ILV): the DMA and every older request have landed, the ILV youngest may still fly
                // (on the very first step the wait could return with DMA pieces in flight: it uses vmcnt(0)).
                const bool swap = cur != s;
                if (swap) {
                    __syncthreads();                       // every wave is done with the table that is there
                    dma(t.tab[s], tabr, t.Ppad[s]);
                    dma(t.lut[s], lutf, t.lut_floats);
                    if (NCE && t.val_in_lds) dma(t.val[s], valr, t.Ppad[s]);
                    cur = s;
                }
                if (!first) {   // (not the very first step, whose scores the prologue requested)
#pragma unroll
                    for (int i = BT_E4 - ILV; i < BT_E4; ++i) v[i] = load_one(nxt, i, nlim);
                }
                if (swap) {
                    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (ILV == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // this wave's pieces have landed ...
                    __syncthreads();                                        // ... and everybody else's
                }
                first = false;
                // the next step: the round's next live item under this system, else its first item under the next system, else the next
                // round's first item; after the very last step the requests re-read this step's own row (unconditional requests: no
                // phi copies of the score registers, no second wait form)
                {
                    bool found = false;
#pragma unroll
                    for (int r2 = r + 1; r2 < R; ++r2)
                        if (!found && live[r2]) { nxt = a.planes[s] + (size_t)qr[r2] * a.ld + cr[r2] * BT_COLS; nlim = chunk_lim(cr[r2]); found = true; }
                    if (!found) {
                        if (s + 1 < a.S) { nxt = a.planes[s + 1] + (size_t)qr[0] * a.ld + cr[0] * BT_COLS; nlim = chunk_lim(cr[0]); }
                        else if (more) { nxt = a.planes[0] + (size_t)qn * a.ld + cn * BT_COLS; nlim = chunk_lim(cn); }
                    }
                }
                uint32_t idx[NCE ? BT_E4 : 1][2];
                auto groups = [&](auto steps_c) {
                    constexpr int STEPS = decltype(steps_c)::value;
#pragma unroll
                    for (int g = 0; g < BT_E4; g += ILV) {
                        float x[W];
                        int best[W];
#pragma unroll
                        for (int j = 0; j < ILV; ++j) { x[4 * j] = v[g + j].x; x[4 * j + 1] = v[g + j].y; x[4 * j + 2] = v[g + j].z; x[4 * j + 3] = v[g + j].w; }
                        bt_lookup<W, STEPS>(tab, lut, lo_v, inv_w, top, steps, last_pair, x, best);
#pragma unroll
                        for (int k = 0; k < W; ++k) {
                            const int i = g + (k >> 2), e = k & 3;
                            if (!NCE) {
                                const float tr = bt_quot(best[k], Pf, rP);               // hybrid.py:275
                                const float prod = tr * w;                              // fl32(t * fl32(w))      hybrid.py:291 under NumPy 2
                                // a document the system does not list adds nothing: + (+0.0f) leaves every accumulator as it is (one is never
                                // -0.0: the sums start from +0.0); as a bit mask, so that the compiler does not branch around the arithmetic
                                const uint32_t keep = 0u - (uint32_t)((ok >> (4 * i + e)) & 1);
                                acc[r][i][e] = acc[r][i][e] + __uint_as_float(__float_as_uint(prod) & keep);
                            } else if (e & 1) idx[NCE ? i : 0][e >> 1] |= (uint32_t)best[k] << 16;   // two 16-bit indices per register (P <= 65535)
                            else idx[NCE ? i : 0][e >> 1] = (uint32_t)best[k];
                        }
#pragma unroll
                        for (int j = 0; j < ILV; ++j) {
                            // pin the group's sums (or indices) HERE: left alone the compiler sinks every group's value arithmetic to the end of the
                            // step, the 4 indices per group stay live until then, and what the register allocator then spills is the score registers
                            if (!NCE) asm volatile("" : "+v"(acc[r][g + j][0]), "+v"(acc[r][g + j][1]), "+v"(acc[r][g + j][2]), "+v"(acc[r][g + j][3]));
                            else asm volatile("" : "+v"(idx[NCE ? g + j : 0][0]), "+v"(idx[NCE ? g + j : 0][1]));
                        }
                        if (g + ILV < BT_E4) {
#pragma unroll
                            for (int j = 0; j < ILV; ++j) v[g + j] = load_one(nxt, g + j, nlim);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // one group's searches at a time: scheduled across groups, their temporaries push the score registers out
                    }
                };
                switch (steps <= BT_MAX_UNROLLED ? (steps < 1 ? 1 : steps) : 0) {   // (wave-uniform; fewer probes than the table needs would be wrong, more are not)
                    case 1: groups(std::integral_constant<int, 1>{}); break;
                    case 2: groups(std::integral_constant<int, 2>{}); break;
                    case 3: groups(std::integral_constant<int, 3>{}); break;
                    case 4: groups(std::integral_constant<int, 4>{}); break;
                    default: groups(std::integral_constant<int, 0>{}); break;
                }
                if (NCE) {
                    const lds_f32* vt = (const lds_f32*)valr;
                    if (!t.val_in_lds) {   // the values take the table's place
                        __syncthreads();
                        dma(t.val[s], tabr, t.Ppad[s]);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        cur = -1;
                        vt = (const lds_f32*)tabr;
                    }
#pragma unroll
                    for (int i = 0; i < BT_E4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float prod = vt[(idx[NCE ? i : 0][e >> 1] >> (16 * (e & 1))) & 0xffffu] * w;
                            const uint32_t keep = 0u - (uint32_t)((ok >> (4 * i + e)) & 1);
                            acc[r][i][e] = acc[r][i][e] + __uint_as_float(__float_as_uint(prod) & keep);
                        }
                }
                present[r] |= ok;
            };
            step(std::integral_constant<int, 0>{});
            if constexpr (R > 1) { if (live[R - 1]) step(std::integral_constant<int, R - 1>{}); }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r]) continue;
            const size_t rowoff = (size_t)qr[r] * a.ld;
            const int col0 = cr[r] * BT_COLS + my_off();
#pragma unroll
            for (int i = 0; i < BT_E4; ++i) {
                const int j0 = col0 + 4 * BT_T * i;
                if (j0 < a.N) {   // columns [N, ld) of the last float4 are padding of the plane: written, never read
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ((present[r] >> (4 * i + e)) & 1) ? acc[r][i][e] : -INFINITY;
                    *reinterpret_cast<float4*>(fused + rowoff + j0) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        q = qn; c = cn;
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

extern "C" size_t fz_nsf_tables_header_offset(int S, const int32_t* P_h, int norm, int s) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS || !P_h || s < 0 || s >= S || (norm != FZ_NORM_PERCENTILE && norm != FZ_NORM_NCE)) return (size_t)-1;
    const BtPlan p = bt_plan(S, P_h, norm == FZ_NORM_NCE);
    return p.ok ? p.sys_off[s] : (size_t)-1;
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
    bt_prepare_steps_kernel<<<S, 1024, 0, as_stream(stream)>>>(a, static_cast<unsigned char*>(workspace));
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
    t.top = (float)(p.lutb - 1);
    t.lutb = p.lutb; t.tab_cap = p.tab_cap; t.lut_floats = p.lut_floats; t.val_in_lds = p.val_in_lds ? 1 : 0;
    const long long items = (long long)Q * ((N + BT_COLS - 1) / BT_COLS);
    if (items >= (1ll << 31)) return FZ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)(items < 256 ? items : 256);
    static unsigned long long set_pr = 0ull, set_nce = 0ull;
    if (nce) {
        if (int rc = raise_lds_limit((const void*)fuse_nsf_bigtab_kernel<true, 1, BT_R>, p.lds_bytes, set_nce)) return rc;
        fuse_nsf_bigtab_kernel<true, 1, BT_R><<<grid, BT_T, p.lds_bytes, st>>>(a, t, fused);
    } else {
        if (int rc = raise_lds_limit((const void*)fuse_nsf_bigtab_kernel<false, 1, BT_R>, p.lds_bytes, set_pr)) return rc;
        fuse_nsf_bigtab_kernel<false, 1, BT_R><<<grid, BT_T, p.lds_bytes, st>>>(a, t, fused);
    }
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
