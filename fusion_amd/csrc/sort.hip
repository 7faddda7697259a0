// sort.hip -- stable descending row sort (K5a/K6) and top-k (A12) for gfx950.
//
// Reference semantics: Python sorted(items, key=score, reverse=True) is stable
// (bm25.py:104, hybrid.py:306); util.semantic_search / torch.topk + heap merge
// (hybrid.py:103, sentence_transformers.py:346-364) for top-k.
//
// Design (MI355X-first): one workgroup owns one row.  A row (<= 35,840 keys) lives entirely in
// the workgroup's registers (E keys per thread, "wave-striped": wave w, item i, lane l holds
// sequence position w*E*64 + i*64 + l), so HBM sees each key once in and once out.  fp32 keys
// take 4 LSD passes; fp64 keys take 4 passes over their HIGH word plus an in-place repair of the
// (short, adjacent) runs of equal high words -- the same permutation as 8 passes.  Each LSD pass
// over an 8-bit digit is
//   1. rank:     per item, the 64 lanes of a wave find their same-digit peers with 8 ballots
//                (wave64 match-any), and bump a wave-private LDS counter (no atomics, stable);
//   2. scan:     256 digits x NW waves counters -> exclusive prefix (digit-major);
//   3. exchange: scatter keys (then the 16-bit payload) through a T*E*4-byte LDS buffer
//                and read them back in striped order.
// Passes whose digit is constant over the row are skipped.  LDS: up to 156 KiB of the CU's
// 160 KiB, i.e. one 1024-thread workgroup per CU.
//
// Round 4: fp32 rows of 8,193 .. 28,672 columns (a 1,024-thread workgroup's rows) with at least 4,096 keys in them that look like a ranker's scores (few ties) skip the digit passes: BUCKET RANKING --
// a counting sort over 16,384 buckets whose widths follow the row's own density, every key's rank = its bucket's first slot + the
// bucket members below it, one neighbour check on the result (see `bucket_rank` in the kernel).  239 instead of 282 vector
// instructions per key and 42 % fewer LDS bank-conflict cycles (profiles/r04_pmc_sort_{bucket,digits}.json): 0.32 instead of 0.37 ms
// per 1024 x 27,942 cosine scores.  Rows it does not suit are found out early and take the digit passes; same permutation either way.
#include <type_traits>

#include "common.h"

namespace fz {

struct SortArgs {
    const void* keys;            // fp32 or fp64
    const int32_t* init_order;   // nullable [rows][key_row_stride]: column at sequence position r (gather)
    const int32_t* init_rank;    // nullable [rows][key_row_stride]: sequence position of column j, -1 = not in the sequence
                                 //   (same information as init_order, but loaded coalesced and placed through LDS)
    const int32_t* row_len;      // nullable [rows]
    int n_total;                 // elements per row (before chunking)
    long key_row_stride;         // elements between consecutive rows
    int seg_len;                 // element e lives at (e / seg_len) * seg_stride + row*key_row_stride + e % seg_len
    long seg_stride;             //   (seg_len >= n_total -> plain rows)
    int chunks;                  // pseudo-rows per row
    int chunk_len;               // chunk c covers [c*chunk_len, min(n_total,(c+1)*chunk_len))
    int32_t* order;              // nullable
    void* sorted_keys;           // nullable, same type as keys
    int32_t* rank;               // nullable (only chunks == 1)
    long out_row_stride;         // elements between rows of order / sorted_keys
    int out_chunk_stride;        // elements between chunks inside a row
    int out_limit;               // only the first out_limit entries of each pseudo-row are written
    const int32_t* colmap;       // nullable: order value = colmap[row*colmap_row_stride + col]
    long colmap_row_stride;
    const int64_t* idmap;        // nullable: out_ids = idmap[addr(col)] (same segment addressing as keys)
    int64_t id_base;             // else out_ids = id_base + col
    int64_t* out_ids;            // nullable, [rows][out_row_stride] like order
    int32_t* row_flags;          // fp64 keys: [rows*chunks] 1 = the fast form left the row to the generic one
    float* row_stats;            // nullable [4][stats_rows]: mean | unbiased std | min | max of the list's float32 values (the statistics of
    int stats_rows;              //   hybrid.py:254-262), a by-product of having the row in registers; chunks == 1 only
    const int32_t* stats_len;    // nullable [rows]: the statistics cover the first stats_len[row] entries of the SORTED list (a ranking
                                 //   truncated to its top-k: PLAID-style short lists, return_topk); fp32 keys only
    int bucket_rank;             // 1 = rows of a 1024-thread workgroup are ordered by the bucket ranking where it applies (set by the launcher)
    int zero_compact;            // 1 = float64 rows that are mostly exact zeros leave their zeros out of the ordering phases (ZC; set by the launcher)
    int expect_zeros;            // the caller expects such rows (fz_sort_rows_desc_lexical): the launcher picks the SORT_ROWS_ZC instantiation
    // FUSE (fz_sort_rank_fused_desc): there is no key plane -- the float64 key of column j is the rank fusion of hybrid.py:248-252,301-304,
    // formed on load from the S rank planes exactly as fuse_rank_kernel (fuse.hip) forms it: 0.0 + sum over the systems, in system order, of
    // 1/(60 + r + 1) (rrf) or (n - r + 1)/n (bcf) over the systems that list the document (r >= 0); -inf when none does
    const int32_t* fuse_ranks[FZ_MAX_SYSTEMS];   // [rows][key_row_stride] each
    const int32_t* fuse_lens;    // [S][fuse_rows] list lengths (bcf's n)
    int fuse_S, fuse_method, fuse_rows;
    double* fuse_gen_plane;      // [rows][key_row_stride]: where the fused scores of a row the fast form FLAGS are written out for the generic launch
    int fuse_first_is_pos;       // placed form with init_rank == fuse_ranks[0] (every list full: first-insertion order = system 0's ranking):
                                 //   the position just loaded IS system 0's rank, its plane is not read a second time
};

// rows the bucket ranking ordered | of those, rows whose neighbour check swapped a pair back | rows it gave up on after starting
// (ties, a crowd at the floor, an overfull bucket, a check it could not settle).  Read by fz_sort_bucket_rank_rows (tests, tools).
__device__ unsigned long long g_bucket_rank_rows[3];
// rows whose zeros were compacted away | rows that were looked at and kept whole (too few zeros).  Read by fz_sort_zero_compact_rows.
__device__ unsigned long long g_zero_compact_rows[2];

// Bucket ranking (see sort_rows_kernel): fine / coarse bucket counts and the words of LDS its tables take in the counter area.
constexpr int BR_FINE = 16384, BR_COARSE = 1024, BR_WORDS = BR_FINE / 2 + BR_COARSE + 1024;   // + one bit per slot (bucket starts)
constexpr int BR_MIN_KEYS = 4096, BR_MAX_BUCKET = 128;
constexpr int BR_STRAIGHT = 6;                       // bucket members the rank phase reads without a loop (8: 1 % slower, 4: 3 %)
template <int T, int E, int KW> struct SortLds {
    static constexpr bool br = T == 1024 && KW == 1 && (size_t)(32 + T * E + BR_WORDS) * 4 <= 160 * 1024;
    static constexpr size_t area = (size_t)(T / 64) * 256 * 4 + (KW == 2 ? (size_t)T * E : 0);   // counters (+ fp64: one move byte per slot)
    static constexpr size_t bytes = (size_t)(32 + T * E) * 4 + (br && area < (size_t)BR_WORDS * 4 ? (size_t)BR_WORDS * 4 : area);
};

// 1.0 / x in float64 for an integer-valued x in [1, 2^20]: the division's own Newton-Raphson steps (LLVM lowers an f64 division to
// v_div_scale x2, v_rcp_f64, five v_fma_f64, v_mul_f64, v_div_fmas, v_div_fixup; with a numerator of 1.0 and a denominator that needs no
// scaling the scale / fixup steps are identities and the multiplication is by 1.0) -- seven instructions instead of twelve, the same
// correctly rounded quotient: tests/test_gpu_rank_fused.py compares every x the sort can see (and 2^20 beyond) with the IEEE division.
__device__ __forceinline__ double recip_small_int_f64(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);       // the residual of the quotient (the numerator is 1.0: q = y)
    return __builtin_fma(e, y, y);
}
__global__ void rrf_terms_kernel(int count, int fast, double* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < count) out[r] = fast ? recip_small_int_f64((double)(60 + r + 1)) : 1.0 / (double)(60 + r + 1);
}

// FUSE: the fused scores of the rows the fast form flagged (practically never: > 2,048 keys sharing a high word), written out as a plain
// float64 row for the generic eight-pass launch -- whose FUSE instantiation (~500 B of scratch per lane) made even an EMPTY launch cost
// 40 us: above ROCr's per-queue scratch limit the dispatch allocates its scratch anew (283 MB of page zeroing per launch, rocprofv3).
// The arithmetic is fuse_rank_kernel's (IEEE division); every other row's workgroup exits at once.
__global__ __launch_bounds__(256) void fuse_flagged_rows_kernel(SortArgs a) {
    const int row = blockIdx.x;                      // one workgroup per row: rows + 1 wave launches to find out that nothing is flagged
    if (a.row_flags[row] == 0) return;
    for (int j = threadIdx.x; j < a.n_total; j += blockDim.x) {
        const size_t off = (size_t)row * a.key_row_stride + j;
        double acc = 0.0;
        bool present = false;
        for (int s = 0; s < a.fuse_S; ++s) {
            const int r = a.fuse_ranks[s][off];
            if (r >= 0) {
                const double n = (double)a.fuse_lens[(size_t)s * a.fuse_rows + row];
                acc = acc + (a.fuse_method == FZ_RRF ? 1.0 / (double)(60 + r + 1) : (n - (double)r + 1.0) / n);
                present = true;
            }
        }
        a.fuse_gen_plane[off] = present ? acc : -(double)INFINITY;
    }
}

// GEN (fp64 only): the generic eight-pass form, run as a second launch for the rows the fast form flags (see below).
// MODE: which of the kernel's many callers an instantiation serves -- what is not served is gone at compile time, and with it the row
// pointers and flags that otherwise stay live from the prologue to the output phase (they were the shipped hot kernels' spills):
//   SORT_ANY   every feature (chunked long rows, segmented top-k lists, column / id maps, id outputs);
//   SORT_ROWS  whole rows of one plane: the rankers' sorts and the final order (identity / gathered / placed sequence, statistics);
//   SORT_FUSE  SORT_ROWS whose float64 keys are formed from rank planes on load (fz_sort_rank_fused_desc), no statistics;
//   SORT_ROWS_ZC  SORT_ROWS for float64 rows the caller expects to be mostly exact zeros (a lexical ranker's scores:
//              fz_sort_rows_desc_lexical): the zeros stay out of the ordering phases (ZC below).  Its own instantiation because the
//              code that serves it costs a row WITHOUT zeros ~5 % (measured: one bit per key in the load phase, a scalar branch per
//              four items in every loop of the ordering phases, a handful of block-uniform branches) -- rows nobody expects zeros in
//              keep the SORT_ROWS kernel as it was.
enum { SORT_ANY = 0, SORT_ROWS = 1, SORT_FUSE = 2, SORT_ROWS_ZC = 3 };
template <int T, int E, int KW, bool GEN, int MODE = SORT_ANY>
__global__ __launch_bounds__(T) void sort_rows_kernel(SortArgs a) {
    constexpr bool FUSE = MODE == SORT_FUSE, LEAN = MODE != SORT_ANY;
    static_assert(!GEN || KW == 2, "the generic form exists for fp64 keys only");
    static_assert(!FUSE || KW == 2, "rank fusion forms float64 keys");
    static_assert(!(FUSE && GEN), "rows the fused fast form flags are sorted by the plain generic form from a materialised score row");
    constexpr int NW = T / 64;
    constexpr uint32_t SENT = 0xffffffffu;
    // ZC -- ZERO COMPACTION (round 6; float64 rows of one plane: BM25's ranking sort).  A lexical score row is mostly EXACT ZEROS (documents
    // that share no term with the query: ~60 % of the LLeQA-shaped bench rows), and a zero needs no sorting: its place in the list is
    // (keys above zero) + (zeros before it in sequence order).  When at least 3/8 of a row of >= 4,096 keys are +-0.0 the load phase's keys
    // are compacted -- the non-zero keys move, in sequence order, into a COMPACT striped layout of Ee = ceil(non-zeros / T) items per thread
    // (rounded up to 4) -- and the digit passes, the run detection, the low-word hand-over and the repair run over Ee items instead of E:
    // their cost is per item slot, whatever the slot holds.  The output phase puts sorted non-zero slot r at list position r (r < P = keys
    // above zero) or r + Z, and every zero at P + its index among the zeros; same stable permutation, bit for bit (a zero ties with nothing
    // but zeros, and those keep their order).  1024 x 27,942 BM25-like rows, 60 % zeros: 0.46 -> 0.2x ms (tools/bench_sort_zeros.py).
    constexpr bool ZC = KW == 2 && !GEN && MODE == SORT_ROWS_ZC && T == 1024 && E >= 16;
    [[maybe_unused]] int Ee = E;                  // items per thread in the ordering phases (ZC rows that were compacted; else E)
    // Loops over the items of the COMPACT layout stop at Ee (a multiple of 4: one scalar branch per four items; ~1 % on a row that kept its
    // zeros.  Instantiating the ordering phases once per path instead -- compacted / whole -- made hipcc spill 176 registers: measured, not kept).
#define ITEM_GUARD(i) if constexpr (ZC) { if ((((i) & 3) == 0) && (i) >= Ee) break; }
    constexpr int LG = (KW == 2) ? 7 : 14;         // global loads in flight per thread before the first use (one HBM latency per group)
    constexpr int LGF = FUSE ? 4 : LG;             // ... in the gathered / plain load phase (FUSE: the per-key sums need the registers)
    constexpr int WALK = 16;                        // longest equal-high-word run the fp64 repair re-sorts in place
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* misc = smem;                         // [32]
    uint32_t* exch = smem + 32;                    // [T*E]
    uint32_t* cnt = smem + 32 + T * E;             // [NW*256] (plain LDS pointer: volatile would lower to flat sc0 sc1 accesses); the bucket
                                                   //   ranking's tables (BR_WORDS words) start here too, over the fp64 move bytes behind it

    // LEAN serves whole rows only (the launcher sees to it): no chunks, segments, column / id maps -- known at compile time
    const int prow = blockIdx.x;
    const int row = LEAN ? prow : prow / a.chunks;
    const int chunk = LEAN ? 0 : prow - row * a.chunks;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int c0 = LEAN ? 0 : chunk * a.chunk_len;
    // this thread's first slot; SLOT_FRESH() makes it opaque again so that the 28 per-item slot numbers are re-derived (one add)
    // in every phase instead of being kept live -- 28 registers -- from the first phase to the last
    int slot0 = w * E * 64 + lane;
#define SLOT_FRESH() asm volatile("" : "+v"(slot0))
    if constexpr (GEN) { if (a.row_flags[prow] == 0) return; }                       // block-uniform: the fast form finished this row
    else if constexpr (KW == 2) { if (threadIdx.x == 0) a.row_flags[prow] = 0; }

    int m_row = a.row_len ? a.row_len[row] : a.n_total;
    m_row = m_row < 0 ? 0 : (m_row > a.n_total ? a.n_total : m_row);
    int m = m_row - c0;
    m = m < 0 ? 0 : (m > a.chunk_len ? a.chunk_len : m);
    const int m_all = m;          // (ZC: `m` becomes the number of non-zero keys for the ordering phases; the output phase works on the whole list)
    // statistics by-product (block-uniform): over the whole list from the load phase, or -- stats_len -- over the first slen entries
    // of the sorted list from the registers of the output phase
    const bool st_on = !FUSE && a.row_stats && (LEAN || a.chunks == 1);
    int slen = (st_on && a.stats_len) ? a.stats_len[row] : m;
    slen = slen < 0 ? 0 : (slen > m ? m : slen);
    const bool st_prefix = st_on && slen < m;
    const bool st_load = !GEN && st_on && !st_prefix;
    if (m == 0) {        // block-uniform: empty (pseudo-)row
        if (st_on && threadIdx.x == 0) {
            a.row_stats[row] = NAN; a.row_stats[a.stats_rows + row] = NAN;     // torch.mean / torch.std of an empty tensor
            a.row_stats[2 * a.stats_rows + row] = 0.f; a.row_stats[3 * a.stats_rows + row] = 0.f;
        }
        return;
    }

    // uniform row bases + 32-bit per-lane indices (saddr+voffset addressing; no 64-bit per-item addresses)
    const size_t krow = (size_t)row * a.key_row_stride;
    const bool seg = !LEAN && a.seg_len < a.n_total;  // block-uniform: segmented rows ([G][rows][k] top-k lists)
    const float* __restrict__ kf = reinterpret_cast<const float*>(a.keys) + krow;
    const uint2* __restrict__ kd = reinterpret_cast<const uint2*>(a.keys) + krow;   // a double as its (low, high) words
    const int32_t* __restrict__ init_row = a.init_order ? a.init_order + krow : nullptr;
    const int32_t* __restrict__ irow = a.init_rank ? a.init_rank + krow : nullptr;
    auto elem = [&](int col) -> long {  // element offset of column `col` relative to the row base
        return seg ? (long)(col / a.seg_len) * a.seg_stride + (col % a.seg_len) : (long)col;
    };
    // outputs
    const int lim = m < a.out_limit ? m : a.out_limit;
    const size_t obase = (size_t)row * a.out_row_stride + (size_t)chunk * a.out_chunk_stride;
    int32_t* __restrict__ o_order = a.order ? a.order + obase : nullptr;
    int64_t* __restrict__ o_ids = (!LEAN && a.out_ids) ? a.out_ids + obase : nullptr;
    float* __restrict__ o_kf = a.sorted_keys ? reinterpret_cast<float*>(a.sorted_keys) + obase : nullptr;
    uint32_t* __restrict__ o_kw = a.sorted_keys ? reinterpret_cast<uint32_t*>(reinterpret_cast<double*>(a.sorted_keys) + obase) : nullptr;   // fp64 keys, word by word
    int32_t* __restrict__ o_rank = a.rank ? a.rank + (size_t)row * a.out_row_stride : nullptr;
    const int32_t* __restrict__ cmap = (!LEAN && a.colmap) ? a.colmap + (size_t)row * a.colmap_row_stride : nullptr;
    const int64_t* __restrict__ imap = (!LEAN && a.idmap) ? a.idmap + krow : nullptr;

    // Register state, TWO words per key at any time (E = 28 keys x 3 words does not fit the 128 VGPRs of a 1024-thread
    // workgroup): fp32 -> ks = the key; fp64 -> ks = the HIGH key word during the passes, the LOW key word afterwards
    // (GEN: low word, four passes, high word, four passes, low word).
    uint32_t ks[E];
    uint32_t meta[E];  // low 16: payload (column inside the chunk); high 16: scratch (offset / destination)

    // fp64 key of the element at column `col` (re-derivable at any time from global memory: L2 / Infinity-Cache hits)
    auto key64 = [&](uint2 v) -> uint64_t { return desc_key_f64(__hiloint2double((int)v.y, (int)v.x)); };
    // FUSE: a key costs S divisions to form, and the fp64 form needs every key's words twice (high words for the passes, low words for
    // the repair).  Scratch use of the row's two outputs, both written by nobody else before the output phase:
    //   ORDER output (`stash`)      the LOW sort words of the load phase, indexed by the column / slot the thread itself loaded (it reads
    //                               back what it wrote; 4 B out + 4 B in per key from L2 / Infinity Cache instead of S divisions), live
    //                               until the low-word phase has published them through LDS (the barrier behind that loop);
    //   SORTED-SCORE output (o_kw)  word p < lim = the HIGH half of the sorted key at rank p, parked compactly at the head of the row once the
    //                               digit passes are done, live until the output phase has read ALL of them back (its s_waitcnt + barrier)
    //                               and stores every entry as one 8-byte word.
    // Placed rows that do not fill the row (entries of `order` beyond the list belong to the caller) and calls without an order output
    // form the low words again instead of parking them.
    uint32_t* const stash = reinterpret_cast<uint32_t*>(a.order ? a.order + obase : nullptr);
    const bool use_stash = FUSE && !GEN && stash != nullptr && (!irow || m == a.n_total);   // block-uniform
    // FUSE: the fused float64 scores of B columns (col < 0: not an element, its value is never used), in two halves so that callers can
    // put other work between them: fuse_issue = the rank loads of the FIRST round (N1 = 0, 1 or 2 systems from `s` on), fuse_finish =
    // the adds in system order from 0.0 like fuse_rank_kernel's -- r0v (r0: system 0's ranks are in the caller's registers: the placed
    // form's positions), the first round, then the remaining systems two at a time (loads, then adds).
    auto fuse_issue = [&](auto bt, auto n1_tag, const int (&col)[decltype(bt)::value], int s, int (&ra)[decltype(bt)::value],
                          int (&rb)[decltype(bt)::value]) __attribute__((always_inline)) {
        constexpr int B = decltype(bt)::value, N1 = decltype(n1_tag)::value;
        if constexpr (FUSE && N1 > 0) {
            const int32_t* __restrict__ pa = a.fuse_ranks[s] + krow;
            const int32_t* __restrict__ pb = a.fuse_ranks[N1 > 1 ? s + 1 : s] + krow;
#pragma unroll
            for (int k = 0; k < B; ++k) { const int c = col[k] < 0 ? 0 : col[k]; ra[k] = pa[c]; if constexpr (N1 > 1) rb[k] = pb[c]; }
        }
    };
    auto fuse_finish = [&](auto bt, auto n1_tag, const int (&col)[decltype(bt)::value], const int (&r0v)[decltype(bt)::value], const bool r0,
                           int s, int (&ra)[decltype(bt)::value], int (&rb)[decltype(bt)::value],
                           uint2 (&out)[decltype(bt)::value]) __attribute__((always_inline)) {
        constexpr int B = decltype(bt)::value, N1 = decltype(n1_tag)::value;
        if constexpr (FUSE) {
            double acc[B];
            uint32_t pres = 0u;
#pragma unroll
            for (int k = 0; k < B; ++k) acc[k] = 0.0;
            auto add = [&](int k, int r, double n) {
                if (r >= 0) {
                    const double c = a.fuse_method == FZ_RRF ? recip_small_int_f64((double)(60 + r + 1))   // hybrid.py:252: 1 / (60 + idx + 1)
                                                             : (n - (double)r + 1.0) / n;                  // hybrid.py:249 (sic)
                    acc[k] = acc[k] + c;
                    pres |= 1u << k;
                }
            };
            auto len_of = [&](int sx) -> double { return a.fuse_method == FZ_BCF ? (double)a.fuse_lens[(size_t)sx * a.fuse_rows + row] : 0.0; };
            if (r0) {
                const double n0 = len_of(0);
#pragma unroll
                for (int k = 0; k < B; ++k) if (col[k] >= 0) add(k, r0v[k], n0);
            }
            if constexpr (N1 > 0) {
                const double na = len_of(s), nb = len_of(N1 > 1 ? s + 1 : s);
#pragma unroll
                for (int k = 0; k < B; ++k) if (col[k] >= 0) { add(k, ra[k], na); if constexpr (N1 > 1) add(k, rb[k], nb); }
            }
            for (int s2 = s + N1; N1 == 2 && s2 < a.fuse_S; s2 += 2) {                          // (block-uniform; S > 3 only)
                const bool two = s2 + 1 < a.fuse_S;
                if (two) fuse_issue(bt, std::integral_constant<int, 2>{}, col, s2, ra, rb); else fuse_issue(bt, std::integral_constant<int, 1>{}, col, s2, ra, rb);
                __builtin_amdgcn_sched_barrier(0);
                const double na = len_of(s2), nb = len_of(two ? s2 + 1 : s2);
#pragma unroll
                for (int k = 0; k < B; ++k) if (col[k] >= 0) { add(k, ra[k], na); if (two) add(k, rb[k], nb); }
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const double v = ((pres >> k) & 1u) ? acc[k] : -(double)INFINITY;
                out[k] = make_uint2((uint32_t)__double2loint(v), (uint32_t)__double2hiint(v));
            }
        }
    };
    // how many systems the first round loads: 2 when there are two left after r0 (one memory round trip for S <= 2 + r0), else 1 or 0
    auto fuse_vals = [&](auto bt, const int (&col)[decltype(bt)::value], const int (&r0v)[decltype(bt)::value], const bool r0,
                         uint2 (&out)[decltype(bt)::value]) __attribute__((always_inline)) {
        constexpr int B = decltype(bt)::value;
        if constexpr (FUSE) {
            const int s = r0 ? 1 : 0, left = a.fuse_S - s;
            int ra[B], rb[B];
            auto go = [&](auto n1_tag) __attribute__((always_inline)) {
                fuse_issue(bt, n1_tag, col, s, ra, rb);
                __builtin_amdgcn_sched_barrier(0);
                fuse_finish(bt, n1_tag, col, r0v, r0, s, ra, rb, out);
            };
            if (left >= 2) go(std::integral_constant<int, 2>{});
            else if (left == 1) go(std::integral_constant<int, 1>{});
            else go(std::integral_constant<int, 0>{});
        }
    };

    if (threadIdx.x < 8) misc[8 + threadIdx.x] = (threadIdx.x == 1 || threadIdx.x == 3) ? 0xffffffffu : 0u;  // [8] = or, [9] = and of the sort words; [10] = max, [11] = min, [12] .. [15] = sums (bucket ranking); ZC: [10] = or, [11] = and of the NON-ZERO keys' words, [12] = keys above zero
    uint32_t orw = 0, andw = 0xffffffffu;
    [[maybe_unused]] uint32_t zm = 0u;                                 // ZC: bit i = this thread's item i holds +-0.0
    const bool zc_try = ZC && a.zero_compact && !irow && !init_row && m >= 4096 && a.order;   // block-uniform (the order output: where the low words are parked)
    if (irow) {
        // placed sequence: column j sits at sequence position init_rank[j].  Keys and positions are read coalesced by
        // column; the sort word, then the payload, are scattered to exch[position] and read back in striped order -- the
        // random walk happens in LDS instead of as 27,942 uncoalesced HBM reads per row.  The positions are read twice
        // (second time from L2) rather than held in 28 registers.
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) exch[(slot0 + i * 64)] = SENT;
        __syncthreads();
        SLOT_FRESH();
        if constexpr (!FUSE) {
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += LG) {
            uint32_t pos_t[LG], lo_t[LG], hi_t[LG];
#pragma unroll
            for (int i = i0; i < (i0 + LG < E ? i0 + LG : E); ++i) {
                const int j = (slot0 + i * 64);
                const bool in = j < a.n_total;
                pos_t[i - i0] = (uint32_t)(in ? irow[j] : -1);
                if (KW == 1) lo_t[i - i0] = __float_as_uint(kf[in ? j : 0]);
                else { const uint2 v = kd[in ? j : 0]; lo_t[i - i0] = v.x; hi_t[i - i0] = v.y; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = i0; i < (i0 + LG < E ? i0 + LG : E); ++i) {
                const int j = (slot0 + i * 64);
                const uint32_t kw = KW == 1 ? desc_key_f32(__uint_as_float(lo_t[i - i0]))
                                            : (uint32_t)(key64(make_uint2(lo_t[i - i0], hi_t[i - i0])) >> (GEN ? 0 : 32));
                if (j < a.n_total && pos_t[i - i0] < (uint32_t)m) exch[pos_t[i - i0]] = kw;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        } else {
            // FUSE: a batch's keys take S divisions each to form -- the NEXT batch's loads (positions + the first round's ranks) are issued
            // before that arithmetic, not after it: two register stages, the batch loop fully unrolled
            constexpr int NB = (E + LG - 1) / LG;
            const int s0 = a.fuse_first_is_pos ? 1 : 0, left = a.fuse_S - s0;                   // block-uniform
            auto run = [&](auto n1_tag) __attribute__((always_inline)) {
                int col_t[2][LG], pos_i[2][LG], ra[2][LG], rb[2][LG];
                auto issue = [&](int b, int st) __attribute__((always_inline)) {
#pragma unroll
                    for (int k = 0; k < LG; ++k) {
                        const int j = (slot0 + (b * LG + k) * 64);
                        const bool in = b * LG + k < E && j < a.n_total;
                        col_t[st][k] = in ? j : -1;
                        pos_i[st][k] = in ? irow[j] : -1;
                    }
                    fuse_issue(std::integral_constant<int, LG>{}, n1_tag, col_t[st], s0, ra[st], rb[st]);
                };
                issue(0, 0);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int st = b & 1;
                    if (b + 1 < NB) issue(b + 1, st ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    uint2 v_t[LG];
                    fuse_finish(std::integral_constant<int, LG>{}, n1_tag, col_t[st], pos_i[st], a.fuse_first_is_pos != 0, s0, ra[st], rb[st], v_t);
#pragma unroll
                    for (int k = 0; k < LG; ++k) {
                        const int j = col_t[st][k];
                        const uint64_t k64 = key64(v_t[k]);
                        if (!GEN && use_stash && j >= 0) stash[j] = (uint32_t)k64;
                        if (j >= 0 && (uint32_t)pos_i[st][k] < (uint32_t)m) exch[pos_i[st][k]] = (uint32_t)(k64 >> (GEN ? 0 : 32));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (left >= 2) run(std::integral_constant<int, 2>{});
            else if (left == 1) run(std::integral_constant<int, 1>{});
            else run(std::integral_constant<int, 0>{});
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) ks[i] = exch[(slot0 + i * 64)];
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) exch[(slot0 + i * 64)] = 0xffffu;
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += 2 * LG) {
            uint32_t pos_t[2 * LG];
#pragma unroll
            for (int i = i0; i < (i0 + 2 * LG < E ? i0 + 2 * LG : E); ++i) {
                const int j = (slot0 + i * 64);
                pos_t[i - i0] = (uint32_t)(j < a.n_total ? irow[j] : -1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = i0; i < (i0 + 2 * LG < E ? i0 + 2 * LG : E); ++i) {
                const int j = (slot0 + i * 64);
                if (j < a.n_total && pos_t[i - i0] < (uint32_t)m) exch[pos_t[i - i0]] = (uint32_t)j;   // payload = own column (< 65535)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            meta[i] = exch[(slot0 + i * 64)];
            if (meta[i] != 0xffffu) { orw |= ks[i]; andw &= ks[i]; }   // a real element landed in this slot
        }
        __syncthreads();
    } else {
        double s1 = 0.0, s2 = 0.0, sn = 0.0, x0 = 0.0;   // row statistics (only when a.row_stats, whole-list form)
        if (!FUSE && st_load) {
            float f0;
            if (KW == 1) f0 = kf[elem(init_row ? 0 : c0)];
            else { const uint2 v0 = kd[elem(init_row ? 0 : c0)]; f0 = (float)__hiloint2double((int)v0.y, (int)v0.x); }
            x0 = (f0 == f0 && fabsf(f0) != INFINITY) ? (double)f0 : 0.0;   // any finite sample of the row keeps the one-pass sums stable
        }
        SLOT_FRESH();
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += LGF) {
            uint32_t lo_t[LGF], hi_t[KW == 2 ? LGF : 1];
            [[maybe_unused]] int col_t[FUSE ? LGF : 1];
#pragma unroll
            for (int i = i0; i < (i0 + LGF < E ? i0 + LGF : E); ++i) {
                // branch-free: out-of-range lanes load a safe element (column c0 exists because m > 0) and discard it
                const int p = (slot0 + i * 64);
                const bool valid = p < m;
                int col = valid ? c0 + p : c0;
                if (init_row) col = init_row[col];
                const bool ok = valid && (unsigned)col < (unsigned)a.n_total;
                // payload: the source column (gathered sequence) or the column inside the chunk
                meta[i] = ok ? (uint32_t)(init_row ? col : col - c0) : 0xffffu;
                col = ok ? col : c0;
                if constexpr (FUSE) col_t[i - i0] = ok ? col : -1;
                else if (KW == 1) lo_t[i - i0] = __float_as_uint(kf[elem(col)]);
                else { const uint2 v = kd[elem(col)]; lo_t[i - i0] = v.x; hi_t[KW == 2 ? i - i0 : 0] = v.y; }
            }
            if constexpr (FUSE) {
#pragma unroll
                for (int k = (i0 + LGF < E ? LGF : E - i0); k < LGF; ++k) col_t[k] = -1;
                uint2 v_t[LGF];
                if (init_row) __builtin_amdgcn_sched_barrier(0);
                fuse_vals(std::integral_constant<int, LGF>{}, col_t, col_t, false, v_t);
#pragma unroll
                for (int k = 0; k < LGF; ++k) { lo_t[k] = v_t[k].x; hi_t[k] = v_t[k].y; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = i0; i < (i0 + LGF < E ? i0 + LGF : E); ++i) {
                const bool ok = meta[i] != 0xffffu;
                uint32_t kw;
                if constexpr (KW == 1) kw = desc_key_f32(__uint_as_float(lo_t[i - i0]));
                else {
                    const uint64_t k64 = key64(make_uint2(lo_t[i - i0], hi_t[KW == 2 ? i - i0 : 0]));
                    kw = (uint32_t)(k64 >> (GEN ? 0 : 32));
                    if constexpr (FUSE && !GEN) { if (use_stash && ok) stash[(slot0 + i * 64)] = (uint32_t)k64; }
                    if constexpr (ZC) zm |= (uint32_t)(ok && ((hi_t[i - i0] << 1) | lo_t[i - i0]) == 0u) << i;   // +-0.0 (all the load phase pays for ZC)
                }
                ks[i] = ok ? kw : SENT;
                orw |= ok ? kw : 0u; andw &= ok ? kw : 0xffffffffu;
                if (st_load) {   // the value as the normalisations see it: float32 (BM25's float64 scores rounded, hybrid.py:261)
                    const float xf = KW == 1 ? __uint_as_float(lo_t[i - i0]) : (float)__hiloint2double((int)hi_t[KW == 2 ? i - i0 : 0], (int)lo_t[i - i0]);
                    const double dd = ok ? (double)xf - x0 : 0.0;
                    s1 += dd; s2 = fma(dd, dd, s2); sn += ok ? 1.0 : 0.0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();   // misc[8..9] initialised
        if (st_load) {   // block-uniform
            // z-score statistics of the row (torch.mean / torch.std of the float32 scores, hybrid.py:261-262): one pass, shifted by
            // the row's first value, fp64 throughout; wave sums by shuffles, the NW partial triples through the (still unused)
            // counter area, folded in wave order by every thread identically
            s1 = wave_reduce_sum(s1); s2 = wave_reduce_sum(s2); sn = wave_reduce_sum(sn);
            double* red = reinterpret_cast<double*>(cnt);
            if (lane == 0) { red[w] = s1; red[NW + w] = s2; red[2 * NW + w] = sn; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double S1 = 0.0, S2 = 0.0, n = 0.0;
                for (int i = 0; i < NW; ++i) { S1 += red[i]; S2 += red[NW + i]; n += red[2 * NW + i]; }
                const double mean = n > 0.0 ? x0 + S1 / n : (double)NAN;
                const double var = n > 1.0 ? (S2 - S1 * S1 / n) / (n - 1.0) : (double)NAN;
                a.row_stats[row] = (float)mean;
                a.row_stats[a.stats_rows + row] = (float)sqrt(var < 0.0 ? 0.0 : var);
            }
            __syncthreads();
        }
    }
    // which bits of the sort word vary over the row?
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { orw |= __shfl_xor(orw, o, 64); andw &= __shfl_xor(andw, o, 64); }
    if (lane == 0) { atomicOr(&misc[8], orw); atomicAnd(&misc[9], andw); }
    if constexpr (ZC) {
        if (zc_try) {   // (block-uniform) per wave: zeros | non-zero keys << 16 -> misc[16 + w]
            int nin = (m - slot0 + 63) >> 6;                   // this thread's items inside the row (identity sequence: slot < m)
            nin = nin < 0 ? 0 : (nin > E ? E : nin);
            const uint32_t zt = (uint32_t)__popc(zm);
            uint32_t pk = zt | (((uint32_t)nin - zt) << 16);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) pk += __shfl_xor(pk, o, 64);
            if (lane == 0) misc[16 + w] = pk;
        }
    }
    __syncthreads();
    uint32_t diff = misc[8] ^ misc[9];                // bits that vary among the REAL keys: a digit without any is skipped
    // Slots beyond the sequence hold the all-ones sentinel and take part in the wave ballots: a bit may only be left out
    // of the match when it is constant over the sentinels too, i.e. constant ONE.
    bool has_sent = irow || init_row || m < T * E;
    uint32_t diff_match = has_sent ? ~misc[9] : diff;
    // ZC: the row's zeros leave the ordering phases (see the note at ZC's definition)
    [[maybe_unused]] int zc_Z = 0, zc_P = 0, zc_zbase = 0, zc_nzbase = 0;   // zeros in the row | keys above zero | zeros / non-zero keys in the waves before this one
    [[maybe_unused]] bool compacted = false;                    // block-uniform
    if constexpr (ZC) {
        if (zc_try) {
            uint32_t nz_all = 0, z_all = 0, nz_before = 0, z_before = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const uint32_t c = misc[16 + ww];
                z_all += c & 0xffffu; nz_all += c >> 16;
                if (ww < w) { z_before += c & 0xffffu; nz_before += c >> 16; }
            }
            if (nz_all > 0u && z_all * 8u >= (uint32_t)m * 3u) {
                compacted = true;
                zc_Z = (int)z_all; zc_zbase = (int)z_before; zc_nzbase = (int)nz_before;
                Ee = (((int)nz_all + T - 1) / T + 3) & ~3;
                // The non-zero keys, in sequence order, to COMPACT positions: (non-zero keys of the waves before) + (of this wave's earlier
                // items) + (of the lanes below).  High sort words through LDS; the LOW sort words -- the ordering phases need them once more,
                // after the digit passes -- are formed again here, while the row is still warm in L2 (the whole-row form re-reads all of it
                // from the Infinity Cache / HBM a hundred microseconds later), and parked by compact position in the row's ORDER output,
                // which nobody writes before the output phase.  The payload of the ordering phases is the compact position itself: the
                // columns are handed over once, at the end.
                uint32_t run = nz_before;
                SLOT_FRESH();
#pragma unroll
                for (int i0 = 0; i0 < E; i0 += 7) {
                    uint32_t lo_t[7], hi_t[7], pos_t[7];
#pragma unroll
                    for (int i = i0; i < (i0 + 7 < E ? i0 + 7 : E); ++i) {
                        const int p = slot0 + i * 64;
                        const bool mv = p < m && !((zm >> i) & 1u);
                        const unsigned long long bal = __ballot(mv);
                        const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        run += (uint32_t)__popcll(bal);
                        pos_t[i - i0] = mv ? pos : 0xffffffffu;
                        const uint2 v = kd[mv ? p : 0];
                        lo_t[i - i0] = v.x; hi_t[i - i0] = v.y;
                        if (mv) exch[pos] = ks[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = i0; i < (i0 + 7 < E ? i0 + 7 : E); ++i)
                        if (pos_t[i - i0] != 0xffffffffu) stash[pos_t[i - i0]] = (uint32_t)key64(make_uint2(lo_t[i - i0], hi_t[i - i0]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                slot0 = w * Ee * 64 + lane;                    // from here to the output phase: the compact layout
                m = (int)nz_all;
                uint32_t orz = 0u, andz = 0xffffffffu, pth = 0u;   // or / and of the non-zero keys' words; keys above zero (sign bit of the sort word clear: NaN, +x, +denormal)
                SLOT_FRESH();
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    ITEM_GUARD(i)
                    const int p = slot0 + i * 64;
                    const bool in = p < m;
                    const uint32_t kw = exch[in ? p : 0];
                    ks[i] = in ? kw : SENT;
                    meta[i] = in ? (uint32_t)p : 0xffffu;       // payload = compact position
                    orz |= in ? kw : 0u; andz &= in ? kw : 0xffffffffu; pth += (in && (int)kw >= 0) ? 1u : 0u;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    orz |= __shfl_xor(orz, o, 64); andz &= __shfl_xor(andz, o, 64); pth += __shfl_xor(pth, o, 64);
                }
                if (lane == 0) { atomicOr(&misc[10], orz); atomicAnd(&misc[11], andz); atomicAdd(&misc[12], pth); }
                __syncthreads();
                zc_P = (int)misc[12];
                const uint32_t orn = misc[10], andn = misc[11];
                diff = orn ^ andn;
                has_sent = m < T * Ee;
                diff_match = has_sent ? ~andn : diff;
            }
            if (threadIdx.x == 0) atomicAdd(&g_zero_compact_rows[compacted ? 0 : 1], 1ull);
        }
    }
    // ZC, compacted row: sorted non-zero slot p stands at list position p (above zero) or p + Z (below zero): where its outputs go
    [[maybe_unused]] auto fpos = [&](int p) -> int { if constexpr (ZC) return (compacted && p >= zc_P) ? p + zc_Z : p; else return p; };

    // One stable LSD pass over the 8-bit digit at `shift` of ks (which travels with the payload).
    // dmask: the digit's bits that vary over the row.
    auto radix_pass = [&](const int shift, const uint32_t dmask) {
        // ---- 1. rank inside the wave ------------------------------------------------
        uint32_t* my = cnt + w * 256;
        my[lane] = 0; my[lane + 64] = 0; my[lane + 128] = 0; my[lane + 192] = 0;
        // FULL (wave-uniform: every bit of the digit varies) is one straight-line block over the 28 items, no scalar branch per bit
        auto rank_items = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                ITEM_GUARD(i)
                const uint32_t kw = ks[i];
                const uint32_t d = (kw >> shift) & 0xffu;
                // wave64 match-any on the digit's VARYING bits (a bit that is constant over the row cannot separate peers;
                // dmask is wave-uniform, the skip is a scalar branch), 4 VALU per bit: sign-extended bit (0 / -1), ballot,
                // and mask &= ~(ballot ^ sext) as one v_bitop3 per 32-lane half
                uint32_t mlo = 0xffffffffu, mhi = 0xffffffffu;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (!FULL && !((dmask >> b) & 1u)) continue;
                    const uint32_t sx = (uint32_t)__builtin_amdgcn_sbfe((int)kw, (unsigned)(shift + b), 1u);
                    const unsigned long long bal = __ballot(sx != 0u);
                    mlo &= ~(((uint32_t)bal) ^ sx);
                    mhi &= ~(((uint32_t)(bal >> 32)) ^ sx);
                }
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
                const uint32_t npeer = __popc(mlo) + __popc(mhi);
                const uint32_t old = my[d];
                if (below == 0) my[d] = old + npeer;
                // bytes {3,2} <- (old+below), bytes {1,0} <- payload : one v_perm_b32
                meta[i] = __builtin_amdgcn_perm(old + below, meta[i], 0x05040100u);
                // opaque to the optimiser: otherwise hipcc keeps old, below, payload and &my[d] in four separate
                // registers per item across the barrier (6.6 VGPRs/item -> scratch spills at E = 28)
                asm volatile("" : "+v"(meta[i]));
            }
        };
        if (dmask == 0xffu) rank_items(std::true_type{}); else rank_items(std::false_type{});
        __syncthreads();
        // ---- 2. exclusive prefix over (digit, wave), digit-major ----------------------
        uint32_t tot = 0, incl = 0;
        if (threadIdx.x < 256) {
            uint32_t run = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                uint32_t c = cnt[ww * 256 + threadIdx.x];
                cnt[ww * 256 + threadIdx.x] = run;
                run += c;
            }
            tot = run;
            incl = wave_incl_scan_u32(tot, lane);
            if (lane == 63) misc[w] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t base = incl - tot;
            for (int ww = 0; ww < w; ++ww) base += misc[ww];
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) cnt[ww * 256 + threadIdx.x] += base;
        }
        __syncthreads();
        // ---- 3. destination, exchange --------------------------------------------------
        // (sched_barrier every 4 items: without it hipcc hoists all E LDS addresses/values and spills)
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            uint32_t kw = ks[i];
            asm volatile("" : "+v"(kw));  // recompute the digit here instead of keeping &my[d] alive per item
            const uint32_t d = (kw >> shift) & 0xffu;
            const uint32_t dst = my[d] + (meta[i] >> 16);
            meta[i] = __builtin_amdgcn_perm(dst, meta[i], 0x05040100u);
            asm volatile("" : "+v"(meta[i]));
            exch[dst] = kw;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            ks[i] = exch[(slot0 + i * 64)];
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            exch[meta[i] >> 16] = meta[i] & 0xffffu;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            meta[i] = exch[(slot0 + i * 64)];
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    // move (ks, payload) to the slots in meta's high halves (a permutation of [0, m)); slots >= m keep theirs
    auto permute_to_meta_hi = [&]() __attribute__((always_inline)) {
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) if ((slot0 + i * 64) < m) exch[meta[i] >> 16] = ks[i]; }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) if ((slot0 + i * 64) < m) ks[i] = exch[(slot0 + i * 64)]; }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) if ((slot0 + i * 64) < m) exch[meta[i] >> 16] = meta[i] & 0xffffu; }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) if ((slot0 + i * 64) < m) meta[i] = exch[(slot0 + i * 64)]; }
        __syncthreads();
    };
    // fp64: (re)load one key word of every slot's element from global memory, by payload (a gather; rare paths only)
    auto reload_word = [&](bool high) {
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const uint32_t py = meta[i] & 0xffffu;
            const int col = py == 0xffffu ? c0 : (int)py + ((init_row || irow) ? 0 : c0);
            uint2 kv[1];
            if constexpr (FUSE) { const int c1[1] = {col}; fuse_vals(std::integral_constant<int, 1>{}, c1, c1, false, kv); }
            else kv[0] = kd[elem(col)];
            const uint64_t kk = key64(kv[0]);
            ks[i] = py == 0xffffu ? SENT : (high ? (uint32_t)(kk >> 32) : (uint32_t)kk);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- bucket ranking: the stable order by the 32-bit sort word WITHOUT the digit passes ------------------------------------------
    // A row of a ranker's scores is a sample of a smooth distribution: spread over 16,384 buckets whose widths follow the row's own
    // density, a key shares its bucket with one or two others, and its rank is the bucket's first slot plus the number of bucket
    // members that precede it -- a handful of comparisons where the four digit passes spend a ballot per key bit.
    //   0. range of the sort words; t(w) = w counted from the row's first word, with the unused exponents between the smallest positive
    //      and negative magnitudes cut out and magnitudes 24 binades under the largest clamped; 1,024 equal COARSE buckets (c = t >> s1);
    //   1. a histogram of 4 keys per thread over the coarse buckets -> coarse bucket c is cut into nsub[c] ~ its share of the keys FINE
    //      buckets: fine(w) = base[c] + floor(frac * nsub[c]), rem(w) = the next 16 bits of that product.  (fine, rem) is a monotone
    //      function of w, exact (distinct words, distinct values) wherever a coarse bucket holds more than a few dozen keys;
    //   2. counting sort by fine bucket: packed 16-bit LDS counters (atomic add), exclusive scan, a second atomic add on the
    //      offsets hands out the slots; a slot holds rem << 16 | sequence position; a bitmap marks the buckets' first slots;
    //   3. the thread of SLOT d counts the members of d's bucket below d's entry: that entry's rank (16 bits per slot);
    //   4. words, then payloads, travel owner -> entry slot -> rank through LDS, and every slot checks its left neighbour: two DISTINCT
    //      words that shared (fine, rem) stand in position order instead of word order;
    //   5. such a pair is swapped back.  Anything else -- three in a row; and before that: rows with heavy ties, a crowd at the floor, a
    //      bucket over BR_MAX_BUCKET keys, holes in the sequence, fewer than three varying key bytes -- is left to the digit passes: the
    //      arrangement is at every point a stable permutation by a coarsening of the word, so they finish it to the same result.
    // All integer arithmetic on the sort word: NaN / inf / signed zeros are whatever desc_key made of them.
    constexpr bool BR = !GEN && SortLds<T, E, KW>::br;
    auto bucket_rank = [&]() __attribute__((always_inline)) -> bool {
      if constexpr (!BR) return false; else {
        uint32_t* cntF = cnt;                       // [BR_FINE / 2]: two 16-bit counters per word (fine bucket 2k low, 2k + 1 high)
        uint32_t* tab = cnt + BR_FINE / 2;          // [BR_COARSE]: sample count, then base << 16 | nsub
        uint32_t* bm = tab + BR_COARSE;             // [T*E/32 + 1]: bit p = slot p is the first of its bucket; bit m closes the last one
        const int t = threadIdx.x;
        auto wg_excl_scan = [&](uint32_t v, uint32_t& total) -> uint32_t {   // exclusive prefix over the workgroup's threads
            const uint32_t incl = wave_incl_scan_u32(v, lane);
            if (lane == 63) misc[16 + w] = incl;
            __syncthreads();
            uint32_t base = 0, tot = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { const uint32_t x = misc[16 + ww]; base += ww < w ? x : 0u; tot += x; }
            __syncthreads();
            total = tot;
            return base + incl - v;
        };
        // 0. range; holes (a slot below m without an element: placed / gathered sequences only); equal neighbours among a thread's keys
        //    (a row that is half zeros is turned away here, before its samples queue up on one LDS address)
        uint32_t mn = 0xffffffffu, mx = 0u, eq = 0u;
        bool hole = false;
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if ((slot0 + i * 64) < m) {
                mn = ks[i] < mn ? ks[i] : mn; mx = ks[i] > mx ? ks[i] : mx;
                hole |= (meta[i] & 0xffffu) == 0xffffu;
                if (i > 0) eq += ks[i] == ks[i > 0 ? i - 1 : 0] ? 1u : 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a1 = __shfl_xor(mn, o, 64), b1 = __shfl_xor(mx, o, 64);
            mn = a1 < mn ? a1 : mn; mx = b1 > mx ? b1 : mx;
            eq += __shfl_xor(eq, o, 64);
        }
        if (lane == 0) { atomicMin(&misc[11], mn); atomicMax(&misc[10], mx); atomicAdd(&misc[13], eq); }   // initialised before the load phase
        {
            uint4* z = reinterpret_cast<uint4*>(cnt);
            for (int x = t; x < BR_WORDS / 4; x += T) z[x] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (__syncthreads_or(hole ? 1 : 0)) return false;
        if (misc[13] > (uint32_t)m / 16u) return false;                // more than ~6 % equal neighbours (block-uniform)
        // The sort word of a float: 0x7fffffff - |bits| for x >= 0, 0x80000000 + |bits| for x < 0 -- between the smallest positive and the
        // smallest negative value of the row lie all the exponents nobody uses.  Magnitudes more than 24 binades below the row's largest are
        // clamped to that floor (they share one bucket value; a stray one is the check's business) and the gap is cut out: t counts sort
        // words from the row's first, over at most 2 x 24 binades.
        const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[11]), w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[10]);
        const uint32_t magp = w_lo < 0x80000000u ? 0x7fffffffu - w_lo : 0u, magn = w_hi >= 0x80000000u ? w_hi - 0x80000000u : 0u;
        const uint32_t magmax = magp > magn ? magp : magn;
        constexpr uint32_t FLOOR = 24u << (KW == 1 ? 23 : 20);         // (fp64: the high word, exponent at bit 20)
        const uint32_t eps = magmax > FLOOR ? magmax - FLOOR : 0u;
        const uint32_t zp = 0x7fffffffu - eps, zn = 0x80000000u + eps;
        auto t_of = [&](uint32_t kw, uint32_t base_p, uint32_t base_n) -> uint32_t {   // base_n = base_p + the gap's width
            const bool neg = kw >= 0x80000000u;
            const uint32_t lo = kw < zp ? kw : zp, hi = kw > zn ? kw : zn;
            return (neg ? hi : lo) - (neg ? base_n : base_p);
        };
        const uint32_t tmin = t_of(w_lo, 0u, 2u * eps);
        const uint32_t range = t_of(w_hi, 0u, 2u * eps) - tmin;
        if (range == 0u) return false;              // (every key at the floor)
        const uint32_t tb_p = tmin, tb_n = tmin + 2u * eps;
        const int nbits = 32 - __builtin_clz(range);
        const int s1 = nbits > 10 ? nbits - 10 : 0;
        const uint32_t shl = (uint32_t)(32 - s1) & 31u, fmask = s1 ? 0xffffffffu : 0u;
        // 1. sample: four keys per thread into the coarse histogram -- and into a 2,048-slot hash table of sort words (the still unused
        //    fine counters): a sample that finds its own word there is a TIE.  A row with heavy ties (SPLADE's or BM25's zeros, a few hundred
        //    distinct values) would fill single buckets and serialise the histogram's atomics on single counters: found out here, cheaply
        {
            uint32_t ns = 0, dup = 0;
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; i += (E + 3) / 4) {
                const bool in = (slot0 + i * 64) < m;
                if (in) {
                    atomicAdd(&tab[t_of(ks[i], tb_p, tb_n) >> s1], 1u);
                    dup += atomicExch(&cntF[(ks[i] * 2654435761u) >> 21], ks[i]) == ks[i] ? 1u : 0u;
                }
                ns += (uint32_t)__popcll(__ballot(in));
            }
            dup = wave_reduce_sum(dup);
            if (lane == 0) { atomicAdd(&misc[14], ns); atomicAdd(&misc[15], dup); }
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; i += (E + 3) / 4) if ((slot0 + i * 64) < m) cntF[(ks[i] * 2654435761u) >> 21] = 0u;
        if (misc[15] > misc[14] / 32u) return false;                   // (block-uniform)
        {
            static_assert(BR_COARSE == T, "one thread per coarse bucket");
            const uint32_t nsamp = misc[14], mine = tab[t];                               // (at least one sample: m >= BR_MIN_KEYS)
            // a crowd at the floor (a heavy tail, or zeros under larger scores) is one bucket value: the digit passes' business
            const bool at_floor = eps != 0u && ((w_lo < 0x80000000u && (uint32_t)t == (zp - tb_p) >> s1) || (w_hi >= 0x80000000u && (uint32_t)t == (zn - tb_n) >> s1));
            if (__syncthreads_or((at_floor && mine > nsamp / 64u) ? 1 : 0)) return false;
            const float kf = (float)(BR_FINE - BR_COARSE - 16) / (float)nsamp;
            const uint32_t nsub = 1u + (uint32_t)((float)mine * kf);
            uint32_t total;
            const uint32_t base = wg_excl_scan(nsub, total);
            tab[t] = (base << 16) | nsub;
        }
        __syncthreads();
        auto fine_rem = [&](uint32_t kw, uint32_t& fine, uint32_t& rem) {
            const uint32_t tt = t_of(kw, tb_p, tb_n);
            const uint32_t tb = tab[tt >> s1];
            const uint32_t f24 = ((tt << shl) & fmask) >> 8;                              // the position inside the coarse bucket, 24 bits
            const uint64_t prod = (uint64_t)f24 * (uint64_t)(tb & 0xffffu);                // < 2^38 (v_mul_u32_u24 + v_mul_hi_u32_u24)
            fine = (tb >> 16) + (uint32_t)(prod >> 24);
            rem = (uint32_t)(prod >> 8) & 0xffffu;
        };
        // 2. fine histogram, scan, placement
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            if ((slot0 + i * 64) < m) {
                uint32_t fine, rem;
                fine_rem(ks[i], fine, rem);
                atomicAdd(&cntF[fine >> 1], 1u << ((fine & 1u) * 16u));
            }
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        {
            static_assert(BR_FINE == 16 * T, "sixteen fine buckets (eight packed words) per thread");
            uint4* my4 = reinterpret_cast<uint4*>(cntF + 8 * t);
            const uint4 q0 = my4[0], q1 = my4[1];
            uint32_t wv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            uint32_t tot = 0, big = 0, sq = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t lo = wv[k] & 0xffffu, hi = wv[k] >> 16;
                tot += lo + hi; big = lo > big ? lo : big; big = hi > big ? hi : big;
                sq += lo * lo + hi * hi;
            }
            sq = wave_reduce_sum(sq);
            if (lane == 0) atomicAdd(&misc[12], sq);                   // the ranking's work: sum of squared bucket sizes
            uint32_t total;
            uint32_t run = wg_excl_scan(tot, total);
            uint32_t st0 = run;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t lo = wv[k] & 0xffffu, hi = wv[k] >> 16;
                if (lo) atomicOr(&bm[st0 >> 5], 1u << (st0 & 31u));
                st0 += lo;
                if (hi) atomicOr(&bm[st0 >> 5], 1u << (st0 & 31u));
                st0 += hi;
            }
            if (t == T - 1) atomicOr(&bm[(uint32_t)m >> 5], 1u << ((uint32_t)m & 31u));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t lo = wv[k] & 0xffffu, hi = wv[k] >> 16;
                wv[k] = run | ((run + lo) << 16);
                run += lo + hi;
            }
            my4[0] = make_uint4(wv[0], wv[1], wv[2], wv[3]); my4[1] = make_uint4(wv[4], wv[5], wv[6], wv[7]);
            if (__syncthreads_or(big > (uint32_t)BR_MAX_BUCKET ? 1 : 0)) return false;
            if (misc[12] > 8u * (uint32_t)m) return false;             // (ties: a few hundred distinct values)
        }
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int p = slot0 + i * 64;
            if (p < m) {
                uint32_t fine, rem;
                fine_rem(ks[i], fine, rem);
                const uint32_t sh = (fine & 1u) * 16u;
                const uint32_t old = atomicAdd(&cntF[fine >> 1], 1u << sh);     // the offset word: start before, end after the last member
                const uint32_t dst = (old >> sh) & 0xffffu;
                exch[dst] = (rem << 16) | (uint32_t)p;
                meta[i] = __builtin_amdgcn_perm(dst, meta[i], 0x05040100u);      // where this key's entry went
            }
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // 3. rank, slot by slot: the thread of slot d finds d's bucket in the bitmap of bucket starts and counts the members below its
        //    entry -- the lanes of a wave look at 64 consecutive slots, i.e. at the same few buckets (broadcast reads, equal trip counts).
        //    Straight-line for buckets of up to six slots that begin and end within a bitmap word of d's (the rest: rolled loops).
        //    The thread keeps 16 bits per slot: the rank of the entry that sits there.
        static_assert(E % 2 == 0, "two ranks per register");
        uint32_t rk[E / 2];
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int d = slot0 + i * 64;
            const int dd = d < m ? d : 0;
            int wi = dd >> 5;
            const uint32_t mine = exch[dd];
            const uint32_t Wp = wi ? bm[wi - 1] : 0u, W0 = bm[wi], Wn = bm[wi + 1];
            const uint32_t low = (2u << (dd & 31)) - 1u;               // bits 0 .. d & 31
            uint32_t z = W0 & low, z2 = W0 & ~low;
            int ws = wi, we = wi;
            if (z == 0u) { z = Wp; --ws; }
            if (z2 == 0u) { z2 = Wn; ++we; }
            if (z == 0u || z2 == 0u) {                                 // a bucket of more than 32 slots around d (rare)
                while (z == 0u) { --ws; z = bm[ws]; }                  // (bit 0 of word 0 is set: terminates)
                while (z2 == 0u) { ++we; z2 = bm[we]; }                // (bit m is set: terminates)
            }
            const uint32_t st = (uint32_t)(ws * 32 + 31 - __builtin_clz(z));
            const uint32_t en = (uint32_t)(we * 32 + __builtin_ctz(z2));
            uint32_t v[BR_STRAIGHT];
#pragma unroll
            for (int e = 0; e < BR_STRAIGHT; ++e) v[e] = exch[st + e];           // (beyond the bucket: other entries or the words behind exch[]; not counted)
            const uint32_t n = en - st;
            uint32_t r = st;
#pragma unroll
            for (int e = 0; e < BR_STRAIGHT; ++e) r += ((uint32_t)e < n && v[e] < mine) ? 1u : 0u;
            if (n > (uint32_t)BR_STRAIGHT) {
                for (uint32_t j = st + (uint32_t)BR_STRAIGHT; j < en; ++j) r += exch[j] < mine ? 1u : 0u;
            }
            if (i & 1) rk[i >> 1] |= r << 16; else rk[i >> 1] = r;
            if (i & 1) { asm volatile("" : "+v"(rk[i >> 1])); __builtin_amdgcn_sched_barrier(0); }
        }
        __syncthreads();
        // 4. two hops through LDS, words then payloads: owner -> its entry's slot -> that slot's rank.  Then every slot checks its left
        //    neighbour: two DISTINCT words that shared (fine, rem) may stand in position order instead of word order.
        bool bad = false;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) if ((slot0 + i * 64) < m) exch[meta[i] >> 16] = round ? (meta[i] & 0xffffu) : ks[i];
            __syncthreads();
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) if ((slot0 + i * 64) < m) { if (round) meta[i] = exch[(slot0 + i * 64)]; else ks[i] = exch[(slot0 + i * 64)]; }
            __syncthreads();
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) if ((slot0 + i * 64) < m) exch[(rk[i >> 1] >> ((i & 1) * 16)) & 0xffffu] = round ? meta[i] : ks[i];
            __syncthreads();
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int p = slot0 + i * 64;
                if (p < m) {
                    if (round) meta[i] = exch[p];
                    else { ks[i] = exch[p]; bad |= p > 0 && exch[p > 0 ? p - 1 : 0] > ks[i]; }
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(bad ? 1 : 0)) return true;
        // 5. (about one row in a hundred) an inverted neighbour pair is swapped back; anything a single round of disjoint swaps does not
        //    settle is left to the digit passes
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) if ((slot0 + i * 64) < m) exch[(slot0 + i * 64)] = ks[i];
        __syncthreads();
        bad = false;
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int p = slot0 + i * 64;
            if (p < m) {
                const bool il = p > 0 && exch[p > 0 ? p - 1 : 0] > ks[i], ir = p + 1 < m && ks[i] > exch[p + 1 < m ? p + 1 : p];
                bad |= il && ir;                                         // three in a row: not a swap
                meta[i] = __builtin_amdgcn_perm((uint32_t)(il ? p - 1 : (ir ? p + 1 : p)), meta[i], 0x05040100u);
            }
        }
        if (__syncthreads_or(bad ? 1 : 0)) return false;               // (ks / meta untouched: still a stable arrangement by a coarsening of the word)
        if (t == 0) atomicAdd(&g_bucket_rank_rows[1], 1ull);
        permute_to_meta_hi();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) if ((slot0 + i * 64) < m) exch[(slot0 + i * 64)] = ks[i];
        __syncthreads();
        bad = false;
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int p = slot0 + i * 64;
            if (p < m) bad |= p > 0 && exch[p > 0 ? p - 1 : 0] > ks[i];
        }
        return !__syncthreads_or(bad ? 1 : 0);
      }
    };

    static_assert(KW == 1 || E <= 32, "fp64: per-thread slot masks are 32 bits");
    uint32_t signm = 0u, nanm = 0u;   // fp64: bit i = slot i's key has its (ascending-key) sign bit set / is the NaN key
    uint32_t contm = 0u;              // fp64: bit i = slot i continues the equal-high-word run of the slot before it
    bool ranked = false;              // block-uniform
    if constexpr (BR) {
        const int npass = (int)((diff & 0xffu) != 0u) + (int)((diff & 0xff00u) != 0u) + (int)((diff & 0xff0000u) != 0u) + (int)((diff >> 24) != 0u);
        if (a.bucket_rank && m >= BR_MIN_KEYS && npass >= 3) {          // (two digit passes -- scores within one binade -- are cheaper than the ranking)
            ranked = bucket_rank();
            if (threadIdx.x == 0) atomicAdd(&g_bucket_rank_rows[ranked ? 0 : 2], 1ull);
        }
    }
    if (!ranked) {
        for (int pass = 0; pass < 4; ++pass) {
            if (((diff >> (pass * 8)) & 0xffu) == 0u) continue;  // constant digit among real keys: order unchanged
            radix_pass(pass * 8, (diff_match >> (pass * 8)) & 0xffu);
        }
    }
    if constexpr (GEN) {
        // generic form: the four passes above ran over the LOW key words; now the high words (re-derived from global
        // memory by payload), four more passes, the high halves of the sorted keys, and the low words again for the output
        reload_word(true);
        uint32_t o1 = 0u, a1 = 0xffffffffu;
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) if ((meta[i] & 0xffffu) != 0xffffu) { o1 |= ks[i]; a1 &= ks[i]; }
        __syncthreads();
        if (threadIdx.x == 0) { misc[8] = 0u; misc[9] = 0xffffffffu; }
        __syncthreads();
        atomicOr(&misc[8], o1); atomicAnd(&misc[9], a1);
        __syncthreads();
        const uint32_t dh = misc[8] ^ misc[9], dh_match = has_sent ? ~misc[9] : dh;
        for (int pass = 0; pass < 4; ++pass)
            if (((dh >> (pass * 8)) & 0xffu) != 0u) radix_pass(pass * 8, (dh_match >> (pass * 8)) & 0xffu);
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int p = (slot0 + i * 64);
            const uint32_t asc = ~ks[i];
            const bool nan = ks[i] == 0u, sgn = (asc >> 31) & 1u;
            signm |= (uint32_t)sgn << i; nanm |= (uint32_t)nan << i;
            if (o_kw && p < lim) o_kw[2 * p + 1] = nan ? 0x7ff80000u : (sgn ? (asc & 0x7fffffffu) : ~asc);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        reload_word(false);
    } else if constexpr (KW == 2) {
        // fp64 keys: the FOUR passes above ordered the rows by the HIGH key word only (sign, exponent, 20 mantissa bits).
        // Two keys that share a high word but differ in the low word may now be out of order -- but they are ADJACENT
        // (a stable sort by the high word leaves each equal-high-word run contiguous and in sequence order), and such
        // a run is almost always 2-3 keys long, except runs of EQUAL keys (BM25's zeros), which need nothing.  So:
        // note which slots continue their left neighbour's run, write the high halves of the sorted keys, swap the
        // high words for the low words (re-derived from global memory by the thread that loaded them, handed over
        // through LDS by payload), and repair the dirty runs in place.  Same permutation as eight LSD passes over the
        // 64-bit key at half the ranking work and a third of the exchanges.  A row the repair cannot take (see below) is
        // flagged instead (row_flags) and redone by the generic eight-pass form of this kernel (GEN), a second launch in
        // which every other row's workgroup exits at once.
        uint32_t* runbits = cnt;   // [T*E/32]: bit (p & 31) of word p >> 5 = "slot p has the same high word as slot p-1"
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) exch[(slot0 + i * 64)] = ks[i]; }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            const int p = (slot0 + i * 64);
            const bool ps = p > 0 && p < m && exch[p - 1] == ks[i];
            contm |= (uint32_t)ps << i;
            const unsigned long long bal = __ballot(ps);
            if (lane == 0) { runbits[(w * Ee + i) * 2] = (uint32_t)bal; runbits[(w * Ee + i) * 2 + 1] = (uint32_t)(bal >> 32); }
            // the sorted key's high half is final now (the repair only moves keys inside runs of EQUAL high words)
            const uint32_t asc = ~ks[i];
            const bool nan = ks[i] == 0u;           // every NaN maps to the all-zero key; no other key has a zero high word
            const bool sgn = (asc >> 31) & 1u;
            signm |= (uint32_t)sgn << i; nanm |= (uint32_t)nan << i;
            // the sorted score's HIGH half is final here, its low half only after the repair: parked COMPACTLY at the head of the row's
            // score output (word p; whole lines) and stored together with the low half as one 8-byte store in the output phase -- two
            // strided 4-byte passes over the row wrote every line of it twice (2 x 229 MB per 1024 x 27,942 batch, rocprofv3 WRITE_SIZE)
            bool parked = false;
            if constexpr (ZC) {
                if (compacted) {   // (block-uniform: a scalar branch, nothing for a row that kept its zeros)
                    parked = true;
                    const int f = p >= zc_P ? p + zc_Z : p;
                    if (o_kw && p < m && f < lim) o_kw[f] = nan ? 0x7ff80000u : (sgn ? (asc & 0x7fffffffu) : ~asc);
                }
            }
            if (!parked) { if (o_kw && p < lim) o_kw[p] = nan ? 0x7ff80000u : (sgn ? (asc & 0x7fffffffu) : ~asc); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- low words: re-derived by the thread that loaded the element, published under its payload ----
        constexpr int LGG = 4;   // (7 in flight measured the same)
        bool zc_low = false;     // block-uniform
        if constexpr (ZC) {
            if (compacted) {     // the low words the compaction parked, by compact position: each thread its own slots (coalesced), published under them
                zc_low = true;
                SLOT_FRESH();
#pragma unroll
                for (int i0 = 0; i0 < E; i0 += 4) {
                    if (i0 >= Ee) break;
                    uint32_t lw[4];
#pragma unroll
                    for (int i = i0; i < (i0 + 4 < E ? i0 + 4 : E); ++i) { const int p = slot0 + i * 64; lw[i - i0] = stash[p < m ? p : 0]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = i0; i < (i0 + 4 < E ? i0 + 4 : E); ++i) { const int p = slot0 + i * 64; if (p < m) exch[p] = lw[i - i0]; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (zc_low) {
        } else if (use_stash) {         // FUSE: the low sort words parked by the load phase, by the slot this thread loaded
            SLOT_FRESH();
#pragma unroll
            for (int i0 = 0; i0 < E; i0 += 7) {
                uint32_t lw[7];
                int pay_t[7];
#pragma unroll
                for (int i = i0; i < (i0 + 7 < E ? i0 + 7 : E); ++i) {
                    const int p = (slot0 + i * 64);
                    int pay;   // payload this slot was loaded with (see the re-deriving form below)
                    if (irow) pay = p < a.n_total ? p : -1;
                    else if (init_row) { int col = p < m ? init_row[c0 + p] : -1; pay = (unsigned)col < (unsigned)a.n_total ? col : -1; }
                    else pay = p < m ? p : -1;
                    pay_t[i - i0] = pay;
                    lw[i - i0] = stash[pay < 0 ? 0 : p];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = i0; i < (i0 + 7 < E ? i0 + 7 : E); ++i)
                    if (pay_t[i - i0] >= 0) exch[pay_t[i - i0]] = lw[i - i0];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        SLOT_FRESH();
#pragma unroll
        for (int i0 = 0; i0 < E; i0 += LGG) {
            uint32_t lo_t[LGG], hi_t[LGG];
            int pay_t[LGG];
            [[maybe_unused]] int col_t[FUSE ? LGG : 1];
#pragma unroll
            for (int i = i0; i < (i0 + LGG < E ? i0 + LGG : E); ++i) {
                const int p = (slot0 + i * 64);
                int pay, col;   // payload this slot was loaded with, and its column in the row
                if (irow) { pay = p < a.n_total ? p : -1; col = pay; }
                else if (init_row) { col = p < m ? init_row[c0 + p] : -1; col = (unsigned)col < (unsigned)a.n_total ? col : -1; pay = col; }
                else { pay = p < m_all ? p : -1; col = pay < 0 ? -1 : c0 + pay; }
                pay_t[i - i0] = pay;
                if constexpr (FUSE) col_t[i - i0] = col;
                else { const uint2 v = kd[elem(col < 0 ? c0 : col)]; lo_t[i - i0] = v.x; hi_t[i - i0] = v.y; }
            }
            if constexpr (FUSE) {
#pragma unroll
                for (int k = (i0 + LGG < E ? LGG : E - i0); k < LGG; ++k) { col_t[k] = -1; pay_t[k] = -1; }
                uint2 v_t[LGG];
                if (init_row) __builtin_amdgcn_sched_barrier(0);
                fuse_vals(std::integral_constant<int, LGG>{}, col_t, col_t, false, v_t);
#pragma unroll
                for (int k = 0; k < LGG; ++k) { lo_t[k] = v_t[k].x; hi_t[k] = v_t[k].y; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = i0; i < (i0 + LGG < E ? i0 + LGG : E); ++i)
                if (pay_t[i - i0] >= 0) exch[pay_t[i - i0]] = (uint32_t)key64(make_uint2(lo_t[i - i0], hi_t[i - i0]));
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            const uint32_t py = meta[i] & 0xffffu;
            ks[i] = py != 0xffffu ? exch[py] : SENT;     // from here on ks = the LOW key word
            if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- repair.  The runs are described completely by LDS state -- runbits ("same high word as the slot before")
        // and the low words by slot in exch[] -- so the work is done by whichever thread is convenient, not by the slots'
        // owners:
        //   1. a misordered neighbour pair inside a run marks the run's head (nearest clear bit to the left) in dirtybits;
        //   2. the thread whose word holds a DIRTY head re-sorts the run: up to WALK + 1 slots by itself (stable counting
        //      sort; the usual case is a pair), leaving every member's move as a signed byte in delta[]; a longer run
        //      (up to 2 T slots) goes on a short list;
        //   3. listed runs are counted by the whole workgroup (slot x of the run: how many members precede it), the new
        //      slot numbers replace the low words in exch[] and delta[] says 0x7e = "look there";
        //   4. every owner moves (low word, payload) to slot p + delta[p] (0x7f: stays).
        // Runs of EQUAL keys (BM25's zeros, ties) are never dirty and cost nothing.  A dirty run longer than 2 T slots,
        // or more than BIGCAP listed runs, flags the row for the generic launch.
        constexpr int NWORDS = T * E / 32, BIGCAP = 64, BIGMAX = 2 * T;
        static_assert(2 * NWORDS + BIGCAP <= NW * 256 && BIGMAX < 4096, "repair state lives in the counter area");
        uint32_t* dirtybits = cnt + NWORDS;
        uint32_t* biglist = cnt + 2 * NWORDS;
        int8_t* delta = reinterpret_cast<int8_t*>(cnt + NW * 256);   // [T*E] bytes
        const int t = threadIdx.x;
        const int nwords = T * Ee / 32;                        // runbits words the run detection wrote (ZC: the compact layout's)
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) { ITEM_GUARD(i) exch[(slot0 + i * 64)] = ks[i]; }
        {
            uint32_t* d32 = reinterpret_cast<uint32_t*>(delta);
#pragma unroll
            for (int i = 0; i < (E + 3) / 4; ++i) { const int x = i * T + t; if (x < T * E / 4) d32[x] = 0x7f7f7f7fu; }
            if (t < NWORDS) dirtybits[t] = 0u;
            if (t == 0) { misc[12] = 0u; misc[13] = 0u; }
        }
        __syncthreads();
        // 1. every owner compares its slot's low word with the left neighbour's (consecutive lanes, consecutive LDS words: the walk
        //    over a runbits word by ONE thread read slots 32 t + k -- one bank for the whole wave -- and cost 25 us per row on rows
        //    with long runs of equal keys, BM25's zeros)
        bool anybad = false;
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            const int p = (slot0 + i * 64);
            const bool cont = (contm >> i) & 1u;
            if (__ballot(cont) == 0ull) continue;              // wave-uniform: nobody in this item continues a run
            const uint32_t prev = exch[p > 0 ? p - 1 : 0];
            if (cont && prev > ks[i]) {                        // out of order in the low word (rare): mark the run's head
                anybad = true;
                const int q = p - 1;                           // the head is the nearest clear runbit at or below slot q
                int wd = q >> 5;
                uint32_t z = ~runbits[wd] & ((2u << (q & 31)) - 1u);
                while (z == 0u) { --wd; z = ~runbits[wd]; }    // terminates: slot 0 never continues a run
                const int h = wd * 32 + 31 - __builtin_clz(z);
                atomicOr(&dirtybits[h >> 5], 1u << (h & 31));
            }
        }
        if (__syncthreads_or(anybad ? 1 : 0)) {
            if (t < NWORDS) {
                uint32_t heads = dirtybits[t];
                while (heads) {
                    const int h = 32 * t + __builtin_ctz(heads);
                    heads &= heads - 1;
                    // end of the run: first slot after h whose bit is clear (slots >= m never continue a run)
                    const int e0 = h + 1;
                    int wd = e0 >> 5;
                    uint32_t z = ~runbits[wd] & ~((1u << (e0 & 31)) - 1u);
                    while (z == 0u && wd + 1 < nwords) { ++wd; z = ~runbits[wd]; }
                    const int L = (z ? wd * 32 + __builtin_ctz(z) : nwords * 32) - h;
                    if (L == 2) { delta[h] = 1; delta[h + 1] = -1; }             // a dirty pair: swap
                    else if (L <= WALK + 1) {                                     // stable counting sort by one thread
                        for (int x = 0; x < L; ++x) {
                            const uint32_t vx = exch[h + x];
                            int before = 0;
                            for (int y = 0; y < L; ++y) { const uint32_t vy = exch[h + y]; before += (vy < vx || (vy == vx && y < x)) ? 1 : 0; }
                            delta[h + x] = (int8_t)(before - x);
                        }
                    } else if (L <= BIGMAX) {
                        const uint32_t idx = atomicAdd(&misc[12], 1u);
                        if (idx < (uint32_t)BIGCAP) biglist[idx] = ((uint32_t)h << 12) | (uint32_t)L;
                        else misc[13] = 1u;
                    } else misc[13] = 1u;
                }
            }
            __syncthreads();
            if (misc[13] != 0u) {                              // block-uniform: leave the row to the generic form
                if (t == 0) a.row_flags[prow] = 1;
                return;
            }
            const int nbig = (int)misc[12];
            for (int bi = 0; bi < nbig; ++bi) {
                const uint32_t ent = biglist[bi];
                const int h = (int)(ent >> 12), L = (int)(ent & 0xfffu);
                int npos[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int x = t + r * T;
                    npos[r] = -1;
                    if (x < L) {
                        const uint32_t vx = exch[h + x];
                        int before = 0;
                        for (int y = 0; y < L; ++y) { const uint32_t vy = exch[h + y]; before += (vy < vx || (vy == vx && y < x)) ? 1 : 0; }
                        npos[r] = h + before;
                    }
                }
                __syncthreads();                               // every low word of the run has been read
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (npos[r] >= 0) { exch[h + t + r * T] = (uint32_t)npos[r]; delta[h + t + r * T] = 0x7e; }
            }
            __syncthreads();
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                ITEM_GUARD(i)
                const int p = (slot0 + i * 64);
                const int d = p < m ? (int)delta[p] : 0x7f;
                const uint32_t np = d == 0x7f ? (uint32_t)p : (d == 0x7e ? exch[p] : (uint32_t)(p + d));
                meta[i] = __builtin_amdgcn_perm(np, meta[i], 0x05040100u);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                   // the slot numbers in exch[] have been read
            permute_to_meta_hi();
        }
    }

    if constexpr (ZC) {
        if (compacted) {
            // the ordering phases carried compact positions: the columns behind them, once.  The threads that loaded the row name their
            // non-zero items' columns by compact position (the compaction's own count, again), the sorted slots pick theirs up.
            uint32_t run = (uint32_t)zc_nzbase;
            const int s_full = w * E * 64 + lane;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int col = s_full + i * 64;
                const bool mv = col < m_all && !((zm >> i) & 1u);
                const unsigned long long bal = __ballot(mv);
                const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                run += (uint32_t)__popcll(bal);
                if (mv) exch[pos] = (uint32_t)col;
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                ITEM_GUARD(i)
                const uint32_t py = meta[i] & 0xffffu;
                meta[i] = py != 0xffffu ? exch[py] : 0xffffu;
            }
            __syncthreads();
        }
    }

    // ---- statistics by-product (block-uniform).  min / max of a list sorted by score are its two ends (a NaN sorts first and makes both
    // NaN, as torch.min / torch.max do): the columns of entries 0 and slen - 1 go through LDS, thread 0 reads their values back (L2 hits).
    // A ranking cut to its first slen entries also takes its mean / unbiased std here: the sorted words go to LDS by rank and are summed by
    // a rolled, strided loop (two live doubles) -- the same sums unrolled over the 28 register-resident keys spilled the sort's hot loops.
    if (st_on) {
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            ITEM_GUARD(i)
            const int p = (slot0 + i * 64);
            const uint32_t col = (meta[i] & 0xffffu) + ((!init_row && !irow) ? (uint32_t)c0 : 0u);
            if constexpr (ZC) {
                if (p < m && fpos(p) == 0) misc[17] = col;
                if (p < m && fpos(p) == slen - 1) misc[18] = col;
            } else {
                if (p == 0) misc[17] = col;
                if (p == slen - 1) misc[18] = col;
            }
        }
        if constexpr (ZC) {   // a list end that is one of the compacted zeros: the thread that loaded that zero names its column (+0.0 or -0.0: the
            if (compacted && (zc_P == 0 || (slen - 1 >= zc_P && slen - 1 < zc_P + zc_Z))) {   // value the whole-row form would read there)
                int zrun = zc_P + zc_zbase;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    const bool zb = (zm >> i) & 1u;
                    const unsigned long long bal = __ballot(zb);
                    const int f = zrun + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    zrun += (int)__popcll(bal);
                    const uint32_t col = (uint32_t)(w * E * 64 + i * 64 + lane);
                    if (zb && f == 0) misc[17] = col;
                    if (zb && f == slen - 1) misc[18] = col;
                }
            }
        }
        double x0 = 0.0;
        const bool pre = KW == 1 && st_prefix;
        if (pre) {
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) exch[(slot0 + i * 64)] = ks[i];
            __syncthreads();
            const float f0 = desc_key_f32_inv(exch[0]);
            x0 = (f0 == f0 && fabsf(f0) != INFINITY) ? (double)f0 : 0.0;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll 1
            for (int p = threadIdx.x; p < slen; p += T) {
                const double dd = (double)desc_key_f32_inv(exch[p]) - x0;
                s1 += dd; s2 = fma(dd, dd, s2);
            }
            s1 = wave_reduce_sum(s1); s2 = wave_reduce_sum(s2);
            double* red = reinterpret_cast<double*>(cnt);
            if (lane == 0) { red[w] = s1; red[NW + w] = s2; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            auto val = [&](int col) -> float {
                if (FUSE) return 0.f;                        // (the launcher rejects row_stats with rank fusion)
                if (KW == 1) return kf[elem(col)];
                const uint2 v = kd[elem(col)];
                return (float)__hiloint2double((int)v.y, (int)v.x);
            };
            float lo = 0.f, hi = 0.f;
            if (slen > 0) { hi = val((int)misc[17]); lo = val((int)misc[18]); if (hi != hi) lo = hi; }
            a.row_stats[2 * a.stats_rows + row] = lo;
            a.row_stats[3 * a.stats_rows + row] = hi;
            if (pre) {
                const double* red = reinterpret_cast<const double*>(cnt);
                double S1 = 0.0, S2 = 0.0;
                for (int i = 0; i < NW; ++i) { S1 += red[i]; S2 += red[NW + i]; }
                const double n = (double)slen;
                const double mean = n > 0.0 ? x0 + S1 / n : (double)NAN;
                const double var = n > 1.0 ? (S2 - S1 * S1 / n) / (n - 1.0) : (double)NAN;
                a.row_stats[row] = (float)mean;
                a.row_stats[a.stats_rows + row] = (float)sqrt(var < 0.0 ? 0.0 : var);
            }
        }
        __syncthreads();   // exch[] and the counter area are free again
    }

    // ---- output (coalesced: consecutive lanes = consecutive ranks) -------------------------
    // rank = inverse permutation.  Scattering it straight to HBM costs as much as the four radix passes
    // (27,942 random 4-byte writes per row); it is inverted in LDS instead and stored coalesced.
    const bool rank_via_lds = o_rank && !cmap && (LEAN || a.chunks == 1);   // block-uniform
    const bool full_row = (m_all == a.n_total) && !init_row && !irow;   // a gathered/placed sequence may skip columns
    if constexpr (ZC) slot0 = w * E * 64 + lane;       // (the pre-fill covers the row's COLUMNS: the full layout)
    if (rank_via_lds && !full_row) {
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) exch[(slot0 + i * 64)] = 0xffffffffu;   // columns outside the sequence
        __syncthreads();
    }
    if constexpr (ZC) slot0 = w * Ee * 64 + lane;      // (the sorted entries: the layout of the ordering phases)
    [[maybe_unused]] uint32_t hiw[(KW == 2 && !GEN) ? E : 1];
    if constexpr (KW == 2 && !GEN) {
        if (o_kw) {   // (block-uniform) the parked high halves back -- ALL of them, then a barrier: the 8-byte stores below land on the words they were parked in
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                ITEM_GUARD(i)
                const int p = (slot0 + i * 64);
                int f = p < lim ? p : 0;
                if constexpr (ZC) { if (compacted) { const int g = p >= zc_P ? p + zc_Z : p; f = (p < m && g < lim) ? g : 0; } }   // (block-uniform branch)
                hiw[i] = o_kw[f];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
        SLOT_FRESH();
#pragma unroll
    for (int i = 0; i < E; ++i) {
        ITEM_GUARD(i)
        int p = (slot0 + i * 64);                              // the sorted slot = the list position ...
        if constexpr (ZC) { if (compacted) p = p >= m ? lim : (p >= zc_P ? p + zc_Z : p); }   // ... (compacted row, block-uniform branch: the zeros stand in between)
        if (p < lim) {
            int col = (int)(meta[i] & 0xffffu);
            if (!init_row && !irow) col += c0;
            const int oc = cmap ? cmap[col] : col;  // -1 = padding candidate of a short top-k chunk
            if (o_order) o_order[p] = oc;
            if (o_ids) o_ids[p] = imap ? imap[elem(col)] : (oc < 0 ? (int64_t)-1 : a.id_base + (int64_t)oc);
            if (KW == 1) { if (o_kf) o_kf[p] = desc_key_f32_inv(ks[i]); }
            else if (o_kw) {
                const uint32_t low = ((nanm >> i) & 1u) ? 0u : (((signm >> i) & 1u) ? ~ks[i] : ks[i]);   // low half of desc_key_f64_inv
                if constexpr (!GEN) reinterpret_cast<uint2*>(o_kw)[p] = make_uint2(low, hiw[i]);
                else o_kw[2 * p] = low;                            // (generic form: its high halves went out in their own pass)
            }
            if (rank_via_lds) exch[col] = (uint32_t)p;          // col < n_total <= T*E
            else if (o_rank && oc >= 0) o_rank[oc] = p;
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ZC) {
        slot0 = w * E * 64 + lane;                             // back to the columns
        if (compacted) {
            // the zeros: list position = (keys above zero) + (zeros before this one in sequence order); value +0.0 (desc_key maps -0.0 there too)
            int zrun = zc_P + zc_zbase;
            SLOT_FRESH();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const bool zb = (zm >> i) & 1u;
                const unsigned long long bal = __ballot(zb);
                const int p = zrun + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                zrun += (int)__popcll(bal);
                if (zb && p < lim) {
                    const int col = slot0 + i * 64;            // (identity sequence, whole rows: the column is the slot)
                    if (o_order) o_order[p] = col;
                    if (o_kw) reinterpret_cast<uint2*>(o_kw)[p] = make_uint2(0u, 0u);
                    if (rank_via_lds) exch[col] = (uint32_t)p;
                    else if (o_rank) o_rank[col] = p;
                }
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (rank_via_lds) {
        __syncthreads();
        SLOT_FRESH();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int j = (slot0 + i * 64);
            if (j < a.n_total) {
                const uint32_t r = exch[j];
                if (full_row || r != 0xffffffffu) o_rank[j] = (int32_t)r;
            }
        }
    }
}

#undef SLOT_FRESH
#undef ITEM_GUARD

// pad the tail of top-k outputs with (-inf, -1)
__global__ void topk_pad_kernel(float* out_scores, int64_t* out_ids, int rows, int k, int have) {
    const int r = blockIdx.y;
    for (int i = have + blockIdx.x * blockDim.x + threadIdx.x; i < k; i += gridDim.x * blockDim.x) {
        out_scores[(size_t)r * k + i] = -INFINITY;
        out_ids[(size_t)r * k + i] = -1;
    }
}

// ---- streaming top-k update -------------------------------------------------------------------------------
// After the first chunk of a corpus shard, the running k-th best score tau[row] bounds what can still enter the
// top-k: chunks arrive in ascending id order, so an element tying with tau has a larger id than the current k-th
// entry and loses; only s > tau (or NaN, which sorts first in this build) survives.  For i.i.d. scores the expected
// number of survivors per row is k * chunk / seen: a few hundred.  One workgroup per row streams the chunk (16-B
// loads), compacts the survivors STABLY (ascending column = ascending id) behind the running list, and the ordinary
// row sort then merges [running k | survivors] (ties: running entries first, then ascending id).
// A row with more survivors than `cap` sets *overflow (checked by the caller, who redoes that chunk exactly).
struct FilterArgs {
    const float* scores; int n; long ld;     // chunk [rows][ld]
    int64_t id_base;
    const float* run_scores;                 // [rows][k] current top-k (sorted desc; -inf padding)
    const int64_t* run_ids;                  // [rows][k]
    int k, cap;
    float* buf_scores; int64_t* buf_ids;     // [rows][k + cap]: running list copied to the front, survivors behind
    int32_t* buf_len;                        // [rows] = k + survivors
    int32_t* overflow;                       // set to 1 if any row exceeded cap
    // append form (fz_topk_filter_append_f32): tau != null -> the threshold comes from tau[row], nothing is copied, survivors go
    // behind the buf_len[row] candidates already in buf_* ([rows][cap], k = 0) and buf_len[row] grows by their number
    const float* tau;
};

// Each of the 4 waves owns a CONTIGUOUS stretch of the row (of a round of 65,536 columns), so "ascending column" = wave order, then step order, then lane order:
// pass 1 streams the stretch from HBM and only counts (a 64-bit mask remembers which 256-column steps had a survivor at all);
// one barrier and a 4-entry scan give every wave its output offset; pass 2 revisits the marked steps (L2 / Infinity-Cache hits)
// and writes.  No barrier and no scan inside the streaming loop.
__global__ __launch_bounds__(256) void topk_filter_kernel(FilterArgs a) {
    constexpr int T = 256, NW = T / 64, STEP = 256, SPAN = NW * 64 * STEP;   // columns per outer round: 64 steps per wave
    __shared__ int wtot[NW];
    const int row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* __restrict__ x = a.scores + (size_t)row * a.ld;
    float* __restrict__ bs = a.buf_scores + (size_t)row * (a.k + a.cap);
    int64_t* __restrict__ bi = a.buf_ids + (size_t)row * (a.k + a.cap);
    float tau;
    int base = 0;
    if (a.tau) { tau = a.tau[row]; base = a.buf_len[row]; }
    else {
        for (int i = threadIdx.x; i < a.k; i += T) { bs[i] = a.run_scores[(size_t)row * a.k + i]; bi[i] = a.run_ids[(size_t)row * a.k + i]; }
        tau = a.run_scores[(size_t)row * a.k + a.k - 1];   // k-th best so far (-inf while the list is short)
    }
    const bool vec = (a.ld % 4 == 0) && ((uintptr_t)a.scores % 16 == 0);
    bool over = false;
    auto load4 = [&](int j0, float (&v)[4]) {
        if (vec && j0 + 3 < a.n) { const float4 f = *reinterpret_cast<const float4*>(x + j0); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (j0 + c < a.n) ? x[j0 + c] : -INFINITY;
        }
    };
    for (int r0 = 0; r0 < a.n; r0 += SPAN) {
        const int nr = min(a.n - r0, SPAN);
        const int steps = (nr + NW * STEP - 1) / (NW * STEP);      // per wave, <= 64
        const int w0 = r0 + w * steps * STEP;                      // this wave's stretch: steps * 256 columns from w0
        // ---- pass 1: count
        int cnt = 0;
        unsigned long long marked = 0ull;
        constexpr int UNR = 8;                                     // 16-byte loads in flight per lane
        auto count_step = [&](int st, const float (&v)[4]) {
            const int j0 = w0 + st * STEP + lane * 4;
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) c += ((j0 + e < a.n) && !(v[e] <= tau)) ? 1 : 0;   // s > tau, or NaN (sorts first in this build)
            cnt += c;
            if (__ballot(c != 0)) marked |= 1ull << st;            // wave-uniform
        };
        int s0 = 0;
        // whole groups of UNR steps inside the row: plain 16-byte loads under no branch (behind a conditional load hipcc waits
        // for every outstanding one, which turns the group into a chain of single round trips)
        if (vec)
            for (; s0 + UNR <= steps && w0 + (s0 + UNR) * STEP <= a.n; s0 += UNR) {
                float4 f[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) f[u] = *reinterpret_cast<const float4*>(x + w0 + (s0 + u) * STEP + lane * 4);
#pragma unroll
                for (int u = 0; u < UNR; ++u) { const float v[4] = {f[u].x, f[u].y, f[u].z, f[u].w}; count_step(s0 + u, v); }
            }
        for (; s0 < steps; ++s0) {                                 // the ragged end (or unaligned rows)
            float v[4];
            load4(w0 + s0 * STEP + lane * 4, v);
            count_step(s0, v);
        }
        cnt = wave_reduce_sum(cnt);
        __syncthreads();                                           // (the previous round's wtot has been read)
        if (lane == 0) wtot[w] = cnt;
        __syncthreads();
        int off = base, tot = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { const int c = wtot[i]; if (i < w) off += c; tot += c; }
        // ---- pass 2: the marked steps again, survivors to their slots.  Four marked steps at a time, their loads issued together
        // (one step at a time is a chain of L2 round trips: at 0.4 % survivors two thirds of the steps are marked and this pass
        // took as long as the HBM pass)
        auto place = [&](int st, const float (&v)[4]) {
            const int j0 = w0 + st * STEP + lane * 4;
            // survivors of lower lanes + own earlier ones, from four ballots (no shuffle chain: this runs once per marked step)
            bool keep[4];
            int pos = off, tot_st = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                keep[e] = (j0 + e < a.n) && !(v[e] <= tau);
                const unsigned long long bal = __ballot(keep[e]);
                pos += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                tot_st += __popcll(bal);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (keep[e]) {
                    if (pos < a.cap) { bs[a.k + pos] = v[e]; bi[a.k + pos] = a.id_base + j0 + e; }
                    else over = true;
                    ++pos;
                }
            off += tot_st;
        };
        // a step is "inner" when all its 256 columns exist: plain 16-byte loads; the (at most one) ragged step goes last, guarded
        const int inner = vec ? max(0, min(steps, (a.n - w0) / STEP)) : 0;
        unsigned long long in_m = inner >= 64 ? marked : (marked & ((1ull << inner) - 1ull));
        unsigned long long rest = marked & ~in_m;
        while (in_m) {
            int st[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // the group's steps; a short last group repeats its last step (loaded, not placed)
                st[u] = in_m ? __builtin_ctzll(in_m) : st[u > 0 ? u - 1 : 0];
                if (in_m) in_m &= in_m - 1; else st[u] |= 0x40000000;
            }
            float4 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) f[u] = *reinterpret_cast<const float4*>(x + w0 + (st[u] & 0xffff) * STEP + lane * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (!(st[u] & 0x40000000)) { const float v[4] = {f[u].x, f[u].y, f[u].z, f[u].w}; place(st[u], v); }
        }
        while (rest) {
            const int st = __builtin_ctzll(rest);
            rest &= rest - 1;
            float v[4];
            load4(w0 + st * STEP + lane * 4, v);
            place(st, v);
        }
        base += tot;
    }
    __syncthreads();   // (append form: every thread has read buf_len[row])
    if (threadIdx.x == 0) a.buf_len[row] = a.k + (base < a.cap ? base : a.cap);
    if (over) atomicExch(a.overflow, 1);
}

// fold: [running k | candidates] of every row side by side for the row sort; afterwards the new threshold and empty candidate lists
__global__ void topk_concat_kernel(const float* run_scores, const int64_t* run_ids, int k, const float* cand_scores, const int64_t* cand_ids,
                                   const int32_t* cand_len, int cap, float* buf_scores, int64_t* buf_ids, int32_t* buf_len) {
    const int row = blockIdx.x;
    const int len = min(cand_len[row], cap);
    float* bs = buf_scores + (size_t)row * (k + cap);
    int64_t* bi = buf_ids + (size_t)row * (k + cap);
    for (int i = threadIdx.x; i < k; i += blockDim.x) { bs[i] = run_scores[(size_t)row * k + i]; bi[i] = run_ids[(size_t)row * k + i]; }
    for (int i = threadIdx.x; i < len; i += blockDim.x) { bs[k + i] = cand_scores[(size_t)row * cap + i]; bi[k + i] = cand_ids[(size_t)row * cap + i]; }
    if (threadIdx.x == 0) buf_len[row] = k + len;
}
__global__ void topk_fold_done_kernel(const float* new_scores, int rows, int k, float* tau, int32_t* cand_len) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    if (tau) tau[row] = new_scores[(size_t)row * k + k - 1];
    cand_len[row] = 0;
}


// ---- rows longer than one workgroup holds (the drop-in path must not stop at 28,672 documents) -------------
// chunk-sort (the kernel above, one workgroup per chunk of the SEQUENCE) + cross-chunk ranking: an element's final
// position = its position in its own sorted chunk + for every other chunk the number of elements that precede it there
// (binary search; ties go to the earlier chunk, i.e. to the earlier sequence position: the stable order).  Exact, any n;
// O(n * chunks * log) work, meant for corpora of 10^5..10^6 documents, not for the LLeQA hot path.
struct LongArgs {
    const void* keys; int key_bits; const int32_t* init_order; const int32_t* init_rank; const int32_t* row_len;
    int rows, n; long ld;
    void* seq_keys; int32_t* seq_cols;            // [rows][n] the incoming sequence, materialised (gathered / placed input only)
    const void* ck; const int32_t* cc;            // [rows][n] chunk-sorted keys / columns
    int chunk_len;
    int32_t* order; void* sorted_keys; int32_t* rank; long out_ld;
};

template <typename K>
__global__ void long_seq_kernel(LongArgs a) {
    const int row = blockIdx.y;
    const K* keys = reinterpret_cast<const K*>(a.keys) + (size_t)row * a.ld;
    K* sk = reinterpret_cast<K*>(a.seq_keys) + (size_t)row * a.n;
    int32_t* sc = a.seq_cols + (size_t)row * a.n;
    int m = a.row_len ? a.row_len[row] : a.n;
    m = m < 0 ? 0 : (m > a.n ? a.n : m);
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < a.n; x += gridDim.x * blockDim.x) {
        if (a.init_order) {                                   // slot x <- column init_order[x]
            const int col = x < m ? a.init_order[(size_t)row * a.ld + x] : -1;
            const bool ok = (unsigned)col < (unsigned)a.n;
            sk[x] = ok ? keys[col] : (K)(-INFINITY);
            sc[x] = ok ? col : -1;
        } else {                                              // column x -> slot init_rank[x]
            const int r = a.init_rank[(size_t)row * a.ld + x];
            if ((unsigned)r < (unsigned)m) { sk[r] = keys[x]; sc[r] = x; }
        }
    }
}

template <typename K> struct DescKey;
template <> struct DescKey<float> { typedef uint32_t T; static __device__ __forceinline__ T of(float v) { return desc_key_f32(v); } };
template <> struct DescKey<double> { typedef uint64_t T; static __device__ __forceinline__ T of(double v) { return desc_key_f64(v); } };

template <typename K>
__global__ void long_merge_kernel(LongArgs a) {
    typedef typename DescKey<K>::T U;
    const int row = blockIdx.y;
    int m = a.row_len ? a.row_len[row] : a.n;
    m = m < 0 ? 0 : (m > a.n ? a.n : m);
    const K* ck = reinterpret_cast<const K*>(a.ck) + (size_t)row * a.n;
    const int32_t* cc = a.cc + (size_t)row * a.n;
    const int CH = a.chunk_len, C = (m + CH - 1) / CH;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < m; x += gridDim.x * blockDim.x) {
        const int c = x / CH, i = x - c * CH;
        const K v = ck[x];
        const U key = DescKey<K>::of(v);
        int pos = i;
        for (int c2 = 0; c2 < C; ++c2) {
            if (c2 == c) continue;
            const K* base = ck + (size_t)c2 * CH;
            const int len = (m - c2 * CH) < CH ? (m - c2 * CH) : CH;
            int lo = 0, hi = len;                              // first index whose key is > key (c2 < c: ties precede) or >= key (c2 > c)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const U k2 = DescKey<K>::of(base[mid]);
                const bool before = c2 < c ? (k2 <= key) : (k2 < key);
                if (before) lo = mid + 1; else hi = mid;
            }
            pos += lo;
        }
        const int col = cc[x];
        if (a.order) a.order[(size_t)row * a.out_ld + pos] = col;
        if (a.sorted_keys) {
            // the key as the short-row kernel writes it back: -0.0 -> +0.0, every NaN -> the canonical quiet NaN
            K o = v;
            if (v != v) o = sizeof(K) == 4 ? (K)__uint_as_float(0x7fc00000u) : (K)__longlong_as_double(0x7ff8000000000000ll);
            else if (v == (K)0) o = (K)0;
            reinterpret_cast<K*>(a.sorted_keys)[(size_t)row * a.out_ld + pos] = o;
        }
        if (a.rank && col >= 0) a.rank[(size_t)row * a.out_ld + col] = pos;
    }
}

template <typename K>
__global__ void long_fill_kernel(K* keys, int32_t* cols, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        keys[i] = (K)(-INFINITY);
        cols[i] = -1;
    }
}

struct SortCfg { int T, E; };

static inline bool pick_cfg(int n, int kw, SortCfg& c) {
    if (n <= 1024) c = {256, 4};
    else if (n <= 4096) c = {256, 16};
    else if (n <= 8192) c = {512, 16};
    else if (n <= 16384) c = {1024, 16};
    else if (n <= 28672) c = {1024, 28};
    else if (n <= 35840 && kw == 1) c = {1024, 35};
    else return false;
    return true;
}

template <int T, int E, int KW, int MODE = SORT_ANY>
static int launch_cfg(const SortArgs& a, int prows, hipStream_t st) {
    constexpr size_t lds = SortLds<T, E, KW>::bytes;
    static_assert(lds <= 160 * 1024, "LDS budget of one CU");
    static unsigned long long lds_set = 0ull, lds_set_gen = 0ull;   // per (T,E,KW,FUSE) instantiation
    constexpr int GEN_MODE = (MODE == SORT_FUSE || MODE == SORT_ROWS_ZC) ? (T == 1024 ? SORT_ROWS : SORT_ANY) : MODE;   // flagged rows of the fused / zero-compacting form: the plain generic kernel
    if (int rc = raise_lds_limit((const void*)sort_rows_kernel<T, E, KW, false, MODE>, lds, lds_set)) return rc;
    if constexpr (KW == 2)
        if (int rc = raise_lds_limit((const void*)sort_rows_kernel<T, E, 2, true, GEN_MODE>, lds, lds_set_gen)) return rc;
    if (KW == 2 && !a.row_flags) return FZ_ERR_WORKSPACE;
    SortArgs b = a;
    {   // FZ_SORT_BUCKET_RANK=0: digit passes only (A/B runs, tests of the two forms against each other)
        const char* e = getenv("FZ_SORT_BUCKET_RANK");
        b.bucket_rank = (e && e[0] == '0') ? 0 : 1;
        b.zero_compact = 1;   // (FZ_SORT_ZERO_COMPACT=0 keeps the launcher from picking the SORT_ROWS_ZC instantiation at all: launch_sort)
    }
    sort_rows_kernel<T, E, KW, false, MODE><<<prows, T, lds, st>>>(b);
    FZ_LAUNCH_CHECK();
    if constexpr (KW == 2) {   // rows the fast form flagged (a dirty run of > 17 equal high words): generic eight passes; all others exit at once
        SortArgs g = a;
        if constexpr (MODE == SORT_FUSE) {   // their fused scores as a plain row first (every other row: nothing), then sorted like any float64 plane
            if (!a.fuse_gen_plane) return FZ_ERR_WORKSPACE;
            fuse_flagged_rows_kernel<<<(unsigned)prows, 256, 0, st>>>(a);
            FZ_LAUNCH_CHECK();
            g.keys = a.fuse_gen_plane; g.fuse_S = 0;
        }
        sort_rows_kernel<T, E, 2, true, GEN_MODE><<<prows, T, lds, st>>>(g);
        FZ_LAUNCH_CHECK();
    }
    return FZ_OK;
}

static int launch_sort(const SortArgs& a, int kw, int prows, int n_chunk, hipStream_t st) {
    SortCfg c;
    if (!pick_cfg(n_chunk, kw, c)) return FZ_ERR_UNSUPPORTED;
    if (a.fuse_S > 0) {   // rank fusion as the load phase (float64 keys, one workgroup per row)
        if (kw != 2 || a.chunks != 1 || a.row_stats) return FZ_ERR_UNSUPPORTED;
#define FZ_SORT_CASE(TT, EE) if (c.T == TT && c.E == EE) return launch_cfg<TT, EE, 2, SORT_FUSE>(a, prows, st);
        FZ_SORT_CASE(256, 4) FZ_SORT_CASE(256, 16) FZ_SORT_CASE(512, 16) FZ_SORT_CASE(1024, 16) FZ_SORT_CASE(1024, 28)
#undef FZ_SORT_CASE
        return FZ_ERR_UNSUPPORTED;
    }
    // whole rows of one plane, 1024-thread configurations (the rankers' sorts and the final order at LLeQA size): the lean instantiation
    static const int lean_env = [] { const char* e = getenv("FZ_SORT_LEAN"); return e ? atoi(e) : -1; }();   // A/B runs: 0 = never, 1 = fp32 rows too
    if (lean_env != 0 && a.chunks == 1 && a.seg_len >= a.n_total && !a.colmap && !a.idmap && !a.out_ids && c.T == 1024) {
        // float64 keys: the lean form holds no spilled register (120 VGPRs; SORT_ANY: 9 spilled) and measures 2.5 % (plain rows) to
        // 7 % (placed rows) faster.  float32 keys stay on SORT_ANY unless FZ_SORT_LEAN=1: hipcc schedules their lean form into MORE
        // spills (122, sixteen reloads inside the pass loop) and it measures 15 % slower (profiles/r05_sort_modes_ab.json)
        const char* zenv = getenv("FZ_SORT_ZERO_COMPACT");   // =0: the plain instantiation for everybody (A/B runs, tests of the two forms; read per launch)
        const bool zc_env = !(zenv && zenv[0] == '0');
        if (kw == 2 && a.expect_zeros && zc_env && a.order && !a.init_order && !a.init_rank) {   // a lexical ranker's rows: the zero-compacting instantiation
            if (c.E == 16) return launch_cfg<1024, 16, 2, SORT_ROWS_ZC>(a, prows, st);
            if (c.E == 28) return launch_cfg<1024, 28, 2, SORT_ROWS_ZC>(a, prows, st);
        }
        if (kw == 2 && c.E == 16) return launch_cfg<1024, 16, 2, SORT_ROWS>(a, prows, st);
        if (kw == 2 && c.E == 28) return launch_cfg<1024, 28, 2, SORT_ROWS>(a, prows, st);
        if (kw == 1 && lean_env == 1 && c.E == 28) return launch_cfg<1024, 28, 1, SORT_ROWS>(a, prows, st);
    }
#define FZ_SORT_CASE(TT, EE)                                                      \
    if (c.T == TT && c.E == EE) return kw == 1 ? launch_cfg<TT, EE, 1>(a, prows, st) : launch_cfg<TT, EE, 2>(a, prows, st);
    FZ_SORT_CASE(256, 4) FZ_SORT_CASE(256, 16) FZ_SORT_CASE(512, 16) FZ_SORT_CASE(1024, 16) FZ_SORT_CASE(1024, 28)
#undef FZ_SORT_CASE
    if (c.T == 1024 && c.E == 35 && kw == 1) return launch_cfg<1024, 35, 1>(a, prows, st);
    return FZ_ERR_UNSUPPORTED;
}


// ---- top-k form of the fused-list ordering (hybrid.py:306 + main's predictions(1000), :537) -------------------------------------------
// Aggregator.fuse sorts every fused row in full; what main() reads of it is the first 1000 entries.  For that use the row is SELECTED, not
// sorted: one workgroup holds the row's float32 sort keys in registers, bisects the key range for a threshold that at least k and at most
// cap keys do not exceed (one count + workgroup sum per step), and writes the documents at or above it -- the k best, every tie at the
// k-th place and up to cap - k more -- with their fused scores and (negated) first-insertion positions.  Two small row sorts then put
// those candidates into insertion order and, stably, into score order: the first k entries of the full sort, bit for bit.
// float64 fused scores (rrf / bcf / 'none') are selected by their float32 rounding -- rounding is monotone, so the k-th largest rounded
// value is the rounding of the k-th largest value and nothing of the top-k is lost; the candidates keep their float64 scores.
struct SelectArgs {
    const void* fused; int key_bits;   // [rows][ld] float32 (32) or float64 (64)
    const int32_t* pos;                // nullable [rows][ld]: first-insertion position of the column, < 0 = in no list; NULL = the column itself
    int n, ld, k, cap;
    int32_t* cand_cols;                // [rows][cap]
    void* cand_vals;                   // [rows][cap] same type as fused
    float* cand_negpos;                // [rows][cap] -(float)position: descending sort = ascending insertion position
    int32_t* cand_len;                 // [rows]
    int32_t* overflow;                 // set when a row has more than cap candidates (the caller then sorts in full)
};

template <int T, int E, int KW>
__global__ __launch_bounds__(T) void topk_select_kernel(SelectArgs a) {
    constexpr int NW = T / 64;
    __shared__ uint32_t red[2][NW];
    const int row = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t base = (size_t)row * a.ld;
    const float* __restrict__ vf = reinterpret_cast<const float*>(a.fused) + base;
    const double* __restrict__ vd = reinterpret_cast<const double*>(a.fused) + base;
    const int32_t* __restrict__ ps = a.pos ? a.pos + base : nullptr;
    uint32_t key[E];
    uint32_t nvalid = 0u;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int j = w * E * 64 + i * 64 + lane;
        const bool in = j < a.n;
        const int p = in ? (ps ? ps[j] : j) : -1;
        const float v = KW == 1 ? vf[in ? j : 0] : (float)vd[in ? j : 0];
        const bool ok = in && p >= 0;
        key[i] = ok ? desc_key_f32(v) : 0xffffffffu;           // (no real key is all ones: that would be -NaN's pattern, mapped to 0)
        nvalid += ok ? 1u : 0u;
    }
    // wave sums / minima / maxima on the VALU (DPP + permlane swaps: no LDS round trips), partials through parity-buffered LDS slots:
    // one barrier per reduction
    auto wave_u32 = [&](uint32_t v, auto op) __attribute__((always_inline)) -> uint32_t {
        auto dpp = [&](uint32_t x, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xf, 0xf, true); };
        v = op(v, dpp(v, std::integral_constant<int, 0xB1>{})); v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));
        v = op(v, dpp(v, std::integral_constant<int, 0x141>{})); v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));
        float x = __uint_as_float(v), o = x;
        swap16(x, o); v = op(__float_as_uint(x), __float_as_uint(o));
        x = __uint_as_float(v); o = x;
        swap32(x, o);
        return op(__float_as_uint(x), __float_as_uint(o));
    };
    auto add_ = [](uint32_t p, uint32_t q) { return p + q; };
    auto min_ = [](uint32_t p, uint32_t q) { return p < q ? p : q; };
    auto max_ = [](uint32_t p, uint32_t q) { return p > q ? p : q; };
    int par = 0;
    auto block_u32 = [&](uint32_t v, auto op, uint32_t ident) __attribute__((always_inline)) -> uint32_t {
        v = wave_u32(v, op);
        if (lane == 0) red[par][w] = v;
        __syncthreads();
        uint32_t t = ident;
#pragma unroll
        for (int i = 0; i < NW; ++i) t = op(t, red[par][i]);
        par ^= 1;
        return t;
    };
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int i = 0; i < E; ++i) if (key[i] != 0xffffffffu) { kmin = min_(kmin, key[i]); kmax = max_(kmax, key[i]); }
    const uint32_t total = block_u32(nvalid, add_, 0u);
    kmin = block_u32(kmin, min_, 0xffffffffu);
    kmax = block_u32(kmax, max_, 0u);
    const uint32_t need = (uint32_t)a.k < total ? (uint32_t)a.k : total;   // fewer listed documents than k: all of them
    // A threshold key tau with need <= #{key <= tau} <= cap is all the two sorts behind this kernel need -- not the exact k-th smallest
    // key -- so the bisection over the key range stops at the first midpoint whose count falls into that window (k = 1000 of 27,942,
    // cap = 2 k: 8-12 counts instead of one per key bit).  If no midpoint does (a tie run longer than cap - k at the k-th place) it ends
    // at the smallest key with at least `need` keys at or below it, the count exceeds cap and the overflow flag sends the caller to the
    // full sort.
    uint32_t tau = kmax;
    if (total > (uint32_t)a.cap) {
        uint32_t lo = kmin, hi = kmax;                         // invariant: count(<= hi) >= need
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < E; ++i) c += key[i] <= mid ? 1u : 0u;
            c = block_u32(c, add_, 0u);
            if (c < need) lo = mid + 1;
            else { hi = mid; tau = mid; if (c <= (uint32_t)a.cap) break; }
        }
        if (lo >= hi) tau = hi;
    }
    // candidates: every listed document whose key does not exceed the threshold
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) mine += (need > 0 && key[i] <= tau && key[i] != 0xffffffffu) ? 1u : 0u;
    uint32_t incl = wave_incl_scan_u32(mine, lane);
    __syncthreads();                                           // (the last reduction's slots have been read)
    if (lane == 63) red[0][w] = incl;
    __syncthreads();
    uint32_t off = incl - mine, all = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { if (i < w) off += red[0][i]; all += red[0][i]; }
    if (threadIdx.x == 0) {
        a.cand_len[row] = (int32_t)(all < (uint32_t)a.cap ? all : (uint32_t)a.cap);
        if (all > (uint32_t)a.cap) atomicExch(a.overflow, 1);
    }
    const size_t cb = (size_t)row * a.cap;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (need > 0 && key[i] <= tau && key[i] != 0xffffffffu) {
            if (off < (uint32_t)a.cap) {
                const int j = w * E * 64 + i * 64 + lane;
                a.cand_cols[cb + off] = j;
                a.cand_negpos[cb + off] = -(float)(ps ? ps[j] : j);
                if (KW == 1) reinterpret_cast<float*>(a.cand_vals)[cb + off] = vf[j];
                else reinterpret_cast<double*>(a.cand_vals)[cb + off] = vd[j];
            }
            ++off;
        }
    }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_select_topk_f(const void* fused, int key_bits, const int32_t* pos, int rows, int n, int ld, int k, int cap, int32_t* cand_cols,
                                void* cand_vals, float* cand_negpos, int32_t* cand_len, int32_t* overflow, void* stream) {
    if ((key_bits != 32 && key_bits != 64) || rows < 0 || n < 0 || ld < n || k <= 0 || cap < k) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    if (!cand_len || !overflow) return FZ_ERR_ARG;
    if (n == 0) { FZ_HIP_TRY(hipMemsetAsync(cand_len, 0, (size_t)rows * 4, as_stream(stream))); return FZ_OK; }
    if (!fused || !cand_cols || !cand_vals || !cand_negpos) return FZ_ERR_ARG;
    if (n > 1024 * 28) return FZ_ERR_UNSUPPORTED;   // one workgroup holds the row (and positions stay exact in float32)
    SelectArgs a{fused, key_bits, pos, n, ld, k, cap, cand_cols, cand_vals, cand_negpos, cand_len, overflow};
    hipStream_t st = as_stream(stream);
#define FZ_SEL(TT, EE) { if (key_bits == 32) topk_select_kernel<TT, EE, 1><<<rows, TT, 0, st>>>(a); else topk_select_kernel<TT, EE, 2><<<rows, TT, 0, st>>>(a); }
    if (n <= 256 * 4) FZ_SEL(256, 4)
    else if (n <= 256 * 16) FZ_SEL(256, 16)
    else if (n <= 1024 * 16) FZ_SEL(1024, 16)
    else FZ_SEL(1024, 28)
#undef FZ_SEL
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_sort_bucket_rank_rows(uint64_t* counts3, int reset) {
    if (counts3) {
        unsigned long long h[3];
        FZ_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bucket_rank_rows), sizeof h));
        for (int i = 0; i < 3; ++i) counts3[i] = (uint64_t)h[i];
    }
    if (reset) {
        const unsigned long long z[3] = {0ull, 0ull, 0ull};
        FZ_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bucket_rank_rows), z, sizeof z));
    }
    return FZ_OK;
}

extern "C" int fz_sort_zero_compact_rows(uint64_t* counts2, int reset) {
    if (counts2) {
        unsigned long long h[2];
        FZ_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_zero_compact_rows), sizeof h));
        counts2[0] = (uint64_t)h[0]; counts2[1] = (uint64_t)h[1];
    }
    if (reset) {
        const unsigned long long z[2] = {0ull, 0ull};
        FZ_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_zero_compact_rows), z, sizeof z));
    }
    return FZ_OK;
}

extern "C" int fz_sort_max_n(void) { return 35840; }   // single-workgroup rows (the fast path), fp32 keys; fp64 keys: 28672.  Longer rows: chunk-sort + merge
extern "C" int fz_sort_max_n_f64(void) { return 28672; }


static size_t long_ws_bytes(int key_bits, int rows, int n) {
    const size_t ksz = key_bits / 8;
    const int CH = key_bits == 32 ? 35840 : 28672;
    const size_t C = ((size_t)n + CH - 1) / CH;
    return (size_t)rows * n * (ksz + 4) * 2 + (size_t)rows * C * 4 + 512;
}

static int sort_long_rows(const void* keys, int key_bits, const int32_t* init_order, const int32_t* init_rank, const int32_t* row_len,
                          int rows, int n, int ld, int32_t* order, void* sorted_keys, int32_t* rank, void* workspace,
                          size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < long_ws_bytes(key_bits, rows, n)) return FZ_ERR_WORKSPACE;
    const size_t ksz = key_bits / 8;
    const int CH = key_bits == 32 ? 35840 : 28672;
    const int C = (n + CH - 1) / CH;
    char* ws = reinterpret_cast<char*>(workspace);
    void* seq_keys = ws; ws += (size_t)rows * n * ksz;
    void* ck = ws; ws += (size_t)rows * n * ksz;
    int32_t* seq_cols = reinterpret_cast<int32_t*>(ws); ws += (size_t)rows * n * 4;
    int32_t* cc = reinterpret_cast<int32_t*>(ws); ws += (size_t)rows * n * 4;
    int32_t* flags = reinterpret_cast<int32_t*>(ws);
    LongArgs L{};
    L.keys = keys; L.key_bits = key_bits; L.init_order = init_order; L.init_rank = init_rank; L.row_len = row_len;
    L.rows = rows; L.n = n; L.ld = ld; L.seq_keys = seq_keys; L.seq_cols = seq_cols; L.ck = ck; L.cc = cc; L.chunk_len = CH;
    L.order = order; L.sorted_keys = sorted_keys; L.rank = rank; L.out_ld = ld;
    const bool seq = init_order || init_rank;
    dim3 grid((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), (unsigned)rows);
    if (seq) {
        if (init_rank) {   // slots no column claims: (-inf, -1)
            if (key_bits == 32) long_fill_kernel<float><<<1024, 256, 0, st>>>((float*)seq_keys, seq_cols, (size_t)rows * n);
            else long_fill_kernel<double><<<1024, 256, 0, st>>>((double*)seq_keys, seq_cols, (size_t)rows * n);
            FZ_LAUNCH_CHECK();
        }
        if (key_bits == 32) long_seq_kernel<float><<<grid, 256, 0, st>>>(L); else long_seq_kernel<double><<<grid, 256, 0, st>>>(L);
        FZ_LAUNCH_CHECK();
    }
    SortArgs a{};
    a.keys = seq ? seq_keys : keys; a.row_len = row_len;
    a.n_total = n; a.key_row_stride = seq ? n : ld; a.seg_len = n; a.seg_stride = 0;
    a.chunks = C; a.chunk_len = CH;
    a.order = cc; a.sorted_keys = ck; a.out_row_stride = n; a.out_chunk_stride = CH; a.out_limit = CH;
    a.colmap = seq ? seq_cols : nullptr; a.colmap_row_stride = n;
    a.row_flags = flags;
    int rc = launch_sort(a, key_bits / 32, rows * C, CH, st);
    if (rc != FZ_OK) return rc;
    if (key_bits == 32) long_merge_kernel<float><<<grid, 256, 0, st>>>(L); else long_merge_kernel<double><<<grid, 256, 0, st>>>(L);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" size_t fz_sort_workspace_bytes(int key_bits, int rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    if (n > (key_bits == 32 ? 35840 : 28672)) return long_ws_bytes(key_bits, rows, n);
    return key_bits == 64 ? (size_t)rows * 4 : 0;
}

extern "C" int fz_sort_rows_desc(const void* keys, int key_bits, const int32_t* init_order, const int32_t* row_len, int rows,
                                 int n, int ld, int32_t* order, void* sorted_keys, int32_t* rank, float* row_stats, const int32_t* stats_len,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if ((key_bits != 32 && key_bits != 64) || rows < 0 || n < 0 || ld < n) return FZ_ERR_ARG;
    if (rows == 0 || n == 0) return FZ_OK;      // nothing to do (empty tensors carry null pointers)
    if (!keys) return FZ_ERR_ARG;
    if (stats_len && !row_stats) return FZ_ERR_ARG;
    if (stats_len && key_bits != 32) return FZ_ERR_UNSUPPORTED;   // prefix statistics come from the sorted fp32 keys in registers (fz_row_stats_f32 otherwise)
    if (n > (key_bits == 32 ? 35840 : 28672)) {   // longer than one workgroup's registers: chunk-sort + cross-chunk ranking
        if (row_stats) return FZ_ERR_UNSUPPORTED; // the statistics by-product exists for single-workgroup rows only (fz_row_stats_f32 otherwise)
        return sort_long_rows(keys, key_bits, init_order, nullptr, row_len, rows, n, ld, order, sorted_keys, rank, workspace, workspace_bytes,
                              as_stream(stream));
    }
    SortArgs a{};
    a.keys = keys; a.init_order = init_order; a.row_len = row_len;
    a.n_total = n; a.key_row_stride = ld; a.seg_len = n; a.seg_stride = 0;
    a.chunks = 1; a.chunk_len = n;
    a.order = order; a.sorted_keys = sorted_keys; a.rank = rank; a.row_stats = row_stats; a.stats_rows = rows; a.stats_len = stats_len;
    a.out_row_stride = ld; a.out_chunk_stride = 0; a.out_limit = n;
    if (key_bits == 64) { if (!workspace || workspace_bytes < fz_sort_workspace_bytes(64, rows, n)) return FZ_ERR_WORKSPACE; a.row_flags = (int32_t*)workspace; }
    return launch_sort(a, key_bits / 32, rows, n, as_stream(stream));
}

// A lexical ranker's rows (BM25 / TF-IDF scores, bm25.py:100-106: every document that shares no term with the query scores exactly 0.0): the
// same stable descending sort of float64 rows in the identity sequence, through the instantiation that leaves a row's zeros out of the
// ordering phases when they are at least 3/8 of its >= 4,096 keys (ZC in sort_rows_kernel).  Same outputs, bit for bit, as fz_sort_rows_desc.
extern "C" int fz_sort_rows_desc_lexical(const double* keys, const int32_t* row_len, int rows, int n, int ld, int32_t* order, double* sorted_keys,
                                         int32_t* rank, float* row_stats, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || n < 0 || ld < n) return FZ_ERR_ARG;
    if (rows == 0 || n == 0) return FZ_OK;
    if (!keys) return FZ_ERR_ARG;
    if (n > 28672 || n <= 8192)   // rows of other workgroup sizes: nothing to gain, the general entry point
        return fz_sort_rows_desc(keys, 64, nullptr, row_len, rows, n, ld, order, sorted_keys, rank, row_stats, nullptr, workspace, workspace_bytes, stream);
    SortArgs a{};
    a.keys = keys; a.row_len = row_len;
    a.n_total = n; a.key_row_stride = ld; a.seg_len = n; a.seg_stride = 0;
    a.chunks = 1; a.chunk_len = n;
    a.order = order; a.sorted_keys = sorted_keys; a.rank = rank; a.row_stats = row_stats; a.stats_rows = rows;
    a.out_row_stride = ld; a.out_chunk_stride = 0; a.out_limit = n;
    a.expect_zeros = 1;
    if (!workspace || workspace_bytes < fz_sort_workspace_bytes(64, rows, n)) return FZ_ERR_WORKSPACE;
    a.row_flags = (int32_t*)workspace;
    return launch_sort(a, 2, rows, n, as_stream(stream));
}

extern "C" int fz_sort_rows_desc_placed(const void* keys, int key_bits, const int32_t* init_rank, const int32_t* row_len, int rows,
                                        int n, int ld, int32_t* order, void* sorted_keys, int32_t* rank, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if ((key_bits != 32 && key_bits != 64) || rows < 0 || n < 0 || ld < n) return FZ_ERR_ARG;
    if (rows == 0 || n == 0) return FZ_OK;
    if (!keys || !init_rank) return FZ_ERR_ARG;
    if (n > (key_bits == 32 ? 35840 : 28672))
        return sort_long_rows(keys, key_bits, nullptr, init_rank, row_len, rows, n, ld, order, sorted_keys, rank, workspace, workspace_bytes,
                              as_stream(stream));
    SortArgs a{};
    a.keys = keys; a.init_rank = init_rank; a.row_len = row_len;
    a.n_total = n; a.key_row_stride = ld; a.seg_len = n; a.seg_stride = 0;
    a.chunks = 1; a.chunk_len = n;
    a.order = order; a.sorted_keys = sorted_keys; a.rank = rank;
    a.out_row_stride = ld; a.out_chunk_stride = 0; a.out_limit = n;
    if (key_bits == 64) { if (!workspace || workspace_bytes < fz_sort_workspace_bytes(64, rows, n)) return FZ_ERR_WORKSPACE; a.row_flags = (int32_t*)workspace; }
    return launch_sort(a, key_bits / 32, rows, n, as_stream(stream));
}

// Rank fusion + final order in one kernel (hybrid.py:248-252 + :301-306): what fz_fuse_rank_f64 followed by fz_sort_rows_desc[_placed] on its
// float64 plane computes -- the same fused scores bit for bit (formed per key on load by the expression of fuse_rank_kernel), the same
// stable order -- without the [rows][ld] float64 plane ever being written or read (229 MB out + 458 MB in at Q = 1024, N = 27,942).
static size_t fused_flags_bytes(int rows) { return ((size_t)rows * 4 + 255) / 256 * 256; }
// row flags + one float64 row per row: where a row the fast form cannot finish (flagged; practically never) gets its fused scores written
// out for the generic launch.  Untouched otherwise: the bytes are reserved, not moved.
extern "C" size_t fz_sort_rank_fused_workspace_bytes(int rows, int n, int ld) {
    if (rows <= 0 || n <= 0 || ld < n) return 0;
    return fused_flags_bytes(rows) + (size_t)rows * ld * 8;
}

extern "C" int fz_sort_rank_fused_desc(const int32_t* const* ranks_h, const int32_t* lens, int S, int method, const int32_t* init_order,
                                       const int32_t* init_rank, const int32_t* row_len, int rows, int n, int ld, int32_t* order,
                                       double* sorted_scores, int32_t* rank, void* workspace, size_t workspace_bytes, void* stream) {
    if (S <= 0 || S > FZ_MAX_SYSTEMS || rows < 0 || n < 0 || ld < n) return FZ_ERR_ARG;
    if (method != FZ_RRF && method != FZ_BCF) return FZ_ERR_ARG;
    if (init_order && init_rank) return FZ_ERR_ARG;
    if (rows == 0 || n == 0) return FZ_OK;
    if (!ranks_h || !lens) return FZ_ERR_ARG;
    if (n > 28672) return FZ_ERR_UNSUPPORTED;      // longer rows: fz_fuse_rank_f64 + fz_sort_rows_desc (chunk-sort + merge)
    SortArgs a{};
    for (int s = 0; s < S; ++s) {
        if (!ranks_h[s]) return FZ_ERR_ARG;
        a.fuse_ranks[s] = ranks_h[s];
    }
    a.fuse_lens = lens; a.fuse_S = S; a.fuse_method = method; a.fuse_rows = rows;
    a.fuse_first_is_pos = (init_rank && init_rank == ranks_h[0]) ? 1 : 0;
    a.init_order = init_order; a.init_rank = init_rank; a.row_len = row_len;
    a.n_total = n; a.key_row_stride = ld; a.seg_len = n; a.seg_stride = 0;
    a.chunks = 1; a.chunk_len = n;
    a.order = order; a.sorted_keys = sorted_scores; a.rank = rank;
    a.out_row_stride = ld; a.out_chunk_stride = 0; a.out_limit = n;
    if (!workspace || workspace_bytes < fz_sort_rank_fused_workspace_bytes(rows, n, ld)) return FZ_ERR_WORKSPACE;
    a.row_flags = (int32_t*)workspace;
    a.fuse_gen_plane = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + fused_flags_bytes(rows));
    return launch_sort(a, 2, rows, n, as_stream(stream));
}

// Diagnostic: out[r] = 1 / (60 + r + 1) in float64 for r < count -- fast != 0: by the division-free form the fused sort's load phase uses
// (recip_small_int_f64), else by the IEEE division fuse_rank_kernel uses.  Tests compare the two (and NumPy) over every rank.
extern "C" int fz_rrf_terms_f64(int count, int fast, double* out, void* stream) {
    if (count < 0 || count > (1 << 20)) return FZ_ERR_ARG;
    if (count == 0) return FZ_OK;
    if (!out) return FZ_ERR_ARG;
    rrf_terms_kernel<<<(count + 255) / 256, 256, 0, as_stream(stream)>>>(count, fast, out);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// ---- top-k: chunk-sort-truncate levels until one workgroup can finish the row ---------------
static const int TOPK_CHUNK = 28672;
extern "C" int fz_topk_max_k(void) { return 8192; }

static void topk_plan(int n, int k, int* levels_out, size_t* elems_out) {
    // level l: cur columns -> chunks of TOPK_CHUNK -> kk = min(k, TOPK_CHUNK) survivors per chunk
    int cur = n, levels = 0;
    size_t elems = 0;
    while (cur > 35840) {
        int nch = (cur + TOPK_CHUNK - 1) / TOPK_CHUNK;
        int kk = k < TOPK_CHUNK ? k : TOPK_CHUNK;
        cur = nch * kk;
        elems += (size_t)cur;
        ++levels;
    }
    *levels_out = levels;
    *elems_out = elems;
}

__global__ void topk_fill_kernel(float* keys, int32_t* cols, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        keys[i] = -INFINITY;
        cols[i] = -1;
    }
}

extern "C" size_t fz_topk_workspace_bytes(int rows, int n, int k) {
    if (rows <= 0 || n <= 0 || k <= 0) return 0;
    int levels; size_t elems;
    topk_plan(n, k, &levels, &elems);
    return (size_t)rows * elems * 8 + 256;  // fp32 score + int32 column per surviving candidate
}

extern "C" int fz_topk_rows_f32(const float* scores, int rows, int n, int ld, int k, int64_t id_base, float* out_scores,
                                int64_t* out_ids, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || n < 0 || ld < n || k <= 0) return FZ_ERR_ARG;
    if (k > fz_topk_max_k()) return FZ_ERR_UNSUPPORTED;
    if (rows == 0) return FZ_OK;
    if (!out_scores || !out_ids || (!scores && n > 0)) return FZ_ERR_ARG;
    hipStream_t st = as_stream(stream);
    const int have = n < k ? n : k;
    if (have < k) {
        dim3 g((unsigned)((k - have + 255) / 256), (unsigned)rows);
        topk_pad_kernel<<<g, 256, 0, st>>>(out_scores, out_ids, rows, k, have);
        FZ_LAUNCH_CHECK();
    }
    if (n == 0) return FZ_OK;
    if (workspace_bytes < fz_topk_workspace_bytes(rows, n, k)) return FZ_ERR_WORKSPACE;
    if (n > 35840 && !workspace) return FZ_ERR_WORKSPACE;

    const float* cur_keys = scores;
    const int32_t* cur_cols = nullptr;
    long cur_stride = ld;
    int cur = n;
    char* ws = reinterpret_cast<char*>(workspace);
    while (cur > 35840) {
        const int per = TOPK_CHUNK;
        int nch = (cur + per - 1) / per;
        int kk = k < per ? k : per;
        int next = nch * kk;
        float* nk = reinterpret_cast<float*>(ws); ws += (size_t)rows * next * 4;
        int32_t* nc = reinterpret_cast<int32_t*>(ws); ws += (size_t)rows * next * 4;
        if (cur - (nch - 1) * per < kk) {
            // the short last chunk leaves a tail unwritten: make it (-inf, col -1) so it cannot win
            topk_fill_kernel<<<1024, 256, 0, st>>>(nk, nc, (size_t)rows * next);
            FZ_LAUNCH_CHECK();
        }
        SortArgs a{};
        a.keys = cur_keys; a.n_total = cur; a.key_row_stride = cur_stride; a.seg_len = cur; a.chunks = nch; a.chunk_len = per;
        a.order = nc; a.sorted_keys = nk; a.out_row_stride = next; a.out_chunk_stride = kk; a.out_limit = kk;
        a.colmap = cur_cols; a.colmap_row_stride = cur_stride;
        int rc = launch_sort(a, 1, rows * nch, per, st);
        if (rc != FZ_OK) return rc;
        cur_keys = nk; cur_cols = nc; cur_stride = next; cur = next;
    }
    SortArgs a{};
    a.keys = cur_keys; a.n_total = cur; a.key_row_stride = cur_stride; a.seg_len = cur; a.chunks = 1; a.chunk_len = cur;
    a.sorted_keys = out_scores; a.out_ids = out_ids; a.id_base = id_base; a.out_row_stride = k; a.out_limit = have;
    a.colmap = cur_cols; a.colmap_row_stride = cur_stride;
    return launch_sort(a, 1, rows, cur, st);
}

extern "C" size_t fz_topk_update_workspace_bytes(int rows, int k, int cap) {
    if (rows <= 0 || k <= 0 || cap <= 0) return 0;
    return (size_t)rows * (k + cap) * (4 + 8) + (size_t)rows * 4 + 256;
}

/* One streaming step of the chunked top-k (sentence_transformers.py:346-364): merge a new chunk of scores into the
 * running per-row top-k.  run_* [rows][k] in, new_* [rows][k] out (distinct buffers).  *overflow (device int32,
 * zeroed by the caller) becomes 1 if some row had more than `cap` candidates above its running threshold: the caller
 * must then redo this chunk with fz_topk_rows_f32 + fz_topk_merge (exact, slower).  k + cap <= 35840. */
extern "C" int fz_topk_update_f32(const float* scores, int rows, int n, int ld, int64_t id_base, const float* run_scores,
                                  const int64_t* run_ids, int k, int cap, float* new_scores, int64_t* new_ids, int32_t* overflow,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || n < 0 || ld < n || k <= 0 || cap <= 0) return FZ_ERR_ARG;
    if ((long)k + cap > 35840) return FZ_ERR_UNSUPPORTED;
    if (rows == 0) return FZ_OK;                   // empty tensors carry null pointers
    if (!run_scores || !run_ids || !new_scores || !new_ids || !overflow || (!scores && n > 0)) return FZ_ERR_ARG;
    if (!workspace || workspace_bytes < fz_topk_update_workspace_bytes(rows, k, cap)) return FZ_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    FilterArgs f{};
    f.scores = scores; f.n = n; f.ld = ld; f.id_base = id_base; f.run_scores = run_scores; f.run_ids = run_ids; f.k = k; f.cap = cap;
    f.buf_ids = reinterpret_cast<int64_t*>(ws); ws += (size_t)rows * (k + cap) * 8;
    f.buf_scores = reinterpret_cast<float*>(ws); ws += (size_t)rows * (k + cap) * 4;
    f.buf_len = reinterpret_cast<int32_t*>(ws);
    f.overflow = overflow;
    topk_filter_kernel<<<rows, 256, 0, st>>>(f);
    FZ_LAUNCH_CHECK();
    SortArgs a{};
    a.keys = f.buf_scores; a.row_len = f.buf_len; a.n_total = k + cap; a.key_row_stride = k + cap; a.seg_len = k + cap;
    a.chunks = 1; a.chunk_len = k + cap;
    a.sorted_keys = new_scores; a.out_ids = new_ids; a.idmap = f.buf_ids; a.out_row_stride = k; a.out_limit = k;
    return launch_sort(a, 1, rows, k + cap, st);
}

/* The streaming step in two halves, so that several chunks can share one sort: fz_topk_filter_append_f32 appends the chunk's
 * scores above tau[row] (or NaN) to the row's candidate list (ascending id inside the chunk; chunks must be fed in ascending id
 * order); fz_topk_fold_f32 merges [running k | candidates] into the new running list, writes the new threshold
 * tau[row] = k-th best and empties the candidate lists.  cand_* [rows][cap], cand_len [rows] (zeroed before the first call). */
extern "C" int fz_topk_filter_append_f32(const float* scores, int rows, int n, int ld, int64_t id_base, const float* tau, float* cand_scores,
                                         int64_t* cand_ids, int32_t* cand_len, int cap, int32_t* overflow, void* stream) {
    if (rows < 0 || n < 0 || ld < n || cap <= 0) return FZ_ERR_ARG;
    if (rows == 0 || n == 0) return FZ_OK;
    if (!scores || !tau || !cand_scores || !cand_ids || !cand_len || !overflow) return FZ_ERR_ARG;
    FilterArgs f{};
    f.scores = scores; f.n = n; f.ld = ld; f.id_base = id_base; f.k = 0; f.cap = cap; f.tau = tau;
    f.buf_scores = cand_scores; f.buf_ids = cand_ids; f.buf_len = cand_len; f.overflow = overflow;
    topk_filter_kernel<<<rows, 256, 0, as_stream(stream)>>>(f);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

// Candidates that arrive in no particular order (the GEMM's filter epilogue appends them as its waves finish): the score sort is
// stable, so equal scores come out in ARRIVAL order.  The sort therefore writes the first k + TIE_MARGIN entries to a scratch
// list and this pass puts every run of equal scores that reaches into the first k into ascending id order (running-list entries
// have smaller ids than any later candidate, so they stay in front).  A run still open at the end of the scratch list, or longer
// than TIE_RUN_MAX, sets *overflow: the caller redoes the search on the exact path.
constexpr int TIE_MARGIN = 64, TIE_RUN_MAX = 512;

__global__ __launch_bounds__(256) void topk_tiefix_kernel(const float* tmp_s, const int64_t* tmp_i, const int32_t* buf_len, int k, int stride,
                                                          float* out_s, int64_t* out_i, int32_t* overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tie_lds[];
    int64_t* li = reinterpret_cast<int64_t*>(tie_lds);                 // [stride]
    uint32_t* lk = reinterpret_cast<uint32_t*>(li + stride);           // [stride] sort keys of the scores (NaN == NaN, -0 == +0)
    const int row = blockIdx.x;
    const int have = buf_len[row];
    const int L = min(have, stride);
    const float* ts = tmp_s + (size_t)row * stride;
    const int64_t* ti = tmp_i + (size_t)row * stride;
    for (int p = threadIdx.x; p < L; p += blockDim.x) { lk[p] = desc_key_f32(ts[p]); li[p] = ti[p]; }
    __syncthreads();
    float* os = out_s + (size_t)row * k;
    int64_t* oi = out_i + (size_t)row * k;
    bool bad = false;
    for (int p = threadIdx.x; p < max(L, k); p += blockDim.x) {
        if (p >= L) { if (p < k) { os[p] = -INFINITY; oi[p] = -1; } continue; }      // short list: (-inf, -1) padding
        const uint32_t key = lk[p];
        int a = p, b = p + 1;
        while (a > 0 && lk[a - 1] == key && p - a < TIE_RUN_MAX) --a;
        if (a >= k) continue;                                                        // the run lies entirely behind the cut
        while (b < L && lk[b] == key && b - p < TIE_RUN_MAX) ++b;
        if ((a > 0 && lk[a - 1] == key) || (b < L && lk[b] == key) || (b == L && have > L)) { bad = true; continue; }
        const int64_t id = li[p];
        int rank = 0;
        for (int j = a; j < b; ++j) rank += (li[j] < id) || (li[j] == id && j < p);
        const int dst = a + rank;
        if (dst < k) { os[dst] = ts[p]; oi[dst] = id; }
    }
    if (bad) atomicExch(overflow, 1);
}

extern "C" size_t fz_topk_fold_workspace_bytes(int rows, int k, int cap) {
    if (rows <= 0 || k <= 0 || cap <= 0) return 0;
    return fz_topk_update_workspace_bytes(rows, k, cap) + (size_t)rows * (k + TIE_MARGIN) * (4 + 8) + 256;
}

extern "C" int fz_topk_fold_f32(const float* run_scores, const int64_t* run_ids, int rows, int k, const float* cand_scores, const int64_t* cand_ids,
                                int32_t* cand_len, int cap, int unordered, float* new_scores, int64_t* new_ids, float* tau_out, int32_t* overflow,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || k <= 0 || cap <= 0) return FZ_ERR_ARG;
    if ((long)k + cap > 35840) return FZ_ERR_UNSUPPORTED;
    if (rows == 0) return FZ_OK;
    if (!run_scores || !run_ids || !cand_scores || !cand_ids || !cand_len || !new_scores || !new_ids || (unordered && !overflow)) return FZ_ERR_ARG;
    if (!workspace || workspace_bytes < fz_topk_fold_workspace_bytes(rows, k, cap)) return FZ_ERR_WORKSPACE;
    hipStream_t st = as_stream(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    int64_t* buf_ids = reinterpret_cast<int64_t*>(ws); ws += (size_t)rows * (k + cap) * 8;
    int64_t* tmp_ids = reinterpret_cast<int64_t*>(ws); ws += (size_t)rows * (k + TIE_MARGIN) * 8;
    float* buf_scores = reinterpret_cast<float*>(ws); ws += (size_t)rows * (k + cap) * 4;
    float* tmp_scores = reinterpret_cast<float*>(ws); ws += (size_t)rows * (k + TIE_MARGIN) * 4;
    int32_t* buf_len = reinterpret_cast<int32_t*>(ws);
    topk_concat_kernel<<<rows, 256, 0, st>>>(run_scores, run_ids, k, cand_scores, cand_ids, cand_len, cap, buf_scores, buf_ids, buf_len);
    FZ_LAUNCH_CHECK();
    const int lim = unordered ? k + TIE_MARGIN : k;
    SortArgs a{};
    a.keys = buf_scores; a.row_len = buf_len; a.n_total = k + cap; a.key_row_stride = k + cap; a.seg_len = k + cap;
    a.chunks = 1; a.chunk_len = k + cap;
    a.sorted_keys = unordered ? tmp_scores : new_scores; a.out_ids = unordered ? tmp_ids : new_ids; a.idmap = buf_ids;
    a.out_row_stride = lim; a.out_limit = lim;
    if (int rc = launch_sort(a, 1, rows, k + cap, st)) return rc;
    if (unordered) {
        const size_t lds = (size_t)lim * (8 + 4);
        if (lds > 60 * 1024) return FZ_ERR_UNSUPPORTED;
        topk_tiefix_kernel<<<rows, 256, lds, st>>>(tmp_scores, tmp_ids, buf_len, k, lim, new_scores, new_ids, overflow);
        FZ_LAUNCH_CHECK();
    }
    topk_fold_done_kernel<<<(rows + 255) / 256, 256, 0, st>>>(new_scores, rows, k, tau_out, cand_len);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_topk_merge(const float* in_scores, const int64_t* in_ids, int G, int rows, int k, float* out_scores,
                             int64_t* out_ids, void* stream) {
    if (G <= 0 || rows < 0 || k <= 0) return FZ_ERR_ARG;
    if (rows != 0 && (!in_scores || !in_ids || !out_scores || !out_ids)) return FZ_ERR_ARG;   // empty tensors carry null pointers
    if ((long)G * k > 35840) return FZ_ERR_UNSUPPORTED;
    if (rows == 0) return FZ_OK;
    SortArgs a{};
    a.keys = in_scores; a.n_total = G * k; a.key_row_stride = k; a.seg_len = k; a.seg_stride = (long)rows * k;
    a.chunks = 1; a.chunk_len = G * k;
    a.sorted_keys = out_scores; a.out_ids = out_ids; a.idmap = in_ids; a.out_row_stride = k; a.out_limit = k;
    return launch_sort(a, 1, rows, G * k, as_stream(stream));
}
