// maxsim.hip -- K2: exact ColBERT late interaction on gfx950.
//
// Reference: Ranker.multi_vector_search (hybrid.py:108-137) -> colbert-ai Searcher.search_all; the score
// PLAID approximates is   s(q,d) = sum_{i<Lq} max_{t in d} <Q[q][i], D[t]>    on 128-d
// L2-normalised token vectors (run_colbert.sh:26-27), query padded to 64 tokens (hybrid.py:129).
//
// Mapping: v_mfma_f32_16x16x32_f16 with A = 16 DOCUMENT tokens (rows) x 32 dims, B = 32 dims x 16 QUERY tokens (cols), four
// k-steps for the 128 dims.  The C layout puts the query token on the lane (col = lane & 15) and the 16 document tokens in the
// 4 accumulator registers x 4 lane groups (row = 4 (lane >> 4) + r), so
//     max over document tokens = in-lane max of 4 registers per row block (v_max3), the four lane groups combined ONCE per
//                                document by permlane32 / permlane16 swaps that transpose 8 partial-max registers into 2,
//     sum over query tokens    = one 16-lane DPP sum per register + two swap-and-add steps, once per (query, document).
// A workgroup = 8 waves; every wave keeps the B fragments of 8 column blocks (2 queries x 64 tokens) in 128 VGPRs for the
// whole kernel, and all 8 waves consume the same 32-token document tile (two row blocks whose MFMA chains interleave), which
// goes global -> LDS once per workgroup by LDS-DMA (16-B chunks XOR-swizzled by row so that ds_read_b128 of 16 rows is
// conflict-free), a group of four tiles ahead in an 8-slot ring, one barrier per group.  Tiles are aligned to document starts
// (rows past the end are masked to -inf); a document's last tile with at most 16 tokens left runs its first row block only.
// Workgroups that share a 32-document range run back-to-back on one XCD, so the range (about 2.4 MB) is fetched from HBM once
// and served from that XCD's L2 to the other query groups.
//
// Where it stands (profiles/r03_pmc_maxsim.json, r03_maxsim_ab.jsonl): 1.42-1.46 PFLOP/s of exact
// sum-of-lengths FLOPs at Q = 1024 = 0.57-0.58 of the 2.5 PF nominal peak, matrix pipe busy 62 % of the SIMD cycles, 1.3 other
// vector instructions per 16-cycle MFMA, LDS conflicts 0.06 % -- what MI355X_MICROARCH.md measures for an LDS-fed 16x16x32 loop
// on random data (1.40-1.43 PF: the chip holds its clock down under matrix load).  What alignment to document starts costs: with
// the LLeQA-shaped lengths of the bench 2.5 % of the row-block slots are padding (7.5 of ~300 rows per document), so tiles
// packed across document boundaries could add at most that much -- 1.43 -> <= 1.47 PF -- for a piece-wise epilogue in every
// tile; not built (round 4, DESIGN.md 'Tried').  The round-2 32x32x16 form is kept as tools/ablate/maxsim_32x32x16.hip.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"

namespace fz {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MS_DIM = 128;
constexpr int MS_WAVES = 8;
constexpr int MS_BLOCKS_PER_WAVE = 4;   // query blocks of 32 tokens held per wave
constexpr int MS_DOCS_PER_WG = 32;
constexpr int MS_TILE_BYTES = 32 * MS_DIM * 2;  // 8 KiB
constexpr int MS_TABLE = MS_DOCS_PER_WG * 16;   // tile-table entries per workgroup (32 documents x 512 tokens)
constexpr int MS_GROUP = 4;                     // tiles per group: one workgroup barrier per group
constexpr int MS_RING = 2 * MS_GROUP;           // LDS tile slots: the group being read + the group being filled

struct MaxSimArgs {
    const _Float16* Qtok;   // [Q][Lq][128]
    const _Float16* Dtok;   // [sumL][128]
    const int64_t* Doff;    // [N+1]
    float* scores; int lds;
    int Q, Lq, N;
    int QB;                 // Lq / 32
    int QG;                 // query groups = ceil(Q*QB / (MS_WAVES*MS_BLOCKS_PER_WAVE))
    int DR;                 // document ranges = ceil(N / docs_per_wg)
    int docs_per_wg;        // <= MS_DOCS_PER_WG, chosen so that docs_per_wg * ceil(max_doc_len/32) <= MS_TABLE tiles
    int max_doc_len;        // tokens beyond this are ignored (the reference's doc_maxlen, hybrid.py:129)
    int64_t sumL;
};

// One workgroup = 8 waves x (up to) 8 column blocks of 16 query tokens; see the header.
__global__ __launch_bounds__(512, 2) void maxsim_kernel(MaxSimArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[MS_RING][MS_TILE_BYTES];   // ring of 8 KiB tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep what derives from it in SGPRs
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int qg = idx % a.QG;
    const int dr = (idx / a.QG) * 8 + x;
    if (dr >= a.DR) return;

    // ---- this wave's queries.  A group covers 8 waves x qpw queries; the queries of a group that is not full (the last one) are
    //      spread evenly over the waves (at Q = 195, Lq = 64: three waves with one query each instead of two waves with 2 + 1), so
    //      that the group's tile stream is not paced by waves that carry a full load next to idle ones ----
    const int qpw = MS_BLOCKS_PER_WAVE / a.QB;                 // queries a wave can hold (Lq = 32 / 64 / 128: 4 / 2 / 1)
    const int q_first = qg * MS_WAVES * qpw;
    const int nq_group = min(a.Q - q_first, MS_WAVES * qpw);
    const int per_wave = (nq_group + MS_WAVES - 1) / MS_WAVES;  // <= qpw
    const int q0 = q_first + w * per_wave;                      // this wave's first query
    const int nq = max(0, min(per_wave, q_first + nq_group - q0));   // ... and how many it has (wave-uniform)
    const int ncb = nq * a.QB * 2;                              // live column blocks (16 query tokens each) of this wave: 0 .. 8
    const bool wave_has_queries = nq > 0;

    // ---- B fragments: up to 8 column blocks x 4 k-steps of 32 dims, resident for the whole kernel (128 VGPRs) ----
    // lane l holds B[k = 8 (l >> 4) + j][col = l & 15] = Qtok[token l & 15 of the block][dim 32 ks + 8 (l >> 4) + j]
    constexpr int NCB = 2 * MS_BLOCKS_PER_WAVE;
    f16x8 bq[NCB][4];
#pragma unroll
    for (int b = 0; b < NCB; ++b) {
        const bool okb = b < ncb;
        const size_t tok = okb ? ((size_t)q0 * a.QB * 2 + b) * 16 + (lane & 15) : 0;   // a query's tokens are consecutive in [Q * Lq]
        const _Float16* src = a.Qtok + tok * MS_DIM + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 v = *reinterpret_cast<const f16x8*>(src + 32 * ks);
            if (!okb) v = (f16x8)(_Float16)0;
            bq[b][ks] = v;
        }
    }

    const int d_begin = dr * a.docs_per_wg;
    const int d_end = (d_begin + a.docs_per_wg < a.N) ? d_begin + a.docs_per_wg : a.N;

    // ---- tile table of this document range, built ONCE into LDS ------------------------------------------
    // (walking Doff[] with scalar global loads per tile put ~0.5 us of load latency on the critical path of every
    //  0.45 us of MFMA work).  Tiles are aligned to document starts; entry k: first token, valid rows, document, last flag.
    __shared__ int64_t s_tok[MS_TABLE + 4];
    __shared__ int s_meta[MS_TABLE + 4];      // doc_local | rows_valid << 8 | last << 16
    __shared__ int s_len[MS_DOCS_PER_WG];
    __shared__ int s_ntiles;
    if (tid < 64) {   // wave 0; lanes >= MS_DOCS_PER_WG idle along
        const int d = d_begin + lane;
        const bool live = lane < a.docs_per_wg && d < d_end;
        const int64_t t0 = live ? a.Doff[d] : 0;
        int len = live ? (int)(a.Doff[d + 1] - t0) : 0;
        if (len > a.max_doc_len) len = a.max_doc_len;   // caller's doc_maxlen (hybrid.py:129)
        const int nt = (len + 31) >> 5;
        int incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const int excl = incl - nt;
        for (int i = 0; i < nt; ++i) {
            const int rows = (len - 32 * i) < 32 ? (len - 32 * i) : 32;
            s_tok[excl + i] = t0 + 32 * i;
            s_meta[excl + i] = lane | (rows << 8) | ((i == nt - 1) ? (1 << 16) : 0);
        }
        if (lane < MS_DOCS_PER_WG) s_len[lane] = live ? len : -1;
        if (lane == 63) s_ntiles = incl;
    }
    __syncthreads();
    const int ntiles = s_ntiles;
    // An empty document scores 0 for every query (sum of an empty max := 0, as in the oracle).
    if (lane < nq)
        for (int i = 0; i < MS_DOCS_PER_WG; ++i)
            if (s_len[i] == 0) a.scores[(size_t)(q0 + lane) * a.lds + d_begin + i] = 0.f;
    if (ntiles == 0) return;   // block-uniform

    // ---- tile ring, filled by LDS-DMA.  Tile k lives in slot k % MS_RING; a tile is 32 rows x 256 B, row r's 16-B chunks
    // XOR-swizzled by r & 15 (conflict-free ds_read_b128 of 16 rows x one chunk).  global_load_lds writes lane l of a wave to
    // (wave-uniform LDS base) + 16 l, so the swizzle is applied on the GLOBAL side: wave w fills rows 4w .. 4w+3, lane l sits
    // at chunk position l & 15 of row 4w + (l >> 4) and fetches the chunk whose swizzled position that is.  No staging
    // registers, no ds_write, and the loads of a whole group are in flight for a whole group of MFMA work:
    //     start of group g:  issue the DMA of group g + 1 (its slots held group g - 1, which every wave has left);
    //     end of group g:    __syncthreads() (= vmcnt(0) + barrier): group g + 1 has landed and is visible to all waves.
    // hipcc drains vmcnt before every LDS access it can see while a DMA is pending (it cannot tell the slots apart), which
    // would put the whole L2 / HBM latency back in front of every tile.  So inside a group NO LDS access is visible to it:
    // the group's tile metadata and the next group's token offsets are read before the DMAs are issued, and the A
    // fragments are fetched by ds_read_b128 written as inline asm, with their own s_waitcnt lgkmcnt(0).
    auto dma_tile = [&](int64_t tok0, int slot) {
        const int row = 4 * w + (lane >> 4), cpos = lane & 15;
        int64_t tok = tok0 + row;
        tok = tok < a.sumL ? tok : a.sumL - 1;   // rows past the end of the corpus: clamp (they are masked)
        const _Float16* g = a.Dtok + (size_t)tok * MS_DIM + ((cpos ^ (row & 15)) << 3);
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(&tile[slot][w * 1024]), 16, 0, 0);
    };
    {
        int64_t toks[MS_GROUP];
#pragma unroll
        for (int i = 0; i < MS_GROUP; ++i) toks[i] = s_tok[i < ntiles ? i : 0];
#pragma unroll
        for (int i = 0; i < MS_GROUP; ++i) if (i < ntiles) dma_tile(toks[i], i);
    }
    __syncthreads();
    const uint32_t tile_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&tile[0][0]);

    // A-fragment addresses: lane l reads rows (l & 15) and 16 + (l & 15), dims 32 ks + 8 (l >> 4) .. + 7 = chunk 4 ks + (l >> 4) of the
    // row at its swizzled position (16 lanes = 16 rows x one chunk each = 16 distinct positions: conflict-free).  The four per-lane
    // byte offsets are computed ONCE; the tile slot inside the group's half of the ring and the row block go into the instruction's
    // immediate offset, and the ring half is toggled once per group: the tile loop spends no vector instruction on addressing.
    uint32_t aaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aaddr[ks] = tile_lds + (lane & 15) * 256 + (((4 * ks + (lane >> 4)) ^ (lane & 15)) << 4);

    float run[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b) run[b] = -INFINITY;

    for (int k0 = 0; k0 < ntiles; k0 += MS_GROUP) {
      int metas[MS_GROUP];
      {
        int64_t toks[MS_GROUP];
#pragma unroll
        for (int i = 0; i < MS_GROUP; ++i) {
            metas[i] = __builtin_amdgcn_readfirstlane(s_meta[k0 + i < ntiles ? k0 + i : 0]);
            toks[i] = s_tok[k0 + MS_GROUP + i < ntiles ? k0 + MS_GROUP + i : 0];
        }
#pragma unroll
        for (int i = 0; i < MS_GROUP; ++i)
            if (k0 + MS_GROUP + i < ntiles) dma_tile(toks[i], ((k0 + MS_GROUP) & (MS_RING - 1)) + i);
      }
      // one tile of the group; GI (compile time) = its slot in the group's half of the ring
      auto tile_body = [&](auto GI) __attribute__((always_inline)) {
        constexpr int gi = decltype(GI)::value;
        const int meta = metas[gi];
        const int rows_valid = (meta >> 8) & 0xff;
        const bool last_tile_of_doc = (meta >> 16) & 1;
        // a wave without queries (the tail of the last query group) has nothing to multiply: it only keeps feeding the ring and
        // meeting the barriers
        if (!wave_has_queries) return;

        f16x8 af[2][4];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[rb][ks]) : "v"(aaddr[ks]), "n"(gi * MS_TILE_BYTES + rb * 4096));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]));

        // ---- 8 column blocks x (2 row blocks x 4 k-steps of v_mfma_f32_16x16x32_f16): the two row blocks' chains interleave (no
        //      MFMA waits for the one before it); the 8-way max of a column block is 4 v_max3 ------------------------------------
        auto chain = [&](int cb, f32x4& lo, f32x4& hi) __attribute__((always_inline)) {
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][0], bq[cb][0], (f32x4)0.f, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][0], bq[cb][0], (f32x4)0.f, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][ks], bq[cb][ks], lo, 0, 0, 0);
                hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][ks], bq[cb][ks], hi, 0, 0, 0);
            }
        };
        auto fold = [&](f32x4 lo, f32x4 hi, int cb) __attribute__((always_inline)) {
            if (rows_valid < 32) {   // wave-uniform: the last tile of a document.  Row of register r: 4 (lane >> 4) + r (+ 16)
                const int r0 = 4 * (lane >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r0 + r >= rows_valid) lo[r] = -INFINITY;
                    if (16 + r0 + r >= rows_valid) hi[r] = -INFINITY;
                }
            }
            float m = run[cb];
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, lo[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, hi[r]);
            run[cb] = m;
        };
        if (rows_valid <= 16) {
            // the last tile of a document with at most 16 tokens left: its second row block is all padding -- only the first one's chains
            // run (half the MFMAs of the tile; tiles are aligned to document starts, so on average 16 of a document's rows are padding)
#pragma unroll
            for (int cb = 0; cb < NCB; cb += 2) {
                if (cb < ncb) {
                    f32x4 aL = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][0], bq[cb][0], (f32x4)0.f, 0, 0, 0);
                    f32x4 bL = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][0], bq[cb + 1][0], (f32x4)0.f, 0, 0, 0);
#pragma unroll
                    for (int ks = 1; ks < 4; ++ks) {
                        aL = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][ks], bq[cb][ks], aL, 0, 0, 0);
                        bL = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][ks], bq[cb + 1][ks], bL, 0, 0, 0);
                    }
                    const int r0 = 4 * (lane >> 4);
                    float ma = run[cb], mb = run[cb + 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ma = fmaxf(ma, r0 + r < rows_valid ? aL[r] : -INFINITY);
                        mb = fmaxf(mb, r0 + r < rows_valid ? bL[r] : -INFINITY);
                    }
                    run[cb] = ma; run[cb + 1] = mb;
                }
            }
        } else if (ncb == NCB) {   // wave-uniform: the full load (every wave of every full query group)
            f32x4 aL, aH, bL, bH;
            chain(0, aL, aH);
            chain(1, bL, bH);
            fold(aL, aH, 0);
            chain(2, aL, aH);
            fold(bL, bH, 1);
            chain(3, bL, bH);
            fold(aL, aH, 2);
            chain(4, aL, aH);
            fold(bL, bH, 3);
            chain(5, bL, bH);
            fold(aL, aH, 4);
            chain(6, aL, aH);
            fold(bL, bH, 5);
            chain(7, bL, bH);
            fold(aL, aH, 6);
            fold(bL, bH, 7);
        } else {            // a partly loaded wave of the last query group: its live column blocks only (pairs: ncb is even)
#pragma unroll
            for (int cb = 0; cb < NCB; cb += 2) {
                if (cb < ncb) {
                    f32x4 aL, aH, bL, bH;
                    chain(cb, aL, aH);
                    chain(cb + 1, bL, bH);
                    fold(aL, aH, cb);
                    fold(bL, bH, cb + 1);
                }
            }
        }
        // ---- end of a document: finish the max over the four 16-lane rows and sum over the query tokens, transposing as we go:
        //      P(A, B) (permlane32 swap + max) leaves [A's rows (0|2), (1|3), B's (0|2), (1|3)], S(X, Y) (permlane16 swap + max) of two
        //      such registers leaves one finished column block per 16-lane row -- 8 registers -> 2, then one 16-lane sum per register
        //      and two more swap steps add the rows up.  28 VALU instead of 64 for the block-by-block form.
        if (last_tile_of_doc) {
            auto P = [&](float A, float B) __attribute__((always_inline)) -> float { swap32(A, B); return fmaxf(A, B); };
            auto S = [&](float X, float Y) __attribute__((always_inline)) -> float { swap16(X, Y); return fmaxf(X, Y); };
            float t0 = row16_sum(S(P(run[0], run[2]), P(run[1], run[3])));     // rows: column blocks 0, 1, 2, 3 (sums over their 16 tokens)
            float t1 = row16_sum(S(P(run[4], run[6]), P(run[5], run[7])));     // rows: column blocks 4, 5, 6, 7
#pragma unroll
            for (int b = 0; b < NCB; ++b) run[b] = -INFINITY;
            float o0 = t0, o1 = t1;
            swap16(t0, o0); swap16(t1, o1);
            t0 += o0; t1 += o1;                                                  // rows (0,1): blocks 0+1 | rows (2,3): blocks 2+3
            const int d = d_begin + (meta & 0xff);
            if (a.QB == 1) {        // Lq = 32: a query is two column blocks -- lanes 0 and 32 hold two queries per register
                if ((lane & 31) == 0) {
                    const int h = lane >> 5;
                    if (h < nq) a.scores[(size_t)(q0 + h) * a.lds + d] = t0;
                    if (2 + h < nq) a.scores[(size_t)(q0 + 2 + h) * a.lds + d] = t1;
                }
            } else {
                o0 = t0; o1 = t1;
                swap32(t0, o0); swap32(t1, o1);
                t0 += o0; t1 += o1;                                              // all four rows: (b0 + b1) + (b2 + b3)
                if (lane == 0) {
                    if (a.QB == 2) {                                             // Lq = 64: one query per register
                        a.scores[(size_t)q0 * a.lds + d] = t0;
                        if (nq > 1) a.scores[(size_t)(q0 + 1) * a.lds + d] = t1;
                    } else a.scores[(size_t)q0 * a.lds + d] = t0 + t1;           // Lq = 128: one query per wave
                }
            }
        }
      };
      if (k0 + 0 < ntiles) { tile_body(std::integral_constant<int, 0>{});
      if (k0 + 1 < ntiles) { tile_body(std::integral_constant<int, 1>{});
      if (k0 + 2 < ntiles) { tile_body(std::integral_constant<int, 2>{});
      if (k0 + 3 < ntiles) { tile_body(std::integral_constant<int, 3>{}); } } } }
      // the next group lives in the other half of the ring
      const uint32_t flip = (k0 & MS_GROUP) ? (uint32_t)(-MS_GROUP * MS_TILE_BYTES) : (uint32_t)(MS_GROUP * MS_TILE_BYTES);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) aaddr[ks] += flip;
      __syncthreads();   // group boundary: the next group's tiles have landed (vmcnt(0)) and everyone has left this group's
    }
}

}  // namespace fz

using namespace fz;

extern "C" int fz_maxsim_f16(const void* Qtok, const void* Dtok, const int64_t* Doff, int64_t sumL, int max_doc_len, int Q, int Lq, int N,
                             int dim, float* scores, int lds, void* stream) {
    if (Q < 0 || N < 0 || Lq <= 0 || lds < N) return FZ_ERR_ARG;
    if ((Q != 0 && N != 0) && (!Qtok || !Doff || !scores)) return FZ_ERR_ARG;   // empty tensors carry null pointers
    if (!Dtok && sumL != 0) return FZ_ERR_ARG;   // an empty token matrix (every document empty) has no pointer to give
    if (dim != MS_DIM) return FZ_ERR_UNSUPPORTED;
    if (Lq % 32 != 0 || (MS_BLOCKS_PER_WAVE % (Lq / 32)) != 0) return FZ_ERR_UNSUPPORTED;  // Lq in {32, 64, 128}
    if (((uintptr_t)Qtok % 16) || ((uintptr_t)Dtok % 16)) return FZ_ERR_UNSUPPORTED;
    if (Q == 0 || N == 0) return FZ_OK;
    hipStream_t st = as_stream(stream);
    if (sumL < 0 || max_doc_len <= 0) return FZ_ERR_ARG;
    if (max_doc_len > 32 * MS_TABLE) return FZ_ERR_UNSUPPORTED;   // one document must fit the tile table (16,384 tokens)
    if (sumL == 0) {  // every document empty: all scores 0
        FZ_HIP_TRY(hipMemset2DAsync(scores, (size_t)lds * 4, 0, (size_t)N * 4, (size_t)Q, st));
        return FZ_OK;
    }
    MaxSimArgs a{};
    a.Qtok = reinterpret_cast<const _Float16*>(Qtok);
    a.Dtok = reinterpret_cast<const _Float16*>(Dtok);
    a.Doff = Doff; a.scores = scores; a.lds = lds; a.Q = Q; a.Lq = Lq; a.N = N; a.sumL = sumL;
    a.QB = Lq / 32;
    const int blocks_per_wg = MS_WAVES * MS_BLOCKS_PER_WAVE;
    a.QG = (Q * a.QB + blocks_per_wg - 1) / blocks_per_wg;
    a.max_doc_len = max_doc_len;
    const int tiles_per_doc = (max_doc_len + 31) / 32;
    a.docs_per_wg = MS_TABLE / tiles_per_doc < MS_DOCS_PER_WG ? MS_TABLE / tiles_per_doc : MS_DOCS_PER_WG;
    a.DR = (N + a.docs_per_wg - 1) / a.docs_per_wg;
    const long nblk = 8L * a.QG * ((a.DR + 7) / 8);
    if (nblk > 0x7fffffffL) return FZ_ERR_UNSUPPORTED;
    maxsim_kernel<<<(unsigned)nblk, 512, 0, st>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
