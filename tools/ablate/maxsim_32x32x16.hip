// maxsim.hip -- K2: exact ColBERT late interaction on gfx950.
//
// Reference: Ranker.multi_vector_search (hybrid.py:108-137) -> colbert-ai Searcher.search_all; the score
// PLAID approximates is   s(q,d) = sum_{i<Lq} max_{t in d} <Q[q][i], D[t]>    on 128-d
// L2-normalised token vectors (run_colbert.sh:26-27), query padded to 64 tokens (hybrid.py:129).
//
// Mapping: v_mfma_f32_32x32x16_f16 with A = 32 DOCUMENT tokens (rows), B = 32 QUERY tokens (cols), K = dim.
// The C layout puts the query token on the lane (col = lane&31) and the 32 document tokens in the 16
// accumulator registers x 2 lane halves, so
//     max over document tokens = in-lane max of 16 registers (+ one exchange between lane l and l^32),
//     sum over query tokens    = a wave reduction over 32 lanes, once per (query, document).
// A workgroup = 8 waves; every wave keeps the B fragments of 4 query blocks (2 queries x 64 tokens)
// in 128 VGPRs for the whole kernel, and all 8 waves consume the same document tile, which goes
// global -> LDS once per workgroup by LDS-DMA (16-B chunks XOR-swizzled by row so that ds_read_b128 of
// 32 rows is conflict-free), a group of four tiles ahead.  Tiles are aligned to document starts (rows past the end are masked
// to -inf).  Workgroups that share a 32-document range run back-to-back on one XCD, so the range (about
// 2.4 MB) is fetched from HBM once and served from that XCD's L2 to the other query groups.
//
// Where it stands (profiles/r02_pmc_maxsim.json, Q = 195): 9.1e8 MFMAs per launch (0.89 useful: query-block and tile
// padding), SQ_VALU_MFMA_BUSY_CYCLES = 64 % of the SIMD cycles at an effective clock of 1.99 GHz (GRBM_GUI_ACTIVE / 8 /
// time) -- 1.12-1.21 PFLOP/s = 45-48 % of the 2.5 PF nominal peak, 90-97 % of the 1,247 TFLOP/s MI355X_MICROARCH.md
// measures for a dense bf16 MFMA loop on random data (same busy fraction, same clock: the chip holds its clock down under
// matrix load).  Measured and without effect on the time: barrier every 2 / 4 / 8 tiles, register staging vs LDS-DMA,
// A fragments refilled in place under the last MFMA chain (slower: 0.39), waves 4-7 staggered by s_sleep 4..24.
#include <hip/hip_fp16.h>

#include "common.h"

namespace fz {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__global__ __launch_bounds__(512, 2) void maxsim_kernel(MaxSimArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[MS_RING][MS_TILE_BYTES];   // ring of 8 KiB tiles
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int qg = idx % a.QG;
    const int dr = (idx / a.QG) * 8 + x;
    if (dr >= a.DR) return;

    // ---- B fragments: this wave's 4 query blocks, all of K, resident for the whole kernel ----
    // lane l holds B[k = 8*(l>>5) + j][col = l&31] = Qtok[token l&31 of the block][dim 16*ks + 8*(l>>5) + j]
    const int blk0 = (qg * MS_WAVES + w) * MS_BLOCKS_PER_WAVE;   // global query-block index of this wave's first block
    const int nblk_total = a.Q * a.QB;
    const bool wave_has_queries = blk0 < nblk_total;   // wave-uniform
    f16x8 bq[MS_BLOCKS_PER_WAVE][8];
#pragma unroll
    for (int b = 0; b < MS_BLOCKS_PER_WAVE; ++b) {
        const int blk = blk0 + b;
        const bool okb = blk < nblk_total;
        const size_t tok = okb ? (size_t)blk * 32 + (lane & 31) : 0;   // blocks are consecutive 32-token slices of [Q*Lq]
        const _Float16* src = a.Qtok + tok * MS_DIM + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            f16x8 v = *reinterpret_cast<const f16x8*>(src + 16 * ks);
            if (!okb) v = (f16x8)(_Float16)0;
            bq[b][ks] = v;
        }
    }

    const int d_begin = dr * a.docs_per_wg;
    const int d_end = (d_begin + a.docs_per_wg < a.N) ? d_begin + a.docs_per_wg : a.N;

    // ---- tile walk: tiles are aligned to document starts -------------------------------
    // vals[b]: per-block sums already reduced over the wave; lane 0 writes one score per query.
    // An empty document scores 0 for every query (sum of an empty max := 0, as in the oracle).
    auto write_scores = [&](int d, const float* vals) {
        if (lane == 0) {
            const int qpw = MS_BLOCKS_PER_WAVE / a.QB;   // queries per wave
            for (int qi = 0; qi < qpw; ++qi) {
                const int q = blk0 / a.QB + qi;
                if (q < a.Q) {
                    float s = 0.f;
                    for (int b = 0; b < a.QB; ++b) s += vals[qi * a.QB + b];
                    a.scores[(size_t)q * a.lds + d] = s;
                }
            }
        }
    };
    const float zeros[MS_BLOCKS_PER_WAVE] = {0.f, 0.f, 0.f, 0.f};

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
    for (int i = 0; i < MS_DOCS_PER_WG; ++i)
        if (s_len[i] == 0) write_scores(d_begin + i, zeros);   // empty documents
    if (ntiles == 0) return;   // block-uniform

    // ---- tile ring, filled by LDS-DMA.  Tile k lives in slot k % MS_RING; a tile is 32 rows x 256 B, row r's 16-B chunks
    // XOR-swizzled by r & 15 (conflict-free ds_read_b128 of 32 rows).  global_load_lds writes lane l of a wave to
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

    float run[MS_BLOCKS_PER_WAVE];
#pragma unroll
    for (int b = 0; b < MS_BLOCKS_PER_WAVE; ++b) run[b] = -INFINITY;

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
#pragma unroll
      for (int gi = 0; gi < MS_GROUP; ++gi) {
        const int k = k0 + gi;
        if (k >= ntiles) break;
        const int meta = metas[gi];
        const int rows_valid = (meta >> 8) & 0xff;
        const bool last_tile_of_doc = (meta >> 16) & 1;
        // a wave whose four query blocks all lie past the last query (the tail of the last query group: at Q = 195, six of its
        // eight waves) has nothing to multiply: it only keeps feeding the ring and meeting the barriers
        if (!wave_has_queries) continue;

        // ---- A fragments from LDS: lane l -> row l&31, k = 16*ks + 8*(l>>5) .. +7 ------------
        f16x8 af[8];
        {
            // the lane id is re-derived here (two v_mbcnt) rather than kept: at 256 registers one long-lived value more is a
            // spill, and a spill's reload is a vmcnt(0) -- which would also wait for the DMAs in flight
            uint32_t l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            const uint32_t a_h = l >> 5, a_x = l & 15;
            const uint32_t base = tile_lds + (l & 31) * 256 + (uint32_t)(((k0 & (MS_RING - 1)) + gi) * MS_TILE_BYTES);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint32_t addr = base + (((2 * ks + a_h) ^ a_x) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(af[ks]) : "v"(addr));
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
        }
        // ---- 4 query blocks x 8 k-steps, two accumulators in ping-pong: the 16-way max of block b runs on the
        //      VALU while the MFMA chain of block b+1 occupies the matrix pipe ---------------------------------
        auto chain = [&](int b) -> f32x16 {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bq[b][ks], acc, 0, 0, 0);
            return acc;
        };
        auto fold = [&](f32x16 acc, int b) {
            if (rows_valid < 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {   // row of register r = (r&3) + 8*(r>>2) + 4*(lane>>5)
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row >= rows_valid) acc[r] = -INFINITY;
                }
            }
            float m = run[b];
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[r]);
            run[b] = m;
        };
        {
            f32x16 accA = chain(0);
            f32x16 accB = chain(1);
            fold(accA, 0);
            accA = chain(2);
            fold(accB, 1);
            accB = chain(3);
            fold(accA, 2);
            fold(accB, 3);
        }
        // ---- end of a document: finish max over the two lane halves, sum over query tokens
        if (last_tile_of_doc) {
            float sums[MS_BLOCKS_PER_WAVE];
#pragma unroll
            for (int b = 0; b < MS_BLOCKS_PER_WAVE; ++b) {
                // all on the VALU (DPP + permlane swaps): the ds_bpermute form put six LDS round trips per block on the
                // critical path of every document
                float m = run[b], other = m;
                swap32(m, other);                 // m = [lo, lo], other = [hi, hi]
                m = fmaxf(m, other);              // both halves now hold the column max
                m = row16_sum(m);                 // per 16 columns
                other = m;
                swap16(m, other);                 // rows (0,1) and (2,3) paired
                sums[b] = m + other;              // sum over the 32 columns, in every lane
                run[b] = -INFINITY;
            }
            write_scores(d_begin + (meta & 0xff), sums);
        }
      }
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
