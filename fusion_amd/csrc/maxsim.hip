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
// in 128 VGPRs for the whole kernel, and all 8 waves consume the same document tile, which is staged
// global -> LDS once per workgroup (16-B chunks XOR-swizzled by row so that ds_read_b128 of 32 rows is
// conflict-free) and double-buffered.  Tiles are aligned to document starts (rows past the end are masked
// to -inf).  Workgroups that share a 32-document range run back-to-back on one XCD, so the range (about
// 2.4 MB) is fetched from HBM once and served from that XCD's L2 to the other query groups.
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
    __shared__ __attribute__((aligned(16))) unsigned char tile[4][MS_TILE_BYTES];   // ring of 4 x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int x = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int qg = idx % a.QG;
    const int dr = (idx / a.QG) * 8 + x;
    if (dr >= a.DR) return;

    // ---- B fragments: this wave's 4 query blocks, all of K, resident for the whole kernel ----
    // lane l holds B[k = 8*(l>>5) + j][col = l&31] = Qtok[token l&31 of the block][dim 16*ks + 8*(l>>5) + j]
    const int blk0 = (qg * MS_WAVES + w) * MS_BLOCKS_PER_WAVE;   // global query-block index of this wave's first block
    const int nblk_total = a.Q * a.QB;
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
    // staging: 512 threads x one 16-B chunk = one 32-token tile; thread -> (row = tid/16, chunk = tid%16)
    const int srow = tid >> 4, schunk = tid & 15;
    const int swz = (srow * 256) + ((schunk ^ (srow & 15)) << 4);
    auto stage_load = [&](int64_t tok0) -> uint4 {
        int64_t t = tok0 + srow;
        t = t < a.sumL ? t : a.sumL - 1;   // rows past the end of the corpus: clamp (they are masked)
        return *reinterpret_cast<const uint4*>(a.Dtok + (size_t)t * MS_DIM + schunk * 8);
    };

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

    // ---- tile ring.  Tile k lives in slot k % 4.  Tile k+3 is fetched (global -> registers) during iteration k
    // and written to its slot at the start of iteration k+1, i.e. two tiles ahead of its use, so ONE workgroup
    // barrier every SECOND tile orders both the RAW (slot written -> read two iterations later) and the WAR
    // (slot read -> overwritten two iterations later) hazards.
    *reinterpret_cast<uint4*>(&tile[0][swz]) = stage_load(s_tok[0]);
    if (ntiles > 1) *reinterpret_cast<uint4*>(&tile[1][swz]) = stage_load(s_tok[1]);
    uint4 stage = make_uint4(0, 0, 0, 0);
    if (ntiles > 2) stage = stage_load(s_tok[2]);
    __syncthreads();

    float run[MS_BLOCKS_PER_WAVE];
#pragma unroll
    for (int b = 0; b < MS_BLOCKS_PER_WAVE; ++b) run[b] = -INFINITY;

    for (int k = 0; k < ntiles; ++k) {
        const int meta = __builtin_amdgcn_readfirstlane(s_meta[k]);
        const int rows_valid = (meta >> 8) & 0xff;
        const bool last_tile_of_doc = (meta >> 16) & 1;
        if (k + 2 < ntiles) *reinterpret_cast<uint4*>(&tile[(k + 2) & 3][swz]) = stage;   // tile k+2: read at iteration k+2
        if (k + 3 < ntiles) stage = stage_load(s_tok[k + 3]);

        // ---- A fragments from LDS: lane l -> row l&31, k = 16*ks + 8*(l>>5) .. +7 ------------
        f16x8 af[8];
        {
            const int r = lane & 31, h = lane >> 5;
            const unsigned char* base = &tile[k & 3][r * 256];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const f16x8*>(base + (((2 * ks + h) ^ (r & 15)) << 4));
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
        if (k & 1) __syncthreads();   // every second tile (see the ring invariant above)
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
