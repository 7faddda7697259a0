// score.hip -- K1: single-vector (DPR / SPLADE) scoring for gfx950.
//
// Reference: Ranker.single_vector_search (hybrid.py:77-106) -> util.semantic_search(..., score_function=util.cos_sim)
// (sentence-transformers 2.2.2; in-tree mirror splade/base.py:186-197): F.normalize both sides, torch.mm.
//
//   fz_normalize_rows_f32   Y = X / max(||X||, 1e-12), one wave per row, fp64 norm
//   fz_dot_scores_f32       S = Qn . Dn^T with v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate:
//                           the 1e-4 score contract rules out bf16/fp16 inputs, SURVEY 7 "hard parts")
//
// GEMM structure: 128x128 output tile per 256-thread workgroup (2x2 waves, each 64x64 = 2x2 MFMA tiles, 64 accumulator
// VGPRs), K-step 32, both operands K-contiguous ("NT").  The grid is persistent (two workgroups per CU) and each workgroup
// runs its tiles as ONE stream of k-tiles: HBM -> registers two k-tiles ahead, registers -> LDS one k-tile ahead (rows padded
// to 36 floats so that ds_read_b128 of 16 consecutive rows is bank-conflict-free), fragments LDS -> registers one group of 8
// k's ahead of the MFMAs.  Within each group of 8 k's, lanes 0-31 take k 0..3 and lanes 32-63 take k 4..7 as one ds_read_b128
// per operand tile; MFMA #kk then contracts {k=kk, k=4+kk}: the permutation is the same on both operands, so the sum is
// unchanged.  The k-loop carries no vector ALU work besides one 64-bit pointer bump per load and pair of k-tiles (a kernel
// instantiation without the ragged-d masks serves d % 64 == 0).  Workgroup -> tile mapping is XCD-aware: the QB query
// blocks of one corpus tile run back-to-back on one XCD (workgroup id % 8), so a corpus tile is fetched from HBM once and hit
// in that XCD's L2 afterwards.  What each of these steps bought is in DESIGN.md section 5 (K1).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace fz {

// ---- normalisation ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ X, int rows, int d, int ldx,
                                                             float* __restrict__ Y, int ldy, int vec) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (size_t)row * ldx;
    float* y = Y + (size_t)row * ldy;
    double ss = 0.0;
    if (vec) {
        for (int k = lane * 4; k < d; k += 256) {
            float4 f = *reinterpret_cast<const float4*>(x + k);
            ss += (double)f.x * (double)f.x; ss += (double)f.y * (double)f.y;
            ss += (double)f.z * (double)f.z; ss += (double)f.w * (double)f.w;
        }
    } else {
        for (int k = lane; k < d; k += 64) ss += (double)x[k] * (double)x[k];
    }
    ss = wave_reduce_sum(ss);
    float nrm = (float)sqrt(ss);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;
    if (vec) {
        for (int k = lane * 4; k < d; k += 256) {
            float4 f = *reinterpret_cast<const float4*>(x + k);
            *reinterpret_cast<float4*>(y + k) = make_float4(f.x / nrm, f.y / nrm, f.z / nrm, f.w / nrm);
        }
    } else {
        for (int k = lane; k < d; k += 64) y[k] = x[k] / nrm;
    }
}

// ---- fp32 MFMA GEMM --------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 32, LDT = BK + 4;  // LDS row = 36 floats = 144 B; BN (128 or 64) is a template parameter

struct GemmArgs {
    const float* A; int lda;   // queries  [Q][lda]
    const float* B; int ldb;   // corpus   [N][ldb]
    float* C; int ldc;         // scores   [Q][ldc]
    int Q, N, d, QB, TN;       // QB: query blocks of 128 rows run as whole-height tiles
    int full, halves;          // tile ids < full are whole 128x128 tiles; then `halves` 128x64 half tiles (at most one per workgroup)
    int tail_ids, tail_row0;   // ids of the tiles of a last query block of 1..96 rows (0: none), which start at row tail_row0 ...
    int tail_rows;             // ... and are 32, 64 or 96 rows high
    // straggler rows (round 5): the last 1..4 query rows beyond a multiple of 32 ride INSIDE the 64-row tail tiles as a 4-row SLAB (see
    // gemm_stream's SLAB form); Q above counts the rows of the tiles only
    int slab_row0, slab_rows;  // first straggler row, how many (0: none)
    // FILTER form (fz_dot_scores_filter_f32): no score plane; what beats a query's threshold goes to its candidate list
    const float* tau;          // [QB * 128]: tau[q] for q < Q, +inf beyond
    float* cand_s;             // [Q][cap]
    int64_t* cand_i;           // [Q][cap]
    int32_t* cand_len;         // [Q] number of candidates so far (may run past cap: the excess is dropped and *overflow set)
    int32_t* overflow;
    int cap;
    int64_t id_base;           // document id of corpus row 0
    // SPLADE form (fz_splade_head_max_f32): A = packed hidden rows [T][d], B = vocabulary projection [V][d]; no logits plane
    const float* bias;         // [N] decoder bias
    const int32_t* cu_rows;    // [nseq + 1] first packed row of every sequence
    int nseq;
    float* pool; int ldp;      // [nseq][ldp] zero-initialised: pool[s][v] = max over the sequence's rows of log1p(relu(logit))
};

enum { EPI_STORE = 0, EPI_FILTER = 1, EPI_SPLADE = 2 };

// One workgroup's share of the tiles: ids first, first + step, ... below `end`, all of one shape (128 x BN).  The k-tiles of
// ALL those tiles form one stream through the software pipeline: the operand loads of a tile's first two k-tiles are issued
// during the last two k-tiles of the tile before it, and its score stores drain under the next tile's MFMAs -- a workgroup
// pays the pipeline fill once per launch, not once per tile.
template <int BN, int MI /* 32-row MFMA blocks per wave */, int WN /* waves along the corpus side: 2 (2 x 2 waves) or 4 (1 x 4) */, bool RAGGED /* d is not a whole number of k-tile pairs */, int EPI,
          bool FLAT = false /* tile id b = corpus columns [b BN, (b + 1) BN) of the row band at row_origin (the 7-row-block cover) */,
          int NSLAB = 0 /* the batch's last 1..8 rows ride along as NSLAB 4-row slabs on v_mfma_f32_4x4x1_16b_f32 (64-row tail tiles only) */>
__device__ __forceinline__ void gemm_stream(const GemmArgs& g, float* lds, int first, const int end, const int step, const int qblocks, const int row_origin) {
    constexpr int WM = 4 / WN;           // waves along the query side
    constexpr int BMT = 32 * MI * WM;    // tile height: 128 (whole query blocks: 2 x 2 waves, MI = 2), 64 (MI = 1) or 32 (1 x 4 waves, MI = 1)
    constexpr int AROWS = BMT / 32;      // staging float4 per thread for the query tile
    constexpr int NI = BN / (32 * WN);   // 32x32 MFMA tiles per wave along N (wave tile = 32 MI x BN / WN)
    constexpr int BROWS = BN / 32;       // staging float4 per thread for the corpus tile
    // Straggler rows as SLABS.  A batch a few rows over a multiple of 32 -- the LLeQA test split is 195 = 6 x 32 + 3, the dev
    // split 201 = 6 x 32 + 9 -- paid a whole 32-row MFMA block for them.  v_mfma_f32_4x4x1_16b_f32 multiplies 16 independent 4 x 4 blocks with
    // K = 1: with the SAME four query rows in every block's A operand and 64 different corpus columns in the B operands one instruction adds one k
    // to 4 rows x 64 columns -- an eighth of a row block's work.  The stragglers ride inside the 64-row tail tiles (2 x 2 waves, wave tile 32 rows
    // x 64 columns): their (up to 8) query rows are staged behind the corpus tile, the waves of the upper row pair read the corpus tile a second
    // time as "lane = column" with every fragment group and issue that group's slab MFMAs BETWEEN its 32x32x2's (a dependent chain of 4x4x1's on its
    // own stalls on every link: as a block behind the tile's MFMAs a slab cost a third of a k-tile instead of an eighth) in the tile stream's own
    // k order -- 0, 4, 1, 5, 2, 6, 3, 7: lanes 0-31 of a 32x32x2 hold k, lanes 32-63 k + 4 -- so that a straggler's score is the same chain of
    // fused multiply-adds, bit for bit.
    constexpr bool SLAB = NSLAB > 0;
    static_assert(!SLAB || (MI == 1 && WN == 2 && BN == 128 && EPI == EPI_STORE && !FLAT && NSLAB <= 2), "slabs ride in the 64-row tail tiles of the plain GEMM");
    constexpr int SROWS = 4 * NSLAB;     // staged straggler rows
    // [stage][A: 128 rows | B: BN rows | straggler rows][LDT]
    constexpr int BUF = (BMT + BN + SROWS) * LDT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WN, wc = w % WN;

    // tile id -> (query row, corpus column) of its corner; ids of the XCD padding decode to nothing
    auto decode = [&](int b, int& row0, int& col0) -> bool {
        if constexpr (FLAT) { row0 = row_origin; col0 = b * BN; return col0 < g.N; }
        int half = 0;
        if (BN == 64) {   // workgroup i of the half-tile round: tile (i/16)*8 + i%8 (same XCD as i), half (i/8)%2
            const int h = b - g.full;
            b = g.full + ((h >> 4) << 3 | (h & 7));
            half = (h >> 3) & 1;
        }
        const int x = b & 7, idx = b >> 3;
        const int dt = (idx / qblocks) * 8 + x;     // 128-wide corpus tile
        row0 = row_origin + (idx % qblocks) * BMT;
        col0 = dt * 128 + half * 64;
        return dt < g.TN;
    };
    auto next_tile = [&](int b, int& row0, int& col0) -> int {   // first decodable id >= b of this workgroup's sequence, or -1
        for (; b < end; b += step)
            if (decode(b, row0, col0)) return b;
        return -1;
    };
    int crow, ccol, lrow, lcol;
    int cur_b = next_tile(first, crow, ccol);
    if (cur_b < 0) return;

    // staging: thread -> (row = tid/8 + 32*i, k4 = tid%8).  Rows past the end are clamped to the last one: their products are
    // computed and never stored.  Global loads carry no predicate and sit under no branch (after a conditional load hipcc waits
    // for ALL outstanding loads at the next use of any staging register); a k-tile past d re-reads the row's last float4 and is
    // zeroed on its way into LDS.  The k-tile count is rounded up to even so that the two register stages and the two LDS stages
    // keep their roles from one tile to the next.
    const int srow = tid >> 3, sk = (tid & 7) * 4;
    // addresses = wave-uniform tile base (SGPRs, advanced per k-tile on the scalar unit) + a 32-bit per-lane offset that is fixed
    // for the whole tile: the k-loop spends no vector instruction on addressing
    const float* abase;
    const float* bbase;
    int32_t oa[AROWS], ob[BROWS];
    [[maybe_unused]] const float* const sbase = g.A + (size_t)g.slab_row0 * g.lda;                      // the straggler rows: the same for every tile
    [[maybe_unused]] const int32_t os = SLAB ? (min(srow, max(g.slab_rows, 1) - 1) * g.lda + sk) * 4 : 0;   // (threads of rows >= SROWS load a copy and store nothing)
    auto point_at = [&](int row0, int col0) {
        row0 = min(row0, g.Q - 1);         // (a half tile may start past the last corpus row)
        col0 = min(col0, g.N - 1);
        abase = g.A + (size_t)row0 * g.lda;
        bbase = g.B + (size_t)col0 * g.ldb;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) oa[i] = (min(srow + 32 * i, g.Q - 1 - row0) * g.lda + sk) * 4;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) ob[i] = (min(srow + 32 * i, g.N - 1 - col0) * g.ldb + sk) * 4;
    };
    const int KT = ((g.d + 2 * BK - 1) / (2 * BK)) * 2;
    struct Stage { float4 a[AROWS], b[BROWS]; float4 s; };
    // `edge` (compile time): this k-tile may reach past d -- only the last two of a tile can; all the others take the plain path
    auto gload = [&](Stage& r, int kt, auto edge) __attribute__((always_inline)) {
        const char* ab = reinterpret_cast<const char*>(abase + kt * BK);
        const char* bb = reinterpret_cast<const char*>(bbase + kt * BK);
        // d % 4 == 0: a float4 is entirely inside or outside.  Outside: step back to the row's last float4 (zeroed in sstore)
        const int32_t back = (RAGGED && decltype(edge)::value) ? max(0, kt * BK + sk + 4 - g.d) * 4 : 0;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) r.a[i] = *reinterpret_cast<const float4*>(ab + (ptrdiff_t)(oa[i] - back));
#pragma unroll
        for (int i = 0; i < BROWS; ++i) r.b[i] = *reinterpret_cast<const float4*>(bb + (ptrdiff_t)(ob[i] - back));
        if constexpr (SLAB) r.s = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sbase + kt * BK) + (ptrdiff_t)(os - back));
    };
    auto sstore = [&](const Stage& r, int kt, auto edge) __attribute__((always_inline)) {   // kt: the k-tile the registers hold
        float* As = lds + (kt & 1) * BUF;
        float* Bs = As + BMT * LDT;
        const uint32_t m = RAGGED && kt * BK + sk >= g.d ? 0u : ~0u;
        auto keep = [&](float4 v) {
            if (!(RAGGED && decltype(edge)::value)) return v;
            return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                               __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
        };
#pragma unroll
        for (int i = 0; i < AROWS; ++i) *reinterpret_cast<float4*>(As + (srow + 32 * i) * LDT + sk) = keep(r.a[i]);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(Bs + (srow + 32 * i) * LDT + sk) = keep(r.b[i]);
        if constexpr (SLAB) { if (srow < SROWS) *reinterpret_cast<float4*>(Bs + (BN + srow) * LDT + sk) = keep(r.s); }
    };

    f32x16 acc[MI][NI];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    [[maybe_unused]] f32x4 sacc[SLAB ? NSLAB : 1];
    auto clear = [&]() {
        if constexpr (SLAB) {
#pragma unroll
            for (int r = 0; r < NSLAB; ++r) sacc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    };

    // fragments of one group of 8 k's: lanes 0-31 hold k 0..3, lanes 32-63 k 4..7 (see the header)
    struct Frag { float4 a[MI], b[NI]; float4 sb[SLAB ? 2 : 1], sa[SLAB ? NSLAB : 1][2]; };   // (SLAB: the group's corpus rows as lane = column, its slab rows)
    const int fr = lane & 31, fh = (lane >> 5) * 4;
    const int aoff = (wr * (32 * MI) + fr) * LDT + fh, boff = BMT * LDT + (wc * (BN / WN) + fr) * LDT + fh;
    auto fread = [&](Frag& f, int buf, int kg) __attribute__((always_inline)) {
        const float* As = lds + buf * BUF + aoff + kg * 8;
        const float* Bs = lds + buf * BUF + boff + kg * 8;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) f.a[mi] = *reinterpret_cast<const float4*>(As + mi * 32 * LDT);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) f.b[ni] = *reinterpret_cast<const float4*>(Bs + ni * 32 * LDT);
        if constexpr (SLAB) {
            if (wr == 0) {                                                   // (wave-uniform: the upper row pair's waves carry the slabs of their 64 columns)
                const float* Bt = lds + buf * BUF + BMT * LDT;
                const float* bl = Bt + (wc * 64 + lane) * LDT + kg * 8;      // lane = corpus column: its row of the tile
                f.sb[0] = *reinterpret_cast<const float4*>(bl); f.sb[1] = *reinterpret_cast<const float4*>(bl + 4);
#pragma unroll
                for (int r = 0; r < NSLAB; ++r) {
                    const float* al = Bt + (BN + 4 * r + (lane & 3)) * LDT + kg * 8;   // lane & 3 = row inside the slab
                    f.sa[r][0] = *reinterpret_cast<const float4*>(al); f.sa[r][1] = *reinterpret_cast<const float4*>(al + 4);
                }
            }
        }
    };
    auto fmma = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float av = kk == 0 ? f.a[mi].x : kk == 1 ? f.a[mi].y : kk == 2 ? f.a[mi].z : f.a[mi].w;
                    const float bv = kk == 0 ? f.b[ni].x : kk == 1 ? f.b[ni].y : kk == 2 ? f.b[ni].z : f.b[ni].w;
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
                    if constexpr (SLAB) {
                        if (wr == 0 && ni < NSLAB) {                         // slab ni's two k's of this step (k, then k + 4) behind a 32x32x2: its chain never waits
                            const float4 a0 = f.sa[ni][0], a1 = f.sa[ni][1], b0 = f.sb[0], b1 = f.sb[1];
                            const float x0 = kk == 0 ? a0.x : kk == 1 ? a0.y : kk == 2 ? a0.z : a0.w, y0 = kk == 0 ? b0.x : kk == 1 ? b0.y : kk == 2 ? b0.z : b0.w;
                            const float x1 = kk == 0 ? a1.x : kk == 1 ? a1.y : kk == 2 ? a1.z : a1.w, y1 = kk == 0 ? b1.x : kk == 1 ? b1.y : kk == 2 ? b1.z : b1.w;
                            sacc[ni] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, y0, sacc[ni], 0, 0, 0);
                            sacc[ni] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, y1, sacc[ni], 0, 0, 0);
                        }
                    }
                }
    };

    // One k-tile of the pipeline = four fragment groups.  HBM -> registers runs TWO k-tiles ahead (two register stages: operand
    // lines that miss in L2 are missed by every workgroup of the XCD at once, so nobody is left to cover a late load), registers
    // -> LDS one k-tile ahead (two LDS stages).  The one barrier of a k-tile sits BEFORE its last group of MFMAs: by then every
    // wave has read all its fragments of the current stage into registers and written its part of the next one, so right after
    // the barrier it fetches the next stage's first fragments and covers that latency with the MFMAs it still owes.
    Frag f0, f1;
    Stage r0, r1;
    auto ktile = [&](const int kt /* being multiplied */, const int kload /* fetched into rl */, const int kstore /* held by rs */, Stage& rl, const Stage& rs, auto edge) __attribute__((always_inline)) {
        const int cur = kt & 1;
        gload(rl, kload, edge);
        fread(f1, cur, 1);
        fmma(f0);
        fread(f0, cur, 2);
        fmma(f1);
        // ONE fence per k-tile, here: without it hipcc sinks the HBM loads to the end of the k-tile and hoists the waits for them
        // to its start (a zero-deep prefetch); with more of them it can no longer run the fragment reads ahead of the MFMAs.
        __builtin_amdgcn_sched_barrier(0);
        sstore(rs, kstore, edge);            // k-tile kt + 1, or the next tile's 0: into the stage read one k-tile ago, whose reads
        fread(f1, cur, 3);                   // all ended before the previous barrier
        fmma(f0);
        __syncthreads();
        fread(f0, cur ^ 1, 0);
        fmma(f1);
    };

    constexpr std::true_type EDGE{};
    constexpr std::false_type PLAIN{};
    point_at(crow, ccol);
    gload(r0, 0, EDGE);
    gload(r1, 1, EDGE);
    sstore(r0, 0, EDGE);
    __syncthreads();
    fread(f0, 0, 0);
    while (true) {
        clear();
        int kt = 0;
        for (; kt + 4 < KT; kt += 2) {
            ktile(kt, kt + 2, kt + 1, r0, r1, PLAIN);
            ktile(kt + 1, kt + 3, kt + 2, r1, r0, PLAIN);
        }
        if (kt + 2 < KT) {                   // loads and stores k-tiles KT-2 and KT-1: the ones a ragged d reaches into
            ktile(kt, kt + 2, kt + 1, r0, r1, EDGE);
            ktile(kt + 1, kt + 3, kt + 2, r1, r0, EDGE);
        }
        // the last two k-tiles fetch the first two of the next tile (after the last tile: of the same one again, unused)
        const int nxt = next_tile(cur_b + step, lrow, lcol);
        if (nxt >= 0) point_at(lrow, lcol);
        ktile(KT - 2, 0, KT - 1, r0, r1, EDGE);
        ktile(KT - 1, 1, 0, r1, r0, EDGE);
        // The epilogues read the accumulators behind branches.  hipcc pads 'MFMA write -> read' inside a basic block, but across a branch it
        // has been seen to leave a reader too close (tools/check_mfma_hazards.py: attn_varlen_kernel, and 11-17 of the 18 wait states a
        // 16-pass MFMA needs on some epilogue paths here): one full-length wait per TILE (80 of ~40,000 cycles) makes every path safe.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
        __builtin_amdgcn_sched_barrier(0);

        // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        if constexpr (EPI == EPI_FILTER) {
            // Threshold filter in the epilogue: a score enters its query's candidate list iff it beats tau[q] (or is NaN) -- the
            // score plane is never written.  One accumulator register of the wave = 32 documents x 2 queries (lane halves), a "step".
            // Pass A counts per step and half by ballot and parks the counts in lane `step` of one register; then the lanes reserve
            // their steps' slots with ONE round of atomics (64 steps in parallel); pass B recomputes the ballots and stores.
            // Candidates of one (query, step) are in document order; ACROSS steps, waves and workgroups the order is arbitrary:
            // fz_topk_fold_f32(unordered) re-establishes "ties by ascending id".
            constexpr int STEPS = MI * NI * 16;
            const int h = lane >> 5;
            const int qw = crow + wr * (32 * MI) + 4 * h;               // + mi*32 + (r&3) + 8*(r>>2)
            const int dw = ccol + wc * (BN / WN) + (lane & 31);         // + ni*32
            bool over = false;
            // `edge` (compile time): the tile reaches past the last query or document and every score is bounds-checked; inside the
            // matrix -- the usual tile -- a score costs ONE compare per pass (the two passes are ~3 % of the kernel)
            auto passes = [&](auto edge) __attribute__((always_inline)) {
                auto beats = [&](int mi, int ni, int r, const float4& t4) __attribute__((always_inline)) -> bool {
                    const float tq = (r & 3) == 0 ? t4.x : (r & 3) == 1 ? t4.y : (r & 3) == 2 ? t4.z : t4.w;
                    const bool b = !(acc[mi][ni][r] <= tq);
                    if constexpr (decltype(edge)::value) return b && dw + ni * 32 < g.N && qw + mi * 32 + (r & 3) + 8 * (r >> 2) < g.Q;
                    else return b;
                };
                uint32_t counts = 0u;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float4 t4 = *reinterpret_cast<const float4*>(g.tau + qw + mi * 32 + 8 * rg);   // this lane half's thresholds, 4 rows
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const int r = rg * 4 + rr;
                                const unsigned long long bal = __ballot(beats(mi, ni, r, t4));
                                const uint32_t c = (uint32_t)__popc((uint32_t)bal) | ((uint32_t)__popc((uint32_t)(bal >> 32)) << 16);
                                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(counts) : "s"(c), "i"((mi * NI + ni) * 16 + r));   // c is wave-uniform
                            }
                    }
                // lane s owns step s = (mi*NI + ni)*16 + r: queries qa (lanes 0-31 of the step) and qa + 4 (lanes 32-63)
                int base_lo = 0, base_hi = 0;
                if (lane < STEPS) {
                    const int r = lane & 15, mi = lane / (16 * NI);
                    const int qa = crow + wr * (32 * MI) + mi * 32 + (r & 3) + 8 * (r >> 2);
                    const int c_lo = counts & 0xffff, c_hi = counts >> 16;
                    if (c_lo) base_lo = atomicAdd(&g.cand_len[qa], c_lo);
                    if (c_hi) base_hi = atomicAdd(&g.cand_len[qa + 4], c_hi);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float4 t4 = *reinterpret_cast<const float4*>(g.tau + qw + mi * 32 + 8 * rg);   // (again: 32 registers are not to spare)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const int r = rg * 4 + rr, step = (mi * NI + ni) * 16 + r;
                                const bool keep = beats(mi, ni, r, t4);
                                const unsigned long long bal = __ballot(keep);
                                if (bal == 0ull) continue;                     // wave-uniform
                                const int b_lo = __builtin_amdgcn_readlane(base_lo, step), b_hi = __builtin_amdgcn_readlane(base_hi, step);
                                const uint32_t mine = h ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                                const int pos = (h ? b_hi : b_lo) + __popc(mine & ((1u << (lane & 31)) - 1u));
                                if (keep) {
                                    const int q = qw + mi * 32 + (r & 3) + 8 * (r >> 2);
                                    if (pos < g.cap) {
                                        g.cand_s[(size_t)q * g.cap + pos] = acc[mi][ni][r];
                                        g.cand_i[(size_t)q * g.cap + pos] = g.id_base + dw + ni * 32;
                                    } else over = true;
                                }
                            }
                    }
            };
            // (rows past Q carry tau = +inf, but a NaN score beats even that: the last query block keeps its row check)
            if (ccol + BN <= g.N && crow + BMT <= g.Q) passes(std::false_type{});   // workgroup-uniform
            else passes(std::true_type{});
            if (over) atomicExch(g.overflow, 1);
        } else if constexpr (EPI == EPI_SPLADE) {
            // SPLADE-max pooling as the epilogue of the vocabulary projection (splade/splade.py:88-99: amax over the tokens of
            // log1p(relu(logits))): the [T, V] logits are never written.  log1p o relu is monotone, so the max over a sequence's rows is
            // taken on the raw dot products, the bias added and the transform applied ONCE per (sequence, column), and the result --
            // a non-negative float, whose bit pattern orders like an unsigned integer -- goes into the zero-initialised pool[s][v] by
            // atomicMax.  A wave's 64 rows are cut at the sequence boundaries (cu_rows: packed rows, sequences back to back); per piece:
            // an in-lane max over the accumulator registers whose row lies in the piece, one permlane32 swap to join the lane halves,
            // and one atomic per column from lanes 0-31.
            const int R0 = crow + wr * (32 * MI);
            const int Rend = min(R0 + 32 * MI, g.Q);
            if (R0 < Rend) {      // wave-uniform
                int lo_s = 0, hi_s = g.nseq;             // first sequence that ends after R0 (scalar binary search)
                while (lo_s < hi_s) { const int mid = (lo_s + hi_s) >> 1; if (g.cu_rows[mid + 1] > R0) hi_s = mid; else lo_s = mid + 1; }
                const int h4 = 4 * (lane >> 5);
                for (int sq = lo_s; sq < g.nseq; ++sq) {
                    const int a0 = g.cu_rows[sq], a1 = g.cu_rows[sq + 1];
                    if (a0 >= Rend) break;
                    const int lo = max(a0, R0) - R0, hi = min(a1, Rend) - R0;     // piece [lo, hi) of the wave's rows
                    if (lo >= hi) continue;                                           // (an empty sequence)
                    float m[NI];
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) m[ni] = -INFINITY;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + h4;
                            const bool in = row >= lo && row < hi;
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) m[ni] = fmaxf(m[ni], in ? acc[mi][ni][r] : -INFINITY);
                        }
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        float other = m[ni];
                        swap32(m[ni], other);                 // m = [lo half, lo half], other = [hi half, hi half]
                        const float best = fmaxf(m[ni], other);
                        const int c = ccol + wc * (BN / WN) + ni * 32 + (lane & 31);
                        if (lane < 32 && c < g.N) {
                            const float logit = best + g.bias[c];
                            const float v = log1pf(fmaxf(logit, 0.0f));
                            if (v > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(g.pool + (size_t)sq * g.ldp + c), __float_as_uint(v));
                        }
                    }
                }
            }
        } else {
        if constexpr (SLAB) {
            // C/D of the 4x4x1: register v = row v of the block, lane = 4 * block + column -> lane l holds column l of the wave's 64
            const int c = ccol + wc * 64 + lane;
            if (wr == 0 && c < g.N) {
#pragma unroll
                for (int r = 0; r < NSLAB; ++r)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (4 * r + v < g.slab_rows) g.C[(size_t)(g.slab_row0 + 4 * r + v) * g.ldc + c] = sacc[r][v];
            }
        }
        const bool whole = crow + BMT <= g.Q && ccol + BN <= g.N;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int c = ccol + wc * (BN / WN) + ni * 32 + (lane & 31);
                const int q0 = crow + wr * (32 * MI) + mi * 32 + 4 * (lane >> 5);
                float* cp = g.C + (size_t)q0 * g.ldc + c;
                if (whole) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mi][ni][r];
                } else if (c < g.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (q0 + (r & 3) + 8 * (r >> 2) < g.Q) cp[(size_t)((r & 3) + 8 * (r >> 2)) * g.ldc] = acc[mi][ni][r];
                }
            }
        }
        if (nxt < 0) break;
        cur_b = nxt; crow = lrow; ccol = lcol;
    }
}

// Workgroup -> tile map.  The grid is PERSISTENT: two workgroups per CU (LDS-bound), workgroup i takes tile ids i, i + grid, ...
// XCD-aware: workgroups i and i+8 share an XCD (round-robin dispatch) and grid % 8 == 0, so the QB query blocks of one corpus tile
// are consecutive ids on one XCD, run at about the same time, and the corpus tile is fetched from HBM once.  The first `g.full`
// ids are 128x128 tiles; the rest is the LAST partial round cut into 128x64 halves, so that the tail occupies every CU for half a
// tile time instead of half the CUs for a whole one (1792 equal tiles on 512 resident slots otherwise cost 4 rounds for 3.5 of work).
// Query batches that are not a multiple of 128 rows: a last block of up to 96 rows runs as tiles of its own height -- 32 or 96 rows
// (1 x 4 waves, one or three MFMA row blocks each) or 64 rows (2 x 2 waves, one row block each) -- whose ids follow the whole tiles in every
// workgroup's sequence: 195 queries are 128 + 96 rows of MFMA work instead of 256, 64 or fewer half of what a 128-row tile costs.  Every
// shape accumulates a score as the same chain of fused multiply-adds in the same k order: a query's scores do not depend on the tile
// its row fell into.
template <bool RAGGED, int EPI>
__global__ __launch_bounds__(256, 2) void dot_scores_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b0 = (int)blockIdx.x, G = (int)gridDim.x;
    gemm_stream<128, 2, 2, RAGGED, EPI>(g, lds, b0, g.full, G, g.QB, 0);
    if (b0 < g.halves) {
        __syncthreads();                     // the half tile restarts the pipeline in stage 0
        gemm_stream<64, 2, 2, RAGGED, EPI>(g, lds, g.full + b0, g.full + b0 + 1, 1, g.QB, 0);
    }
    if constexpr (EPI != EPI_SPLADE) {
        if (g.tail_ids > 0) {                // this workgroup's id sequence b0, b0 + G, ... continues through the last query block's tiles
            int t = b0;
            if (t < g.full) t += ((g.full - t + G - 1) / G) * G;
            t -= g.full;
            if (t < g.tail_ids) {            // (workgroup-uniform)
                __syncthreads();
                if (g.tail_rows == 32) gemm_stream<128, 1, 4, RAGGED, EPI>(g, lds, t, g.tail_ids, G, 1, g.tail_row0);
                else if (g.tail_rows == 64) {
                    if constexpr (EPI == EPI_STORE) {
                        if (g.slab_rows > 0) gemm_stream<128, 1, 2, RAGGED, EPI, false, 1>(g, lds, t, g.tail_ids, G, 1, g.tail_row0);
                        else gemm_stream<128, 1, 2, RAGGED, EPI>(g, lds, t, g.tail_ids, G, 1, g.tail_row0);
                    } else gemm_stream<128, 1, 2, RAGGED, EPI>(g, lds, t, g.tail_ids, G, 1, g.tail_row0);
                }
                else gemm_stream<128, 3, 4, RAGGED, EPI>(g, lds, t, g.tail_ids, G, 1, g.tail_row0);
            }
        }
    }
}

// 193..224 queries (the LLeQA test and dev batches: 195, 201) against a corpus that fits the chip in one round: 7 row blocks x ceil(N / 32) column
// blocks of MFMA work -- 23.9 block pairs per CU at N = 27,942 -- which no single tile shape spreads evenly (128- and 96-row tiles of 128 columns:
// 219 + 219 tiles on 256 CUs, the busiest CU holds two whole tiles).  Two shapes of 24 blocks each do: rows 0..127 as 128 x 192 tiles (2 x 2 waves,
// 2 x 3 MFMA tiles each), rows 128.. as 96 x 256 tiles (1 x 4 waves, 3 x 2 each): 146 + 110 = 256 equal workgroups, one per CU (92 / 101 KB of LDS).
// Same k-order fmaf chain per score as every other shape.
template <bool RAGGED>
__global__ __launch_bounds__(256, 1) void dot_scores_cover7_kernel(GemmArgs g, int nP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x;
    if (b < nP) gemm_stream<192, 2, 2, RAGGED, EPI_STORE, true>(g, lds, b, b + 1, 1, 1, 0);
    else gemm_stream<256, 3, 4, RAGGED, EPI_STORE, true>(g, lds, b - nP, b - nP + 1, 1, 1, 128);
}

}  // namespace fz

using namespace fz;

extern "C" int fz_normalize_rows_f32(const float* X, int rows, int d, int ldx, float* Y, int ldy, void* stream) {
    if (rows < 0 || d <= 0 || ldx < d || ldy < d) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;                   // empty tensors carry null pointers
    if (!X || !Y) return FZ_ERR_ARG;
    const int vec = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0);
    normalize_rows_kernel<<<(rows + 3) / 4, 256, 0, as_stream(stream)>>>(X, rows, d, ldx, Y, ldy, vec);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

static int launch_gemm(GemmArgs& g, int epi, hipStream_t st) {
    // the last 1..4 rows over a multiple of 32 ride as ONE slab inside the 64-row tail tiles: Q % 128 in 65 .. 68 (195: the LLeQA test batch).
    // Measured (profiles/r05_gemm_slabs_ab.json): Q = 195 0.0907 -> 0.0867 ms (0.587 -> 0.614 of the fp32-MFMA peak), Q = 193 0.0888 -> 0.0852; two
    // slabs (Q = 197 .. 200) bring nothing over the 7-row-block cover (0.629 vs 0.636), three (Q = 201) lose: those batches keep the cover.
    static const bool slabs_on = [] { const char* e = getenv("FZ_GEMM_SLABS"); return !(e && e[0] == '0'); }();   // (A/B runs)
    g.slab_rows = 0; g.slab_row0 = 0;
    if (epi == EPI_STORE && slabs_on && g.Q % BM > 64 && g.Q % BM <= 68) {
        g.slab_rows = g.Q % BM - 64;
        g.slab_row0 = g.Q - g.slab_rows;
        g.Q = g.slab_row0;                                 // the tiles see the rows before the stragglers only
    }
    // rows -> whole 128-row query blocks + a last block in its own height class (32, 64 or 96 rows; more than 96: a padded whole block)
    int nfull = g.Q / BM;
    const int rem = g.Q % BM;
    int tail = 0;
    if (epi == EPI_SPLADE || rem > 96) nfull += rem > 0;
    else if (rem > 0) tail = (rem + 31) / 32 * 32;
    g.QB = nfull;
    g.TN = (g.N + 127) / 128;
    const long TN8 = 8L * ((g.TN + 7) / 8);             // corpus-tile ids incl. the XCD padding, which decodes to nothing
    const long B = TN8 * nfull;                           // ids of whole tiles
    g.tail_ids = tail ? (int)TN8 : 0; g.tail_row0 = nfull * BM; g.tail_rows = tail;
    const long Bt = g.tail_ids;
    static int cus[64];
    int dev = 0;
    FZ_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 64) return FZ_ERR_UNSUPPORTED;
    if (!cus[dev]) FZ_HIP_TRY(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    // per-lane offsets are signed 32-bit byte offsets inside one operand tile (at most 256 rows)
    if (256.0 * g.lda * 4 >= 2147483648.0 || 256.0 * g.ldb * 4 >= 2147483648.0) return FZ_ERR_UNSUPPORTED;
    if (epi == EPI_STORE && g.Q > 192 && g.Q <= 224) {   // the 7-row-block cover, when its tiles fit the chip in one round
        const long nP = (g.N + 191) / 192, nQ = (g.N + 255) / 256;
        if (nP + nQ <= cus[dev]) {
            constexpr size_t lds_c7 = 2 * (96 + 256) * LDT * sizeof(float);
            static unsigned long long c7_set[2] = {0ull, 0ull};
            const bool rg = g.d % (2 * BK) != 0;
            if (rg) {
                if (int rc = raise_lds_limit((const void*)dot_scores_cover7_kernel<true>, lds_c7, c7_set[1])) return rc;
                dot_scores_cover7_kernel<true><<<(unsigned)(nP + nQ), 256, lds_c7, st>>>(g, (int)nP);
            } else {
                if (int rc = raise_lds_limit((const void*)dot_scores_cover7_kernel<false>, lds_c7, c7_set[0])) return rc;
                dot_scores_cover7_kernel<false><<<(unsigned)(nP + nQ), 256, lds_c7, st>>>(g, (int)nP);
            }
            FZ_LAUNCH_CHECK();
            return FZ_OK;
        }
    }
    const long slots = 2L * cus[dev] / 8 * 8;            // resident workgroups (two per CU), a multiple of the 8 XCDs
    if (slots <= 0 || B + Bt > 0x3fffffffL) return FZ_ERR_UNSUPPORTED;
    long R = B % slots;                                  // the partial last round ...
    if (R > slots / 2 || B < slots || Bt > 0) R = 0;     // ... is only worth halving when it is at most half full (and nothing follows it)
    g.full = (int)(B - R);
    g.halves = (int)(2 * R);
    const long ids = B + Bt;
    const long nblk = ids < slots ? ids : slots;
    if (nblk <= 0) return FZ_OK;
    constexpr size_t lds_db = 2 * (BM + 128) * LDT * sizeof(float);
    // per-lane offsets are signed 32-bit byte offsets inside one 128-row operand tile
    if (128.0 * g.lda * 4 >= 2147483648.0 || 128.0 * g.ldb * 4 >= 2147483648.0) return FZ_ERR_UNSUPPORTED;
    static unsigned long long lds_set[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    const bool ragged = g.d % (2 * BK) != 0;
#define FZ_GEMM_LAUNCH(RG, EP)                                                                                                   \
    {                                                                                                                            \
        if (int rc = raise_lds_limit((const void*)dot_scores_kernel<RG, EP>, lds_db, lds_set[3 * RG + EP])) return rc;           \
        dot_scores_kernel<RG, EP><<<(unsigned)nblk, 256, lds_db, st>>>(g);                                                       \
    }
    if (ragged && epi == EPI_FILTER) FZ_GEMM_LAUNCH(true, EPI_FILTER)
    else if (ragged && epi == EPI_SPLADE) FZ_GEMM_LAUNCH(true, EPI_SPLADE)
    else if (ragged) FZ_GEMM_LAUNCH(true, EPI_STORE)
    else if (epi == EPI_FILTER) FZ_GEMM_LAUNCH(false, EPI_FILTER)
    else if (epi == EPI_SPLADE) FZ_GEMM_LAUNCH(false, EPI_SPLADE)
    else FZ_GEMM_LAUNCH(false, EPI_STORE)
#undef FZ_GEMM_LAUNCH
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_dot_scores_f32(const float* Qn, int ldq, const float* Dn, int ldd, int Q, int N, int d, float* scores, int lds,
                                 void* stream) {
    if (Q < 0 || N < 0 || d <= 0 || ldq < d || ldd < d || lds < N) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;           // empty tensors carry null pointers
    if (!Qn || !Dn || !scores) return FZ_ERR_ARG;
    // 16-byte vector staging: the Python binding pads embeddings to a multiple of 4 floats
    if ((d % 4) || (ldq % 4) || (ldd % 4) || ((uintptr_t)Qn % 16) || ((uintptr_t)Dn % 16)) return FZ_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = Qn; g.lda = ldq; g.B = Dn; g.ldb = ldd; g.C = scores; g.ldc = lds;
    g.Q = Q; g.N = N; g.d = d;
    return launch_gemm(g, EPI_STORE, as_stream(stream));
}

extern "C" int fz_dot_scores_filter_f32(const float* Qn, int ldq, const float* Dn, int ldd, int Q, int N, int d, int64_t id_base,
                                        const float* tau_padded, float* cand_scores, int64_t* cand_ids, int32_t* cand_len, int cap,
                                        int32_t* overflow, void* stream) {
    if (Q < 0 || N < 0 || d <= 0 || ldq < d || ldd < d || cap <= 0) return FZ_ERR_ARG;
    if (Q == 0 || N == 0) return FZ_OK;
    if (!Qn || !Dn || !tau_padded || !cand_scores || !cand_ids || !cand_len || !overflow) return FZ_ERR_ARG;
    if ((d % 4) || (ldq % 4) || (ldd % 4) || ((uintptr_t)Qn % 16) || ((uintptr_t)Dn % 16) || ((uintptr_t)tau_padded % 16)) return FZ_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = Qn; g.lda = ldq; g.B = Dn; g.ldb = ldd;
    g.Q = Q; g.N = N; g.d = d;
    g.tau = tau_padded; g.cand_s = cand_scores; g.cand_i = cand_ids; g.cand_len = cand_len; g.cap = cap; g.id_base = id_base; g.overflow = overflow;
    return launch_gemm(g, EPI_FILTER, as_stream(stream));
}

// SPLADE head: pool[s][v] = max over the rows t of sequence s of log1p(relu(<X[t], W[v]> + bias[v])), the logits never materialised.
extern "C" int fz_splade_head_max_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const int32_t* cu_rows, int nseq, int T,
                                      int V, int d, float* pool, int ldp, void* stream) {
    if (T < 0 || V < 0 || nseq < 0 || d <= 0 || ldx < d || ldw < d || ldp < V) return FZ_ERR_ARG;
    if (T == 0 || V == 0 || nseq == 0) return FZ_OK;
    if (!X || !W || !bias || !cu_rows || !pool) return FZ_ERR_ARG;
    if ((d % 4) || (ldx % 4) || (ldw % 4) || ((uintptr_t)X % 16) || ((uintptr_t)W % 16)) return FZ_ERR_UNSUPPORTED;
    GemmArgs g{};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw;
    g.Q = T; g.N = V; g.d = d;
    g.bias = bias; g.cu_rows = cu_rows; g.nseq = nseq; g.pool = pool; g.ldp = ldp;
    return launch_gemm(g, EPI_SPLADE, as_stream(stream));
}
