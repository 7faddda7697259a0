// score.hip -- K1: single-vector (DPR / SPLADE) scoring for gfx950.
//
// Reference: Ranker.single_vector_search (hybrid.py:77-106) -> util.semantic_search(..., score_function=util.cos_sim)
// (sentence-transformers 2.2.2; in-tree mirror splade/base.py:186-197): F.normalize both sides, torch.mm.
//
//   fz_normalize_rows_f32   Y = X / max(||X||, 1e-12), one wave per row, fp64 norm
//   fz_dot_scores_f32       S = Qn . Dn^T with v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate:
//                           the 1e-4 score contract rules out bf16/fp16 inputs, SURVEY 7 "hard parts")
//
// GEMM structure: 128x128 output tile per 256-thread workgroup (2x2 waves, each 64x64 = 2x2 MFMA tiles,
// 64 accumulator VGPRs), K-step 32, both operands K-contiguous ("NT").  Tiles are staged global -> registers
// -> LDS (double-buffered, rows padded to 36 floats so that ds_read_b128 of 16 consecutive rows is
// bank-conflict-free).  Within each group of 8 k's, lanes 0-31 take k 0..3 and lanes 32-63 take k 4..7 as one
// ds_read_b128 per operand tile; MFMA #kk then contracts {k=kk, k=4+kk}: the permutation is the same on both
// operands, so the sum is unchanged.  Workgroup -> tile mapping is XCD-aware: the QB query blocks of one
// corpus tile run back-to-back on one XCD (blockIdx % 8), so a corpus tile is fetched from HBM once and hit
// in that XCD's L2 afterwards.
#include <stdlib.h>

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
    int Q, N, d, QB, TN;
    int full;                  // block ids < full run whole 128x128 tiles, the others 128x64 halves
};

template <int BN, bool DB /* double-buffered LDS (2 workgroups/CU) vs single buffer (3/CU) */>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int row0, const int col0, float* lds) {
    constexpr int NI = BN / 64;          // MFMA tiles per wave along N (wave tile = 64 x BN/2)
    constexpr int BROWS = BN / 32;       // staging float4 per thread for the corpus tile
    // [buf][A: 128 rows | B: BN rows][LDT]
    constexpr int BUF = (BM + BN) * LDT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;

    // staging: thread -> (row = tid/8 + 32*i, k4 = tid%8)
    const int srow = tid >> 3, sk = (tid & 7) * 4;
    float4 ra[4], rb[BROWS];
    auto gload = [&](int kt) {
        const int k = kt * BK + sk;
        const bool kin = k < g.d;  // d % 4 == 0: a float4 is entirely inside or outside
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int qa = row0 + srow + 32 * i;
            ra[i] = (kin && qa < g.Q) ? *reinterpret_cast<const float4*>(g.A + (size_t)qa * g.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            const int nb = col0 + srow + 32 * i;
            rb[i] = (kin && nb < g.N) ? *reinterpret_cast<const float4*>(g.B + (size_t)nb * g.ldb + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
        float* As = lds + buf * BUF;
        float* Bs = As + BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(As + (srow + 32 * i) * LDT + sk) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(Bs + (srow + 32 * i) * LDT + sk) = rb[i];
    };

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int KT = (g.d + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    const int fr = lane & 31, fh = (lane >> 5) * 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = DB ? (kt & 1) : 0;
        if (kt + 1 < KT) gload(kt + 1);
        const float* As = lds + buf * BUF + (wr * 64 + fr) * LDT + fh;
        const float* Bs = lds + buf * BUF + BM * LDT + (wc * (BN / 2) + fr) * LDT + fh;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 a[2], b[NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const float4*>(As + mi * 32 * LDT + kg * 8);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const float4*>(Bs + ni * 32 * LDT + kg * 8);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const float av = kk == 0 ? a[mi].x : kk == 1 ? a[mi].y : kk == 2 ? a[mi].z : a[mi].w;
                        const float bv = kk == 0 ? b[ni].x : kk == 1 ? b[ni].y : kk == 2 ? b[ni].z : b[ni].w;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
                    }
        }
        if (DB) {
            if (kt + 1 < KT) sstore(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();                       // every wave is done reading the tile
            if (kt + 1 < KT) sstore(0);
            __syncthreads();
        }
    }

    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int c = col0 + wc * (BN / 2) + ni * 32 + (lane & 31);
            if (c < g.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = row0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (q < g.Q) g.C[(size_t)q * g.ldc + c] = acc[mi][ni][r];
                }
            }
        }
}

// Workgroup -> tile map.  XCD-aware: blocks b and b+8 share an XCD (round-robin dispatch), so the QB query blocks of
// one corpus tile are consecutive block ids on one XCD and the corpus tile is fetched from HBM once.  Blocks are
// dispatched in id order; the first `g.full` ids run 128x128 tiles, the rest are the LAST partial round of tiles
// cut into 128x64 halves, so that the tail occupies every CU for half a tile time instead of half the CUs for a
// whole one (1792 equal tiles on 512 resident slots otherwise cost 4 rounds for 3.5 rounds of work).
template <bool DB>
__global__ __launch_bounds__(256, DB ? 2 : 3) void dot_scores_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int b = blockIdx.x, half = -1;
    if (b >= g.full) { const int h = b - g.full; b = g.full + (h >> 1); half = h & 1; }
    const int x = b & 7, idx = b >> 3;
    const int qb = idx % g.QB;
    const int dt = (idx / g.QB) * 8 + x;     // 128-wide corpus tile
    if (dt >= g.TN) return;
    if (half < 0) gemm_tile<128, DB>(g, qb * BM, dt * 128, lds);
    else gemm_tile<64, DB>(g, qb * BM, dt * 128 + half * 64, lds);
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
    g.QB = (Q + BM - 1) / BM;
    g.TN = (N + 127) / 128;
    const long B = 8L * g.QB * ((g.TN + 7) / 8);        // block ids of whole tiles (incl. the XCD padding, which exits at once)
    const long slots = 512;                              // resident workgroups on 256 CUs (two double-buffered workgroups per CU)
    long R = B % slots;                                  // the partial last round ...
    if (R > slots / 2 || B < slots) R = 0;               // ... is only worth halving when it is at most half full
    g.full = (int)(B - R);
    const long nblk = (B - R) + 2 * R;
    if (nblk > 0x7fffffffL) return FZ_ERR_UNSUPPORTED;
    constexpr size_t lds_db = 2 * (BM + 128) * LDT * sizeof(float);
    static unsigned long long lds_set = 0ull;
    if (int rc = raise_lds_limit((const void*)dot_scores_kernel<true>, lds_db, lds_set)) return rc;
    dot_scores_kernel<true><<<(unsigned)nblk, 256, lds_db, as_stream(stream)>>>(g);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
