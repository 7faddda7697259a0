// encoder.hip -- the per-sequence parts of the bi-encoder forward (hybrid.py:97-102: SentenceTransformer.encode of a
// BERT/CamemBERT backbone + mean Pooling), for PACKED (padding-free) token rows.
//
// The Linear layers stay hipBLASLt GEMMs (fp32 MFMA at ~86 % of peak already); what runs here is everything between
// them that the padded formulation pays for twice (padding tokens in every GEMM, a gather/scatter around attention):
//
//   fz_attn_varlen_f32     softmax(q k^T / sqrt(64)) v per (sequence, head) straight from the fused-QKV activations of
//                          packed rows: fp32 MFMA (16x16x4), online softmax over 16-key tiles (see the kernel's comment);
//   fz_add_layernorm_f32   LayerNorm(x + residual): one wave per row, row held in registers, one HBM pass.
//   fz_segment_mean_f32 / fz_segment_splade_max_f32    mean Pooling / SPLADE-max pooling over each sequence's rows.
#include "common.h"

namespace fz {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// Reductions across the four 16-lane groups of a wave on the VALU (common.h: swap16 / swap32) instead of two ds_bpermute
// round trips through the LDS pipe: every lane ends up with the result of its column.
__device__ __forceinline__ float xgroup_max(float x) {
    float y = x;
    swap16(x, y);
    x = y = fmaxf(x, y);
    swap32(x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float xgroup_sum(float x) {
    float y = x;
    swap16(x, y);
    x = y = x + y;
    swap32(x, y);
    return x + y;
}

struct AttnArgs {
    const void* qkv;    // [T][ld]: q | k | v, each H*64 wide; float32, or float16 in the QH instantiation
    int ld;
    const int4* strips;  // (first row of the sequence, its length L, first query of this strip of <= 16 * FZ_ATTN_NQ queries, unused)
    int n_strips;
    int H;
    float* out;  // [T][ldo]
    int ldo;
    float scale_log2e;   // scale * log2(e): softmax runs in the base-2 domain
    _Float16* out16;     // non-null: the context rows go out as float16 instead ([T][ldo] halves: the next Linear's operand in the mixed-precision forward)
};

#define FZ_ATTN_NQ 2   // 16-query sub-strips per wave; the strip table is cut every 16 * FZ_ATTN_NQ queries
#define ATT_LDT 68   // LDS row stride in floats: 64 + 4 -> the 16 lanes of one ds_read_b128 phase hit 64 distinct banks

// One wave = one (32-query strip = two 16-query sub-strips, head); the four waves of a workgroup take four consecutive strips of the table for ONE
// head -- consecutive strips mostly belong to one sequence, so its waves read the same K/V rows (L1 hits) -- and every wave
// is live whatever the sequence lengths.  No barriers: each wave turns its tiles through its own LDS slice.
//
//   * S^T = K Q^T: a query is a lane COLUMN, so the row max / row sum of the softmax are 3 in-register steps + two
//     cross-group shuffles, and the C registers of S^T are, unchanged, the B operand of O^T = V^T P^T (register i of lane
//     (q, g) holds key 4 g + i of the tile: contraction step i pairs the four groups' keys) -- probabilities never move;
//   * online softmax over 16-key tiles in the base-2 domain (v_exp_f32; one FMA rounding per element, the rounding of the
//     running maximum is common to a row and cancels in p / sum p);
//   * global -> MFMA layout: S^T needs lane = token, registers = dims, i.e. the transpose of the row-major activations
//     (a direct load would touch 32 cache lines per instruction).  Q/K tiles are read as whole 256-B head rows (4 rows per
//     wave-instruction) and turned through LDS; V needs lane = dim: direct dword loads.  All loads are bounds-checked
//     buffer loads whose range ends with the sequence: the rows a partial tile reaches past it read 0 without touching
//     memory (unchecked, 32-row tiles moved 1.4x the algorithmic bytes at the LLeQA length mix);
//   * v_mfma_f32_16x16x4_f32 rather than 32x32x2: tiles cut at 16 waste 24 % fewer MFMA cycles on padding at that mix,
//     and a wave needs 72 VGPRs per 16 queries instead of 128 per 32.
// Layouts (16x16x4): A lane l -> row l%16, k-slot l/16; B lane l -> col l%16, k-slot l/16; C reg r -> row 4(l/16)+r, col l%16.
//   S^T step i pairs dim 16*(l/16) + i of key row l%16 with the same dim of query l%16;
//   O^T step i takes register i of S^T (key 4(l/16) + i of the tile) against V[that key][4 (l%16) + t] (output tile t = dims 4 r + t).
// NQ = 16-query sub-strips per wave.  NQ = 2: every K/V tile a wave loads serves 32 queries -- half the re-reads of the
// sequence's K/V rows through the load path (the kernel's bound, see above) for 112 instead of 72 VGPRs.
// QH: the fused-QKV rows are float16 (the mixed-precision forward's Linear output): same lane -> element map, 8-byte loads, converted on the way
// into LDS / the V registers; the arithmetic is the float32 kernel's.
template <int NQ, bool QH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NQ == 1 ? 7 : 4, 8))) void attn_varlen_kernel(AttnArgs a) {
    constexpr int ES = QH ? 2 : 4;   // bytes per qkv element
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * ATT_LDT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int grp = blockIdx.x / a.H;
    const int h = blockIdx.x - grp * a.H;
    const int strip = grp * 4 + wave;
    if (strip >= a.n_strips) return;
    const int4 st = a.strips[strip];
    const int tok0 = st.x, L = st.y, q0 = st.z;
    if (q0 >= L) return;
    const int r = lane & 15, kg = lane >> 4;
    const int hid = a.H * 64;
    const int ldb = a.ld * ES;   // row pitch in bytes
    // one descriptor per wave: base = the sequence's first row, this head's q columns; range = the sequence
    const char* base = reinterpret_cast<const char*>(a.qkv) + ((size_t)tok0 * a.ld + h * 64) * ES;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (L * a.ld - h * 64) * ES, 0x00020000);

    float* my = lds + wave * (16 * ATT_LDT);
    const int voff_t = kg * ldb + r * (4 * ES);           // tile load: lane = (row kg of 4, elements 4 r .. 4 r + 3 of the head's 64) per instruction
    float* const wr = my + kg * ATT_LDT + r * 4;
    const float* const rd = my + r * ATT_LDT + kg * 16;
    const int voff_v = (4 * kg) * ldb + r * (4 * ES);

    auto load_tile = [&](int row0, int col_bytes, float (&f)[16]) {   // rows row0..row0+15 -> f[i] = [row r][16 kg + i]
        if constexpr (QH) {
            i32x2 raw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_t, (row0 + 4 * i) * ldb + col_bytes, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const h4v hv = __builtin_bit_cast(h4v, raw[i]);
                *reinterpret_cast<float4*>(wr + 4 * i * ATT_LDT) = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
            }
        } else {
            i32x4 raw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_t, (row0 + 4 * i) * ldb + col_bytes, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<i32x4*>(wr + 4 * i * ATT_LDT) = raw[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(rd + 4 * i);
            f[4 * i] = t.x; f[4 * i + 1] = t.y; f[4 * i + 2] = t.z; f[4 * i + 3] = t.w;
        }
    };

    const bool second = NQ == 2 && q0 + 16 < L;   // wave-uniform: the strip's second 16 queries exist
    float qf[NQ][16];
    float m[NQ], l[NQ];   // running max (base-2 domain) and sum of this lane's query
    f32x4 o[NQ][4];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (u == 0 || second) load_tile(q0 + 16 * u, 0, qf[u]);
        m[u] = -INFINITY; l[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int j0 = 0; j0 < L; j0 += 16) {
        float kf[16];
        f32x4 v[4];      // [step]; component t feeds dim tile t
        load_tile(j0, hid * ES, kf);
        // V: one 16-byte load per key and lane -- lane (r, kg) takes dims 4r..4r+3 of key 4 kg + i, i.e. output tile t holds the
        // dims 4 r + t (any assignment of dims to MFMA rows is as good as another; this one reads whole 256-B head rows)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if constexpr (QH) {
                const h4v hv = __builtin_bit_cast(h4v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff_v, (j0 + i) * ldb + 2 * hid * ES, 0));
                v[i] = f32x4{(float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w};
            } else {
                v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_v, (j0 + i) * ldb + 2 * hid * 4, 0));
            }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if (u == 1 && !second) break;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[u][i], s, 0, 0, 0);
            // s[g] = <q_{q0+16u+r}, k_j>, j = j0 + 4 kg + g.  Keys past the sequence (a last, partial tile: read as zeros) drop out -- as selects
            // in EVERY tile, not under `if (j0 + 16 > L)`: with the branch hipcc (ROCm 7.2) left the first reader of the MFMA chain's result,
            // the tile maximum below, at the head of the join block with no wait states behind the branch (tools/check_mfma_hazards.py).  The
            // maximum then saw the score registers before the last MFMA had written them: harmless (any shift works in a softmax; the
            // probabilities were computed later, from the finished registers) but different from run to run in the last bits.
            {
                const int left = L - (j0 + 4 * kg);
#pragma unroll
                for (int g = 0; g < 4; ++g) s[g] = g < left ? s[g] : -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = xgroup_max(mx);
            const float mnew = fmaxf(m[u], mx * a.scale_log2e);   // finite: key j0 < L is in this tile
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mnew);   // first tile: exp2(-inf) = 0
            float psum = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s[g] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[g], a.scale_log2e, -mnew));
                psum += s[g];
            }
            psum = xgroup_sum(psum);
            l[u] = l[u] * alpha + psum;
            m[u] = mnew;
            if (j0 > 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) o[u][t] *= alpha;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t) o[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i][t], s[i], o[u][t], 0, 0, 0);
        }
    }
    // o[u][t][g] = O[query 16 u + r][16 kg + 4 g + t] -> rows through the LDS slice -> whole 256-B rows out
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (u == 1 && !second) break;
        const float inv = 1.0f / l[u];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(my + r * ATT_LDT + 16 * kg + 4 * g) =
                make_float4(o[u][0][g] * inv, o[u][1][g] * inv, o[u][2][g] * inv, o[u][3][g] * inv);
        const size_t o0 = (size_t)(tok0 + q0 + 16 * u) * a.ldo + h * 64 + r * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * i + kg;
            const float4 t4 = *reinterpret_cast<const float4*>(my + row * ATT_LDT + r * 4);
            if (q0 + 16 * u + row < L) {
                if (a.out16) *reinterpret_cast<h4v*>(a.out16 + o0 + (size_t)row * a.ldo) = h4v{(_Float16)t4.x, (_Float16)t4.y, (_Float16)t4.z, (_Float16)t4.w};
                else *reinterpret_cast<float4*>(a.out + o0 + (size_t)row * a.ldo) = t4;
            }
        }
    }
}

// The attention of the MIXED-PRECISION forward with float16 matrix operands -- what torch.autocast does to a BERT layer (the reference's ColBERT:
// colbert-ai's Checkpoint runs query() / doc() under autocast): q k^T and p v are float16 matmuls with float32 accumulation, the softmax between them
// float32.  Same strips, same lane = query column softmax as attn_varlen_kernel, but on v_mfma_f32_16x16x32_f16 (S^T = K Q^T: two MFMAs per 16-key
// tile instead of sixteen 16x16x4_f32) and v_mfma_f32_16x16x16_f16 (O^T = V^T P^T: four instead of sixteen), and with NO LDS at all:
//   * the 16x16x32 operand layout -- lane l: row / column l % 16, k = 8 (l / 16) .. + 7 -- is a 16-byte piece of a head row: Q and K tiles are
//     loaded straight into MFMA registers (dims 8 kg .. 8 kg + 7, and the same + 32 for the second MFMA; both operands permute the contraction
//     index alike);
//   * the C registers of S^T (lane (r, kg): keys 4 kg + i of query r) are, rounded to float16, the B operand of the 16x16x16 MFMA -- probabilities
//     never move, as in the float32 kernel;
//   * V: lane (r, kg) loads dims 4 r .. 4 r + 3 of keys 4 kg + i (8 bytes each, whole 128-byte head rows per instruction); output tile t takes the
//     dims 4 r + t, so its A operand is element t of the four loads: a 4 x 4 transpose of halves inside the lane (8 v_perm per tile);
//   * O^T comes out with lane (r, kg) holding dims 16 kg .. 16 kg + 15 of query r: two 16-byte stores per lane, no transpose.
// Bounds-checked buffer loads as before (range = the sequence).  No branch between an MFMA and a reader of its result (tools/check_mfma_hazards.py).
template <int NQ>
__global__ __launch_bounds__(256) void attn_varlen_amp_kernel(AttnArgs a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int grp = blockIdx.x / a.H;
    const int h = blockIdx.x - grp * a.H;
    const int strip = grp * 4 + wave;
    if (strip >= a.n_strips) return;
    const int4 st = a.strips[strip];
    const int tok0 = st.x, L = st.y, q0 = st.z;
    if (q0 >= L) return;
    const int r = lane & 15, kg = lane >> 4;
    const int hid = a.H * 64;
    const int ldb = a.ld * 2;   // row pitch in bytes
    const char* base = reinterpret_cast<const char*>(a.qkv) + ((size_t)tok0 * a.ld + h * 64) * 2;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (L * a.ld - h * 64) * 2, 0x00020000);
    const int voff_qk = r * ldb + kg * 16;           // row r of the tile, dims 8 kg .. 8 kg + 7
    const int voff_v = (4 * kg) * ldb + r * 8;       // key 4 kg (+ i), dims 4 r .. 4 r + 3

    const bool second = NQ == 2 && q0 + 16 < L;      // wave-uniform: the strip's second 16 queries exist
    h8v qf[NQ][2];
    float m[NQ], l[NQ];
    f32x4 o[NQ][4];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            qf[u][hf] = (u == 0 || second) ? __builtin_bit_cast(h8v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_qk, (q0 + 16 * u) * ldb + hf * 64, 0)) : h8v{};
        m[u] = -INFINITY; l[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int j0 = 0; j0 < L; j0 += 16) {
        h8v kf[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) kf[hf] = __builtin_bit_cast(h8v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_qk, j0 * ldb + hid * 2 + hf * 64, 0));
        i32x2 vr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff_v, (j0 + i) * ldb + 2 * hid * 2, 0);
        // A operands of the four output tiles: element t of the four keys' loads
        h4v va[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t sel = (t & 1) ? 0x07060302u : 0x05040100u;
            const int w = t >> 1;
            const i32x2 x = {(int)__builtin_amdgcn_perm((uint32_t)vr[1][w], (uint32_t)vr[0][w], sel), (int)__builtin_amdgcn_perm((uint32_t)vr[3][w], (uint32_t)vr[2][w], sel)};
            va[t] = __builtin_bit_cast(h4v, x);
        }
        const int left = L - (j0 + 4 * kg);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if (u == 1 && !second) break;
            f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[u][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[u][1], s, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) s[g] = g < left ? s[g] : -INFINITY;     // keys past the sequence (read as zeros) drop out; a select, not a branch
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = xgroup_max(mx);
            const float mnew = fmaxf(m[u], mx * a.scale_log2e);
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mnew);                // first tile: exp2(-inf) = 0, and o is 0
            float psum = 0.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s[g] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[g], a.scale_log2e, -mnew));
                psum += s[g];
            }
            psum = xgroup_sum(psum);
            l[u] = l[u] * alpha + psum;
            m[u] = mnew;
            const h4v pb = {(_Float16)s[0], (_Float16)s[1], (_Float16)s[2], (_Float16)s[3]};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o[u][t] *= alpha;
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(va[t], pb, o[u][t], 0, 0, 0);
            }
        }
    }
    // o[u][t][g] = O[query 16 u + r][16 kg + 4 g + t]: 16 consecutive dims per lane
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (u == 1 && !second) break;
        const float inv = 1.0f / l[u];
        h8v lo, hi;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) { lo[4 * g + t] = (_Float16)(o[u][t][g] * inv); hi[4 * g + t] = (_Float16)(o[u][t][g + 2] * inv); }
        if (q0 + 16 * u + r < L) {
            _Float16* op = a.out16 + (size_t)(tok0 + q0 + 16 * u + r) * a.ldo + h * 64 + 16 * kg;
            *reinterpret_cast<h8v*>(op) = lo;
            *reinterpret_cast<h8v*>(op + 8) = hi;
        }
    }
}

// LayerNorm(x + res) * gamma + beta; one wave per row, VPL float4 per lane (d <= 256 * VPL).  XH: x is float16 (a mixed-precision Linear's
// output; the sum and the normalisation are float32 all the same); out16 non-null: the result is stored a second time as float16, the
// operand of the next Linear (the float32 copy stays the residual stream).
template <int VPL, bool XH>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const void* __restrict__ xv, int ldx, const float* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int rows, int d, float* __restrict__ out, int ldo, _Float16* __restrict__ out16, int ldo16) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = d >> 2;
    float4 v[VPL];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            typedef float f4v __attribute__((ext_vector_type(4)));   // both inputs are streamed once: non-temporal loads
            float4 t;
            if (XH) {
                const h4v tv = __builtin_nontemporal_load(reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(xv) + (size_t)row * ldx) + c);
                t = make_float4((float)tv.x, (float)tv.y, (float)tv.z, (float)tv.w);
            } else {
                const f4v tv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(xv) + (size_t)row * ldx) + c);
                t = make_float4(tv.x, tv.y, tv.z, tv.w);
            }
            if (res) {
                const f4v u = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(res + (size_t)row * ldr) + c);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[i] = t;
            sum += (t.x + t.y) + (t.z + t.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_reduce_sum(sum) / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_reduce_sum(sq) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
            float4 y;
            y.x = (v[i].x - mean) * rstd * g.x + b.x;
            y.y = (v[i].y - mean) * rstd * g.y + b.y;
            y.z = (v[i].z - mean) * rstd * g.z + b.z;
            y.w = (v[i].w - mean) * rstd * g.w + b.w;
            reinterpret_cast<float4*>(out + (size_t)row * ldo)[c] = y;
            if (out16) reinterpret_cast<h4v*>(out16 + (size_t)row * ldo16)[c] = h4v{(_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
        }
    }
}

// erf-GELU of the FFN activations (BERT / CamemBERT "gelu"): y = 0.5 x (1 + erf(x / sqrt 2)), the expression and float precision of
// torch.nn.functional.gelu.  The input is streamed exactly once (non-temporal loads); the output feeds the next GEMM (plain stores).
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(x) + i);
        float4 o;
        o.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752440f));
        o.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752440f));
        o.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752440f));
        o.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752440f));
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// The same on float16 activations (mixed-precision forward): float32 arithmetic, one rounding on the way out -- what torch.nn.functional.gelu
// does with a float16 tensor.  8 values per lane and step.
__global__ __launch_bounds__(256) void gelu_h_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const h8v v = __builtin_nontemporal_load(reinterpret_cast<const h8v*>(x) + i);
        h8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)v[j];
            o[j] = (_Float16)(0.5f * f * (1.0f + erff(f * 0.70710678118654752440f)));
        }
        reinterpret_cast<h8v*>(y)[i] = o;
    }
}

// Embedding sum + LayerNorm of a BERT/RoBERTa embedding block for packed rows: out[t] = LN(word[ids[t]] + pos[pos_ids[t]] + type0)
// (token type 0 everywhere: single-segment inputs).  One wave per row, one pass; VPL float4 per lane.
template <int VPL>
__global__ __launch_bounds__(256) void embed_layernorm_kernel(const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type0,
                                                              const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int rows,
                                                              int d, float* __restrict__ out, int ldo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = d >> 2;
    const float4* w = reinterpret_cast<const float4*>(word + (size_t)ids[row] * d);
    const float4* p = reinterpret_cast<const float4*>(pos + (size_t)pos_ids[row] * d);
    const float4* ty = reinterpret_cast<const float4*>(type0);
    float4 v[VPL];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float4 a = w[c], b = p[c], t = ty[c];
            float4 x;   // (word + type) + position: the order of the HF embedding modules
            x.x = (a.x + t.x) + b.x; x.y = (a.y + t.y) + b.y; x.z = (a.z + t.z) + b.z; x.w = (a.w + t.w) + b.w;
            v[i] = x;
            sum += (x.x + x.y) + (x.z + x.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_reduce_sum(sum) / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_reduce_sum(sq) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c], b = reinterpret_cast<const float4*>(beta)[c];
            float4 y;
            y.x = (v[i].x - mean) * rstd * g.x + b.x;
            y.y = (v[i].y - mean) * rstd * g.y + b.y;
            y.z = (v[i].z - mean) * rstd * g.z + b.z;
            y.w = (v[i].w - mean) * rstd * g.w + b.w;
            reinterpret_cast<float4*>(out + (size_t)row * ldo)[c] = y;
        }
    }
}

// Per-sequence column reductions over packed rows; one workgroup per (sequence, 1024-column slab).
//   MODE 0: out[b] = mean of rows [cu[b], cu[b+1])                     (sentence-transformers Pooling(mean))
//   MODE 1: out[b] = log1p(relu(max of the rows))  = max_t log1p(relu(x_t)), log1p o relu being monotone
//           (SPLADE-max, splade/splade.py:88-99); an empty sequence gives 0 in both modes.
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ cu, int d,
                                                             float* __restrict__ out, int ldo) {
    const int b = blockIdx.x;
    const int t0 = cu[b], t1 = cu[b + 1];
    const float inv = t1 > t0 ? 1.0f / (float)(t1 - t0) : 0.0f;
    const int c = (blockIdx.y * 256 + threadIdx.x) * VEC;
    if (c >= d) return;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = MODE == 0 ? 0.0f : -INFINITY;
    for (int t = t0; t < t1; ++t) {
        float u[VEC];
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(x + (size_t)t * ldx + c);
            u[0] = q.x; u[1 % VEC] = q.y; u[2 % VEC] = q.z; u[3 % VEC] = q.w;
        } else {
            u[0] = x[(size_t)t * ldx + c];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = MODE == 0 ? acc[i] + u[i] : fmaxf(acc[i], u[i]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[(size_t)b * ldo + c + i] = MODE == 0 ? acc[i] * inv : log1pf(fmaxf(acc[i], 0.0f));
}

}  // namespace fz

using namespace fz;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int fz_attn_varlen_f32(const float* qkv, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                                  float* out, int ldo, void* stream) {
    if (n_strips < 0 || H <= 0 || !(scale > 0.0f)) return FZ_ERR_ARG;
    if (n_strips == 0) return FZ_OK;
    if (!qkv || !strips || !out) return FZ_ERR_ARG;
    if (head_dim != 64) return FZ_ERR_UNSUPPORTED;
    if (ld < 3 * H * 64 || ldo < H * 64) return FZ_ERR_ARG;
    if ((ld & 3) || (ldo & 3) || !aligned16(qkv) || !aligned16(out) || !aligned16(strips)) return FZ_ERR_UNSUPPORTED;
    if ((long long)ld * 4 * 16384 > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;   // in-sequence byte offsets are 32-bit
    AttnArgs a{qkv, ld, reinterpret_cast<const int4*>(strips), n_strips, H, out, ldo, scale * 1.4426950408889634f, nullptr};
    const long long grid = (long long)((n_strips + 3) / 4) * H;
    if (grid > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    attn_varlen_kernel<FZ_ATTN_NQ, false><<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_attn_varlen_f16(const void* qkv, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                                 void* out, int ldo, void* stream) {
    if (n_strips < 0 || H <= 0 || !(scale > 0.0f)) return FZ_ERR_ARG;
    if (n_strips == 0) return FZ_OK;
    if (!qkv || !strips || !out) return FZ_ERR_ARG;
    if (head_dim != 64) return FZ_ERR_UNSUPPORTED;
    if (ld < 3 * H * 64 || ldo < H * 64) return FZ_ERR_ARG;
    if ((ld & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(qkv) & 7) || (reinterpret_cast<uintptr_t>(out) & 7) || !aligned16(strips)) return FZ_ERR_UNSUPPORTED;
    if ((long long)ld * 2 * 16384 > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    AttnArgs a{qkv, ld, reinterpret_cast<const int4*>(strips), n_strips, H, nullptr, ldo, scale * 1.4426950408889634f, reinterpret_cast<_Float16*>(out)};
    const long long grid = (long long)((n_strips + 3) / 4) * H;
    if (grid > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    attn_varlen_kernel<FZ_ATTN_NQ, true><<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_attn_varlen_f16_amp(const void* qkv, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                                     void* out, int ldo, void* stream) {
    if (n_strips < 0 || H <= 0 || !(scale > 0.0f)) return FZ_ERR_ARG;
    if (n_strips == 0) return FZ_OK;
    if (!qkv || !strips || !out) return FZ_ERR_ARG;
    if (head_dim != 64) return FZ_ERR_UNSUPPORTED;
    if (ld < 3 * H * 64 || ldo < H * 64) return FZ_ERR_ARG;
    if ((ld & 7) || (ldo & 7) || !aligned16(qkv) || !aligned16(out) || !aligned16(strips)) return FZ_ERR_UNSUPPORTED;   // 16-byte pieces of the rows
    if ((long long)ld * 2 * 16384 > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    AttnArgs a{qkv, ld, reinterpret_cast<const int4*>(strips), n_strips, H, nullptr, ldo, scale * 1.4426950408889634f, reinterpret_cast<_Float16*>(out)};
    const long long grid = (long long)((n_strips + 3) / 4) * H;
    if (grid > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    attn_varlen_amp_kernel<FZ_ATTN_NQ><<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_add_layernorm_f32(const float* x, int ldx, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                                    int rows, int d, float* out, int ldo, void* stream) {
    if (rows < 0 || d <= 0) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    if (!x || !gamma || !beta || !out) return FZ_ERR_ARG;
    if (ldx < d || ldo < d || (res && ldr < d)) return FZ_ERR_ARG;
    if ((d & 3) || (ldx & 3) || (ldo & 3) || (res && (ldr & 3)) || d > 4096 || !aligned16(x) || !aligned16(out) || !aligned16(gamma) ||
        !aligned16(beta) || (res && !aligned16(res)))
        return FZ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = as_stream(stream);
    if (d <= 1024)
        add_layernorm_kernel<4, false><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo, nullptr, 0);
    else
        add_layernorm_kernel<16, false><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo, nullptr, 0);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_add_layernorm_x16(const void* x, int ldx, const float* res, int ldr, const float* gamma, const float* beta, float eps,
                                    int rows, int d, float* out, int ldo, void* out16v, int ldo16, void* stream) {
    _Float16* out16 = reinterpret_cast<_Float16*>(out16v);
    if (rows < 0 || d <= 0) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    if (!x || !gamma || !beta || !out) return FZ_ERR_ARG;
    if (ldx < d || ldo < d || (res && ldr < d) || (out16 && ldo16 < d)) return FZ_ERR_ARG;
    if ((d & 3) || (ldx & 3) || (ldo & 3) || (res && (ldr & 3)) || (out16 && (ldo16 & 3)) || d > 4096 || (reinterpret_cast<uintptr_t>(x) & 7) ||
        !aligned16(out) || !aligned16(gamma) || !aligned16(beta) || (res && !aligned16(res)) || (out16 && (reinterpret_cast<uintptr_t>(out16) & 7)))
        return FZ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = as_stream(stream);
    if (d <= 1024)
        add_layernorm_kernel<4, true><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo, out16, ldo16);
    else
        add_layernorm_kernel<16, true><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo, out16, ldo16);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_gelu_f32(const float* x, float* y, size_t count, void* stream) {
    if (count == 0) return FZ_OK;
    if (!x || !y) return FZ_ERR_ARG;
    if ((count & 3) || !aligned16(x) || !aligned16(y)) return FZ_ERR_UNSUPPORTED;
    const size_t n4 = count >> 2;
    const size_t blocks = (n4 + 255) / 256;
    gelu_kernel<<<(unsigned)(blocks < 256 * 32 ? blocks : 256 * 32), 256, 0, as_stream(stream)>>>(x, y, n4);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_gelu_f16(const void* xv, void* yv, size_t count, void* stream) {
    const _Float16* x = reinterpret_cast<const _Float16*>(xv);
    _Float16* y = reinterpret_cast<_Float16*>(yv);
    if (count == 0) return FZ_OK;
    if (!x || !y) return FZ_ERR_ARG;
    if ((count & 7) || !aligned16(x) || !aligned16(y)) return FZ_ERR_UNSUPPORTED;
    const size_t n8 = count >> 3;
    const size_t blocks = (n8 + 255) / 256;
    gelu_h_kernel<<<(unsigned)(blocks < 256 * 32 ? blocks : 256 * 32), 256, 0, as_stream(stream)>>>(x, y, n8);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_embed_layernorm_f32(const float* word, const float* pos, const float* type0, const int64_t* ids, const int64_t* pos_ids,
                                      const float* gamma, const float* beta, float eps, int rows, int d, float* out, int ldo, void* stream) {
    if (rows < 0 || d <= 0) return FZ_ERR_ARG;
    if (rows == 0) return FZ_OK;
    if (!word || !pos || !type0 || !ids || !pos_ids || !gamma || !beta || !out || ldo < d) return FZ_ERR_ARG;
    if ((d & 3) || (ldo & 3) || d > 4096 || !aligned16(word) || !aligned16(pos) || !aligned16(type0) || !aligned16(gamma) || !aligned16(beta) ||
        !aligned16(out))
        return FZ_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t s = as_stream(stream);
    if (d <= 1024)
        embed_layernorm_kernel<4><<<grid, 256, 0, s>>>(word, pos, type0, ids, pos_ids, gamma, beta, eps, rows, d, out, ldo);
    else
        embed_layernorm_kernel<16><<<grid, 256, 0, s>>>(word, pos, type0, ids, pos_ids, gamma, beta, eps, rows, d, out, ldo);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

template <int MODE>
static int launch_segment_reduce(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream) {
    if (B < 0 || d <= 0) return FZ_ERR_ARG;
    if (B == 0) return FZ_OK;
    if (!x || !cu_rows || !out || ldx < d || ldo < d) return FZ_ERR_ARG;
    const bool vec = !(d & 3) && !(ldx & 3) && aligned16(x);
    const int per = vec ? 1024 : 256;
    dim3 grid((unsigned)B, (unsigned)((d + per - 1) / per));
    if (vec)
        segment_reduce_kernel<MODE, 4><<<grid, 256, 0, as_stream(stream)>>>(x, ldx, cu_rows, d, out, ldo);
    else
        segment_reduce_kernel<MODE, 1><<<grid, 256, 0, as_stream(stream)>>>(x, ldx, cu_rows, d, out, ldo);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}

extern "C" int fz_segment_mean_f32(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream) {
    return launch_segment_reduce<0>(x, ldx, cu_rows, B, d, out, ldo, stream);
}

extern "C" int fz_segment_splade_max_f32(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream) {
    return launch_segment_reduce<1>(x, ldx, cu_rows, B, d, out, ldo, stream);
}
