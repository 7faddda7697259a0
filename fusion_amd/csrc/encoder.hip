// encoder.hip -- the per-sequence parts of the bi-encoder forward (hybrid.py:97-102: SentenceTransformer.encode of a
// BERT/CamemBERT backbone + mean Pooling), for PACKED (padding-free) token rows.
//
// The Linear layers stay hipBLASLt GEMMs (fp32 MFMA at ~86 % of peak already); what runs here is everything between
// them that the padded formulation pays for twice (padding tokens in every GEMM, a gather/scatter around attention):
//
//   fz_attn_varlen_f32     softmax(q k^T / sqrt(64)) v per (sequence, head) straight from the fused-QKV activations of
//                          packed rows, fp32 MFMA, flash-style online softmax over 32-key tiles, no LDS:
//                            S^T = K Q^T   so that a query is a LANE: row max / row sum are 16 in-register steps + one
//                                          cross-half shuffle instead of 80 shuffles;
//                            O^T = V^T P^T the C-layout registers of S^T ARE the B operand of this product (register r of
//                                          lane (q, half) holds key 8(r/4) + 4 half + r%4: contraction step r pairs the
//                                          two halves' keys), so probabilities never move.
//   fz_add_layernorm_f32   LayerNorm(x + residual): one wave per row, row held in registers, one HBM pass.
//   fz_segment_mean_f32    mean Pooling over each sequence's rows.
#include "common.h"

namespace fz {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct AttnArgs {
    const float* qkv;   // [T][ld]: q | k | v, each H*64 wide
    int ld;
    const int4* blocks;  // (first row of the sequence, its length L, first query of this block of <= 64 queries, unused)
    int n_blocks;
    int H;
    float* out;  // [T][ldo]
    int ldo;
    float scale_log2e;   // scale * log2(e): softmax runs in the base-2 domain
};

#define ATT_LDT 68   // LDS row stride in floats: 64 + 4 -> the 16 lanes of one ds_read_b128 phase hit 64 distinct banks

// One wave = one (32-query strip, head); a workgroup = the two strips of one 64-query block x two heads (the strips share
// their K/V rows through L1/L2; a strip past the end of the sequence exits at once).  No barriers: every wave transposes
// through its own LDS slice.
//
// Global -> MFMA layout.  S^T = K Q^T needs lane = token, registers = dims (the contraction), i.e. the transpose of the
// row-major activations: a direct load would touch 64 different cache lines per instruction.  Tiles are therefore read
// as whole 256-B head rows (4 rows per wave-instruction, bounds-checked buffer loads: rows past the end of the sequence
// read 0) and turned through LDS.  V needs lane = dim: direct dword loads are
// already two full lines per instruction.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_varlen_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * ATT_LDT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int hpairs = (a.H + 1) >> 1;
    const int blk = blockIdx.x / hpairs;
    const int h = (blockIdx.x - blk * hpairs) * 2 + (wave >> 1);
    const int4 st = a.blocks[blk];
    const int tok0 = st.x, L = st.y, q0 = st.z + 32 * (wave & 1);
    if (h >= a.H || q0 >= L) return;
    const int r = lane & 31, half = lane >> 5;
    const int hid = a.H * 64;
    const int ldb = a.ld * 4;   // row pitch in bytes

    // everything below addresses the sequence through one descriptor: base = its first row, this head's q columns; the
    // range ends with the sequence's last row, so the rows a partial tile reaches past it read 0 WITHOUT touching memory
    // (tiles are 32 rows whatever the length: unchecked loads moved 1.4x the algorithmic bytes at the LLeQA length mix)
    const float* base = a.qkv + (size_t)tok0 * a.ld + h * 64;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (L * a.ld - h * 64) * 4, 0x00020000);

    float* my = lds + wave * (32 * ATT_LDT);
    const int ld_row = lane >> 4, ld_c4 = lane & 15;                 // coalesced tile load: 4 rows x 16 float4 per instruction
    const int voff_t = ld_row * ldb + ld_c4 * 16;
    float* const wr = my + ld_row * ATT_LDT + ld_c4 * 4;
    const float* const rd = my + r * ATT_LDT + half * 32;
    const int voff_v = (4 * half) * ldb + r * 4;

    auto load_tile = [&](int row0, int col_bytes, float (&f)[32]) {   // rows row0..row0+31, 64 floats at col_bytes -> f = [row r][half*32 + kk]
        i32x4 raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_t, (row0 + 4 * i) * ldb + col_bytes, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<i32x4*>(wr + 4 * i * ATT_LDT) = raw[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(rd + 4 * i);
            f[4 * i] = t.x; f[4 * i + 1] = t.y; f[4 * i + 2] = t.z; f[4 * i + 3] = t.w;
        }
    };

    float qf[32];
    load_tile(q0, 0, qf);
    float m = -INFINITY, l = 0.0f;   // running max (base-2 domain) and sum of this lane's query
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.0f; o1[i] = 0.0f; }

    for (int j0 = 0; j0 < L; j0 += 32) {
        float kf[32];
        load_tile(j0, hid * 4, kf);
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kk], qf[kk], s, 0, 0, 0);
        // s[g] = <q_{q0+r}, k_j>, j = j0 + 8(g/4) + 4 half + g%4.  V in the matching layout: row = dim r (and 32 + r)
        float v0[16], v1[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int so = (j0 + 8 * (g >> 2) + (g & 3)) * ldb + 2 * hid * 4;
            v0[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff_v, so, 0));
            v1[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff_v + 128, so, 0));
        }
        if (j0 + 32 > L) {   // last, partial tile: keys past the sequence (read as zeros) drop out
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (j0 + 8 * (g >> 2) + (g & 3) + 4 * half >= L) s[g] = -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int g = 1; g < 16; ++g) mx = fmaxf(mx, s[g]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx * a.scale_log2e);   // finite: key j0 < L is in this tile
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);   // first tile: exp2(-inf) = 0
        float psum = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            // one rounding: the error of mnew itself is common to the whole row and cancels in p / sum(p)
            s[g] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[g], a.scale_log2e, -mnew));
            psum += s[g];
        }
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = mnew;
        if (j0 > 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[g], s[g], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[g], s[g], o1, 0, 0, 0);
        }
    }
    // O^T (lane = query, registers = dims) -> rows through the LDS slice -> whole 256-B rows out
    const float inv = 1.0f / l;
    float* const ow = my + r * ATT_LDT + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(ow + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
        *reinterpret_cast<float4*>(ow + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
    float* const op = a.out + (size_t)(tok0 + q0) * a.ldo + h * 64 + ld_c4 * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + ld_row;
        const float4 t = *reinterpret_cast<const float4*>(my + row * ATT_LDT + ld_c4 * 4);
        if (q0 + row < L) *reinterpret_cast<float4*>(op + (size_t)row * a.ldo) = t;
    }
}

// LayerNorm(x + res) * gamma + beta; one wave per row, VPL float4 per lane (d <= 256 * VPL)
template <int VPL>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ res, int ldr,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int rows, int d, float* __restrict__ out, int ldo) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = d >> 2;
    float4 v[VPL];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            float4 t = reinterpret_cast<const float4*>(x + (size_t)row * ldx)[c];
            if (res) {
                const float4 u = reinterpret_cast<const float4*>(res + (size_t)row * ldr)[c];
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

extern "C" int fz_attn_varlen_f32(const float* qkv, int ld, const int32_t* blocks, int n_blocks, int H, int head_dim, float scale,
                                  float* out, int ldo, void* stream) {
    if (n_blocks < 0 || H <= 0 || !(scale > 0.0f)) return FZ_ERR_ARG;
    if (n_blocks == 0) return FZ_OK;
    if (!qkv || !blocks || !out) return FZ_ERR_ARG;
    if (head_dim != 64) return FZ_ERR_UNSUPPORTED;
    if (ld < 3 * H * 64 || ldo < H * 64) return FZ_ERR_ARG;
    if ((ld & 3) || (ldo & 3) || !aligned16(qkv) || !aligned16(out) || !aligned16(blocks)) return FZ_ERR_UNSUPPORTED;
    if ((long long)ld * 4 * 16384 > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;   // in-sequence byte offsets are 32-bit
    AttnArgs a{qkv, ld, reinterpret_cast<const int4*>(blocks), n_blocks, H, out, ldo, scale * 1.4426950408889634f};
    const long long grid = (long long)n_blocks * ((H + 1) / 2);
    if (grid > 0x7fffffffLL) return FZ_ERR_UNSUPPORTED;
    attn_varlen_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a);
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
        add_layernorm_kernel<4><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo);
    else
        add_layernorm_kernel<16><<<grid, 256, 0, s>>>(x, ldx, res, ldr, gamma, beta, eps, rows, d, out, ldo);
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
