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

struct AttnArgs {
    const float* qkv;   // [T][ld]: q | k | v, each H*64 wide
    int ld;
    const int4* strips;  // (first row of the sequence, its length L, first query of this strip, unused)
    int n_strips;
    int H;
    float* out;  // [T][ldo]
    int ldo;
    float scale;
};

// one wave = one (strip of 32 queries, head); 4 waves of a workgroup = 4 consecutive heads of the same strip (their
// rows are contiguous in memory).  No LDS, no barriers.
__global__ __launch_bounds__(256) void attn_varlen_kernel(AttnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hgroups = (a.H + 3) >> 2;
    const int strip = blockIdx.x / hgroups;
    const int h = (blockIdx.x - strip * hgroups) * 4 + wave;
    if (h >= a.H) return;
    const int4 st = a.strips[strip];
    const int tok0 = st.x, L = st.y, q0 = st.z;
    const int r = lane & 31, half = lane >> 5;
    const int hid = a.H * 64;
    const float* base = a.qkv + (size_t)tok0 * a.ld + h * 64;

    // Q fragment (B operand of S^T): query q0 + r, contraction slots kk -> dim half*32 + kk
    float qf[32];
    {
        const int qi = min(q0 + r, L - 1);
        const float4* qp = reinterpret_cast<const float4*>(base + (size_t)qi * a.ld + half * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = qp[i];
            qf[4 * i] = t.x; qf[4 * i + 1] = t.y; qf[4 * i + 2] = t.z; qf[4 * i + 3] = t.w;
        }
    }
    float m = -INFINITY, l = 0.0f;
    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.0f; o1[i] = 0.0f; }

    for (int j0 = 0; j0 < L; j0 += 32) {
        // K fragment (A operand): key j0 + r, same contraction slots as Q
        float kf[32];
        {
            const int kj = min(j0 + r, L - 1);
            const float4* kp = reinterpret_cast<const float4*>(base + (size_t)kj * a.ld + hid + half * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t = kp[i];
                kf[4 * i] = t.x; kf[4 * i + 1] = t.y; kf[4 * i + 2] = t.z; kf[4 * i + 3] = t.w;
            }
        }
        // V fragments (A operand of O^T): row = dim r (and 32 + r), contraction slot (reg, half) -> key j(reg, half)
        float v0[16], v1[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int j = min(j0 + 8 * (g >> 2) + (g & 3) + 4 * half, L - 1);
            const float* vp = base + (size_t)j * a.ld + 2 * hid + r;
            v0[g] = vp[0];
            v1[g] = vp[32];
        }
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kk], qf[kk], s, 0, 0, 0);
        // s[g] = <q_{q0+r}, k_j>, j = j0 + 8(g/4) + 4 half + g%4
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int j = j0 + 8 * (g >> 2) + (g & 3) + 4 * half;
            s[g] = j < L ? s[g] * a.scale : -INFINITY;
            mx = fmaxf(mx, s[g]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);          // finite: key j0 < L is in this tile
        const float alpha = __expf(m - mnew);     // first tile: exp(-inf) = 0
        float psum = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            s[g] = __expf(s[g] - mnew);
            psum += s[g];
        }
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = mnew;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[g], s[g], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[g], s[g], o1, 0, 0, 0);
        }
    }
    if (q0 + r < L) {
        const float inv = 1.0f / l;
        float* op = a.out + (size_t)(tok0 + q0 + r) * a.ldo + h * 64 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            *reinterpret_cast<float4*>(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
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

// out[b] = mean of rows [cu[b], cu[b+1]) (zeros for an empty sequence); one workgroup per sequence
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ cu, int d,
                                                           float* __restrict__ out, int ldo) {
    const int b = blockIdx.x;
    const int t0 = cu[b], t1 = cu[b + 1];
    const float inv = t1 > t0 ? 1.0f / (float)(t1 - t0) : 0.0f;
    for (int c = threadIdx.x; c < (d >> 2); c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = t0; t < t1; ++t) {
            const float4 u = reinterpret_cast<const float4*>(x + (size_t)t * ldx)[c];
            acc.x += u.x; acc.y += u.y; acc.z += u.z; acc.w += u.w;
        }
        reinterpret_cast<float4*>(out + (size_t)b * ldo)[c] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

}  // namespace fz

using namespace fz;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int fz_attn_varlen_f32(const float* qkv, int ld, const int32_t* strips, int n_strips, int H, int head_dim, float scale,
                                  float* out, int ldo, void* stream) {
    if (n_strips < 0 || H <= 0) return FZ_ERR_ARG;
    if (n_strips == 0) return FZ_OK;
    if (!qkv || !strips || !out) return FZ_ERR_ARG;
    if (head_dim != 64) return FZ_ERR_UNSUPPORTED;
    if (ld < 3 * H * 64 || ldo < H * 64) return FZ_ERR_ARG;
    if ((ld & 3) || (ldo & 3) || !aligned16(qkv) || !aligned16(out) || !aligned16(strips)) return FZ_ERR_UNSUPPORTED;
    AttnArgs a{qkv, ld, reinterpret_cast<const int4*>(strips), n_strips, H, out, ldo, scale};
    const long long grid = (long long)n_strips * ((H + 3) / 4);
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

extern "C" int fz_segment_mean_f32(const float* x, int ldx, const int32_t* cu_rows, int B, int d, float* out, int ldo, void* stream) {
    if (B < 0 || d <= 0) return FZ_ERR_ARG;
    if (B == 0) return FZ_OK;
    if (!x || !cu_rows || !out || ldx < d || ldo < d) return FZ_ERR_ARG;
    if ((d & 3) || (ldx & 3) || (ldo & 3) || !aligned16(x) || !aligned16(out)) return FZ_ERR_UNSUPPORTED;
    segment_mean_kernel<<<(unsigned)B, 256, 0, as_stream(stream)>>>(x, ldx, cu_rows, d, out, ldo);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
