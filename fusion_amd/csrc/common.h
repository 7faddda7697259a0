// common.h -- shared device/host helpers for libfusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fusion_hip.h"

#define FZ_MAX_SYSTEMS 8
#define FZ_WAVE 64

namespace fz {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return FZ_ERR_HIP;
}

#define FZ_HIP_TRY(expr)                                  \
    do {                                                  \
        hipError_t _e = (expr);                           \
        if (_e != hipSuccess) return ::fz::hip_fail(_e);  \
    } while (0)

// every launcher ends with this: surfaces launch-configuration errors loudly
#define FZ_LAUNCH_CHECK()                                 \
    do {                                                  \
        hipError_t _e = hipGetLastError();                \
        if (_e != hipSuccess) return ::fz::hip_fail(_e);  \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T, int N>
struct PtrPack {
    T p[N];
};

// ---- wave-level primitives (64 lanes) -------------------------------------------------
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// order-preserving bit transforms: larger float  <=>  SMALLER unsigned key (descending sort
// becomes an ascending LSD radix sort).  -0.0 == +0.0; every NaN maps to key 0 (sorts first).
__device__ __forceinline__ uint32_t desc_key_f32(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;          // NaN first
    if (u == 0x80000000u) u = 0u;                             // -0.0 -> +0.0
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ float desc_key_f32_inv(uint32_t k) {
    if (k == 0u) return __uint_as_float(0x7fc00000u);
    uint32_t asc = ~k;
    uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t desc_key_f64(double f) {
    uint64_t u = (uint64_t)__double_as_longlong(f);
    if ((u & 0x7fffffffffffffffull) > 0x7ff0000000000000ull) return 0ull;
    if (u == 0x8000000000000000ull) u = 0ull;
    uint64_t asc = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    return ~asc;
}
__device__ __forceinline__ double desc_key_f64_inv(uint64_t k) {
    if (k == 0ull) return __longlong_as_double(0x7ff8000000000000ll);
    uint64_t asc = ~k;
    uint64_t u = (asc & 0x8000000000000000ull) ? (asc & 0x7fffffffffffffffull) : ~asc;
    return __longlong_as_double((long long)u);
}

}  // namespace fz
