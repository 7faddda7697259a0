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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: remember per device (bit mask, one per call site)
// where it has been set; devices >= 64 set it on every call.  Returns FZ_OK or FZ_ERR_HIP.
inline int raise_lds_limit(const void* kernel, size_t bytes, unsigned long long& done_mask) {
    int dev = 0;
    FZ_HIP_TRY(hipGetDevice(&dev));
    if (dev < 64 && ((done_mask >> dev) & 1ull)) return FZ_OK;
    FZ_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (dev < 64) done_mask |= 1ull << dev;
    return FZ_OK;
}

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

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// ---- VALU-only cross-lane steps (no ds_bpermute round trip through the LDS pipe) ------------------------------
// DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_mirror = 0x140, row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over each 16-lane row; every lane of the row ends up with it
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);   // quads are uniform: mirroring a half row brings the other quad's sum
    v += dpp_f32<0x140>(v);   // half rows are uniform: mirroring the row brings the other half's sum
    return v;
}
// gfx950 v_permlane16_swap / v_permlane32_swap: swap16(d, s) -> d = [d0 s0 d2 s2], s = [d1 s1 d3 s3] by 16-lane rows;
// swap32(d, s) -> d = [d.lo s.lo], s = [d.hi s.hi] (probed: tools/micro/permlane_probe.hip).  Written as asm: through
// the builtin, hipcc (ROCm 7.2) treats the two results of swap(x, x) as one value and drops the second.  The s_nops cover
// the VALU-write -> permlane-read and permlane-write -> VALU-read hazards the compiler would otherwise handle itself.
__device__ __forceinline__ void swap16(float& d, float& s) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(d), "+v"(s)); }
__device__ __forceinline__ void swap32(float& d, float& s) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(d), "+v"(s)); }

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {   // DPP moves 32-bit registers: the two halves of a double separately
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void swap16(double& d, double& s) {
    float dl = __int_as_float(__double2loint(d)), dh = __int_as_float(__double2hiint(d));
    float sl = __int_as_float(__double2loint(s)), sh = __int_as_float(__double2hiint(s));
    swap16(dl, sl); swap16(dh, sh);
    d = __hiloint2double(__float_as_int(dh), __float_as_int(dl));
    s = __hiloint2double(__float_as_int(sh), __float_as_int(sl));
}
__device__ __forceinline__ void swap32(double& d, double& s) {
    float dl = __int_as_float(__double2loint(d)), dh = __int_as_float(__double2hiint(d));
    float sl = __int_as_float(__double2loint(s)), sh = __int_as_float(__double2hiint(s));
    swap32(dl, sl); swap32(dh, sh);
    d = __hiloint2double(__float_as_int(dh), __float_as_int(dl));
    s = __hiloint2double(__float_as_int(sh), __float_as_int(sl));
}
// Whole-wave (64 lanes) reductions on the VALU: 4 DPP steps inside every 16-lane row (quad_perm xor 1, xor 2, half-row mirror,
// row mirror: after each step the lanes of the group just closed hold the same value, so mirroring pairs distinct groups), then
// the two permlane swaps across rows.  Every lane ends up with the result.  (The __shfl_xor forms compile to ds_bpermute: six
// dependent LDS round trips per value.)
template <typename T, typename F>
__device__ __forceinline__ T wave_reduce_valu(T v, F op) {
    if constexpr (sizeof(T) == 4) {
        v = op(v, dpp_f32<0xB1>(v)); v = op(v, dpp_f32<0x4E>(v)); v = op(v, dpp_f32<0x141>(v)); v = op(v, dpp_f32<0x140>(v));
    } else {
        v = op(v, dpp_f64<0xB1>(v)); v = op(v, dpp_f64<0x4E>(v)); v = op(v, dpp_f64<0x141>(v)); v = op(v, dpp_f64<0x140>(v));
    }
    T o = v;
    swap16(v, o);
    v = op(v, o);
    o = v;
    swap32(v, o);
    return op(v, o);
}
// the same over ONE 16-lane row only (every lane of the row gets its row's result)
template <typename T, typename F>
__device__ __forceinline__ T row16_reduce_valu(T v, F op) {
    if constexpr (sizeof(T) == 4) {
        v = op(v, dpp_f32<0xB1>(v)); v = op(v, dpp_f32<0x4E>(v)); v = op(v, dpp_f32<0x141>(v)); v = op(v, dpp_f32<0x140>(v));
    } else {
        v = op(v, dpp_f64<0xB1>(v)); v = op(v, dpp_f64<0x4E>(v)); v = op(v, dpp_f64<0x141>(v)); v = op(v, dpp_f64<0x140>(v));
    }
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
