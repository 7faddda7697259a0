// util.hip -- status strings and small elementwise helpers.
#include "common.h"

namespace fz {
thread_local int g_last_hip_error = 0;

__global__ void fill_i32_kernel(int32_t* p, size_t count, int32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void f64_to_f32_kernel(const double* s, float* d, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
}  // namespace fz

using namespace fz;

extern "C" const char* fz_strerror(int status) {
    switch (status) {
        case FZ_OK: return "ok";
        case FZ_ERR_ARG: return "invalid argument";
        case FZ_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case FZ_ERR_HIP: return "HIP runtime error (see fz_last_hip_error)";
        case FZ_ERR_WORKSPACE: return "workspace missing or too small";
        default: return "unknown status";
    }
}
extern "C" int fz_last_hip_error(void) { return g_last_hip_error; }
extern "C" int fz_abi_version(void) { return 6; }

extern "C" int fz_fill_i32(int32_t* p, size_t count, int32_t value, void* stream) {
    if (!p && count) return FZ_ERR_ARG;
    if (!count) return FZ_OK;
    size_t blocks = (count + 255) / 256;
    fill_i32_kernel<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, as_stream(stream)>>>(p, count, value);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
extern "C" int fz_f64_to_f32(const double* src, float* dst, size_t count, void* stream) {
    if ((!src || !dst) && count) return FZ_ERR_ARG;
    if (!count) return FZ_OK;
    size_t blocks = (count + 255) / 256;
    f64_to_f32_kernel<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, as_stream(stream)>>>(src, dst, count);
    FZ_LAUNCH_CHECK();
    return FZ_OK;
}
