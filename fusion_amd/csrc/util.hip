// util.hip -- status strings and small elementwise helpers.
#include <dlfcn.h>

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
extern "C" int fz_abi_version(void) { return 19; }

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

// ---- C1: the one collective of the sharded configuration, behind the C ABI ------------------------------------------------
// ncclAllGather of the per-shard [Q][k] (score, id) lists + the local G-way merge (fz_topk_merge).  RCCL is NOT linked into
// this library: the two entry points it needs are looked up at the first call -- in the process image first (a host that
// already runs RCCL, e.g. PyTorch, brings its own copy and its communicators must be used with it), then in librccl.so.
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
nccl_allgather_fn g_allgather = nullptr;
nccl_allgather_fn find_allgather() {
    if (g_allgather) return g_allgather;
    void* f = dlsym(RTLD_DEFAULT, "ncclAllGather");
    if (!f) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) f = dlsym(h, "ncclAllGather");
    }
    g_allgather = reinterpret_cast<nccl_allgather_fn>(f);
    return g_allgather;
}
}  // namespace

extern "C" size_t fz_topk_allgather_workspace_bytes(int world, int Q, int k) {
    if (world <= 0 || Q <= 0 || k <= 0) return 0;
    return (size_t)world * Q * k * (sizeof(float) + sizeof(int64_t)) + 256;
}

extern "C" int fz_topk_allgather(const float* local_scores, const int64_t* local_ids, int Q, int k, void* rccl_comm, int world,
                                 float* out_scores, int64_t* out_ids, void* workspace, size_t workspace_bytes, void* stream) {
    if (Q < 0 || k <= 0 || world <= 0) return FZ_ERR_ARG;
    if (Q == 0) return FZ_OK;
    if (!local_scores || !local_ids || !out_scores || !out_ids || !rccl_comm) return FZ_ERR_ARG;
    if ((long)world * k > 35840) return FZ_ERR_UNSUPPORTED;   // one merge row (fz_topk_merge)
    if (!workspace || workspace_bytes < fz_topk_allgather_workspace_bytes(world, Q, k)) return FZ_ERR_WORKSPACE;
    nccl_allgather_fn ag = find_allgather();
    if (!ag) return FZ_ERR_UNSUPPORTED;                         // no RCCL in the process and no librccl.so to load
    int64_t* gi = reinterpret_cast<int64_t*>(workspace);          // [world][Q][k]
    float* gs = reinterpret_cast<float*>(gi + (size_t)world * Q * k);
    hipStream_t st = as_stream(stream);
    const size_t count = (size_t)Q * k;
    // ncclDataType_t: ncclInt64 = 4, ncclFloat32 = 7 (rccl.h)
    if (ag(local_scores, gs, count, 7, rccl_comm, st) != 0) return FZ_ERR_HIP;
    if (ag(local_ids, gi, count, 4, rccl_comm, st) != 0) return FZ_ERR_HIP;
    return fz_topk_merge(gs, gi, world, Q, k, out_scores, out_ids, stream);
}
