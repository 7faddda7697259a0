// Read-side ceiling: S planes of Q x ld floats streamed with float4 loads, nothing written but one float per workgroup.
// mode 0: flat grid (one float4 per plane per thread, like fuse_rank / fuse_nsf_elem4); mode 1: grid-stride with 4 float4 in flight.
#include <hip/hip_runtime.h>
struct P { const float4* p[4]; };
extern "C" __global__ __launch_bounds__(256) void read_flat(P a, int S, long long n4, float* sink) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n4) for (int s = 0; s < S; ++s) { const float4 v = a.p[s][i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}
extern "C" __global__ __launch_bounds__(256) void read_stride(P a, int S, long long n4, float* sink) {
    float acc = 0.f;
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * step) {
        for (int s = 0; s < S; ++s) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (i + u * step < n4) ? a.p[s][i + u * step] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}
extern "C" int read_bw(const float* p0, const float* p1, const float* p2, const float* p3, int S, long long n4, int mode, float* sink, void* stream) {
    P a; a.p[0] = (const float4*)p0; a.p[1] = (const float4*)p1; a.p[2] = (const float4*)p2; a.p[3] = (const float4*)p3;
    if (mode == 0) read_flat<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, S, n4, sink);
    else read_stride<<<256 * 8, 256, 0, (hipStream_t)stream>>>(a, S, n4, sink);
    return (int)hipGetLastError();
}
