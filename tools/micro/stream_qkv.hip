// Streaming reference for the attention kernel's byte count: out[t][c] = q[t][c] + k[t][c] + v[t][c], fully coalesced.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void stream_qkv_kernel(const float4* qkv, int T, int hid4, float4* out) {
    const long long n = (long long)T * hid4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = i / hid4; const int c = (int)(i - t * hid4);
        const float4* row = qkv + t * 3 * hid4;
        const float4 a = row[c], b = row[hid4 + c], d = row[2 * hid4 + c];
        out[i] = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);
    }
}
extern "C" int stream_qkv(const float* qkv, int T, int hid, float* out, void* stream) {
    stream_qkv_kernel<<<256 * 16, 256, 0, (hipStream_t)stream>>>((const float4*)qkv, T, hid / 4, (float4*)out);
    return (int)hipGetLastError();
}
