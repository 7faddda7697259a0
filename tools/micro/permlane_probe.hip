// Probe of v_permlane16_swap / v_permlane32_swap semantics (gfx950): prints, per lane, both results for input = lane id.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
    unsigned u = threadIdx.x;
    auto a = __builtin_amdgcn_permlane16_swap(u, u + 100, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(u, u + 100, false, false);
    o[threadIdx.x * 4 + 0] = a[0]; o[threadIdx.x * 4 + 1] = a[1]; o[threadIdx.x * 4 + 2] = b[0]; o[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { for (int l = 0; l < 64; l += 1) printf("%u ", h[l * 4 + j]); printf("\n"); }
    return 0;
}
