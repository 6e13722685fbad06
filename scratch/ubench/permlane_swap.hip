// semantics of v_permlane32_swap / v_permlane16_swap on gfx950 (which lanes of which operand end up where): prints the lane maps
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o)
{
    const unsigned a = threadIdx.x, b = threadIdx.x + 100;      // a: the first operand (vdst), b: the second (src0)
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const u32x2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y; o[128 + threadIdx.x] = s.x; o[192 + threadIdx.x] = s.y;
}
int main()
{
    unsigned *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char *name[4] = {"permlane32_swap .x", "permlane32_swap .y", "permlane16_swap .x", "permlane16_swap .y"};
    for (int r = 0; r < 4; r++) {
        std::printf("%s:", name[r]);
        for (int l = 0; l < 64; l += 8) std::printf(" [%d]=%u", l, h[64 * r + l]);
        std::printf("\n");
    }
    return 0;
}
