// Issue cost of the vector instructions the packet search is made of (gfx950): cycles per wave64 instruction per SIMD with W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ITER 2000
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int lanesel)
{
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 1.0001f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&b));
            if (OP == 2) asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == 3) asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (OP == 5) asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 6) asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (OP == 7) asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 8) asm volatile("v_readlane_b32 s20, %0, 3\n v_mul_f32 %1, s20, %1\n v_readlane_b32 s21, %2, 5\n v_mul_f32 %3, s21, %3\n v_readlane_b32 s22, %4, 7\n v_mul_f32 %5, s22, %5\n v_readlane_b32 s23, %6, 9\n v_mul_f32 %7, s23, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23");
            if (OP == 9) asm volatile("v_sub_f32 %0, s20, %0\n v_sub_f32 %1, s21, %1\n v_sub_f32 %2, s22, %2\n v_sub_f32 %3, s23, %3\n v_sub_f32 %4, s20, %4\n v_sub_f32 %5, s21, %5\n v_sub_f32 %6, s22, %6\n v_sub_f32 %7, s23, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21", "s22", "s23");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP>
double run(int waves_per_simd, float *d)
{
    const int blocks = 256 * waves_per_simd;      // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)waves_per_simd * ITER * REP;
    return ms * 1e-3 * 2.4e9 / insts_per_simd;     // cycles per instruction per SIMD at 2.4 GHz
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const char *names[] = {"v_mul_f32", "v_pk_mul_f32", "v_readlane_b32", "v_cmp_lt_f32 (sgpr dst)", "v_cndmask_b32", "v_max3_f32", "v_min_u32 dpp row_shr", "v_fma_f32", "readlane+dependent v_mul", "v_sub_f32 sgpr operand"};
    for (int w : {1, 2, 4, 8}) {
        printf("waves/SIMD %d:", w);
        printf(" %s %.2f |", names[0], run<0>(w, d)); printf(" %s %.2f |", names[1], run<1>(w, d)); printf(" %s %.2f |", names[2], run<2>(w, d));
        printf(" %s %.2f |", names[3], run<3>(w, d)); printf(" %s %.2f |", names[4], run<4>(w, d)); printf(" %s %.2f |", names[5], run<5>(w, d));
        printf(" %s %.2f |", names[6], run<6>(w, d)); printf(" %s %.2f |", names[7], run<7>(w, d)); printf(" %s %.2f |", names[8], run<8>(w, d)); printf(" %s %.2f\n", names[9], run<9>(w, d));
    }
    return 0;
}
