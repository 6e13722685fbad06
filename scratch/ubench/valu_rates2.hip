// More issue costs (gfx950), cycles per wave64 instruction per SIMD at a nominal 2.4 GHz, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
#define R8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, const float4 *tq)
{
    __shared__ float4 s_pts[64];
    if (threadIdx.x < 64) s_pts[threadIdx.x] = tq[threadIdx.x];
    __syncthreads();
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 1.0001f;
    int n = 0;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) asm volatile(R8("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 1) asm volatile(R8("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n") :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b) : "vcc");
            if (OP == 2) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n" R8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (OP == 3) asm volatile(R8("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 4) {   // ds_read_b128 broadcast (all lanes one address) + 3 sub + 3 mul + 2 add on VGPRs: the LDS form of one point test (no compare)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float4 q; { const volatile float *vp = (const volatile float *)&s_pts[(n + j) & 63]; q.x = vp[0]; q.y = vp[1]; q.z = vp[2]; q.w = 0.f; }
                    const float dx = a0 - q.x, dy = a1 - q.y, dz = a2 - q.z;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    a3 = fminf(a3, d2);
                }
                n += 8;
            }
            if (OP == 5) {   // readlane form of the same
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a4), (n + j) & 63)), qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a5), (n + j) & 63)),
                                qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a6), (n + j) & 63));
                    const float dx = a0 - qx, dy = a1 - qy, dz = a2 - qz;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    a3 = fminf(a3, d2);
                }
                n += 8;
            }
            if (OP == 6) asm volatile(R8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (OP == 7) asm volatile(R8("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (OP == 8) asm volatile(R8("v_sub_f32 %0, s20, %0\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_sub_f32 %3, s21, %3\n v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, s22, %6\n v_min_f32 %7, %7, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21", "s22");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP>
double run(int waves_per_simd, float *d, const float4 *tq, double per_iter)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, tq);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, tq);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * ITER * per_iter);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    float4 *tq; hipMalloc(&tq, 64 * 16); hipMemset(tq, 0, 64 * 16);
    for (int w : {1, 2, 4, 8}) {
        printf("waves/SIMD %d:", w);
        printf(" v_sub vgpr %.2f |", run<0>(w, d, tq, 512)); printf(" v_cmp_lt e32->vcc %.2f |", run<1>(w, d, tq, 512)); printf(" v_cndmask e32 %.2f |", run<2>(w, d, tq, 513));
        printf(" v_min_f32 %.2f |", run<3>(w, d, tq, 512)); printf(" LDS-broadcast point test (cycles per point) %.2f |", run<4>(w, d, tq, 64)); printf(" readlane point test (cycles per point) %.2f |", run<5>(w, d, tq, 64));
        printf(" v_mov dpp %.2f |", run<6>(w, d, tq, 512)); printf(" mixed VOP2 vgpr %.2f |", run<7>(w, d, tq, 512)); printf(" mixed VOP2, 3 of 8 with sgpr %.2f\n", run<8>(w, d, tq, 512));
    }
    return 0;
}
