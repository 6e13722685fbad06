// Issue costs, part 3 (gfx950): v_cndmask forms, LDS broadcast reads, full point tests.  Cycles per instruction (or per point) per SIMD at a nominal 2.4 GHz.
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
    int n = 0, slot = 0;
    unsigned long long tie = 0;
    for (int it = 0; it < ITER; it++) {
        if (OP >= 3) { a0 += 1e-3f; a1 -= 1e-3f; a2 += 2e-3f; a3 = 1e30f; }      // (a new query every 64 points: nothing to hoist)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            if (OP == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n" R8("v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %4, %8, vcc\n v_cndmask_b32 %4, %5, %8, vcc\n v_cndmask_b32 %5, %6, %8, vcc\n v_cndmask_b32 %6, %7, %8, vcc\n v_cndmask_b32 %7, %0, %8, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (OP == 1) asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n" R8("v_cndmask_b32 %0, %1, %8, s[20:21]\n v_cndmask_b32 %1, %2, %8, s[20:21]\n v_cndmask_b32 %2, %3, %8, s[20:21]\n v_cndmask_b32 %3, %4, %8, s[20:21]\n v_cndmask_b32 %4, %5, %8, s[20:21]\n v_cndmask_b32 %5, %6, %8, s[20:21]\n v_cndmask_b32 %6, %7, %8, s[20:21]\n v_cndmask_b32 %7, %0, %8, s[20:21]\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");
            if (OP == 2) asm volatile(R8("v_cndmask_b32 %0, %1, %8, vcc\n v_sub_f32 %1, %2, %8\n v_mul_f32 %2, %3, %8\n v_add_f32 %3, %4, %8\n v_cndmask_b32 %4, %5, %8, vcc\n v_sub_f32 %5, %6, %8\n v_mul_f32 %6, %7, %8\n v_add_f32 %7, %0, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (OP == 3) {   // LDS broadcast: ds_read_b128 of one address in all lanes + distance + min
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 q = s_pts[(n + j) & 63];
                    const float dx = a0 - q.x, dy = a1 - q.y, dz = a2 - q.z;
                    a3 = fminf(a3, (dx * dx + dy * dy) + dz * dz);
                }
                n += 8;
            }
            if (OP == 4) {   // the packet kernel's point test as written (readlane operands): lt / eq ballots, select best and slot
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a4), (n + j) & 63)), qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a5), (n + j) & 63)),
                                qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a6), (n + j) & 63));
                    const float dx = a0 - qx, dy = a1 - qy, dz = a2 - qz;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const bool lt = d2 < a3;
                    tie |= __ballot(d2 == a3);
                    a3 = lt ? d2 : a3;
                    slot = lt ? j : slot;
                }
                n += 8;
            }
            if (OP == 5) {   // the same from LDS
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 q = s_pts[(n + j) & 63];
                    const float dx = a0 - q.x, dy = a1 - q.y, dz = a2 - q.z;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    const bool lt = d2 < a3;
                    tie |= __ballot(d2 == a3);
                    a3 = lt ? d2 : a3;
                    slot = lt ? j : slot;
                }
                n += 8;
            }
            if (OP == 6) {   // LDS, select-free: best by v_min, winner recovered afterwards (one compare per point, ballot of equality kept)
                float m = a3;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float4 q = s_pts[(n + j) & 63];
                    const float dx = a0 - q.x, dy = a1 - q.y, dz = a2 - q.z;
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    m = fminf(m, d2);
                    a7 = d2;      // (kept for the recovery below: stands for 8 live values)
                }
                if (__ballot(m < a3)) { slot = (a7 == m) ? 7 : slot; a3 = m; }
                n += 8;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + slot + (float)tie;
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
        printf(" cndmask vcc (non-self) %.2f |", run<0>(w, d, tq, 513)); printf(" cndmask sgpr-pair %.2f |", run<1>(w, d, tq, 513)); printf(" 2 cndmask + 6 VOP2 %.2f |", run<2>(w, d, tq, 512));
        printf(" per point: LDS dist+min %.1f |", run<3>(w, d, tq, 64)); printf(" readlane full test %.1f |", run<4>(w, d, tq, 64)); printf(" LDS full test %.1f |", run<5>(w, d, tq, 64));
        printf(" LDS min-only %.1f\n", run<6>(w, d, tq, 64));
    }
    return 0;
}
