// host_solve.cpp -- host entry points of the solve (reference ICP/func.cpp:76-102); the arithmetic lives in solve_core.h,
// shared with the device-side solve of k_final_reduce.
#include "symmicp.h"
#include "host_solve.h"
#include "solve_core.h"

namespace symmicp {

int solve_quirks(const symmicp_sums &S, float pbar[3], float qbar[3], float a[3], float t[3], float *rcond, float out16[16])
{
    return solve::solve_quirks(S, pbar, qbar, a, t, rcond, out16);
}

int solve_paper(const symmicp_sums &S, const float pivot[3], float pbar[3], float qbar[3], float a[3], float t[3], float *rcond, float out16[16])
{
    return solve::solve_paper(S, pivot, pbar, qbar, a, t, rcond, out16);
}

int solve_p2p(const symmicp_sums &S, const float pivot[3], float *rcond, float out16[16])
{
    return solve::solve_p2p(S, pivot, rcond, out16);
}

void mat4_mul(const float A[16], const float B[16], float C[16]) { solve::mat4_mul(A, B, C); }

}  // namespace symmicp
