// host_solve.h -- host part of estimateTransformSymm (reference ICP/func.cpp:76-102)
#pragma once
#include "symmicp.h"
namespace symmicp {
int solve_quirks(const symmicp_sums &S, float pbar[3], float qbar[3], float a[3], float t[3], float *rcond, float out16[16]);
int solve_paper(const symmicp_sums &S, const float pivot[3], float pbar[3], float qbar[3], float a[3], float t[3],
                float *rcond, float out16[16]);
int solve_p2p(const symmicp_sums &S, const float pivot[3], float *rcond, float out16[16]);
void mat4_mul(const float A[16], const float B[16], float C[16]);
}  // namespace symmicp
