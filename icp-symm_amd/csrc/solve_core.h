// solve_core.h -- the O(1) part of estimateTransformSymm (reference ICP/func.cpp:76-102): from one 40-double
// reduction record to the 4x4 increment.  ONE source for both sides: host_solve.cpp wraps these functions for the host
// (the tested reference, north_star: "the 6x6 solve stays on the host"), and k_final_reduce calls the same code on the
// device so that a converged alignment can run pass after pass without a host round trip (kernels_pass.hip).
//
// The reference solves two N x 3 least-squares problems with a JacobiSVD pseudo-inverse (func.cpp:64-73, :87-88).  For
// full column rank that is the normal-equation solution, and the normal-equation blocks (M^T M, N^T N, M^T N, M^T c,
// N^T c) are exactly what the GPU pass accumulates in fp64.  So 3 x 3 (QUIRKS) or 6 x 6 (PAPER) symmetric systems are
// solved in fp64 and the affine is composed in fp32 the way Eigen::Affine3f::translate/rotate do (post-multiply).
// fp32 expressions stay unfused (-ffp-contract=off on both sides); the two sides differ only in the last bits of
// sin / cos / atan / sqrt of their math libraries.
#pragma once
#include <cmath>
#include "symmicp.h"

#if defined(__HIPCC__)
#define SYMMICP_HD __host__ __device__
#else
#define SYMMICP_HD
#endif
#pragma clang fp contract(off)

namespace symmicp {
namespace solve {

SYMMICP_HD inline bool finite32(float v) { return (v - v) == 0.0f; }      // false for inf and NaN

// Symmetric positive (semi)definite solve: Cholesky with the conditioning read
// from a Jacobi eigenvalue sweep of the same matrix.  N <= 6.
template <int N>
struct SymSolver {
    double A[N][N];

    // eigenvalues by cyclic Jacobi rotations (matrix is tiny; 30 sweeps is plenty)
    SYMMICP_HD void eigenvalues(double w[N]) const
    {
        double B[N][N];
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) B[i][j] = A[i][j];
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0.0, diag = 0.0;
            for (int i = 0; i < N; ++i) {
                diag += B[i][i] * B[i][i];
                for (int j = i + 1; j < N; ++j) off += B[i][j] * B[i][j];
            }
            if (off <= 1e-30 * diag) break;      // converged far below fp64 resolution of the eigenvalues
            for (int p = 0; p < N - 1; ++p)
                for (int q = p + 1; q < N; ++q) {
                    if (B[p][q] == 0.0) continue;
                    const double tau = (B[q][q] - B[p][p]) / (2.0 * B[p][q]);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                    for (int k = 0; k < N; ++k) {
                        const double kp = B[k][p], kq = B[k][q];
                        B[k][p] = c * kp - s * kq;
                        B[k][q] = s * kp + c * kq;
                    }
                    for (int k = 0; k < N; ++k) {
                        const double pk = B[p][k], qk = B[q][k];
                        B[p][k] = c * pk - s * qk;
                        B[q][k] = s * pk + c * qk;
                    }
                }
        }
        for (int i = 0; i < N; ++i) w[i] = B[i][i];
    }

    // returns reciprocal condition number (|lambda|min / |lambda|max); x = A^-1 b when it is usable
    // exact_rc == false (device-driven loop: one GPU thread, where 30 Jacobi sweeps would cost more than a whole pass): the
    // ratio of the smallest to the largest Cholesky pivot instead, an upper bound of |lambda|min / |lambda|max; the caller
    // hands anything suspicious back to the host's exact form.
    SYMMICP_HD double solve(const double b[N], double x[N], bool exact_rc = true) const
    {
        double rc = 0.0;
        if (exact_rc) {
            double w[N];
            eigenvalues(w);
            double wmin = (double)INFINITY, wmax = 0.0;
            for (int i = 0; i < N; ++i) {
                wmin = fmin(wmin, fabs(w[i]));
                wmax = fmax(wmax, fabs(w[i]));
            }
            rc = wmax > 0.0 ? wmin / wmax : 0.0;
        }
        // Cholesky A = L L^T; one reciprocal per pivot instead of a division per element (the device runs this in ONE thread behind every
        // pass of a device-driven loop: 33 fp64 divisions at ~12 dependent instructions each were a third of its solve)
        double L[N][N], inv[N];
        for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) L[i][j] = 0.0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < N && ok; ++j) {
            double d = A[j][j];
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            if (!(d > 0.0)) { ok = false; break; }
            L[j][j] = sqrt(d);
            inv[j] = 1.0 / L[j][j];
            for (int i = j + 1; i < N; ++i) {
                double s = A[i][j];
                for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
                L[i][j] = s * inv[j];
            }
        }
        if (!ok) {
            for (int i = 0; i < N; ++i) x[i] = (double)NAN;
            return 0.0;
        }
        if (!exact_rc) {
            double dmin = (double)INFINITY, dmax = 0.0;
            for (int i = 0; i < N; ++i) { const double d = L[i][i] * L[i][i]; dmin = fmin(dmin, d); dmax = fmax(dmax, d); }
            rc = dmax > 0.0 ? dmin / dmax : 0.0;
        }
        double y[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double s = b[i];
            for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
            y[i] = s * inv[i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
            x[i] = s * inv[i];
        }
        return rc;
    }
};

struct Blocks {
    double MtM[3][3], NtN[3][3], MtN[3][3], Mtc[3], Ntc[3];
    double sp[3], sq[3], cnt;
};

SYMMICP_HD inline Blocks unpack(const symmicp_sums &S)
{
    double G[6][6];
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) { G[r][c] = S.s[k]; G[c][r] = S.s[k]; ++k; }
    Blocks B;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { B.MtM[r][c] = G[r][c]; B.NtN[r][c] = G[r + 3][c + 3]; B.MtN[r][c] = G[r][c + 3]; }
        B.Mtc[r] = S.s[21 + r]; B.Ntc[r] = S.s[24 + r];
        B.sp[r] = S.s[27 + r]; B.sq[r] = S.s[30 + r];
    }
    B.cnt = S.s[34];
    return B;
}

// fp32 affine with Eigen's post-multiplying translate()/rotate()
struct Affine3 {
    float L[3][3];
    float T[3];
    SYMMICP_HD Affine3()
    {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) L[r][c] = (r == c) ? 1.f : 0.f; T[r] = 0.f; }
    }
    SYMMICP_HD void translate(const float v[3])
    {   // translation += linear * v
        for (int r = 0; r < 3; ++r) T[r] += (L[r][0] * v[0] + L[r][1] * v[1]) + L[r][2] * v[2];
    }
    SYMMICP_HD void rotate(const float R[3][3])
    {   // linear = linear * R
        float P[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) P[r][c] = (L[r][0] * R[0][c] + L[r][1] * R[1][c]) + L[r][2] * R[2][c];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) L[r][c] = P[r][c];
    }
    SYMMICP_HD void store(float out16[16]) const
    {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) out16[4 * r + c] = L[r][c]; out16[4 * r + 3] = T[r]; }
        out16[12] = out16[13] = out16[14] = 0.f; out16[15] = 1.f;
    }
};

// Eigen::AngleAxisf(angle, axis).toRotationMatrix()
SYMMICP_HD inline void angle_axis(float angle, const float ax[3], float R[3][3])
{
    const float s = sin(angle), c = cos(angle);
    const float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const float ca[3] = {(1.f - c) * ax[0], (1.f - c) * ax[1], (1.f - c) * ax[2]};
    float tmp = ca[0] * ax[1];
    R[0][1] = tmp - sa[2]; R[1][0] = tmp + sa[2];
    tmp = ca[0] * ax[2];
    R[0][2] = tmp + sa[1]; R[2][0] = tmp - sa[1];
    tmp = ca[1] * ax[2];
    R[1][2] = tmp - sa[0]; R[2][1] = tmp + sa[0];
    R[0][0] = ca[0] * ax[0] + c; R[1][1] = ca[1] * ax[1] + c; R[2][2] = ca[2] * ax[2] + c;
}


// func.cpp:85-99, reference arithmetic as written
SYMMICP_HD inline int solve_quirks(const symmicp_sums &S, float pbar[3], float qbar[3], float a[3], float t[3], float *rcond, float out16[16], bool exact_rc = true)
{
    const Blocks B = unpack(S);
    if (!(B.cnt > 0.0)) return SYMMICP_ERR_DEGENERATE;
    float t0[3];
    for (int k = 0; k < 3; ++k) {
        pbar[k] = (float)(B.sp[k] / B.cnt);      // colwise().mean() held as Vector3f (func.cpp:85)
        qbar[k] = (float)(B.sq[k] / B.cnt);
        t0[k] = qbar[k] - pbar[k];               // func.cpp:86
    }
    SymSolver<3> sm, sn;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { sm.A[r][c] = B.MtM[r][c]; sn.A[r][c] = B.NtN[r][c]; }
    double rhs[3], xa[3], xt[3];
    for (int r = 0; r < 3; ++r) {                // a = argmin |M a + (N t0 + c)|   (func.cpp:87)
        double s = B.Mtc[r];
        for (int c = 0; c < 3; ++c) s += B.MtN[r][c] * (double)t0[c];
        rhs[r] = -s;
    }
    const double rc1 = sm.solve(rhs, xa, exact_rc);
    for (int k = 0; k < 3; ++k) a[k] = (float)xa[k];
    for (int r = 0; r < 3; ++r) {                // t = argmin |N t + (M a + c)|    (func.cpp:88)
        double s = B.Ntc[r];
        for (int c = 0; c < 3; ++c) s += B.MtN[c][r] * (double)a[c];
        rhs[r] = -s;
    }
    const double rc2 = sn.solve(rhs, xt, exact_rc);
    for (int k = 0; k < 3; ++k) t[k] = (float)xt[k];
    const double rc = rc1 < rc2 ? rc1 : rc2;
    if (rcond) *rcond = (float)rc;

    // func.cpp:91-99: transform = T(-pbar) R T(t cos) R T(qbar) by successive post-multiplication
    const float na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    const float theta = atan(na);                                     // :93
    const float ax[3] = {a[0] / na, a[1] / na, a[2] / na};                 // :96 (NaN when na == 0, as the reference)
    float R[3][3];
    angle_axis(theta, ax, R);
    const float ct = cos(theta);
    const float mp[3] = {-pbar[0], -pbar[1], -pbar[2]};
    const float tc[3] = {t[0] * ct, t[1] * ct, t[2] * ct};
    Affine3 X;
    X.translate(mp);       // :95
    X.rotate(R);           // :96
    X.translate(tc);       // :97
    X.rotate(R);           // :98
    X.translate(qbar);     // :99
    X.store(out16);
    if (!(rc > 1e-10)) return SYMMICP_ERR_DEGENERATE;
    for (int k = 0; k < 16; ++k)
        if (!finite32(out16[k])) return SYMMICP_ERR_DEGENERATE;
    return SYMMICP_OK;
}

// Rusinkiewicz 2019 as the reference's comments intend it (func.cpp:84,94): centred rows, joint
// 6 x 6 system, T(qbar) R T(t cos) R T(-pbar).  The pass accumulated un-centred sums about
// `pivot`; the centring is applied here algebraically:
//   m~ = m - s x n, c~ = c - d.n with s = pbar+qbar, d = pbar-qbar (means about the pivot).
SYMMICP_HD inline int solve_paper(const symmicp_sums &S, const float pivot[3], float pbar[3], float qbar[3], float a[3], float t[3],
                float *rcond, float out16[16], bool exact_rc = true)
{
    const Blocks B = unpack(S);
    if (!(B.cnt >= 6.0)) return SYMMICP_ERR_DEGENERATE;
    double pb[3], qb[3], s[3], d[3];
    for (int k = 0; k < 3; ++k) { pb[k] = B.sp[k] / B.cnt; qb[k] = B.sq[k] / B.cnt; s[k] = pb[k] + qb[k]; d[k] = pb[k] - qb[k]; }
    const double K[3][3] = {{0, -s[2], s[1]}, {s[2], 0, -s[0]}, {-s[1], s[0], 0}};   // K n = s x n
    double KN[3][3], MNKt[3][3], KNKt[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double x = 0, y = 0;
            for (int k = 0; k < 3; ++k) { x += K[r][k] * B.NtN[k][c]; y += B.MtN[r][k] * K[c][k]; }
            KN[r][c] = x; MNKt[r][c] = y;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double x = 0;
            for (int k = 0; k < 3; ++k) x += KN[r][k] * K[c][k];
            KNKt[r][c] = x;
        }
    SymSolver<6> sys;
    double rhs[6], x6[6];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            sys.A[r][c] = B.MtM[r][c] - MNKt[r][c] - MNKt[c][r] + KNKt[r][c];
            sys.A[r][c + 3] = B.MtN[r][c] - KN[r][c];
            sys.A[c + 3][r] = sys.A[r][c + 3];
            sys.A[r + 3][c + 3] = B.NtN[r][c];
        }
    for (int r = 0; r < 3; ++r) {
        double x = B.Mtc[r], y = B.Ntc[r];
        for (int k = 0; k < 3; ++k) {
            x += -B.MtN[r][k] * d[k] - K[r][k] * B.Ntc[k] + KN[r][k] * d[k];
            y += -B.NtN[r][k] * d[k];
        }
        rhs[r] = -x; rhs[r + 3] = -y;
    }
    const double rc = sys.solve(rhs, x6, exact_rc);
    if (rcond) *rcond = (float)rc;
    for (int k = 0; k < 3; ++k) {
        a[k] = (float)x6[k]; t[k] = (float)x6[k + 3];
        pbar[k] = (float)(pb[k] + (pivot ? (double)pivot[k] : 0.0));
        qbar[k] = (float)(qb[k] + (pivot ? (double)pivot[k] : 0.0));
    }
    const float na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    const float theta = atan(na);
    float R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (na > 0.f) {
        const float ax[3] = {a[0] / na, a[1] / na, a[2] / na};
        angle_axis(theta, ax, R);
    }
    const float ct = cos(theta);
    const float mp[3] = {-pbar[0], -pbar[1], -pbar[2]};
    const float tc[3] = {t[0] * ct, t[1] * ct, t[2] * ct};
    Affine3 X;
    X.translate(qbar);
    X.rotate(R);
    X.translate(tc);
    X.rotate(R);
    X.translate(mp);
    X.store(out16);
    if (!(rc > 1e-12)) return SYMMICP_ERR_DEGENERATE;
    for (int k = 0; k < 16; ++k)
        if (!finite32(out16[k])) return SYMMICP_ERR_DEGENERATE;
    return SYMMICP_OK;
}

// Closed-form rigid fit of the current pairs (reference ICP/regist.h:8-72, registrateNPoint):
//   H = sum (p - pbar)(q - qbar)^T = sum p q^T - n pbar qbar^T ;  H = U W V^T ;
//   R = V diag(1, 1, det(V U^T)) U^T ;  T = qbar - R pbar.
// The record holds sum p q^T (slots 0..8) and the coordinate sums about `pivot`.  The 3x3 SVD is taken through the
// symmetric eigen-problem of H^T H (Jacobi rotations with eigenvectors).
SYMMICP_HD inline int solve_p2p(const symmicp_sums &S, const float pivot[3], float *rcond, float out16[16])
{
    const double n = S.s[34];
    if (!(n >= 3.0)) return SYMMICP_ERR_DEGENERATE;
    double pb[3], qb[3], H[3][3];
    for (int k = 0; k < 3; ++k) { pb[k] = S.s[27 + k] / n; qb[k] = S.s[30 + k] / n; }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) H[r][c] = S.s[3 * r + c] - n * pb[r] * qb[c];
    // A = H^T H, eigen-decomposition A = V diag(w) V^T
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = H[0][r] * H[0][c] + H[1][r] * H[1][c] + H[2][r] * H[2][c];
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double tau = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) { const double kp = A[k][p], kq = A[k][q]; A[k][p] = c * kp - s * kq; A[k][q] = s * kp + c * kq; }
                for (int k = 0; k < 3; ++k) { const double pk = A[p][k], qk = A[q][k]; A[p][k] = c * pk - s * qk; A[q][k] = s * pk + c * qk; }
                for (int k = 0; k < 3; ++k) { const double kp = V[k][p], kq = V[k][q]; V[k][p] = c * kp - s * kq; V[k][q] = s * kp + c * kq; }
            }
    }
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (A[ord[b]][ord[b]] > A[ord[a]][ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double sig[3], Vs[3][3], U[3][3];
    for (int k = 0; k < 3; ++k) {
        const double w = A[ord[k]][ord[k]];
        sig[k] = sqrt(w > 0.0 ? w : 0.0);
        for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][ord[k]];
    }
    if (rcond) *rcond = sig[0] > 0.0 ? (float)(sig[1] / sig[0]) : 0.f;
    if (!(sig[1] > 1e-12 * sig[0])) return SYMMICP_ERR_DEGENERATE;           // collinear pairs: rotation undetermined
    for (int k = 0; k < 2; ++k)
        for (int r = 0; r < 3; ++r) U[r][k] = (H[r][0] * Vs[0][k] + H[r][1] * Vs[1][k] + H[r][2] * Vs[2][k]) / sig[k];
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    double M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + Vs[r][2] * U[c][2];
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    const double d = det < 0.0 ? -1.0 : 1.0;                                     // regist.h:57-61
    double R[3][3], cs[3], cd[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + d * Vs[r][2] * U[c][2];
    for (int k = 0; k < 3; ++k) { cs[k] = pb[k] + (pivot ? (double)pivot[k] : 0.0); cd[k] = qb[k] + (pivot ? (double)pivot[k] : 0.0); }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out16[4 * r + c] = (float)R[r][c];
        out16[4 * r + 3] = (float)(cd[r] - (R[r][0] * cs[0] + R[r][1] * cs[1] + R[r][2] * cs[2]));   // regist.h:65-67
    }
    out16[12] = out16[13] = out16[14] = 0.f; out16[15] = 1.f;
    for (int k = 0; k < 16; ++k)
        if (!finite32(out16[k])) return SYMMICP_ERR_DEGENERATE;
    return SYMMICP_OK;
}

// transform = incre * transform (myicp.cpp:138), fp32, k sequential
SYMMICP_HD inline void mat4_mul(const float A[16], const float B[16], float C[16])
{
    float T[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = A[4 * r] * B[c];
            s = s + A[4 * r + 1] * B[4 + c];
            s = s + A[4 * r + 2] * B[8 + c];
            s = s + A[4 * r + 3] * B[12 + c];
            T[4 * r + c] = s;
        }
    for (int k = 0; k < 16; ++k) C[k] = T[k];
}

}  // namespace solve
}  // namespace symmicp
