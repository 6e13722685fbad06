/*
 * symmicp_oracle.c -- CPU restatement of the StephenNG59/ICP-symm hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker for the HIP product
 * path (icp-symm_amd/csrc); nothing under icp-symm_amd/ or include/ may link,
 * import or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * What it restates (reference file:line, all under /root/reference/ICP):
 *   orc_rows              func.cpp:43-60   calculateMatrixNotation
 *   orc_eval_diff         func.cpp:19-32   evalDiff(src,tgt)
 *   orc_lls_svd           func.cpp:64-73   solveLLS (thin SVD pseudo-inverse)
 *   orc_estimate_quirks*  func.cpp:76-102  estimateTransformSymm
 *   orc_apply             func.cpp:104-121 applyTransform
 *   orc_align             myicp.cpp:100-150 RegisterSymm loop
 *   orc_normals_knn       myicp.cpp:152-172 estimateNormals (PCL k=10 PCA)
 *   orc_pcd_read          myicp.cpp:20-31  LoadCloud (ASCII/binary PCD v0.7)
 *   orc_kabsch / orc_solve_p2p   regist.h:8-72   registrateNPoint (closed-form rigid fit)
 * plus the capability the reference leaves as a todo (myicp.cpp:128-131,
 * func.cpp:36-40): exact nearest-neighbour correspondence (orc_nn_brute,
 * orc_nn_grid) and the paper-correct solve the reference's own comments
 * describe (func.cpp:84,94; myicp.cpp:220-225) -- orc_solve_paper.
 *
 * PINNING.  The reference cannot be compiled here (it needs PCL 1.9.1, Eigen3,
 * FLANN, VTK, OpenCV; none present) and it ships no tests or golden outputs.
 * The pins this oracle is checked against (tests/test_oracle_pins.py) are the
 * data-level known answers the reference's own files imply:
 *   (1) cat_out.pcd == Rz(pi/4) * cat.pcd + (2.5,0,0), same row order
 *       (generator: main.cpp:43-52, matrix-transform.cpp:82-114);
 *   (2) initial evalDiff(cat, cat_out) == 99242.67;
 *   (3) txt2pcd_bunny1.pcd == za.txt (93 collinear points).
 * Third-party arithmetic whose source is not under /root/reference --
 * Eigen3 JacobiSVD / Affine3f / AngleAxisf (unpinned version bundled with the
 * PCL 1.9.1 Windows all-in-one) and PCL 1.9.1 NormalEstimation / PCDReader --
 * is restated from its published algorithm.  For those boundaries (SVD
 * rounding, PCA-normal rounding) there is no reference output to compare
 * with: PARITY UNPINNED there, pinned only through (1)-(3).
 *
 * Arithmetic conventions (shared, by definition, with the HIP path):
 *   - per-point math is IEEE fp32 with NO fused multiply-add
 *     (compile with -ffp-contract=off), sums are fp64;
 *   - affine apply:  x' = ((X00*x + X01*y) + X02*z) + X03   (row-major X)
 *   - distance:      d2 = (dx*dx + dy*dy) + dz*dz
 *   - NN ties: lowest target index wins.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <ctype.h>

#define ORC_NSUM 40

/* bench.py's all-core CPU baseline: the two O(N) loops of an iteration (exact NN through the grid, the 40-sum reduction)
 * split over this many threads (OpenMP).  1 (default) = the serial code every parity test runs. */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 1 ? n : 1; }
int orc_get_threads(void) { return orc_threads; }

enum { ORC_MODE_QUIRKS = 0, ORC_MODE_PAPER = 1, ORC_MODE_P2P = 2 };
enum { ORC_CORR_IDENTITY = 0, ORC_CORR_BRUTE = 1, ORC_CORR_GRID = 2 };
enum { ORC_SOLVE_GRAM = 0, ORC_SOLVE_LITERAL = 1 };
enum { ORC_APPLY_INCREMENTAL = 0, ORC_APPLY_CUMULATIVE = 1 };
enum { ORC_OK = 0, ORC_ERR_ARG = 1, ORC_ERR_SIZE = 2, ORC_ERR_DEGENERATE = 3, ORC_ERR_IO = 4 };

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

static inline void xform_pt(const float X[16], const float *p, float *o, int with_t)
{
    /* func.cpp:111-118: [x y z 1] padded, X * v; sequential k order, no FMA */
    float w = with_t ? 1.0f : 0.0f;
    float x = p[0], y = p[1], z = p[2];
    o[0] = ((X[0] * x + X[1] * y) + X[2] * z) + X[3] * w;
    o[1] = ((X[4] * x + X[5] * y) + X[6] * z) + X[7] * w;
    o[2] = ((X[8] * x + X[9] * y) + X[10] * z) + X[11] * w;
}

static inline float dist2f(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

static void mat4_mul_f(const float A[16], const float B[16], float C[16])
{
    /* plain fp32 4x4 product, k sequential (Eigen Affine3f * Affine3f) */
    float T[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = A[r * 4 + 0] * B[0 * 4 + c];
            s = s + A[r * 4 + 1] * B[1 * 4 + c];
            s = s + A[r * 4 + 2] * B[2 * 4 + c];
            s = s + A[r * 4 + 3] * B[3 * 4 + c];
            T[r * 4 + c] = s;
        }
    memcpy(C, T, sizeof(T));
}

static void mat4_identity(float X[16])
{
    memset(X, 0, 16 * sizeof(float));
    X[0] = X[5] = X[10] = X[15] = 1.0f;
}

/* ------------------------------------------------------------------ */
/* func.cpp:43-60  rows M_i, N_i, c_i (fp32, materialised)             */
/* ------------------------------------------------------------------ */
static inline void row_f32(const float *p, const float *np, const float *q, const float *nq,
                           float m[3], float n[3], float *c)
{
    n[0] = np[0] + nq[0]; n[1] = np[1] + nq[1]; n[2] = np[2] + nq[2];   /* :51 */
    float sx = p[0] + q[0], sy = p[1] + q[1], sz = p[2] + q[2];
    float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    m[0] = sy * n[2] - sz * n[1];                                       /* :54 cross */
    m[1] = sz * n[0] - sx * n[2];
    m[2] = sx * n[1] - sy * n[0];
    *c = (dx * n[0] + dy * n[1]) + dz * n[2];                           /* :58 dot */
}

void orc_rows(const float *p, const float *np, const float *q, const float *nq, size_t n,
              float *M, float *N, float *c)
{
    for (size_t i = 0; i < n; i++)
        row_f32(p + 3 * i, np + 3 * i, q + 3 * i, nq + 3 * i, M + 3 * i, N + 3 * i, c + i);
}

/* func.cpp:19-32: serial fp32 accumulation of per-pair L2 norms */
float orc_eval_diff(const float *p, const float *q, size_t n)
{
    float diff = 0.f;
    for (size_t i = 0; i < n; i++)
        diff += sqrtf(dist2f(p + 3 * i, q + 3 * i));
    return diff;
}

double orc_eval_diff_f64(const float *p, const float *q, size_t n)
{
    double diff = 0.0;
    for (size_t i = 0; i < n; i++)
        diff += (double)sqrtf(dist2f(p + 3 * i, q + 3 * i));
    return diff;
}

/* func.cpp:104-121 applyTransform on an N x 3 AoS array, in place allowed.
 * with_translation=1 is what the reference does for points AND normals
 * (myicp.cpp:136-137). */
void orc_apply(const float X[16], const float *src, float *dst, size_t n, int with_translation)
{
    for (size_t i = 0; i < n; i++) {
        float o[3];
        xform_pt(X, src + 3 * i, o, with_translation);
        dst[3 * i] = o[0]; dst[3 * i + 1] = o[1]; dst[3 * i + 2] = o[2];
    }
}

/* ------------------------------------------------------------------ */
/* The 40-sum reduction.                    */
/* p, np are the CURRENT source points/normals (already transformed).   */
/* idx == NULL -> identity pairing (myicp.cpp:130); idx[i] < 0 -> skip. */
/* pivot is subtracted (fp32) from p and q before forming rows; the     */
/* reference does no centring (func.cpp:51-58) -> pivot = 0 in QUIRKS.  */
/* Layout: [0..20] upper triangle (row-major) of sum v v^T with         */
/* v=(m,n); [21..26] sum v c; [27..29] sum p; [30..32] sum q;           */
/* [33] sum |p-q|; [34] count; [35] sum c^2; [36] sum |p-q|^2.          */
/* ------------------------------------------------------------------ */
static void reduce_range(const float *p, const float *np, size_t i0, size_t i1, const float *q, const float *nq,
                         const int32_t *idx, const float pv[3], float max_d2, float min_ndot, int p2p, double S[ORC_NSUM])
{
    for (int k = 0; k < ORC_NSUM; k++) S[k] = 0.0;
    for (size_t i = i0; i < i1; i++) {
        long j = idx ? (long)idx[i] : (long)i;
        if (j < 0) continue;
        const float *pi = p + 3 * i, *qj = q + 3 * j;
        float d2 = dist2f(pi, qj);
        if (max_d2 > 0.f && d2 > max_d2) continue;
        if (min_ndot > -1.0f) {
            const float *a = np + 3 * i, *b = nq + 3 * j;
            if ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2] < min_ndot) continue;
        }
        float pc[3] = {pi[0] - pv[0], pi[1] - pv[1], pi[2] - pv[2]};
        float qc[3] = {qj[0] - pv[0], qj[1] - pv[1], qj[2] - pv[2]};
        float m[3], nn[3], c;
        row_f32(pc, np + 3 * i, qc, nq + 3 * j, m, nn, &c);
        if (p2p) {
            /* point-to-point (regist.h:52): slots 0..8 = sum p q^T about the pivot, row-major */
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) S[3 * a + b] += (double)pc[a] * (double)qc[b];
        } else {
            double v[6] = {m[0], m[1], m[2], nn[0], nn[1], nn[2]};
            int k = 0;
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++)
                    S[k++] += v[a] * v[b];
            for (int a = 0; a < 6; a++) S[21 + a] += v[a] * (double)c;
        }
        for (int a = 0; a < 3; a++) { S[27 + a] += (double)pc[a]; S[30 + a] += (double)qc[a]; }
        S[33] += (double)sqrtf(d2);
        S[34] += 1.0;
        if (!p2p) S[35] += (double)c * (double)c;
        S[36] += (double)d2;
    }
}

void orc_reduce40(const float *p, const float *np, size_t n_s,
                  const float *q, const float *nq, size_t n_t,
                  const int32_t *idx, const float pivot[3], float max_d2, float min_ndot, int p2p, double S[ORC_NSUM])
{
    (void)n_t;
    float pv[3] = {0, 0, 0};
    if (pivot) { pv[0] = pivot[0]; pv[1] = pivot[1]; pv[2] = pivot[2]; }
    const int T = orc_threads;
    if (T <= 1) { reduce_range(p, np, 0, n_s, q, nq, idx, pv, max_d2, min_ndot, p2p, S); return; }
    /* all-core baseline: contiguous chunks, partial records added in chunk order (deterministic for a given thread count) */
    double *part = (double *)malloc(sizeof(double) * ORC_NSUM * (size_t)T);
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++)
        reduce_range(p, np, n_s * (size_t)t / (size_t)T, n_s * (size_t)(t + 1) / (size_t)T, q, nq, idx, pv, max_d2, min_ndot, p2p, part + (size_t)t * ORC_NSUM);
    for (int k = 0; k < ORC_NSUM; k++) { double v = 0.0; for (int t = 0; t < T; t++) v += part[(size_t)t * ORC_NSUM + k]; S[k] = v; }
    free(part);
}

/* ------------------------------------------------------------------ */
/* Small dense fp64 linear algebra                                     */
/* ------------------------------------------------------------------ */

/* cyclic Jacobi eigen-decomposition of a symmetric n x n (n<=6) matrix.
 * A is overwritten; eigenvalues in w, eigenvectors in columns of V. */
static void jacobi_eig(int n, double *A, double *V, double *w)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* solve sym pos-def A x = b through its eigen-decomposition (pseudo-inverse
 * semantics of func.cpp:70 for full rank); returns min/max eigenvalue ratio. */
static double sym_solve(int n, const double *A, const double *b, double *x)
{
    double T[36], V[36], w[6];
    memcpy(T, A, sizeof(double) * n * n);
    jacobi_eig(n, T, V, w);
    double wmax = 0, wmin = DBL_MAX;
    for (int i = 0; i < n; i++) { if (fabs(w[i]) > wmax) wmax = fabs(w[i]); if (fabs(w[i]) < wmin) wmin = fabs(w[i]); }
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        double proj = 0;
        for (int i = 0; i < n; i++) proj += V[i * n + k] * b[i];
        double inv = (w[k] != 0.0) ? 1.0 / w[k] : INFINITY; /* no threshold: func.cpp:70 */
        for (int i = 0; i < n; i++) x[i] += V[i * n + k] * proj * inv;
    }
    return wmax > 0 ? wmin / wmax : 0.0;
}

static void unpack_gram(const double S[ORC_NSUM], double G[36], double b[6])
{
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int c = a; c < 6; c++) { G[a * 6 + c] = S[k]; G[c * 6 + a] = S[k]; k++; }
    for (int a = 0; a < 6; a++) b[a] = S[21 + a];
}

/* ------------------------------------------------------------------ */
/* func.cpp:64-73 solveLLS, literal: thin SVD of an N x 3 fp32 matrix   */
/* (one-sided Jacobi), x = V S^-1 U^T b, no singular value threshold.   */
/* ------------------------------------------------------------------ */
int orc_lls_svd(const float *A, const float *b, size_t n, float x[3])
{
    float *U = (float *)malloc(sizeof(float) * 3 * n);
    if (!U) return ORC_ERR_ARG;
    /* U stored column-major: U[k*n + i] */
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) U[(size_t)k * n + i] = A[3 * i + k];
    float V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                float *up = U + (size_t)p * n, *uq = U + (size_t)q * n;
                float alpha = 0, beta = 0, gamma = 0;
                for (size_t i = 0; i < n; i++) { alpha += up[i] * up[i]; beta += uq[i] * uq[i]; gamma += up[i] * uq[i]; }
                if (fabsf(gamma) <= 1e-7f * sqrtf(alpha * beta) || gamma == 0.f) continue;
                rotated = 1;
                float zeta = (beta - alpha) / (2.0f * gamma);
                float t = (zeta >= 0 ? 1.0f : -1.0f) / (fabsf(zeta) + sqrtf(1.0f + zeta * zeta));
                float c = 1.0f / sqrtf(1.0f + t * t), s = c * t;
                for (size_t i = 0; i < n; i++) {
                    float a0 = up[i], a1 = uq[i];
                    up[i] = c * a0 - s * a1;
                    uq[i] = s * a0 + c * a1;
                }
                for (int k = 0; k < 3; k++) {
                    float v0 = V[k * 3 + p], v1 = V[k * 3 + q];
                    V[k * 3 + p] = c * v0 - s * v1;
                    V[k * 3 + q] = s * v0 + c * v1;
                }
            }
        if (!rotated) break;
    }
    float y[3];
    for (int k = 0; k < 3; k++) {
        float *uk = U + (size_t)k * n;
        float s2 = 0, ub = 0;
        for (size_t i = 0; i < n; i++) { s2 += uk[i] * uk[i]; ub += uk[i] * b[i]; }
        /* sigma = sqrt(s2); U_k = uk/sigma; y = (U_k^T b)/sigma = ub / s2 */
        y[k] = ub / s2; /* inf/NaN for a zero singular value, as func.cpp:70 */
    }
    for (int i = 0; i < 3; i++) x[i] = V[i * 3 + 0] * y[0] + V[i * 3 + 1] * y[1] + V[i * 3 + 2] * y[2];
    free(U);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* func.cpp:90-99 composition, literal.  Eigen Affine3f::translate and  */
/* ::rotate POST-multiply, so the matrix built is                       */
/*   T(-pbar) * R(theta,a) * T(t*cos(theta)) * R(theta,a) * T(qbar).    */
/* AngleAxisf -> matrix as Eigen::AngleAxis::toRotationMatrix.          */
/* ------------------------------------------------------------------ */
static void angle_axis_f(float angle, const float ax[3], float R[9])
{
    float s = sinf(angle), c = cosf(angle);
    float sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
    float ca[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
    float tmp;
    tmp = ca[0] * ax[1]; R[0 * 3 + 1] = tmp - sa[2]; R[1 * 3 + 0] = tmp + sa[2];
    tmp = ca[0] * ax[2]; R[0 * 3 + 2] = tmp + sa[1]; R[2 * 3 + 0] = tmp - sa[1];
    tmp = ca[1] * ax[2]; R[1 * 3 + 2] = tmp - sa[0]; R[2 * 3 + 1] = tmp + sa[0];
    R[0] = ca[0] * ax[0] + c; R[4] = ca[1] * ax[1] + c; R[8] = ca[2] * ax[2] + c;
}

static void post_translate_f(float X[16], const float v[3])
{
    /* Eigen: translation() += linear() * v */
    for (int r = 0; r < 3; r++)
        X[r * 4 + 3] += (X[r * 4 + 0] * v[0] + X[r * 4 + 1] * v[1]) + X[r * 4 + 2] * v[2];
}

static void post_rotate_f(float X[16], const float R[9])
{
    float L[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            L[r * 3 + c] = (X[r * 4 + 0] * R[0 * 3 + c] + X[r * 4 + 1] * R[1 * 3 + c]) + X[r * 4 + 2] * R[2 * 3 + c];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) X[r * 4 + c] = L[r * 3 + c];
}

void orc_compose_quirks(const float pbar[3], const float qbar[3], const float a[3], const float t[3], float X[16])
{
    mat4_identity(X);
    float na = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    float theta = atanf(na);                                   /* :93 */
    float ax[3] = {a[0] / na, a[1] / na, a[2] / na};           /* :96 (NaN if na==0) */
    float R[9]; angle_axis_f(theta, ax, R);
    float mp[3] = {-pbar[0], -pbar[1], -pbar[2]};
    float ct = cosf(theta);
    float tc[3] = {t[0] * ct, t[1] * ct, t[2] * ct};
    post_translate_f(X, mp);                                   /* :95 */
    post_rotate_f(X, R);                                       /* :96 */
    post_translate_f(X, tc);                                   /* :97 */
    post_rotate_f(X, R);                                       /* :98 */
    post_translate_f(X, qbar);                                 /* :99 */
}

/* the order the reference's own comment intends (func.cpp:94):
 *   T(qbar) * R * T(t cos) * R * T(-pbar)                              */
void orc_compose_paper(const float pbar[3], const float qbar[3], const float a[3], const float t[3], float X[16])
{
    mat4_identity(X);
    float na = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    float theta = atanf(na);
    float R[9];
    if (na > 0.f) {
        float ax[3] = {a[0] / na, a[1] / na, a[2] / na};
        angle_axis_f(theta, ax, R);
    } else {
        memset(R, 0, sizeof(R)); R[0] = R[4] = R[8] = 1.f;
    }
    float mp[3] = {-pbar[0], -pbar[1], -pbar[2]};
    float ct = cosf(theta);
    float tc[3] = {t[0] * ct, t[1] * ct, t[2] * ct};
    post_translate_f(X, qbar);
    post_rotate_f(X, R);
    post_translate_f(X, tc);
    post_rotate_f(X, R);
    post_translate_f(X, mp);
}

/* ------------------------------------------------------------------ */
/* func.cpp:76-102 estimateTransformSymm -- Gram route (what the HIP     */
/* path does: the two N x 3 least squares become 3 x 3 block systems).   */
/*   t0 = qbar - pbar                                 :86                */
/*   a  = argmin |M a + (N t0 + c)|                   :87                */
/*   t  = argmin |N t + (M a + c)|                    :88                */
/* ------------------------------------------------------------------ */
int orc_solve_quirks_gram(const double S[ORC_NSUM], float pbar[3], float qbar[3], float a_out[3], float t_out[3], double *rcond)
{
    double G[36], b[6];
    unpack_gram(S, G, b);
    double cnt = S[34];
    if (!(cnt > 0)) return ORC_ERR_DEGENERATE;
    double t0[3];
    for (int k = 0; k < 3; k++) {
        /* the reference holds means as Vector3f */
        pbar[k] = (float)(S[27 + k] / cnt);
        qbar[k] = (float)(S[30 + k] / cnt);
    }
    for (int k = 0; k < 3; k++) t0[k] = (double)(float)(qbar[k] - pbar[k]);
    double MtM[9], NtN[9], MtN[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            MtM[r * 3 + c] = G[r * 6 + c];
            NtN[r * 3 + c] = G[(r + 3) * 6 + (c + 3)];
            MtN[r * 3 + c] = G[r * 6 + (c + 3)];
        }
    double rhs[3], a[3], t[3];
    for (int r = 0; r < 3; r++) {
        double s = b[r];
        for (int c = 0; c < 3; c++) s += MtN[r * 3 + c] * t0[c];
        rhs[r] = -s;
    }
    double rc1 = sym_solve(3, MtM, rhs, a);
    float af[3] = {(float)a[0], (float)a[1], (float)a[2]};   /* a_ is a Vector3f */
    for (int r = 0; r < 3; r++) {
        double s = b[3 + r];
        for (int c = 0; c < 3; c++) s += MtN[c * 3 + r] * (double)af[c]; /* N^T M a */
        rhs[r] = -s;
    }
    double rc2 = sym_solve(3, NtN, rhs, t);
    for (int k = 0; k < 3; k++) { a_out[k] = af[k]; t_out[k] = (float)t[k]; }
    if (rcond) *rcond = rc1 < rc2 ? rc1 : rc2;
    if (!(rc1 > 1e-10) || !(rc2 > 1e-10)) return ORC_ERR_DEGENERATE;
    return ORC_OK;
}

/* literal route: materialise M,N,c in fp32, fp32 column means, two SVD LLS */
int orc_solve_quirks_literal(const float *p, const float *np, const float *q, const float *nq, size_t n,
                             float pbar[3], float qbar[3], float a[3], float t[3])
{
    float *M = (float *)malloc(sizeof(float) * 3 * n), *N = (float *)malloc(sizeof(float) * 3 * n);
    float *c = (float *)malloc(sizeof(float) * n), *rhs = (float *)malloc(sizeof(float) * n);
    if (!M || !N || !c || !rhs) return ORC_ERR_ARG;
    orc_rows(p, np, q, nq, n, M, N, c);
    float sp[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { sp[k] += p[3 * i + k]; sq[k] += q[3 * i + k]; }
    for (int k = 0; k < 3; k++) { pbar[k] = sp[k] / (float)n; qbar[k] = sq[k] / (float)n; }
    float t0[3] = {qbar[0] - pbar[0], qbar[1] - pbar[1], qbar[2] - pbar[2]};
    for (size_t i = 0; i < n; i++)
        rhs[i] = -(((N[3 * i] * t0[0] + N[3 * i + 1] * t0[1]) + N[3 * i + 2] * t0[2]) + c[i]);
    orc_lls_svd(M, rhs, n, a);
    for (size_t i = 0; i < n; i++)
        rhs[i] = -(((M[3 * i] * a[0] + M[3 * i + 1] * a[1]) + M[3 * i + 2] * a[2]) + c[i]);
    orc_lls_svd(N, rhs, n, t);
    free(M); free(N); free(c); free(rhs);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Paper-correct solve (Rusinkiewicz 2019; func.cpp:84,94 comments and   */
/* the demeaning of the dead draft myicp.cpp:220-225): joint 6 x 6 on    */
/* centred data.  Sums were taken about `pivot`; the remaining offset    */
/* (pbar, qbar about the pivot) is removed algebraically:                */
/*   m~ = m - s x n,  c~ = c - d.n,  s = pbar+qbar, d = pbar-qbar.       */
/* ------------------------------------------------------------------ */
int orc_solve_paper(const double S[ORC_NSUM], const float pivot[3],
                    float pbar_out[3], float qbar_out[3], float a_out[3], float t_out[3], double *rcond)
{
    double G[36], b[6];
    unpack_gram(S, G, b);
    double cnt = S[34];
    if (!(cnt >= 6)) return ORC_ERR_DEGENERATE;
    double pb[3], qb[3], s[3], d[3];
    for (int k = 0; k < 3; k++) { pb[k] = S[27 + k] / cnt; qb[k] = S[30 + k] / cnt; s[k] = pb[k] + qb[k]; d[k] = pb[k] - qb[k]; }
    /* K = [s]x  (K n = s x n) */
    double K[9] = {0, -s[2], s[1], s[2], 0, -s[0], -s[1], s[0], 0};
    double MtM[9], NtN[9], MtN[9], Mtc[3], Ntc[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) {
            MtM[r * 3 + c] = G[r * 6 + c];
            NtN[r * 3 + c] = G[(r + 3) * 6 + (c + 3)];
            MtN[r * 3 + c] = G[r * 6 + (c + 3)];
        }
        Mtc[r] = b[r]; Ntc[r] = b[3 + r];
    }
    /* helpers */
    double KN[9], KNKt[9], MNKt[9];            /* K NtN ; K NtN K^T ; MtN K^T */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double x = 0, y = 0;
            for (int k = 0; k < 3; k++) { x += K[r * 3 + k] * NtN[k * 3 + c]; y += MtN[r * 3 + k] * K[c * 3 + k]; }
            KN[r * 3 + c] = x; MNKt[r * 3 + c] = y;
        }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double x = 0;
            for (int k = 0; k < 3; k++) x += KN[r * 3 + k] * K[c * 3 + k];
            KNKt[r * 3 + c] = x;
        }
    double Gc[36], bc[6];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            /* M~^T M~ = MtM - MtN K^T - K NtM + K NtN K^T */
            Gc[r * 6 + c] = MtM[r * 3 + c] - MNKt[r * 3 + c] - MNKt[c * 3 + r] + KNKt[r * 3 + c];
            /* M~^T N = MtN - K NtN */
            Gc[r * 6 + (c + 3)] = MtN[r * 3 + c] - KN[r * 3 + c];
            Gc[(c + 3) * 6 + r] = Gc[r * 6 + (c + 3)];
            Gc[(r + 3) * 6 + (c + 3)] = NtN[r * 3 + c];
        }
    for (int r = 0; r < 3; r++) {
        /* M~^T c~ = Mtc - MtN d - K Ntc + K NtN d ;  N^T c~ = Ntc - NtN d */
        double x = Mtc[r], y = Ntc[r];
        for (int k = 0; k < 3; k++) {
            x += -MtN[r * 3 + k] * d[k] - K[r * 3 + k] * Ntc[k] + KN[r * 3 + k] * d[k];
            y += -NtN[r * 3 + k] * d[k];
        }
        bc[r] = -x; bc[3 + r] = -y;
    }
    double x6[6];
    double rc = sym_solve(6, Gc, bc, x6);
    if (rcond) *rcond = rc;
    for (int k = 0; k < 3; k++) {
        a_out[k] = (float)x6[k]; t_out[k] = (float)x6[3 + k];
        pbar_out[k] = (float)(pb[k] + (pivot ? (double)pivot[k] : 0.0));
        qbar_out[k] = (float)(qb[k] + (pivot ? (double)pivot[k] : 0.0));
    }
    if (!(rc > 1e-12)) return ORC_ERR_DEGENERATE;
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Correspondence (the reference's todo, myicp.cpp:128-131).            */
/* idx[i] = argmin_j |X p_i - q_j|^2, ties -> lowest j.                  */
/* ------------------------------------------------------------------ */
void orc_nn_brute(const float X[16], const float *p, size_t n_s, const float *q, size_t n_t,
                  int32_t *idx, float *d2_out)
{
    for (size_t i = 0; i < n_s; i++) {
        float pt[3];
        if (X) xform_pt(X, p + 3 * i, pt, 1); else { pt[0] = p[3 * i]; pt[1] = p[3 * i + 1]; pt[2] = p[3 * i + 2]; }
        float best = INFINITY; int32_t bj = -1;
        for (size_t j = 0; j < n_t; j++) {
            float d2 = dist2f(pt, q + 3 * j);
            if (d2 < best) { best = d2; bj = (int32_t)j; }
        }
        idx[i] = bj;
        if (d2_out) d2_out[i] = best;
    }
}

/* exact NN through a uniform grid (counting sort by cell, expanding cubic
 * shells).  Same result as orc_nn_brute bit for bit: candidate distances use
 * the same fp32 expression, ties -> lowest j, and a shell search only stops
 * once the best distance is strictly inside the proven-empty margin. */
typedef struct {
    float lo[3], h, inv_h;
    int dim[3];
    size_t n;
    uint32_t *cell_start;  /* ncell + 1 */
    int32_t *order;        /* point ids sorted by cell, ascending id inside a cell */
} orc_grid;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

orc_grid *orc_grid_build(const float *q, size_t n, float pts_per_cell)
{
    if (n == 0) return NULL;
    orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { float v = q[3 * i + k]; if (v < lo[k]) lo[k] = v; if (v > hi[k]) hi[k] = v; }
    double ext[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    double emax = ext[0] > ext[1] ? ext[0] : ext[1]; if (ext[2] > emax) emax = ext[2];
    if (emax <= 0) emax = 1.0;
    /* pick h so that the occupied-cell estimate ~ n / pts_per_cell, assuming a 2-D manifold-ish
       cloud falls between area and volume scaling; refine by trying a few sizes */
    double vol = 1; int nd = 0;
    for (int k = 0; k < 3; k++) if (ext[k] > 1e-6 * emax) { vol *= ext[k]; nd++; }
    if (nd == 0) { nd = 1; vol = emax; }
    double h = pow(vol * pts_per_cell / (double)n, 1.0 / nd);
    if (h < emax / 1024.0) h = emax / 1024.0;
    for (int it = 0; it < 8; it++) {
        for (int k = 0; k < 3; k++) { g->dim[k] = (int)floor(ext[k] / h) + 1; }
        double ncell = (double)g->dim[0] * g->dim[1] * g->dim[2];
        if (ncell <= 64e6) break;
        h *= 1.26;
    }
    g->h = (float)h; g->inv_h = (float)(1.0 / h);
    for (int k = 0; k < 3; k++) g->lo[k] = lo[k];
    g->n = n;
    size_t ncell = (size_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->cell_start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    g->order = (int32_t *)malloc(sizeof(int32_t) * n);
    uint32_t *cid = (uint32_t *)malloc(sizeof(uint32_t) * n);
    for (size_t i = 0; i < n; i++) {
        int c[3];
        for (int k = 0; k < 3; k++) c[k] = clampi((int)floorf((q[3 * i + k] - g->lo[k]) * g->inv_h), 0, g->dim[k] - 1);
        cid[i] = (uint32_t)(((size_t)c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0]);
        g->cell_start[cid[i] + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) g->cell_start[c + 1] += g->cell_start[c];
    uint32_t *fill = (uint32_t *)malloc(sizeof(uint32_t) * ncell);
    memcpy(fill, g->cell_start, sizeof(uint32_t) * ncell);
    for (size_t i = 0; i < n; i++) g->order[fill[cid[i]]++] = (int32_t)i; /* ascending id inside a cell */
    free(fill); free(cid);
    return g;
}

void orc_grid_free(orc_grid *g)
{
    if (!g) return;
    free(g->cell_start); free(g->order); free(g);
}

static inline void grid_visit(const orc_grid *g, const float *q, int cx, int cy, int cz, const float *pt,
                              float *best, int32_t *bj)
{
    size_t c = ((size_t)cz * g->dim[1] + cy) * g->dim[0] + cx;
    for (uint32_t k = g->cell_start[c]; k < g->cell_start[c + 1]; k++) {
        int32_t j = g->order[k];
        float d2 = dist2f(pt, q + 3 * (size_t)j);
        if (d2 < *best || (d2 == *best && j < *bj)) { *best = d2; *bj = j; }
    }
}

void orc_nn_grid(const orc_grid *g, const float X[16], const float *p, size_t n_s, const float *q,
                 int32_t *idx, float *d2_out)
{
    int maxdim = g->dim[0] > g->dim[1] ? g->dim[0] : g->dim[1]; if (g->dim[2] > maxdim) maxdim = g->dim[2];
    /* queries are independent: the all-core baseline splits them over threads (results identical to the serial loop) */
#pragma omp parallel for num_threads(orc_threads) schedule(dynamic, 4096) if (orc_threads > 1)
    for (size_t i = 0; i < n_s; i++) {
        float pt[3];
        if (X) xform_pt(X, p + 3 * i, pt, 1); else { pt[0] = p[3 * i]; pt[1] = p[3 * i + 1]; pt[2] = p[3 * i + 2]; }
        int c[3];
        for (int k = 0; k < 3; k++) c[k] = clampi((int)floorf((pt[k] - g->lo[k]) * g->inv_h), 0, g->dim[k] - 1);
        float best = INFINITY; int32_t bj = -1;
        for (int r = 0; r <= maxdim; r++) {
            int x0 = c[0] - r, x1 = c[0] + r, y0 = c[1] - r, y1 = c[1] + r, z0 = c[2] - r, z1 = c[2] + r;
            for (int z = (z0 < 0 ? 0 : z0); z <= (z1 >= g->dim[2] ? g->dim[2] - 1 : z1); z++)
                for (int y = (y0 < 0 ? 0 : y0); y <= (y1 >= g->dim[1] ? g->dim[1] - 1 : y1); y++) {
                    int shell_zy = (z == z0 || z == z1 || y == y0 || y == y1);
                    if (shell_zy) {
                        for (int x = (x0 < 0 ? 0 : x0); x <= (x1 >= g->dim[0] ? g->dim[0] - 1 : x1); x++)
                            grid_visit(g, q, x, y, z, pt, &best, &bj);
                    } else {
                        if (x0 >= 0) grid_visit(g, q, x0, y, z, pt, &best, &bj);
                        if (x1 < g->dim[0] && x1 != x0) grid_visit(g, q, x1, y, z, pt, &best, &bj);
                    }
                }
            /* everything in cells [c-r, c+r] is searched.  Unsearched points lie beyond the
               faces of that block (faces clamped at the grid edge bound nothing). */
            double bound = INFINITY;
            int cl[3] = {x0, y0, z0}, ch[3] = {x1, y1, z1};
            int all_clamped = 1;
            for (int k = 0; k < 3; k++) {
                if (cl[k] > 0) { double f = (double)g->lo[k] + (double)cl[k] * g->h; double dd = (double)pt[k] - f; if (dd < bound) bound = dd; all_clamped = 0; }
                if (ch[k] < g->dim[k] - 1) { double f = (double)g->lo[k] + (double)(ch[k] + 1) * g->h; double dd = f - (double)pt[k]; if (dd < bound) bound = dd; all_clamped = 0; }
            }
            if (all_clamped) break;
            bound -= 1e-3 * g->h;  /* cell assignment is fp32: keep a safety margin */
            if (bj >= 0 && bound > 0 && (double)best < bound * bound * (1.0 - 1e-6)) break;
        }
        idx[i] = bj;
        if (d2_out) d2_out[i] = best;
    }
}

/* ------------------------------------------------------------------ */
/* myicp.cpp:152-172 estimateNormals: k-NN PCA normals, flipped toward   */
/* the viewpoint (0,0,0).  PCL 1.9.1 semantics restated: the query point */
/* is in the tree so it is one of its own k neighbours; normal = eigen-  */
/* vector of the smallest covariance eigenvalue; flip if (vp - p).n < 0. */
/* Brute-force k-NN (oracle sizes only).                                 */
/* ------------------------------------------------------------------ */
int orc_normals_knn(const float *xyz, size_t n, int k, const float vp[3], float *nrm, float *curv)
{
    if (k < 3 || (size_t)k > n || k > 64) return ORC_ERR_ARG;
    float bd[64]; int32_t bi[64];
    for (size_t i = 0; i < n; i++) {
        const float *pi = xyz + 3 * i;
        int cnt = 0;
        for (size_t j = 0; j < n; j++) {
            float d2 = dist2f(pi, xyz + 3 * j);
            if (cnt < k || d2 < bd[cnt - 1]) {
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && bd[pos - 1] > d2) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
                bd[pos] = d2; bi[pos] = (int32_t)j;
                if (cnt < k) cnt++;
            }
        }
        double mu[3] = {0, 0, 0};
        for (int a = 0; a < k; a++) for (int c = 0; c < 3; c++) mu[c] += xyz[3 * (size_t)bi[a] + c];
        for (int c = 0; c < 3; c++) mu[c] /= k;
        double C[9] = {0};
        for (int a = 0; a < k; a++) {
            double d[3];
            for (int c = 0; c < 3; c++) d[c] = xyz[3 * (size_t)bi[a] + c] - mu[c];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] += d[r] * d[c];
        }
        for (int c = 0; c < 9; c++) C[c] /= k;
        double V[9], w[3];
        jacobi_eig(3, C, V, w);
        int m = 0; if (w[1] < w[m]) m = 1; if (w[2] < w[m]) m = 2;
        double nx = V[0 * 3 + m], ny = V[1 * 3 + m], nz = V[2 * 3 + m];
        double nn = sqrt(nx * nx + ny * ny + nz * nz);
        nx /= nn; ny /= nn; nz /= nn;
        double vx = (vp ? vp[0] : 0.0) - pi[0], vy = (vp ? vp[1] : 0.0) - pi[1], vz = (vp ? vp[2] : 0.0) - pi[2];
        if (vx * nx + vy * ny + vz * nz < 0) { nx = -nx; ny = -ny; nz = -nz; }
        nrm[3 * i] = (float)nx; nrm[3 * i + 1] = (float)ny; nrm[3 * i + 2] = (float)nz;
        if (curv) { double tr = w[0] + w[1] + w[2]; curv[i] = tr > 0 ? (float)(fabs(w[m]) / tr) : 0.f; }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* regist.h:8-72 registrateNPoint: closed-form rigid fit of known pairs  */
/* (Kabsch).  S = sum (src-c_s)(dst-c_d)^T ; S = U W V^T ;                */
/* R = V diag(1,1,det(V U^T)) U^T ; T = c_d - R c_s   (dst ~ R src + T).  */
/* The 3x3 SVD is done through the eigen-decomposition of S^T S.          */
/* ------------------------------------------------------------------ */
static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

int orc_kabsch_from_cov(const double S[9], const double cs[3], const double cd[3], double R[9], double T[3])
{
    double StS[9], V[9], w[3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double x = 0;
            for (int k = 0; k < 3; k++) x += S[k * 3 + r] * S[k * 3 + c];
            StS[r * 3 + c] = x;
        }
    jacobi_eig(3, StS, V, w);
    /* sort eigenpairs descending */
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (w[ord[b]] > w[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    double Vs[9], sig[3], U[9];
    for (int k = 0; k < 3; k++) {
        sig[k] = sqrt(w[ord[k]] > 0 ? w[ord[k]] : 0);
        for (int r = 0; r < 3; r++) Vs[r * 3 + k] = V[r * 3 + ord[k]];
    }
    if (!(sig[1] > 1e-12 * sig[0])) return ORC_ERR_DEGENERATE;      /* rank < 2: rotation undetermined */
    for (int k = 0; k < 2; k++)
        for (int r = 0; r < 3; r++) {
            double x = 0;
            for (int c = 0; c < 3; c++) x += S[r * 3 + c] * Vs[c * 3 + k];
            U[r * 3 + k] = x / sig[k];
        }
    /* third left vector: u2 = u0 x u1 (also covers sig[2] ~ 0) */
    U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
    U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
    U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    /* make V right-handed the same way so that S = U diag(sig) V^T stays a valid factorisation up to the sign of the
       last pair, which the determinant fix below absorbs */
    double VUt[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double x = 0;
            for (int k = 0; k < 3; k++) x += Vs[r * 3 + k] * U[c * 3 + k];
            VUt[r * 3 + c] = x;
        }
    double d = det3(VUt) < 0 ? -1.0 : 1.0;                          /* regist.h:57-61 */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            R[r * 3 + c] = Vs[r * 3 + 0] * U[c * 3 + 0] + Vs[r * 3 + 1] * U[c * 3 + 1] + d * Vs[r * 3 + 2] * U[c * 3 + 2];
    for (int r = 0; r < 3; r++) T[r] = cd[r] - (R[r * 3 + 0] * cs[0] + R[r * 3 + 1] * cs[1] + R[r * 3 + 2] * cs[2]);   /* :65-67 */
    return ORC_OK;
}

/* literal entry point: pairs by index, doubles throughout (regist.h works on cv::Point3d) */
int orc_kabsch(const float *src, const float *dst, size_t n, double R[9], double T[3])
{
    if (n < 3) return ORC_ERR_SIZE;
    double cs[3] = {0, 0, 0}, cd[3] = {0, 0, 0}, S[9] = {0};
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { cs[k] += src[3 * i + k]; cd[k] += dst[3 * i + k]; }
    for (int k = 0; k < 3; k++) { cs[k] /= (double)n; cd[k] /= (double)n; }
    for (size_t i = 0; i < n; i++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) S[r * 3 + c] += (src[3 * i + r] - cs[r]) * (dst[3 * i + c] - cd[c]);
    return orc_kabsch_from_cov(S, cs, cd, R, T);
}

/* the same from one reduction record taken in point-to-point mode (sums about `pivot`) */
int orc_solve_p2p(const double S40[ORC_NSUM], const float pivot[3], float X[16])
{
    double n = S40[34];
    if (!(n >= 3)) return ORC_ERR_DEGENERATE;
    double pb[3], qb[3], H[9], cs[3], cd[3], R[9], T[3];
    for (int k = 0; k < 3; k++) { pb[k] = S40[27 + k] / n; qb[k] = S40[30 + k] / n; }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) H[r * 3 + c] = S40[3 * r + c] - n * pb[r] * qb[c];
    for (int k = 0; k < 3; k++) { cs[k] = pb[k] + (pivot ? pivot[k] : 0.0); cd[k] = qb[k] + (pivot ? pivot[k] : 0.0); }
    int st = orc_kabsch_from_cov(H, cs, cd, R, T);
    if (st != ORC_OK) return st;
    mat4_identity(X);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) X[r * 4 + c] = (float)R[r * 3 + c]; X[r * 4 + 3] = (float)T[r]; }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* myicp.cpp:100-150 RegisterSymm loop                                   */
/* ------------------------------------------------------------------ */
typedef struct {
    int32_t mode;            /* ORC_MODE_* */
    int32_t corr;            /* ORC_CORR_* */
    int32_t solve;           /* ORC_SOLVE_* (QUIRKS only) */
    int32_t apply;           /* ORC_APPLY_* */
    int32_t max_iters;       /* myicp.cpp:6  -> 10 */
    float diff_threshold;    /* myicp.cpp:6  -> 1.0 */
    float max_corr_dist;     /* <=0: none */
    int32_t fixed_iters;     /* !=0: ignore the threshold, run exactly max_iters */
    float min_normal_dot;    /* > -1: drop pairs with n_p . n_q below it */
    float eps_rotation, eps_translation;   /* both > 0: also stop on a small increment */
} orc_config;

typedef struct {
    float transform[16];     /* row-major 4x4, maps original source -> target */
    int32_t iters;
    int32_t status;
    float diff_initial, diff_final;
    float diffs[256];        /* diff printed at the top of each iteration (myicp.cpp:125-126) */
    double last_sums[ORC_NSUM];
    double rcond;
} orc_result;

void orc_config_default(orc_config *c)
{
    memset(c, 0, sizeof(*c));
    c->mode = ORC_MODE_QUIRKS; c->corr = ORC_CORR_IDENTITY; c->solve = ORC_SOLVE_GRAM;
    c->apply = ORC_APPLY_INCREMENTAL; c->max_iters = 10; c->diff_threshold = 1.0f; c->min_normal_dot = -2.0f;
}

static void rot_only(const float X[16], float R[16])
{
    memcpy(R, X, 16 * sizeof(float)); R[3] = R[7] = R[11] = 0.f;
}

int orc_align(const orc_config *cfg, const float *src_xyz, const float *src_nrm, size_t n_s,
              const float *tgt_xyz, const float *tgt_nrm, size_t n_t, const float *guess, orc_result *res)
{
    if (!cfg || !src_xyz || !src_nrm || !tgt_xyz || !tgt_nrm || !res) return ORC_ERR_ARG;
    memset(res, 0, sizeof(*res));
    if (cfg->corr == ORC_CORR_IDENTITY && n_s != n_t) return ORC_ERR_SIZE;    /* func.cpp:21 assert */
    if (n_s == 0 || n_t == 0) return ORC_ERR_SIZE;
    float *p = (float *)malloc(sizeof(float) * 3 * n_s), *np = (float *)malloc(sizeof(float) * 3 * n_s);
    int32_t *idx = (cfg->corr != ORC_CORR_IDENTITY) ? (int32_t *)malloc(sizeof(int32_t) * n_s) : NULL;
    orc_grid *grid = (cfg->corr == ORC_CORR_GRID) ? orc_grid_build(tgt_xyz, n_t, 2.0f) : NULL;
    float X[16]; mat4_identity(X);
    if (guess) memcpy(X, guess, sizeof(X));
    int paper = cfg->mode != ORC_MODE_QUIRKS;      /* PAPER and P2P: centred sums, normals rotate only */
    int p2p = cfg->mode == ORC_MODE_P2P;
    /* working copies (myicp.cpp:109-111); a guess is applied up front */
    if (guess) {
        orc_apply(X, src_xyz, p, n_s, 1);
        if (paper) { float R[16]; rot_only(X, R); orc_apply(R, src_nrm, np, n_s, 0); }
        else orc_apply(X, src_nrm, np, n_s, 1);
    } else {
        memcpy(p, src_xyz, sizeof(float) * 3 * n_s); memcpy(np, src_nrm, sizeof(float) * 3 * n_s);
    }
    float pivot[3] = {0, 0, 0};
    if (paper) {
        double s[3] = {0, 0, 0};
        for (size_t i = 0; i < n_t; i++) for (int k = 0; k < 3; k++) s[k] += tgt_xyz[3 * i + k];
        for (int k = 0; k < 3; k++) pivot[k] = (float)(s[k] / (double)n_t);
    }
    float maxd2 = cfg->max_corr_dist > 0 ? cfg->max_corr_dist * cfg->max_corr_dist : 0.f;
    double S[ORC_NSUM];
    int iters = 0, status = ORC_OK;

#define ORC_CORRESPOND_AND_REDUCE()                                                         \
    do {                                                                                    \
        if (cfg->corr == ORC_CORR_BRUTE) orc_nn_brute(NULL, p, n_s, tgt_xyz, n_t, idx, NULL); \
        else if (cfg->corr == ORC_CORR_GRID) orc_nn_grid(grid, NULL, p, n_s, tgt_xyz, idx, NULL); \
        orc_reduce40(p, np, n_s, tgt_xyz, tgt_nrm, n_t, idx, pivot, maxd2, cfg->min_normal_dot, cfg->mode == ORC_MODE_P2P, S); \
    } while (0)

    ORC_CORRESPOND_AND_REDUCE();
    float diff = (float)S[33];                                  /* myicp.cpp:122 */
    res->diff_initial = diff;
    while ((cfg->fixed_iters || diff > cfg->diff_threshold) && iters++ < cfg->max_iters) {  /* :123 */
        if (iters <= 256) res->diffs[iters - 1] = diff;
        float pbar[3], qbar[3], a[3], t[3], Xi[16];
        double rc = 1.0;
        int st;
        if (p2p) {
            st = orc_solve_p2p(S, pivot, Xi);
        } else if (!paper) {
            if (cfg->solve == ORC_SOLVE_LITERAL && cfg->corr == ORC_CORR_IDENTITY)
                st = orc_solve_quirks_literal(p, np, tgt_xyz, tgt_nrm, n_s, pbar, qbar, a, t);
            else
                st = orc_solve_quirks_gram(S, pbar, qbar, a, t, &rc);
            orc_compose_quirks(pbar, qbar, a, t, Xi);            /* func.cpp:90-99 */
        } else {
            st = orc_solve_paper(S, pivot, pbar, qbar, a, t, &rc);
            orc_compose_paper(pbar, qbar, a, t, Xi);
        }
        res->rcond = rc;
        if (st != ORC_OK) { status = st; iters--; break; }
        int bad = 0;
        for (int k = 0; k < 16; k++) if (!isfinite(Xi[k])) bad = 1;
        if (bad) { status = ORC_ERR_DEGENERATE; iters--; break; }
        mat4_mul_f(Xi, X, X);                                    /* myicp.cpp:138 */
        if (cfg->apply == ORC_APPLY_INCREMENTAL) {
            orc_apply(Xi, p, p, n_s, 1);                         /* :136 */
            if (paper) { float R[16]; rot_only(Xi, R); orc_apply(R, np, np, n_s, 0); }
            else orc_apply(Xi, np, np, n_s, 1);                  /* :137 (translation on normals) */
        } else {
            orc_apply(X, src_xyz, p, n_s, 1);
            if (paper) { float R[16]; rot_only(X, R); orc_apply(R, src_nrm, np, n_s, 0); }
            else orc_apply(X, src_nrm, np, n_s, 1);
        }
        ORC_CORRESPOND_AND_REDUCE();
        diff = (float)S[33];                                     /* :141 */
        if (cfg->eps_rotation > 0.f && cfg->eps_translation > 0.f && !cfg->fixed_iters) {
            double tr = ((double)Xi[0] + Xi[5] + Xi[10] - 1.0) * 0.5;
            double ang = acos(tr > 1.0 ? 1.0 : (tr < -1.0 ? -1.0 : tr));
            double tn = sqrt((double)Xi[3] * Xi[3] + (double)Xi[7] * Xi[7] + (double)Xi[11] * Xi[11]);
            if (ang < cfg->eps_rotation && tn < cfg->eps_translation) break;
        }
    }
    if (iters > cfg->max_iters) iters = cfg->max_iters;          /* iters++ overshoots by one on exit */
    memcpy(res->transform, X, sizeof(X));
    memcpy(res->last_sums, S, sizeof(S));
    res->iters = iters; res->status = status; res->diff_final = diff;
    free(p); free(np); free(idx); orc_grid_free(grid);
    return status;
}

/* ------------------------------------------------------------------ */
/* myicp.cpp:20-31 LoadCloud -> PCD v0.7 reader (ASCII and binary),      */
/* keeps x,y,z and, when present, normal_x/y/z.                          */
/* Returns the point count (>=0) or -ORC_ERR_IO.  Call with xyz==NULL to */
/* query the count.                                                      */
/* ------------------------------------------------------------------ */
long orc_pcd_read(const char *path, float *xyz, float *nrm, size_t cap, int *has_normals)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -ORC_ERR_IO;
    char line[4096];
    char fields[32][32]; int nf = 0; int sizes[32], counts[32]; char types[32];
    long npts = -1, width = -1, height = 1; int binary = 0, got_data = 0;
    for (int i = 0; i < 32; i++) { sizes[i] = 4; counts[i] = 1; types[i] = 'F'; }
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        char key[32] = {0};
        if (sscanf(line, "%31s", key) != 1) continue;
        char *rest = line + strlen(key);
        if (!strcmp(key, "FIELDS")) {
            char *tok = strtok(rest, " \t\r\n");
            while (tok && nf < 32) { strncpy(fields[nf], tok, 31); fields[nf][31] = 0; nf++; tok = strtok(NULL, " \t\r\n"); }
        } else if (!strcmp(key, "SIZE")) {
            int k = 0; char *tok = strtok(rest, " \t\r\n"); while (tok && k < 32) { sizes[k++] = atoi(tok); tok = strtok(NULL, " \t\r\n"); }
        } else if (!strcmp(key, "TYPE")) {
            int k = 0; char *tok = strtok(rest, " \t\r\n"); while (tok && k < 32) { types[k++] = tok[0]; tok = strtok(NULL, " \t\r\n"); }
        } else if (!strcmp(key, "COUNT")) {
            int k = 0; char *tok = strtok(rest, " \t\r\n"); while (tok && k < 32) { counts[k++] = atoi(tok); tok = strtok(NULL, " \t\r\n"); }
        } else if (!strcmp(key, "WIDTH")) width = atol(rest);
        else if (!strcmp(key, "HEIGHT")) height = atol(rest);
        else if (!strcmp(key, "POINTS")) npts = atol(rest);
        else if (!strcmp(key, "DATA")) {
            char kind[32] = {0}; sscanf(rest, "%31s", kind);
            if (!strcmp(kind, "binary")) binary = 1;
            else if (!strcmp(kind, "ascii")) binary = 0;
            else { fclose(f); return -ORC_ERR_IO; }
            got_data = 1; break;
        }
    }
    if (!got_data || nf == 0) { fclose(f); return -ORC_ERR_IO; }
    if (npts < 0) npts = width * height;
    int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1;
    int col[32], ncol = 0, off[32], rec = 0;
    for (int k = 0; k < nf; k++) {
        col[k] = ncol; off[k] = rec; ncol += counts[k]; rec += counts[k] * sizes[k];
        if (!strcmp(fields[k], "x")) ix = k; else if (!strcmp(fields[k], "y")) iy = k; else if (!strcmp(fields[k], "z")) iz = k;
        else if (!strcmp(fields[k], "normal_x")) inx = k; else if (!strcmp(fields[k], "normal_y")) iny = k; else if (!strcmp(fields[k], "normal_z")) inz = k;
    }
    if (ix < 0 || iy < 0 || iz < 0) { fclose(f); return -ORC_ERR_IO; }
    int hn = (inx >= 0 && iny >= 0 && inz >= 0);
    if (has_normals) *has_normals = hn;
    if (!xyz) { fclose(f); return npts; }
    if ((size_t)npts > cap) { fclose(f); return -ORC_ERR_SIZE; }
    if (binary) {
        unsigned char *buf = (unsigned char *)malloc((size_t)rec);
        for (long i = 0; i < npts; i++) {
            if (fread(buf, 1, (size_t)rec, f) != (size_t)rec) { free(buf); fclose(f); return -ORC_ERR_IO; }
            memcpy(&xyz[3 * i], buf + off[ix], 4); memcpy(&xyz[3 * i + 1], buf + off[iy], 4); memcpy(&xyz[3 * i + 2], buf + off[iz], 4);
            if (nrm && hn) { memcpy(&nrm[3 * i], buf + off[inx], 4); memcpy(&nrm[3 * i + 1], buf + off[iny], 4); memcpy(&nrm[3 * i + 2], buf + off[inz], 4); }
        }
        free(buf);
    } else {
        for (long i = 0; i < npts; i++) {
            if (!fgets(line, sizeof line, f)) { fclose(f); return -ORC_ERR_IO; }
            double vals[64]; int nv = 0; char *s = line, *e;
            while (nv < 64) { double v = strtod(s, &e); if (e == s) break; vals[nv++] = v; s = e; }
            if (nv < ncol) { fclose(f); return -ORC_ERR_IO; }
            xyz[3 * i] = (float)vals[col[ix]]; xyz[3 * i + 1] = (float)vals[col[iy]]; xyz[3 * i + 2] = (float)vals[col[iz]];
            if (nrm && hn) { nrm[3 * i] = (float)vals[col[inx]]; nrm[3 * i + 1] = (float)vals[col[iny]]; nrm[3 * i + 2] = (float)vals[col[inz]]; }
        }
    }
    fclose(f);
    return npts;
}

int orc_nsum(void) { return ORC_NSUM; }
