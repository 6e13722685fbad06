// kernels_normals.hip -- k-NN PCA normals on gfx950, the pre-step of the path.
// Replaces MyICP::estimateNormals (reference ICP/myicp.cpp:152-172), i.e. PCL 1.9.1
// NormalEstimation<PointXYZ,Normal> with setKSearch(10) over a search::KdTree:
//   per point: the k nearest points of the SAME cloud (the point itself included, it is in
//   the tree), their mean and covariance, the eigenvector of the smallest eigenvalue,
//   flipped so that (viewpoint - p) . n >= 0 (viewpoint defaults to the origin).
// The k-NN set is exact: (d2, row) lexicographic, found by a pruned pre-order walk of the
// same implicit 8-ary box tree the correspondence search uses, its radius bounded from the start by
// the query's neighbours on the Morton curve.  Moments are fp64.
#include "symmicp_internal.h"
#pragma clang fp contract(off)

namespace symmicp {

constexpr int kKnnMax = 16;

__device__ __forceinline__ float n_dist2(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ float n_boxdist2(float px, float py, float pz, const float4 &lo, const float4 &hi)
{
    float dx = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.0f);
    float dy = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.0f);
    float dz = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ bool knn_less(float d2a, int ra, float d2b, int rb)
{
    return d2a < d2b || (d2a == d2b && ra < rb);
}

// cyclic Jacobi on a symmetric 3x3 (fp64); returns the eigenvector of the smallest eigenvalue
__device__ __forceinline__ void smallest_eigvec3(double a00, double a01, double a02, double a11, double a12, double a22,
                                                 double &nx, double &ny, double &nz, double &lam, double &trace)
{
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            double apq = A[p][q];
            if (fabs(apq) < 1e-300) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double kp = A[k][p], kq = A[k][q];
                A[k][p] = c * kp - s * kq; A[k][q] = s * kp + c * kq;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double pk = A[p][k], qk = A[q][k];
                A[p][k] = c * pk - s * qk; A[q][k] = s * pk + c * qk;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double kp = V[k][p], kq = V[k][q];
                V[k][p] = c * kp - s * kq; V[k][q] = s * kp + c * kq;
            }
        }
    }
    double w0 = A[0][0], w1 = A[1][1], w2 = A[2][2];
    trace = w0 + w1 + w2;
    int m = 0;
    lam = w0;
    if (w1 < lam) { lam = w1; m = 1; }
    if (w2 < lam) { lam = w2; m = 2; }
    nx = (m == 0) ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
    ny = (m == 0) ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
    nz = (m == 0) ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

__global__ __launch_bounds__(256) void k_normals_knn(TargetIndex ix, int k, float vx, float vy, float vz,
                                                     float *nrm_out, float *curv_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ix.n) return;
    const float4 me = ix.tq[i];
    const float px = me.x, py = me.y, pz = me.z;
    float bd[kKnnMax];
    int br[kKnnMax], bp[kKnnMax];
#pragma unroll
    for (int j = 0; j < kKnnMax; j++) { bd[j] = __int_as_float(0x7f800000); br[j] = 0x7fffffff; bp[j] = -1; }
    const int last = k - 1;

    // Bound from the neighbours in the sorted order: the query is itself a point of the cloud, and the 2k+1 points around
    // it on the Morton curve are near it in space, so the k-th smallest of their distances bounds the k-th nearest
    // neighbour's distance from above (only the bound is kept, not the points: the walk finds them again and would
    // otherwise have to filter duplicates).  Without it the pre-order walk starts at the far end of the cloud with an
    // infinite radius and prunes nothing until k arbitrary points have been seen.
    float seed = __int_as_float(0x7f800000);
    {
        const uint32_t w = 2u * (uint32_t)k + 1u;
        uint32_t j1 = min(((i > (uint32_t)k) ? i - (uint32_t)k : 0u) + w, ix.n);
        uint32_t j0 = (j1 >= w) ? j1 - w : 0u;
        float sd[kKnnMax];
#pragma unroll
        for (int j = 0; j < kKnnMax; j++) sd[j] = __int_as_float(0x7f800000);
        for (uint32_t jj = j0; jj < j1; jj++) {
            const float4 q = ix.tq[jj];
            const float d2 = n_dist2(px, py, pz, q.x, q.y, q.z);
            float wd = __int_as_float(0x7f800000);
#pragma unroll
            for (int j = 0; j < kKnnMax; j++) if (j == last) wd = sd[j];
            if (d2 < wd) {
#pragma unroll
                for (int j = 0; j < kKnnMax; j++) if (j == last) sd[j] = d2;
#pragma unroll
                for (int j = kKnnMax - 1; j > 0; j--)
                    if (j <= last && sd[j] < sd[j - 1]) { const float t = sd[j]; sd[j] = sd[j - 1]; sd[j - 1] = t; }
            }
        }
#pragma unroll
        for (int j = 0; j < kKnnMax; j++) if (j == last) seed = sd[j];
    }

    int level = ix.top;
    uint32_t node = 0;
    while (true) {
        const float4 *bx = ix.boxes + 2 * ((size_t)ix.level_off[level] + node);
        float4 lo = bx[0], hi = bx[1];
        float d = n_boxdist2(px, py, pz, lo, hi);
        // worst kept distance (slot k-1); static indexing only
        float worst = __int_as_float(0x7f800000);
#pragma unroll
        for (int j = 0; j < kKnnMax; j++) if (j == last) worst = bd[j];
        bool hit = (d <= fminf(worst, seed)) && (lo.x <= hi.x);
        if (hit && level > 0) { level--; node <<= 3; continue; }
        if (hit) {
            uint32_t j0 = node * kLeaf, j1 = min(j0 + kLeaf, ix.n);
            for (uint32_t jj = j0; jj < j1; jj++) {
                float4 q = ix.tq[jj];
                float d2 = n_dist2(px, py, pz, q.x, q.y, q.z);
                int row = __float_as_int(q.w);
                float wd = __int_as_float(0x7f800000); int wr = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < kKnnMax; j++) if (j == last) { wd = bd[j]; wr = br[j]; }
                if (knn_less(d2, row, wd, wr)) {
                    // replace the worst, then bubble toward the front
#pragma unroll
                    for (int j = 0; j < kKnnMax; j++) if (j == last) { bd[j] = d2; br[j] = row; bp[j] = (int)jj; }
#pragma unroll
                    for (int j = kKnnMax - 1; j > 0; j--) {
                        if (j <= last && knn_less(bd[j], br[j], bd[j - 1], br[j - 1])) {
                            float td = bd[j]; bd[j] = bd[j - 1]; bd[j - 1] = td;
                            int tr = br[j]; br[j] = br[j - 1]; br[j - 1] = tr;
                            int tp = bp[j]; bp[j] = bp[j - 1]; bp[j - 1] = tp;
                        }
                    }
                }
            }
        }
        node++;
        while ((node & (kFan - 1)) == 0 && level < ix.top) { node >>= 3; level++; }
        if (level == ix.top && node >= ix.ntop) break;
    }

    // moments over the neighbours in ascending (d2,row) order, fp64
    double mx = 0, my = 0, mz = 0;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kKnnMax; j++)
        if (j < k && bp[j] >= 0) { float4 q = ix.tq[bp[j]]; mx += (double)q.x; my += (double)q.y; mz += (double)q.z; cnt++; }
    const int row_me = __float_as_int(me.w);
    if (cnt < 3) {
        nrm_out[3 * (size_t)row_me] = nrm_out[3 * (size_t)row_me + 1] = nrm_out[3 * (size_t)row_me + 2] = __int_as_float(0x7fc00000);
        if (curv_out) curv_out[row_me] = __int_as_float(0x7fc00000);
        return;
    }
    mx /= cnt; my /= cnt; mz /= cnt;
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
    for (int j = 0; j < kKnnMax; j++)
        if (j < k && bp[j] >= 0) {
            float4 q = ix.tq[bp[j]];
            double dx = (double)q.x - mx, dy = (double)q.y - my, dz = (double)q.z - mz;
            c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
        }
    c00 /= cnt; c01 /= cnt; c02 /= cnt; c11 /= cnt; c12 /= cnt; c22 /= cnt;
    double nx, ny, nz, lam, tr;
    smallest_eigvec3(c00, c01, c02, c11, c12, c22, nx, ny, nz, lam, tr);
    double nn = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= nn; ny /= nn; nz /= nn;
    // flipNormalTowardsViewpoint: (vp - p) . n < 0 -> flip
    double dot = ((double)vx - (double)px) * nx + ((double)vy - (double)py) * ny + ((double)vz - (double)pz) * nz;
    if (dot < 0) { nx = -nx; ny = -ny; nz = -nz; }
    nrm_out[3 * (size_t)row_me] = (float)nx;
    nrm_out[3 * (size_t)row_me + 1] = (float)ny;
    nrm_out[3 * (size_t)row_me + 2] = (float)nz;
    if (curv_out) curv_out[row_me] = tr > 0 ? (float)(fabs(lam) / tr) : 0.f;
}

void launch_normals_knn(const TargetIndex &ix, int k, const float vp[3], float *nrm_out, float *curv_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_normals_knn, dim3((ix.n + 255) / 256), dim3(256), 0, s, ix, k, vp[0], vp[1], vp[2], nrm_out, curv_out);
}

}  // namespace symmicp
