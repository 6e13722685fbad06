// kernels_pass.hip -- the per-iteration kernels of the symmetric-ICP loop for gfx950.
//
// One "pass" = everything the reference does per trip of myicp.cpp:123-142 that
// touches all N points, fused into a single sweep over the source share:
//   applyTransform (func.cpp:104-121)  -> transform p, n_p on the fly (optionally write back)
//   correspondence (myicp.cpp:128-131) -> identity / precomputed / fused exact NN search
//   calculateMatrixNotation (func.cpp:43-60) -> M_i, N_i, c_i in fp32 registers, never stored
//   the O(N) parts of solveLLS / means / evalDiff (func.cpp:19-32,64-73,85)
//                                     -> 37 fp64 sums per thread -> wave64 shuffle tree
//                                        -> LDS across the 4 waves -> one record per block.
// The path is HBM/L2-bound integer+fp32 work; no MFMA.
//
// fp32 expressions here must stay UNFUSED (compiled with -ffp-contract=off) and
// keep the association written: the CPU oracle uses the same expressions, so
// nearest-neighbour choices compare bit for bit.
#include "symmicp_internal.h"
#pragma clang fp contract(off)

namespace symmicp {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float xf_row(const float *m, float x, float y, float z, float w)
{
    // ((m0*x + m1*y) + m2*z) + m3*w      (func.cpp:111-118, k sequential)
    return ((m[0] * x + m[1] * y) + m[2] * z) + m[3] * w;
}

__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// squared distance from a point to an axis-aligned box, same expression shape as
// dist2 so that (in fp32) boxdist2 <= dist2 to every point stored inside the box.
__device__ __forceinline__ float boxdist2(float px, float py, float pz, const float4 &lo, const float4 &hi)
{
    float dx = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.0f);
    float dy = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.0f);
    float dz = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.0f);
    return (dx * dx + dy * dy) + dz * dz;
}

struct Acc {
    double v[kNAcc];
};

__device__ __forceinline__ void acc_zero(Acc &a)
{
#pragma unroll
    for (int k = 0; k < kNAcc; k++) a.v[k] = 0.0;
}

// rows of func.cpp:51-58 for one pair, accumulated in fp64
__device__ __forceinline__ void acc_pair(Acc &a, float px, float py, float pz, float npx, float npy, float npz,
                                         float qx, float qy, float qz, float nqx, float nqy, float nqz,
                                         float d2, const float *pivot)
{
    px -= pivot[0]; py -= pivot[1]; pz -= pivot[2];
    qx -= pivot[0]; qy -= pivot[1]; qz -= pivot[2];
    float nx = npx + nqx, ny = npy + nqy, nz = npz + nqz;            // func.cpp:51
    float sx = px + qx, sy = py + qy, sz = pz + qz;
    float dx = px - qx, dy = py - qy, dz = pz - qz;
    float m0 = sy * nz - sz * ny;                                      // func.cpp:54
    float m1 = sz * nx - sx * nz;
    float m2 = sx * ny - sy * nx;
    float c = (dx * nx + dy * ny) + dz * nz;                           // func.cpp:58
    double v[6] = {(double)m0, (double)m1, (double)m2, (double)nx, (double)ny, (double)nz};
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int s = r; s < 6; s++) { a.v[k] = __builtin_fma(v[r], v[s], a.v[k]); k++; }
    double cd = (double)c;
#pragma unroll
    for (int r = 0; r < 6; r++) a.v[21 + r] = __builtin_fma(v[r], cd, a.v[21 + r]);
    a.v[27] += (double)px; a.v[28] += (double)py; a.v[29] += (double)pz;
    a.v[30] += (double)qx; a.v[31] += (double)qy; a.v[32] += (double)qz;
    a.v[33] += (double)sqrtf(d2);                                      // func.cpp:28
    a.v[34] += 1.0;
    a.v[35] = __builtin_fma(cd, cd, a.v[35]);
    a.v[36] += (double)d2;
}

// block reduction: wave64 shuffle tree, then LDS across waves; fixed order -> deterministic
__device__ __forceinline__ void acc_block_reduce_store(Acc &a, double *partials_row)
{
    __shared__ double red[(kPassThreads / 64) * kNSum];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kNAcc; k++) {
        double x = a.v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) red[wave * kNSum + k] = x;
    }
    __syncthreads();
    if (threadIdx.x < kNSum) {
        double s = 0.0;
        if (threadIdx.x < kNAcc) {
#pragma unroll
            for (int w = 0; w < kPassThreads / 64; w++) s += red[w * kNSum + threadIdx.x];
        }
        partials_row[threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------
// pass, identity pairing (what the reference does: myicp.cpp:130)
// Streaming: 48 B/point read (+24 B/point written with write-back).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPassThreads) void k_pass_identity(PassArgs a, CloudSoA tgt)
{
    Acc acc; acc_zero(acc);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
        float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
        float npx = xf_row(a.X.m + 0, nx, ny, nz, a.X.nrm_w), npy = xf_row(a.X.m + 4, nx, ny, nz, a.X.nrm_w),
              npz = xf_row(a.X.m + 8, nx, ny, nz, a.X.nrm_w);
        if (a.writeback) {
            a.out.x[i] = px; a.out.y[i] = py; a.out.z[i] = pz;
            a.out.nx[i] = npx; a.out.ny[i] = npy; a.out.nz[i] = npz;
        }
        uint32_t j = a.tgt_offset + i;
        float qx = tgt.x[j], qy = tgt.y[j], qz = tgt.z[j];
        float d2 = dist2(px, py, pz, qx, qy, qz);
        if (a.d2_out) a.d2_out[i] = d2;
        if (a.max_d2 > 0.0f && d2 > a.max_d2) continue;
        acc_pair(acc, px, py, pz, npx, npy, npz, qx, qy, qz, tgt.nx[j], tgt.ny[j], tgt.nz[j], d2, a.pivot);
    }
    acc_block_reduce_store(acc, a.partials + (size_t)blockIdx.x * kNSum);
}

// ---------------------------------------------------------------------------
// pass, pairs given by a previous search kernel (brute force): best64[i] holds
// (d2 bits << 32 | target row).  Target rows are gathered as float4.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPassThreads) void k_pass_indexed(PassArgs a, const float4 *__restrict__ tq,
                                                              const float4 *__restrict__ tn)
{
    Acc acc; acc_zero(acc);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
        float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
        float npx = xf_row(a.X.m + 0, nx, ny, nz, a.X.nrm_w), npy = xf_row(a.X.m + 4, nx, ny, nz, a.X.nrm_w),
              npz = xf_row(a.X.m + 8, nx, ny, nz, a.X.nrm_w);
        if (a.writeback) {
            a.out.x[i] = px; a.out.y[i] = py; a.out.z[i] = pz;
            a.out.nx[i] = npx; a.out.ny[i] = npy; a.out.nz[i] = npz;
        }
        unsigned long long b = a.best64[i];
        uint32_t j = (uint32_t)(b & 0xFFFFFFFFull);
        float d2 = __uint_as_float((uint32_t)(b >> 32));
        bool ok = (b != ~0ull);
        if (a.pos_out) a.pos_out[i] = ok ? (int32_t)j : -1;
        if (a.d2_out) a.d2_out[i] = ok ? d2 : __int_as_float(0x7f800000);
        if (!ok) continue;
        if (a.max_d2 > 0.0f && d2 > a.max_d2) continue;
        float4 q = tq[j], nq = tn[j];
        acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, d2, a.pivot);
    }
    acc_block_reduce_store(acc, a.partials + (size_t)blockIdx.x * kNSum);
}

// ---------------------------------------------------------------------------
// exact nearest neighbour of one query in the target index.
//   phase 0: the pair found by the previous pass bounds the search (temporal coherence);
//   phase 1: the 27 Morton cells around the query in the dense cell table ("grid"),
//            skipping cells farther than the current best;
//   phase 2: only if phase 1 cannot prove its answer: stackless pre-order walk of the
//            implicit 8-ary box tree ("linear BVH"), pruned by the current best.
// Result = argmin over ALL target points of (d2, original row), identical to brute force.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

struct Best {
    float d2;
    int32_t pos;      // sorted position
    int32_t row;      // original target row (tie-break key)
};

__device__ __forceinline__ void test_point(Best &b, const float4 &q, int32_t pos, float px, float py, float pz)
{
    float d2 = dist2(px, py, pz, q.x, q.y, q.z);
    int32_t row = __float_as_int(q.w);
    if (d2 < b.d2 || (d2 == b.d2 && row < b.row)) { b.d2 = d2; b.pos = pos; b.row = row; }
}

__device__ __forceinline__ void nn_search(const TargetIndex &ix, float px, float py, float pz, int32_t prev, Best &b)
{
    b.d2 = __int_as_float(0x7f800000); b.pos = -1; b.row = 0x7fffffff;
    if (prev >= 0 && (uint32_t)prev < ix.n) test_point(b, ix.tq[prev], prev, px, py, pz);

    bool proven = false;
    if (ix.glevel > 0) {
        float fx = (px - ix.ox) * ix.inv_h, fy = (py - ix.oy) * ix.inv_h, fz = (pz - ix.oz) * ix.inv_h;
        const float lim = (float)ix.gdim;
        // only queries inside (or within one cell of) the grid use it
        if (fx >= -1.0f && fy >= -1.0f && fz >= -1.0f && fx < lim + 1.0f && fy < lim + 1.0f && fz < lim + 1.0f) {
            int cx = min(max((int)floorf(fx), 0), ix.gdim - 1);
            int cy = min(max((int)floorf(fy), 0), ix.gdim - 1);
            int cz = min(max((int)floorf(fz), 0), ix.gdim - 1);
            const float margin = 2e-3f * ix.h;
            for (int dz = -1; dz <= 1; dz++) {
                int z = cz + dz;
                if (z < 0 || z >= ix.gdim) continue;
                float zlo = ix.oz + (float)z * ix.h;
                float gz = fmaxf(fmaxf(zlo - pz, pz - (zlo + ix.h)) - margin, 0.0f);
                uint32_t mz = spread3((uint32_t)z) << 2;
                for (int dy = -1; dy <= 1; dy++) {
                    int y = cy + dy;
                    if (y < 0 || y >= ix.gdim) continue;
                    float ylo = ix.oy + (float)y * ix.h;
                    float gy = fmaxf(fmaxf(ylo - py, py - (ylo + ix.h)) - margin, 0.0f);
                    uint32_t mzy = mz | (spread3((uint32_t)y) << 1);
                    for (int dx = -1; dx <= 1; dx++) {
                        int x = cx + dx;
                        if (x < 0 || x >= ix.gdim) continue;
                        float xlo = ix.ox + (float)x * ix.h;
                        float gx = fmaxf(fmaxf(xlo - px, px - (xlo + ix.h)) - margin, 0.0f);
                        float g2 = (gx * gx + gy * gy) + gz * gz;
                        if (g2 > b.d2) continue;
                        uint2 r = ix.cells[mzy | spread3((uint32_t)x)];
                        for (uint32_t j = r.x; j < r.y; j++) test_point(b, ix.tq[j], (int32_t)j, px, py, pz);
                    }
                }
            }
            // every point of cells [c-1, c+1] has been seen (or was provably farther than best).
            // Unseen points lie beyond the faces of that block; faces on the grid edge bound nothing.
            float bound = __int_as_float(0x7f800000);
            bool any = false;
            if (cx - 1 > 0) { bound = fminf(bound, px - (ix.ox + (float)(cx - 1) * ix.h)); any = true; }
            if (cy - 1 > 0) { bound = fminf(bound, py - (ix.oy + (float)(cy - 1) * ix.h)); any = true; }
            if (cz - 1 > 0) { bound = fminf(bound, pz - (ix.oz + (float)(cz - 1) * ix.h)); any = true; }
            if (cx + 1 < ix.gdim - 1) { bound = fminf(bound, (ix.ox + (float)(cx + 2) * ix.h) - px); any = true; }
            if (cy + 1 < ix.gdim - 1) { bound = fminf(bound, (ix.oy + (float)(cy + 2) * ix.h) - py); any = true; }
            if (cz + 1 < ix.gdim - 1) { bound = fminf(bound, (ix.oz + (float)(cz + 2) * ix.h) - pz); any = true; }
            if (!any) {
                proven = (b.pos >= 0);   // the block covers the whole grid
            } else {
                bound -= margin;
                proven = (b.pos >= 0) && bound > 0.0f && b.d2 < bound * bound * 0.99999f;
            }
        }
    }
    if (proven) return;

    // tree walk.  Node (level, i); children of (level, i) are (level-1, 8i .. 8i+7).
    int level = ix.top;
    uint32_t i = 0;
    while (true) {
        const float4 *bx = ix.boxes + 2 * ((size_t)ix.level_off[level] + i);
        float4 lo = bx[0], hi = bx[1];
        float d = boxdist2(px, py, pz, lo, hi);
        bool hit = (d <= b.d2) && (lo.x <= hi.x);
        if (hit && level > 0) {
            level--; i <<= 3;
            continue;
        }
        if (hit) {
            uint32_t j0 = i * kLeaf, j1 = min(j0 + kLeaf, ix.n);
            for (uint32_t j = j0; j < j1; j++) test_point(b, ix.tq[j], (int32_t)j, px, py, pz);
        }
        i++;
        while ((i & (kFan - 1)) == 0 && level < ix.top) { i >>= 3; level++; }
        if (level == ix.top && i >= ix.ntop) break;
    }
}

__global__ __launch_bounds__(kPassThreads) void k_pass_tree(PassArgs a, TargetIndex ix)
{
    Acc acc; acc_zero(acc);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
        float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
        float npx = xf_row(a.X.m + 0, nx, ny, nz, a.X.nrm_w), npy = xf_row(a.X.m + 4, nx, ny, nz, a.X.nrm_w),
              npz = xf_row(a.X.m + 8, nx, ny, nz, a.X.nrm_w);
        if (a.writeback) {
            a.out.x[i] = px; a.out.y[i] = py; a.out.z[i] = pz;
            a.out.nx[i] = npx; a.out.ny[i] = npy; a.out.nz[i] = npz;
        }
        Best b;
        nn_search(ix, px, py, pz, a.pos_prev ? a.pos_prev[i] : -1, b);
        if (a.pos_out) a.pos_out[i] = b.pos;
        if (a.d2_out) a.d2_out[i] = b.d2;
        if (b.pos < 0) continue;
        if (a.max_d2 > 0.0f && b.d2 > a.max_d2) continue;
        float4 q = ix.tq[b.pos], nq = ix.tn[b.pos];
        acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, b.d2, a.pivot);
    }
    acc_block_reduce_store(acc, a.partials + (size_t)blockIdx.x * kNSum);
}

// ---------------------------------------------------------------------------
// final reduce: [blocks][40] -> [40], fixed order.  1000 threads = 40 sums x 25 strands.
// Writes the device record and, when given, a host-mapped copy (single-GPU fast path).
// ---------------------------------------------------------------------------
constexpr int kFinalStrands = 25;
__global__ __launch_bounds__(1024) void k_final_reduce(const double *__restrict__ partials, int blocks,
                                                       double *out_dev, double *out_host)
{
    __shared__ double red[kFinalStrands * kNSum];
    const int t = threadIdx.x;
    if (t < kFinalStrands * kNSum) {
        const int k = t % kNSum, strand = t / kNSum;
        double s = 0.0;
        for (int b = strand; b < blocks; b += kFinalStrands) s += partials[(size_t)b * kNSum + k];
        red[strand * kNSum + k] = s;
    }
    __syncthreads();
    if (t < kNSum) {
        double s = 0.0;
        for (int strand = 0; strand < kFinalStrands; strand++) s += red[strand * kNSum + t];
        out_dev[t] = s;
        if (out_host) out_host[t] = s;
    }
}

// ---------------------------------------------------------------------------
// brute-force exact NN (SURVEY 8(a) k_nn_brute): target tiles of 1024 points staged in LDS,
// 4 queries per thread, every lane reads the same LDS address (broadcast, conflict-free).
// grid = (query blocks, target splits); splits are merged with a 64-bit atomicMin on
// (d2 bits << 32 | row): min is order independent, and equal d2 resolves to the lowest row.
// VALU-bound by design (N_s x N_t distance evaluations); used for small clouds and as the
// on-device cross-check of the tree search.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPassThreads) void k_nn_brute(CloudSoA src, uint32_t n_s, Affine X,
                                                          const float4 *__restrict__ tq, uint32_t n_t,
                                                          uint32_t tiles_per_split, unsigned long long *best64)
{
    __shared__ float4 tile[kBruteTile];
    float px[kBruteQ], py[kBruteQ], pz[kBruteQ], bd[kBruteQ];
    uint32_t bj[kBruteQ];
    const uint32_t q0 = blockIdx.x * (kPassThreads * kBruteQ) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kBruteQ; k++) {
        uint32_t i = q0 + k * kPassThreads;
        float x = 0.f, y = 0.f, z = 0.f;
        if (i < n_s) { x = src.x[i]; y = src.y[i]; z = src.z[i]; }
        px[k] = xf_row(X.m + 0, x, y, z, 1.0f); py[k] = xf_row(X.m + 4, x, y, z, 1.0f); pz[k] = xf_row(X.m + 8, x, y, z, 1.0f);
        bd[k] = __int_as_float(0x7f800000); bj[k] = 0xFFFFFFFFu;
    }
    const uint32_t tile0 = blockIdx.y * tiles_per_split;
    const uint32_t ntiles = (n_t + kBruteTile - 1) / kBruteTile;
    const uint32_t tile1 = min(tile0 + tiles_per_split, ntiles);
    for (uint32_t t = tile0; t < tile1; t++) {
        const uint32_t base = t * kBruteTile;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBruteTile / kPassThreads; k++) {
            uint32_t j = base + k * kPassThreads + threadIdx.x;
            // out-of-range slots get a NaN coordinate: every comparison with them is false
            tile[k * kPassThreads + threadIdx.x] = (j < n_t) ? tq[j] : make_float4(__int_as_float(0x7fc00000), 0.f, 0.f, 0.f);
        }
        __syncthreads();
        // rows inside a tile ascend, tiles ascend -> strict '<' keeps the lowest row on ties
#pragma unroll 4
        for (int j = 0; j < kBruteTile; j++) {
            float4 q = tile[j];
#pragma unroll
            for (int k = 0; k < kBruteQ; k++) {
                float d2 = dist2(px[k], py[k], pz[k], q.x, q.y, q.z);
                if (d2 < bd[k]) { bd[k] = d2; bj[k] = base + j; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kBruteQ; k++) {
        uint32_t i = q0 + k * kPassThreads;
        if (i < n_s && bj[k] != 0xFFFFFFFFu) {
            unsigned long long key = ((unsigned long long)__float_as_uint(bd[k]) << 32) | (unsigned long long)bj[k];
            atomicMin(best64 + i, key);
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_pass_identity(const PassArgs &a, CloudSoA tgt, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_pass_identity, dim3(blocks), dim3(kPassThreads), 0, s, a, tgt);
}

void launch_pass_indexed(const PassArgs &a, const float4 *tq, const float4 *tn, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_pass_indexed, dim3(blocks), dim3(kPassThreads), 0, s, a, tq, tn);
}

void launch_pass_tree(const PassArgs &a, const TargetIndex &ix, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_pass_tree, dim3(blocks), dim3(kPassThreads), 0, s, a, ix);
}

void launch_final_reduce(const double *partials, int blocks, double *out_dev, double *out_host_mapped, hipStream_t s)
{
    hipLaunchKernelGGL(k_final_reduce, dim3(1), dim3(1024), 0, s, partials, blocks, out_dev, out_host_mapped);
}

void launch_nn_brute(const CloudSoA &src, uint32_t n_s, const Affine &X, const float4 *tq, uint32_t n_t,
                     unsigned long long *best64, hipStream_t s)
{
    const uint32_t qblocks = (n_s + kPassThreads * kBruteQ - 1) / (kPassThreads * kBruteQ);
    const uint32_t ntiles = (n_t + kBruteTile - 1) / kBruteTile;
    uint32_t splits = 1;
    if (qblocks < 2048) splits = (2048 + qblocks - 1) / qblocks;
    if (splits > ntiles) splits = ntiles;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const uint32_t tiles_per_split = (ntiles + splits - 1) / splits;
    splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
    hipMemsetAsync(best64, 0xFF, sizeof(unsigned long long) * (size_t)n_s, s);
    hipLaunchKernelGGL(k_nn_brute, dim3(qblocks, splits), dim3(kPassThreads), 0, s, src, n_s, X, tq, n_t, tiles_per_split, best64);
}

}  // namespace symmicp
