// kernels_pass.hip -- the per-iteration kernels of the symmetric-ICP loop for gfx950.
//
// One "pass" = everything the reference does per trip of myicp.cpp:123-142 that
// touches all N points.  Identity pairing is one streaming kernel; the nearest-neighbour
// pass is k_search_cells -> k_search_walk -> k_accumulate (see below).  Per point:
//   applyTransform (func.cpp:104-121)  -> transform p, n_p on the fly (optionally write back)
//   correspondence (myicp.cpp:128-131) -> identity / brute force / exact NN through cell table + trees
//   calculateMatrixNotation (func.cpp:43-60) -> M_i, N_i, c_i in fp32 registers, never stored
//   the O(N) parts of solveLLS / means / evalDiff (func.cpp:19-32,64-73,85)
//                                     -> 37 fp64 sums per thread -> wave64 DPP sum
//                                        -> LDS across the 4 waves -> one record per block.
// The path is HBM/L2-bound integer+fp32 work; no MFMA.
//
// fp32 expressions here must stay UNFUSED (compiled with -ffp-contract=off) and
// keep the association written: the CPU oracle uses the same expressions, so
// nearest-neighbour choices compare bit for bit.
#include <cstdlib>
#include "symmicp_internal.h"
#include "device_common.h"
#include "oct_walk.h"
#include "solve_core.h"
#pragma clang fp contract(off)

namespace symmicp {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
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
                                         float d2, const float *pivot, int p2p)
{
    px -= pivot[0]; py -= pivot[1]; pz -= pivot[2];
    qx -= pivot[0]; qy -= pivot[1]; qz -= pivot[2];
    if (p2p) {
        // point-to-point (regist.h:52): 3x3 cross-covariance sums, row-major in slots 0..8
        const double P[3] = {(double)px, (double)py, (double)pz}, Q[3] = {(double)qx, (double)qy, (double)qz};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) a.v[3 * r + c] = __builtin_fma(P[r], Q[c], a.v[3 * r + c]);
        a.v[27] += P[0]; a.v[28] += P[1]; a.v[29] += P[2];
        a.v[30] += Q[0]; a.v[31] += Q[1]; a.v[32] += Q[2];
        a.v[33] += (double)sqrtf(d2);
        a.v[34] += 1.0;
        a.v[36] += (double)d2;
        return;
    }
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

// one step of a sum on the VALU's DPP cross-lane network (no LDS traffic): row_shr 1,2,4,8 builds 16-lane row sums.  A double moves
// as two 32-bit DPP movs; lanes without a source read 0 (bound_ctrl), i.e. add +0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(hi, lo);
}

// wave64 sums of the 37 accumulators by halving: v_permlane32_swap puts the upper half-wave of value k next to the lower half-wave of
// value k + 20 (one add sums both values' halves: lanes 0..31 carry k, lanes 32..63 carry k + 20), v_permlane16_swap does the same
// with 16-lane rows for k and k + 10, and only the last four steps (row_shr 1, 2, 4, 8 inside a row of 16) touch every remaining
// register: 70 fp64 adds and 140 cross-lane moves per wave instead of the 222 + 444 of a full DPP reduction per value (2.4 us of the
// fused pass's 25 on its own stamps).  Lane 16 r + 15 then holds the wave's sum of value k + 10 r.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double swap32_add(double a, double b)
{
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double swap16_add(double a, double b)
{
    const u32x2_t lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}

// block reduction: wave sums, then LDS across the 4 waves; fixed order -> deterministic.
// record k of block b lands at partials[b * kNSum + k]: one contiguous 320-byte burst per block.  (Round 2 stored it transposed,
// [k][block], so that the reduce could read along the blocks: 40 scattered 8-byte stores per block into lines shared with up to 15
// other blocks -- on other XCDs, i.e. other L2s -- and a reduce whose load phase alone took 5.9 us of k_reduce_solve's 11.5.)
__device__ __forceinline__ void acc_block_reduce_store(Acc &a, double *partials, uint32_t nblocks, uint32_t col)
{
    static_assert(kNSum == 40 && kNAcc <= 40, "the halving below pairs value k with k + 20, then k + 10");
    __shared__ double red[(kPassThreads / 64) * kNSum];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double h1[20], h2[10];
#pragma unroll
    for (int k = 0; k < 20; k++) h1[k] = swap32_add(k < kNAcc ? a.v[k] : 0.0, k + 20 < kNAcc ? a.v[k + 20] : 0.0);
#pragma unroll
    for (int k = 0; k < 10; k++) h2[k] = swap16_add(h1[k], h1[k + 10]);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        double x = h2[k];
        x = dpp_add_f64<0x111, 0xf>(x);      // row_shr:1
        x = dpp_add_f64<0x112, 0xf>(x);      // row_shr:2
        x = dpp_add_f64<0x114, 0xf>(x);      // row_shr:4
        x = dpp_add_f64<0x118, 0xf>(x);      // row_shr:8  -> lane 15 of each row holds the row sum
        if ((lane & 15) == 15) red[wave * kNSum + k + 10 * (lane >> 4)] = x;
    }
    __syncthreads();
    if (threadIdx.x < kNSum) {
        double s = 0.0;
        if (threadIdx.x < kNAcc) {
#pragma unroll
            for (int w = 0; w < kPassThreads / 64; w++) s += red[w * kNSum + threadIdx.x];
        }
        partials[(size_t)col * kNSum + threadIdx.x] = s;
    }
    (void)nblocks;
}

__device__ __forceinline__ void acc_block_reduce_store(Acc &a, double *partials, uint32_t nblocks) { acc_block_reduce_store(a, partials, nblocks, blockIdx.x); }

// ---------------------------------------------------------------------------
// pass, identity pairing (what the reference does: myicp.cpp:130)
// Streaming: 48 B/point read (+24 B/point written with write-back).
// ---------------------------------------------------------------------------
// VEC = 4: each thread handles 4 consecutive points per step with 16-byte loads/stores from the planar
// arrays (needs n, the target offset and the array lengths to be multiples of 4 so every column stays
// 16-byte aligned); VEC = 1 is the general form.
template <int VEC>
__global__ __launch_bounds__(kPassThreads) void k_pass_identity(PassArgs a_in, CloudSoA tgt)
{
    if (a_in.loop && a_in.loop->stop) return;
    PassArgs a = a_in;
    if (a_in.loop) a.X = a_in.loop->Xapply;        // device-driven loop: the transform k_reduce_solve left behind
    Acc acc; acc_zero(acc);
    const uint32_t stride = gridDim.x * blockDim.x * VEC;
    for (uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * VEC; i0 < a.n; i0 += stride) {
        float x[VEC], y[VEC], z[VEC], nx[VEC], ny[VEC], nz[VEC], qx[VEC], qy[VEC], qz[VEC], qnx[VEC], qny[VEC], qnz[VEC];
        const uint32_t j0 = a.tgt_offset + i0;
        if (VEC == 4) {
            *reinterpret_cast<float4 *>(x) = *reinterpret_cast<const float4 *>(a.in.x + i0);
            *reinterpret_cast<float4 *>(y) = *reinterpret_cast<const float4 *>(a.in.y + i0);
            *reinterpret_cast<float4 *>(z) = *reinterpret_cast<const float4 *>(a.in.z + i0);
            *reinterpret_cast<float4 *>(nx) = *reinterpret_cast<const float4 *>(a.in.nx + i0);
            *reinterpret_cast<float4 *>(ny) = *reinterpret_cast<const float4 *>(a.in.ny + i0);
            *reinterpret_cast<float4 *>(nz) = *reinterpret_cast<const float4 *>(a.in.nz + i0);
            *reinterpret_cast<float4 *>(qx) = *reinterpret_cast<const float4 *>(tgt.x + j0);
            *reinterpret_cast<float4 *>(qy) = *reinterpret_cast<const float4 *>(tgt.y + j0);
            *reinterpret_cast<float4 *>(qz) = *reinterpret_cast<const float4 *>(tgt.z + j0);
            *reinterpret_cast<float4 *>(qnx) = *reinterpret_cast<const float4 *>(tgt.nx + j0);
            *reinterpret_cast<float4 *>(qny) = *reinterpret_cast<const float4 *>(tgt.ny + j0);
            *reinterpret_cast<float4 *>(qnz) = *reinterpret_cast<const float4 *>(tgt.nz + j0);
        } else {
            x[0] = a.in.x[i0]; y[0] = a.in.y[i0]; z[0] = a.in.z[i0];
            nx[0] = a.in.nx[i0]; ny[0] = a.in.ny[i0]; nz[0] = a.in.nz[i0];
            qx[0] = tgt.x[j0]; qy[0] = tgt.y[j0]; qz[0] = tgt.z[j0];
            qnx[0] = tgt.nx[j0]; qny[0] = tgt.ny[j0]; qnz[0] = tgt.nz[j0];
        }
        float px[VEC], py[VEC], pz[VEC], npx[VEC], npy[VEC], npz[VEC], d2[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            px[k] = xf_row(a.X.m + 0, x[k], y[k], z[k], 1.0f); py[k] = xf_row(a.X.m + 4, x[k], y[k], z[k], 1.0f);
            pz[k] = xf_row(a.X.m + 8, x[k], y[k], z[k], 1.0f);
            npx[k] = xf_row(a.X.m + 0, nx[k], ny[k], nz[k], a.X.nrm_w); npy[k] = xf_row(a.X.m + 4, nx[k], ny[k], nz[k], a.X.nrm_w);
            npz[k] = xf_row(a.X.m + 8, nx[k], ny[k], nz[k], a.X.nrm_w);
            d2[k] = dist2(px[k], py[k], pz[k], qx[k], qy[k], qz[k]);
        }
        if (a.writeback) {
            if (VEC == 4) {
                *reinterpret_cast<float4 *>(a.out.x + i0) = *reinterpret_cast<float4 *>(px);
                *reinterpret_cast<float4 *>(a.out.y + i0) = *reinterpret_cast<float4 *>(py);
                *reinterpret_cast<float4 *>(a.out.z + i0) = *reinterpret_cast<float4 *>(pz);
                *reinterpret_cast<float4 *>(a.out.nx + i0) = *reinterpret_cast<float4 *>(npx);
                *reinterpret_cast<float4 *>(a.out.ny + i0) = *reinterpret_cast<float4 *>(npy);
                *reinterpret_cast<float4 *>(a.out.nz + i0) = *reinterpret_cast<float4 *>(npz);
            } else {
                a.out.x[i0] = px[0]; a.out.y[i0] = py[0]; a.out.z[i0] = pz[0];
                a.out.nx[i0] = npx[0]; a.out.ny[i0] = npy[0]; a.out.nz[i0] = npz[0];
            }
        }
        if (a.d2_out) {
            if (VEC == 4) *reinterpret_cast<float4 *>(a.d2_out + i0) = *reinterpret_cast<float4 *>(d2);
            else a.d2_out[i0] = d2[0];
        }
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            if (a.max_d2 > 0.0f && d2[k] > a.max_d2) continue;
            if (a.min_ndot > -1.0f && (npx[k] * qnx[k] + npy[k] * qny[k]) + npz[k] * qnz[k] < a.min_ndot) continue;
            acc_pair(acc, px[k], py[k], pz[k], npx[k], npy[k], npz[k], qx[k], qy[k], qz[k], qnx[k], qny[k], qnz[k], d2[k], a.pivot, a.p2p);
        }
    }
    acc_block_reduce_store(acc, a.partials, gridDim.x);
}

// ---------------------------------------------------------------------------
// pass, pairs given by a previous search kernel (brute force): best64[i] holds
// (d2 bits << 32 | target row).  Target rows are gathered as float4.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPassThreads) void k_pass_indexed(PassArgs a, const float4 *__restrict__ tn)
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
        const float4 q = tn[2 * (size_t)j], nq = tn[2 * (size_t)j + 1];     // one 32-byte pair record
        if (a.min_ndot > -1.0f && (npx * nq.x + npy * nq.y) + npz * nq.z < a.min_ndot) continue;
        acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, d2, a.pivot, a.p2p);
    }
    acc_block_reduce_store(acc, a.partials, gridDim.x);
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
__device__ __forceinline__ uint2 cell_range(const TargetIndex &ix, uint32_t morton)
{
    const uint32_t blk = ix.ctop[morton >> 9];
    if (blk == 0xFFFFFFFFu) return make_uint2(0u, 0u);
    return ix.cells[(size_t)blk * 512u + (morton & 511u)];
}

__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// ---------------------------------------------------------------------------
// The nearest-neighbour pass (SYMMICP_CORR_TREE) is three kernels plus the final reduce:
//   k_search_cells  thread = query: pair certificate / cell scan / two-round cell probe; what it cannot finish goes
//                   to a work list (64-shard append list)
//   k_search_walk   the work list: one thread per entry on the sparse octree (long lists: the first pass), or one
//                   wave per entry on the run tree (short lists: stragglers).  Not launched after a pass whose list
//                   was empty; the host repairs the pass if the list turns out non-empty (engine.cpp, run_pass)
//   k_accumulate    streaming: rows + 37 fp64 sums from the stored pairs (and the optional write-back)
// Blocks are dealt to XCDs round-robin, so block b is remapped to a contiguous chunk of the Morton-sorted source
// per XCD: each XCD's 4 MB L2 then serves one compact region of the target.
// ---------------------------------------------------------------------------
constexpr int kWaveFrontier = 512;        // frontier nodes per level in the wave-per-entry walk

// Every pair keeps a private copy of its target's 32-byte (point, normal) record, so that k_accumulate streams instead of
// gathering through `pos`: once an alignment has converged almost no pair changes, and the gather was what bounded
// that kernel.  k_search_cells refreshes the copy of a pair it settles itself and marks the copy STALE (w = 2 in the
// normal slot) when it hands the query to the tree walk; k_accumulate then gathers that pair once more and refreshes the
// copy itself (the walk kernel is register-bound and stays untouched).  w = 1: the point has no target at all.
// squared gap (minus the safety margin) between the query coordinate and the cell [lo, lo+h) on one axis
__device__ __forceinline__ float axis_gap2(float p, float origin, int c, float h, float margin)
{
    const float lo = origin + (float)c * h;
    const float g = fmaxf(fmaxf(lo - p, p - (lo + h)) - margin, 0.0f);
    return g * g;
}

// ---------------------------------------------------------------------------
// k_search_cells: the common case of a pass, load-balanced at (query, cell) granularity.
//   phase 1  thread = query: transform, previous-pair bound, the <= 3x3x3 cells its ball overlaps, and which of
//            them can still hold something not farther than the bound.  Each surviving (query, cell) pair becomes
//            a 13-bit work item in LDS (block-wide prefix sum gives the slots).  Queries without a previous pair or
//            with a wider ball go to the work list of the tree-walk kernels.
//   phase 2  thread = work item: one cell range, its points four at a time, result merged per query with an LDS
//            64-bit atomicMin on (d2 bits << 32 | row)  -- min is order independent, ties resolve to the lowest row.
//   phase 3  thread = query: store the pair.
// A wave's cost is now the largest single CELL in it, not the largest sum over a query's cells, and queries with 27
// cells no longer drag their whole wave through 27 steps.
//
// Pair certificates.  The search ball is padded by D = kSlackFrac * h beyond the previous pair's distance, so after
// the scan every target point other than the winner is known to be at least L = min(second-nearest scanned, bound + D)
// away from the query position p_ref (the "clear radius", stored in PassArgs::cert[i].w; 0 = no certificate).  If the
// query later sits at p with |p - p_ref| = delta, any other point is >= L - delta away from it, so while the winner's
// current distance d1' satisfies d1' + delta < L the nearest neighbour is provably the same point and the scan is
// skipped; only the distance is refreshed.  Once ICP has converged almost every pair is certified and the kernel is
// little more than phase 1.  Margins: every fp32 distance here is good to ~2.5e-7 relative (three differences, three
// squares, two sums, one 1-ulp root), so with D(p,x) >= second (1 - 2.5e-7) - delta (1 + 2.5e-7) for every other point x
// the test (d1' + delta) (1 + 2e-6) < second (1 - 1e-6) leaves D(p,x) > D(p,winner) (1 + 2e-6): the fp32 comparison the
// oracle makes cannot come out the other way (it needs 3.5e-7).  Bounds that come from cell faces (lim) carry the
// grid's own absolute margin and keep 1e-5.  Pairs whose two nearest points are closer together than that - a few
// dozen per million - have no certificate and are searched again in every pass.
// ---------------------------------------------------------------------------
// Neighbourhood certificates.  The single certificate above has only the room between the two nearest points, L - d1; while an
// alignment still drifts (the scan-like pair keeps moving by a fraction of its point spacing for a dozen passes) that room is used
// up within a pass or two and the query is scanned again, and a pair whose two nearest points are (nearly) equally far can never be
// certified at all -- at convergence the transform still jitters in its last bits, and ~70 pairs per million fail in every pass.
// So a scan also keeps the whole neighbourhood it saw: S = every target point closer to p_ref than a radius T <= lim, if these are
// at most 8.  Every point outside S is then >= T from p_ref, and while
//     min over S of d(p, s) + delta < T
// the nearest neighbour of p is a member of S: decided by the same fp32 (d2, row) comparison brute force makes, over <= 8 gathers
// instead of a scan.  The room is now T - d1, several times L - d1.  Stored per pair: the sorted positions of S (PassArgs::certk,
// 8 words, 0xFFFFFFFF: empty slot; the winner is one of them) and (T, hint) in PassArgs::hoodr.  Bit 0 of cert.w says that they are
// valid, so every writer of a certificate invalidates them by writing an even word (cert_word).  A successful test moves the
// reference to the current position: T <- T - delta and a fresh single certificate L = min(second nearest of S, T), exact bounds.
// T follows the local point spacing: a scan counts ALL points within its T, kept or not, and leaves hint = T (6.5 / count)^0.4 for the
// pair's next scan (the count grows with T^2 on a surface, T^3 in a noisy slab); a first scan starts from 2 x the previous pair's
// distance.
#ifndef SLACK_FRAC
#define SLACK_FRAC 0.0625f
#endif
// (with the grid at ~4 points per cell: 0.5 / 0.25 / 0.125 / 0.0625 / 0.031 / 0.016 of a cell edge give pass 3 of the 1M surface pair in
// 0.240 / 0.178 / 0.153 / 0.139 / 0.135 / 0.128 ms; the 8M scan pair holds 2 780 iter/s down to 0.0625 and drops to 2 645 / 2 543 below:
// its settling passes want some room.  At ~1 point per cell, rounds 1-2, 0.125 .. 1.0 were within noise.)
constexpr float kSlackFrac = SLACK_FRAC;
constexpr int kHoodSlots = 8;            // |S| <= 8, the winner included
constexpr uint32_t kHoodNone = 0xFFFFFFFFu;

// cert.w for clear radius L (> 0; 0: none) and the neighbourhood flag: never above L
__device__ __forceinline__ float cert_word(float L, bool hood)
{
    const uint32_t u = __float_as_uint(L);
    if (!(L > 0.0f)) return hood ? __uint_as_float(0xBF800001u) : 0.0f;      // -1 with the flag: neighbourhood only
    return __uint_as_float(hood ? ((u - 1u) | 1u) : (u & ~1u));
}

// Neighbourhood test of pair i (its single certificate has failed or does not exist).  In: the query's position, delta^2 = m2 to the
// reference, the current winner (sorted position, d2, row).  True: the nearest neighbour is provably a member of S; pos_w / d2w are
// the winner's, and everything the change needs is stored (position, record copy, moved reference).
__device__ __forceinline__ bool hood_test(const PassArgs &a, const TargetIndex &ix, uint32_t i, float px, float py, float pz, float m2,
                                          int32_t pos_a, float d2a, int32_t rowa, int32_t &pos_w, float &d2w)
{
    const uint4 w0 = a.certk[2 * (size_t)i], w1 = a.certk[2 * (size_t)i + 1];
    const float T = a.hoodr[i].x;
    const uint32_t cand[kHoodSlots] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float4 q[kHoodSlots];
#pragma unroll
    for (int k = 0; k < kHoodSlots; k++) q[k] = ix.tq[min(cand[k], ix.n - 1u)];       // all gathers in flight at once
    unsigned long long kw = ((unsigned long long)__float_as_uint(d2a) << 32) | (unsigned long long)(uint32_t)rowa;
    uint32_t sec = 0x7f800000u;              // d2 bits of the second nearest of S
    pos_w = pos_a;
#pragma unroll
    for (int k = 0; k < kHoodSlots; k++) {
        const float d2 = dist2(px, py, pz, q[k].x, q[k].y, q[k].z);
        if (cand[k] < ix.n && cand[k] != (uint32_t)pos_a && d2 == d2) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(uint32_t)__float_as_int(q[k].w);
            if (key < kw) { sec = (uint32_t)(kw >> 32); kw = key; pos_w = (int32_t)cand[k]; }
            else sec = min(sec, __float_as_uint(d2));
        }
    }
    d2w = __uint_as_float((uint32_t)(kw >> 32));
    const float dl = __builtin_amdgcn_sqrtf(m2);
    if (!((__builtin_amdgcn_sqrtf(d2w) + dl) * 1.000002f < T)) return false;
    if (pos_w != pos_a) {
        a.pos_out[i] = pos_w;
        store_pair_record(a, ix, i, pos_w);
    }
    // move the reference to p: everything outside S is >= T - delta from here
    const float d1 = sqrtf(d2w) * 1.000001f;
    const float Tn = (T - dl * 1.000001f) * 0.999999f;
    const float L = fminf(sqrtf(__uint_as_float(sec)) * 0.999999f, Tn);
    if (Tn > d1 * 1.001f) {
        a.cert[i] = make_float4(px, py, pz, cert_word((L > d1) ? L : 0.0f, true));
        a.hoodr[i].x = Tn;
    } else if (pos_w != pos_a) {
        a.cert[i].w = cert_word(0.0f, true);       // the moved radius would leave no room: reference and T stay, the old single certificate goes
    }
    return true;
}

// One tile of 256 queries (i >= a.n: idle lane).  Called by k_search_cells (tile = block) and by k_pass_fused (the queries its
// streaming phase could not certify).
// from_list: the queries come from k_pass_fused's list -- their certificates have just failed there and are not tried again.
// HOOD: this scan also keeps neighbourhoods (off in the passes right after the first big move, whose certificates the next
// move invalidates anyway: the kernel then runs without the extra LDS and the scattered member stores).
template <bool HOOD>
__device__ __forceinline__ void cells_tile(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, const Affine &X, const uint32_t i,
                                           const uint32_t shard, const bool from_list)
{
    __shared__ float s_px[kPassThreads], s_py[kPassThreads], s_pz[kPassThreads];
    __shared__ unsigned long long s_key[kPassThreads];
    __shared__ uint32_t s_second[kPassThreads];        // d2 bits of the second-nearest scanned point
    __shared__ int32_t s_pos[kPassThreads];
    __shared__ uint32_t s_cell0[kPassThreads];
    __shared__ __align__(8) uint16_t s_items[kPassThreads * 27];       // (the free tail also holds the 8-byte sub-items of crowded cells)
    __shared__ uint32_t s_wsum[kPassThreads / 64 + 1];
    __shared__ uint32_t s_anyw[kPassThreads / 64];     // per-wave flags for block-wide "any" votes
    __shared__ uint32_t s_nmore;                       // sub-items of crowded cells (phase 2)
    constexpr int kHoodLds = HOOD ? kPassThreads : 1;
    __shared__ float s_reach[kHoodLds];                // T of the query's neighbourhood (0: none wanted)
    __shared__ uint32_t s_hcnt[kHoodLds];              // scanned points closer than T (the first 8 go straight to certk)
    __shared__ uint32_t s_qi[kHoodLds];                // the query's index

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool active = i < a.n;
    const float inf = __int_as_float(0x7f800000);
    // the shard a query is appended to follows the QUERY's own tile of 256, whoever scans it: a block of the streaming kernels strides over
    // many tiles, and a shard's capacity (shard_capacity) only covers the tiles with (tile & 63) == shard
    const uint32_t push_shard = from_list ? ((i / kPassThreads) & (uint32_t)(kShards - 1)) : shard;

    // ---- phase 1 ----
    float px = 0.f, py = 0.f, pz = 0.f;
    Best b;
    b.d2 = inf; b.pos = -1; b.row = 0x7fffffff;
    uint32_t mask = 0;                 // cells of this query to scan this round, bit = kx + 3*ky + 9*kz
    uint32_t mask_rest = 0;            // probe: the other cells of the 3x3x3 block, scanned only if the 2x2x2 block cannot prove its best
    bool defer = false, searched = false, probe = false, uncert = false, hood_ok = false;
    float lim = 0.f, lim_full = 0.f;   // everything outside the scanned cells is at least this far from the query
    float reach = 0.f;                 // T of the neighbourhood this scan collects (0: none)
    if (active) {
        // Two memory round trips decide a certified pair: everything addressed by i first (the certificate is loaded
        // whether or not it will be needed), then the previous winner as one 16-byte load.
        const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        const int32_t prev = a.pos_prev ? a.pos_prev[i] : -1;
        float clear = 0.0f, rx = 0.0f, ry = 0.0f, rz = 0.0f;                              // L (0: no certificate), p_ref
        if (a.use_slack) { const float4 ce = a.cert[i]; rx = ce.x; ry = ce.y; rz = ce.z; clear = ce.w; }      // one 16-byte load (bit 0 of w: neighbourhood flag)
        px = xf_row(X.m + 0, x, y, z, 1.0f); py = xf_row(X.m + 4, x, y, z, 1.0f); pz = xf_row(X.m + 8, x, y, z, 1.0f);
        if (prev >= 0 && (uint32_t)prev < ix.n) {
            const float4 q = ix.tq[prev];
            const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
            const int32_t row = __float_as_int(q.w);
            if (d2 <= inf) { b.d2 = d2; b.pos = prev; b.row = row; }                      // (not NaN)
        }
        bool certified = false;
        if (a.use_slack && b.pos >= 0 && !from_list) {
            const float m2 = dist2(px, py, pz, rx, ry, rz);                               // delta^2
            certified = (__builtin_amdgcn_sqrtf(b.d2) + __builtin_amdgcn_sqrtf(m2)) * 1.000002f < clear;     // 1-ulp roots, inside the margin (clear <= 0: none)
            if (!certified && (__float_as_uint(clear) & 1u) && a.certk) {
                int32_t pw; float d2w;
                if (hood_test(a, ix, i, px, py, pz, m2, b.pos, b.d2, b.row, pw, d2w)) {
                    certified = hood_ok = true; b.pos = pw; b.d2 = d2w;
                }
            }
        }
        if (!certified) {               // (certified: same pair, nothing stored -- its distance is evaluated when someone asks: k_pairs_d2)
            uncert = true;
            defer = true;
            if (b.pos >= 0 && ix.glevel > 0) {
                const float margin = 2e-3f * ix.h;
                const float gmax = (float)ix.gdim;
                // any rb >= sqrt(d2) is valid: the hardware square root (1 ulp) with a relative pad
                const float rb = __builtin_amdgcn_sqrtf(b.d2) * 1.00001f;
                lim = rb + kSlackFrac * ix.h;
                const float r = lim + margin;
                const float lx = (px - r - ix.ox) * ix.inv_h, hx = (px + r - ix.ox) * ix.inv_h;
                const float ly = (py - r - ix.oy) * ix.inv_h, hy = (py + r - ix.oy) * ix.inv_h;
                const float lz = (pz - r - ix.oz) * ix.inv_h, hz = (pz + r - ix.oz) * ix.inv_h;
                const int x0 = (int)floorf(fminf(fmaxf(lx, 0.0f), gmax - 1.0f)), x1 = (int)floorf(fminf(fmaxf(hx, 0.0f), gmax - 1.0f));
                const int y0 = (int)floorf(fminf(fmaxf(ly, 0.0f), gmax - 1.0f)), y1 = (int)floorf(fminf(fmaxf(hy, 0.0f), gmax - 1.0f));
                const int z0 = (int)floorf(fminf(fmaxf(lz, 0.0f), gmax - 1.0f)), z1 = (int)floorf(fminf(fmaxf(hz, 0.0f), gmax - 1.0f));
                const int nx = x1 - x0, ny = y1 - y0, nz = z1 - z0;      // extra cells per axis
                if (nx <= 2 && ny <= 2 && nz <= 2) {
                    defer = false;
                    searched = true;
                    const float thr2 = lim * lim * 1.00001f;             // keep every cell that reaches into the padded ball
                    float gx[3], gy[3], gz[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        gx[k] = (k <= nx) ? axis_gap2(px, ix.ox, x0 + k, ix.h, margin) : inf;
                        gy[k] = (k <= ny) ? axis_gap2(py, ix.oy, y0 + k, ix.h, margin) : inf;
                        gz[k] = (k <= nz) ? axis_gap2(pz, ix.oz, z0 + k, ix.h, margin) : inf;
                    }
#pragma unroll
                    for (int kz = 0; kz < 3; kz++)
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                const float g2 = (gx[kx] + gy[ky]) + gz[kz];
                                if (g2 <= thr2) mask |= 1u << (kx + 3 * ky + 9 * kz);
                            }
                    s_cell0[tid] = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)z0 << 20);
                    reach = rb;            // (marks a scan; the radius is chosen below)
                }
            }
            if (defer && b.pos >= 0 && ix.glevel > 0) {
                // The previous pair is too far to bound the scan (the cloud has just moved a lot).  Probe the 27 cells
                // around the query instead: if the best point found there is closer than the nearest outer face of that
                // block, nothing outside the block can beat it and the answer is exact; otherwise it still tightens the
                // bound the tree walk starts from.  Only queries inside the grid probe.
                const float fx = (px - ix.ox) * ix.inv_h, fy = (py - ix.oy) * ix.inv_h, fz = (pz - ix.oz) * ix.inv_h;
                const float gmax = (float)ix.gdim;
                if (fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < gmax && fy < gmax && fz < gmax) {
                    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
                    const int x0 = max(cx - 1, 0), y0 = max(cy - 1, 0), z0 = max(cz - 1, 0);
                    const int x1 = min(cx + 1, ix.gdim - 1), y1 = min(cy + 1, ix.gdim - 1), z1 = min(cz + 1, ix.gdim - 1);
                    // first round: the 2x2x2 cells nearest to the query (the query's cell and, per axis, the neighbour on
                    // the side of the nearer face); bits are relative to (x0, y0, z0) like those of the full block
                    const int ax = max(cx - ((fx - (float)cx) < 0.5f ? 1 : 0), 0), bx = min(ax + 1, ix.gdim - 1);
                    const int ay = max(cy - ((fy - (float)cy) < 0.5f ? 1 : 0), 0), by = min(ay + 1, ix.gdim - 1);
                    const int az = max(cz - ((fz - (float)cz) < 0.5f ? 1 : 0), 0), bz = min(az + 1, ix.gdim - 1);
#pragma unroll
                    for (int kz = 0; kz < 3; kz++)
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                const int X = x0 + kx, Y = y0 + ky, Z = z0 + kz;
                                const uint32_t bit = 1u << (kx + 3 * ky + 9 * kz);
                                if (X <= x1 && Y <= y1 && Z <= z1) {
                                    if (X >= ax && X <= bx && Y >= ay && Y <= by && Z >= az && Z <= bz) mask |= bit;
                                    else mask_rest |= bit;
                                }
                            }
                    s_cell0[tid] = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)z0 << 20);
                    // distance from the query to the nearest face, with cells beyond it, of the small and of the full block
                    float face = inf, face8 = inf;
                    if (x0 > 0) face = fminf(face, px - (ix.ox + (float)x0 * ix.h));
                    if (y0 > 0) face = fminf(face, py - (ix.oy + (float)y0 * ix.h));
                    if (z0 > 0) face = fminf(face, pz - (ix.oz + (float)z0 * ix.h));
                    if (x1 < ix.gdim - 1) face = fminf(face, (ix.ox + (float)(x1 + 1) * ix.h) - px);
                    if (y1 < ix.gdim - 1) face = fminf(face, (ix.oy + (float)(y1 + 1) * ix.h) - py);
                    if (z1 < ix.gdim - 1) face = fminf(face, (ix.oz + (float)(z1 + 1) * ix.h) - pz);
                    if (ax > 0) face8 = fminf(face8, px - (ix.ox + (float)ax * ix.h));
                    if (ay > 0) face8 = fminf(face8, py - (ix.oy + (float)ay * ix.h));
                    if (az > 0) face8 = fminf(face8, pz - (ix.oz + (float)az * ix.h));
                    if (bx < ix.gdim - 1) face8 = fminf(face8, (ix.ox + (float)(bx + 1) * ix.h) - px);
                    if (by < ix.gdim - 1) face8 = fminf(face8, (ix.oy + (float)(by + 1) * ix.h) - py);
                    if (bz < ix.gdim - 1) face8 = fminf(face8, (ix.oz + (float)(bz + 1) * ix.h) - pz);
                    lim = face8 - 2e-3f * ix.h;          // everything outside the scanned block is at least this far away
                    lim_full = face - 2e-3f * ix.h;
                    probe = true;
                    defer = false;
                    reach = __builtin_amdgcn_sqrtf(b.d2) * 1.00001f;
                }
            }
            if (defer) {
                a.pos_out[i] = b.pos;        // provisional: the tree walk starts from this bound
                a.d2_out[i] = b.d2;
                a.cert[i].w = 0.0f;           // pairs found by the walk carry no certificate
                a.pairrec[2 * (size_t)i + 1].w = 2.0f;      // ... and k_accumulate has to fetch their record again
                sl_push(wl.work, push_shard, i);
            }
        }
    }
    if (!from_list) {
        // how many pairs had to be searched this pass: tells the host when the alignment has converged far enough for the fused
        // pass (engine.cpp, batch_eligible); word 1 behind each shard counter of the work list is free (counters sit 16 words apart)
        const unsigned long long mu = __ballot(uncert);
        if (lane == 0 && mu) atomicAdd(wl.work.counts + shard * kShardStride + 1, (uint32_t)__popcll(mu));      // (one word per shard: a single word saturates near 88 atomics per microsecond)
    }
    s_px[tid] = px; s_py[tid] = py; s_pz[tid] = pz;
    s_pos[tid] = b.pos;
    s_key[tid] = (b.pos >= 0) ? (((unsigned long long)__float_as_uint(b.d2) << 32) | (unsigned long long)(uint32_t)b.row) : ~0ull;
    s_second[tid] = 0x7f800000u;
    if (HOOD && reach > 0.0f && a.certk) {
        // radius of the neighbourhood: the pair's hint from its last scan (but beyond the bound on the winner's distance), else twice
        // the previous pair's distance; never beyond what the scan covers
        const float hint = a.hoodr[i].y;
        reach = fminf(probe ? lim_full : lim, hint > 0.0f ? fmaxf(hint, 1.01f * reach) : 2.0f * reach);      // (a probe: cut to what it has scanned in the end)
    } else reach = 0.0f;
    if (HOOD) { s_reach[tid] = reach; s_hcnt[tid] = 0; s_qi[tid] = i; }
    uint32_t total = 0, total_all = 0;
    for (int round = 0; round < 2; round++) {
    // block-wide exclusive prefix sum of the item counts
    const uint32_t cnt = (uint32_t)__popc(mask);
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) s_wsum[wave] = incl;
    if (round == 0) {
        const unsigned long long any3 = __ballot(searched || probe);
        if (lane == 0) s_anyw[wave] = (any3 != 0ull);
    }
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < kPassThreads / 64; w++) {
        const uint32_t ws = s_wsum[w];
        if (w < wave) base += ws;
        total += ws;
    }
    total_all += total;
    if (round == 0 && !ix.dbg) {
        // the common block of a converged pass: every pair certified (or handed to the walk), nothing to scan or store
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < kPassThreads / 64; w++) any |= s_anyw[w];
        if (!any) return;
    }
    {
        uint32_t slot = base + incl - cnt;
        uint32_t m = mask;
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            s_items[slot++] = (uint16_t)((tid << 5) | bit);
        }
        if (tid == 0) s_nmore = 0;
    }
    __syncthreads();
    // ---- phase 2 ----
    // One thread scans one (query, cell) item -- at most kItemMax points of it: the rest of a crowded cell (the inner rings of a scan hold
    // ~100 points per finest cell) is handed over in chunks of kItemMax points, as sub-items in the free tail of the item array that the
    // whole block scans afterwards.  A wave's cost is then bounded by kItemMax points per lane, not by the most crowded cell among its 64 items.
    constexpr uint32_t kItemMax = 16;          // (8 / 32: the same within noise on the 2M scan pair)
    uint2 *s_more = reinterpret_cast<uint2 *>(s_items + ((total + 3u) & ~3u));          // (8-byte aligned: 4 uint16)
    const uint32_t more_cap = (uint32_t)(kPassThreads * 27 - ((total + 3u) & ~3u)) / 4u;
    // scan target points [j0, j1) for query q and merge the result into the query's (nearest, second nearest, position); every thread
    // of the block calls it (have = false: nothing to scan) -- it contains a barrier
    auto scan_merge = [&](bool have, int q, uint32_t j0, uint32_t j1) {
        unsigned long long mykey = ~0ull;
        uint32_t my2nd = 0x7f800000u;          // d2 bits of this range's second-nearest point
        int32_t mypos = -1;
        if (have) {
            const float qx = s_px[q], qy = s_py[q], qz = s_pz[q];
            const float t2 = HOOD ? s_reach[q] * s_reach[q] : 0.0f;
            for (uint32_t j = j0; j < j1; j += 4) {
                float4 t4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t4[k] = ix.tq[min(j + (uint32_t)k, j1 - 1)];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (j + (uint32_t)k < j1) {
                        const float d2 = dist2(qx, qy, qz, t4[k].x, t4[k].y, t4[k].z);
                        if (HOOD && d2 < t2) {                         // a member of the query's neighbourhood
                            const uint32_t hs = atomicAdd(&s_hcnt[q], 1u);
                            if (hs < (uint32_t)kHoodSlots) reinterpret_cast<uint32_t *>(a.certk)[(size_t)s_qi[q] * kHoodSlots + hs] = j + (uint32_t)k;
                        }
                        if (d2 == d2) {
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) |
                                                           (unsigned long long)(uint32_t)__float_as_int(t4[k].w);
                            if (key < mykey) { my2nd = (uint32_t)(mykey >> 32); mykey = key; mypos = (int32_t)(j + k); }
                            else my2nd = min(my2nd, __float_as_uint(d2));
                        }
                    }
                }
            }
            if (mypos >= 0) {
                // Every key that loses against the running minimum -- ours, or the one we displace -- is a candidate for
                // the query's second-nearest distance; the final winner is the only key never offered.  (Deciding
                // "am I the winner" per round would be wrong: a later round can still displace this round's winner.)
                const unsigned long long old = atomicMin(&s_key[q], mykey);
                uint32_t offer;
                if (mykey < old) offer = min((uint32_t)(old >> 32), my2nd);
                else if (mykey == old) offer = my2nd;              // the previous pair found again in its cell
                else offer = (uint32_t)(mykey >> 32);
                atomicMin(&s_second[q], offer);
            }
        }
        __syncthreads();
        if (mypos >= 0 && s_key[q] == mykey) s_pos[q] = mypos;
    };
    for (uint32_t it0 = 0; it0 < total; it0 += kPassThreads) {
        const uint32_t it = it0 + tid;
        int q = 0;
        uint32_t j0 = 0, j1 = 0;
        if (it < total) {
            const uint32_t item = s_items[it];
            q = (int)(item >> 5);
            const int bit = (int)(item & 31u);
            const int kz = bit / 9, ky = (bit - 9 * kz) / 3, kx = bit - 9 * kz - 3 * ky;
            const uint32_t c0 = s_cell0[q];
            const uint32_t cx = (c0 & 1023u) + (uint32_t)kx, cy = ((c0 >> 10) & 1023u) + (uint32_t)ky, cz = (c0 >> 20) + (uint32_t)kz;
            const uint2 rng = cell_range(ix, (spread3(cz) << 2) | (spread3(cy) << 1) | spread3(cx));
            if (ix.dbg) atomicAdd(ix.dbg + 3, (unsigned long long)(rng.y - rng.x));
            j0 = rng.x; j1 = rng.y;
            // a crowded cell: chunks are handed over from the end while the tail has room; what is left stays with this thread
            while (j1 > j0 && j1 - j0 > kItemMax) {
                const uint32_t js = j0 + ((j1 - j0 - 1u) / kItemMax) * kItemMax;      // start of the last chunk
                const uint32_t e = atomicAdd(&s_nmore, 1u);
                if (e >= more_cap) break;
                s_more[e] = make_uint2((uint32_t)q | ((j1 - js) << 8), js);
                j1 = js;
            }
        }
        scan_merge(it < total && j1 > j0, q, j0, j1);
    }
    __syncthreads();
    // sub-items of crowded cells
    {
        const uint32_t nmore = min(s_nmore, more_cap);
        for (uint32_t e0 = 0; e0 < nmore; e0 += kPassThreads) {
            const uint32_t e = e0 + tid;
            int q = 0;
            uint32_t j0 = 0, j1 = 0;
            if (e < nmore) { const uint2 m = s_more[e]; q = (int)(m.x & 255u); j0 = m.y; j1 = m.y + (m.x >> 8); }
            scan_merge(e < nmore && j1 > j0, q, j0, j1);
        }
    }
    __syncthreads();

    // A probe that scanned its 2x2x2 block: done if the best point is closer than the nearest outer face of that block,
    // else the other cells of the 3x3x3 block are scanned in a second round (about one query in ten on a surface cloud).
    bool more = false;
    if (round == 0 && probe) {
        const float d1 = sqrtf(__uint_as_float((uint32_t)(s_key[tid] >> 32))) * 1.00001f;
        // (2 % of room at least: a best that only just clears the small block's face would hold a certificate no jitter survives)
        if (!(d1 * 1.02f < lim)) { more = (mask_rest != 0); lim = lim_full; }
    }
    mask = more ? mask_rest : 0u;
    if (round == 1) break;
    {
        const unsigned long long anym = __ballot(more);
        if (lane == 0) s_anyw[wave] = (anym != 0ull);
        __syncthreads();
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < kPassThreads / 64; w++) any |= s_anyw[w];
        if (!any) break;
    }
    }
    if (ix.dbg) {
        // debug counters: [0] (query,cell) items, [1] certified, [2] cell scans, [6] probes, [7] handed to the walk in phase 1
        if (tid == 0) atomicAdd(ix.dbg + 0, (unsigned long long)total_all);
        const unsigned long long mc = __ballot(active && !defer && !searched && !probe), ms = __ballot(searched), mp = __ballot(probe), md = __ballot(defer);
        const unsigned long long mh = __ballot(hood_ok);
        if (lane == 0) {
            atomicAdd(ix.dbg + 8, (unsigned long long)__popcll(mh));
            atomicAdd(ix.dbg + 1, (unsigned long long)__popcll(mc)); atomicAdd(ix.dbg + 2, (unsigned long long)__popcll(ms));
            atomicAdd(ix.dbg + 6, (unsigned long long)__popcll(mp)); atomicAdd(ix.dbg + 7, (unsigned long long)__popcll(md));
        }
    }

    // ---- phase 3 ----
    if (active && (searched || probe)) {
        const unsigned long long key = s_key[tid];
        const float d1sq = __uint_as_float((uint32_t)(key >> 32));
        const float d1 = sqrtf(d1sq) * 1.000001f;
        a.pos_out[i] = s_pos[tid];
        a.d2_out[i] = d1sq;
        if (probe && !(d1 < lim)) {
            // the probe could not prove its best: the tree walk takes over from this (tighter) bound
            a.cert[i].w = 0.0f;
            a.pairrec[2 * (size_t)i + 1].w = 2.0f;
            sl_push(wl.work, push_shard, i);
        } else {
            // certificate for the following passes
            const float second = sqrtf(__uint_as_float(s_second[tid])) * 0.999999f;
            const float L = fminf(second, lim * 0.99999f);
            // the neighbourhood: everything the scan saw closer than T, if it fits (the winner is inside: T > d1)
            bool hood = false;
            const float Tq = HOOD ? fminf(s_reach[tid], lim) : 0.0f;         // (a probe collected up to its full block's bound: cut to what it scanned)
            if (Tq > 0.0f) {
                const uint32_t hc = s_hcnt[tid];
                const float T = Tq * 0.99999f;
                hood = T > d1 && hc >= 1u && hc <= (uint32_t)kHoodSlots;
                if (hood) {
                    uint32_t *slots = reinterpret_cast<uint32_t *>(a.certk) + (size_t)i * kHoodSlots;
#pragma unroll
                    for (int k = 1; k < kHoodSlots; k++) if ((uint32_t)k >= hc) slots[k] = kHoodNone;
                }
                // the radius that would have held ~6.5 points: count ~ T^2.5 between a surface and a slab
                const float scale = fminf(fmaxf(__powf(6.5f / ((float)hc + 0.5f), 0.4f), 0.4f), 1.5f);
                a.hoodr[i] = make_float2(hood ? T : 0.0f, Tq * scale);
                if (ix.dbg) atomicAdd(ix.dbg + (hood ? 9 : 10), 1ull);
            }
            const float Lw = cert_word((L > d1) ? L : 0.0f, hood);
            a.cert[i] = make_float4(px, py, pz, Lw);
            store_pair_record(a, ix, i, s_pos[tid]);
        }
    }
}

template <bool HOOD>
__global__ __launch_bounds__(kPassThreads) void k_search_cells(PassArgs a, TargetIndex ix, WorkLists wl, uint32_t chunk, uint32_t qshift)
{
    // (tiles dealt to the XCDs in chunks of kCellsChunk: the regions of a cloud differ in cost, see xcd_remap_chunked; tiles past the end idle)
    // A tile is 1 << qshift queries (256, 128 or 64) scanned by all 256 threads.  A workgroup lives as long as its chain of scan rounds
    // (256 items each: three dependent gathers and a barrier): a share of 125 k queries (1M points over 8 ranks) is 488 tiles of 256 --
    // under two per CU, and the launch lasts one workgroup's 15 rounds; tiles of 64 are four times as many workgroups of 4 rounds each.
    const uint32_t tile = xcd_remap_chunked(blockIdx.x, chunk);
    const uint32_t i = threadIdx.x < (1u << qshift) ? (tile << qshift) + threadIdx.x : 0xFFFFFFFFu;
    cells_tile<HOOD>(a, ix, wl, a.X, i, tile & (kShards - 1), false);      // (the shard follows the tile: the lists' capacity assumes an even spread)
}

// ---------------------------------------------------------------------------
// k_pass_fused: the whole pass of a CONVERGED alignment in one kernel.  Once ICP has converged nearly every pair is
// certified, so the pass is a stream: per point 12 B position + 12 B normal + the pair's own 32-byte record copy + the
// 16-byte certificate, all coalesced; certified pairs are accumulated on the spot (rows of func.cpp:51-58, 37 fp64 sums).
// Separate search and accumulate kernels read the source twice and cost two launches.
//   stream   grid-stride over tiles of 256 points: certificate test (the expressions of cells_tile, phase 1: single, then the
//            neighbourhood); a pair that fails both goes to a list in LDS
//   scan     the listed queries run through cells_tile, 256 at a time: exact scan, new pair, certificate and record copy
//   settle   the listed queries once more: those that now have a pair are accumulated from their fresh record copy
// A query the scan has to hand to the tree walk is appended to the work list and NOT accumulated (its record copy is
// marked stale): the pass's record then reports a non-empty list and whoever drives the loop repeats the pass through
// the separate kernels.  So does a block whose LDS list overflows (the early passes of an alignment: they keep the
// separate kernels).  The transform comes from the device-resident loop state when there is one (a.loop): passes can be
// enqueued back to back without the host (k_reduce_solve writes the next transform); a stopped loop returns at once.
// ---------------------------------------------------------------------------
constexpr int kFusedList = 2048;      // (room for the uncertified points of several 256-point tiles between two flushes)

// What the streaming loop of k_pass_fused needs of its arguments, held in VECTOR registers.  The kernel's three argument structs are ~150
// scalars, all live across the loop (the scan behind it needs them); with 102 scalar registers per wave the compiler parked them in the
// lanes of a vector register and fetched them back one v_readlane at a time: 85 of the loop's 291 vector instructions per tile.  Values
// that pass through in_vgpr are opaque to it: they stay where they are (and VGPR operands issue faster than scalar ones).
template <typename T>
__device__ __forceinline__ T in_vgpr(T v)
{
    asm volatile("" : "+v"(v));
    return v;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T *gcol(unsigned long long base)      // a global-memory pointer from its address (an opaque pointer would load as `flat`)
{
    return (__attribute__((address_space(1))) T *)base;
}
__device__ __forceinline__ float4 as_float4(f32x4_t v) { return make_float4(v.x, v.y, v.z, v.w); }
struct HotParams {
    Affine X;
    float pivot[3];
    float max_d2, min_ndot;
    int32_t p2p;
};

__device__ __forceinline__ void fused_accumulate(Acc &acc, const HotParams &h, float nx, float ny, float nz, float px, float py, float pz,
                                                 const float4 &q, const float4 &nq, float d2)
{
    const Affine &X = h.X;
    const float npx = xf_row(X.m + 0, nx, ny, nz, X.nrm_w), npy = xf_row(X.m + 4, nx, ny, nz, X.nrm_w), npz = xf_row(X.m + 8, nx, ny, nz, X.nrm_w);
    if (h.max_d2 > 0.0f && d2 > h.max_d2) return;
    if (h.min_ndot > -1.0f && (npx * nq.x + npy * nq.y) + npz * nq.z < h.min_ndot) return;
    acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, d2, h.pivot, h.p2p);
}

#ifndef FUSED_WAVES
#define FUSED_WAVES 2
#endif
#ifndef COMPACT_WAVES
#define COMPACT_WAVES 5
#endif
#ifndef FUSED_PPT
#define FUSED_PPT 1u      // points per thread of the fused pass's streaming loop
#endif
// ACC = true: the fused pass described above.  ACC = false: only the search part (stream + scan), for the passes of an alignment
// that is still settling: the separate k_search_cells lets every block of 256 queries pay the latency of the whole scan machinery
// for the dozen of them that need it; here a block streams several tiles and scans the failures 256 at a time (the walk and
// k_accumulate follow as usual).  No normals, no record copies read, no sums: registers for 5 waves per SIMD.
template <bool ACC>
__global__ __launch_bounds__(kPassThreads, ACC ? FUSED_WAVES : COMPACT_WAVES) void k_pass_fused(PassArgs a, TargetIndex ix, WorkLists wl)
{
    constexpr int kList = ACC ? kFusedList : kFusedList / 2;      // (the search-only form keeps 5 workgroups per CU: its tile is 256 points)
    __shared__ uint32_t s_list[kList];
    __shared__ uint32_t s_cnt, s_total;
    __shared__ unsigned long long s_col[ACC ? 10 : 1];      // base addresses of the streamed columns (read back at the top of every tile: see in_vgpr)
    // device-driven loop: stop flag and transform come from device memory, written by the kernel just before this one -- a cold round trip.
    // They are requested here and first LOOKED AT behind the first tile's loads (which do not depend on them): one round trip, not two
    // (behind the barrier: scalar loads and LDS stores share a counter, and the barrier waits for it)
    Acc acc;
    if (ACC) acc_zero(acc);
    if (threadIdx.x == 0) { s_cnt = 0; s_total = 0; }
    if (ACC && threadIdx.x == 0) {
        s_col[0] = (unsigned long long)a.in.x; s_col[1] = (unsigned long long)a.in.y; s_col[2] = (unsigned long long)a.in.z;
        s_col[3] = (unsigned long long)a.in.nx; s_col[4] = (unsigned long long)a.in.ny; s_col[5] = (unsigned long long)a.in.nz;
        s_col[6] = (unsigned long long)a.pairrec; s_col[7] = (unsigned long long)a.cert;
    }
    __syncthreads();
    int stop = 0;
    HotParams h;
    if (ACC && a.loop) {
        // (vector loads through a pointer the compiler cannot see through: the transform arrives in VGPRs, behind the first tile's loads like the flag)
        stop = a.loop->stop;
        const float4 *xp = reinterpret_cast<const float4 *>(&in_vgpr(a.loop)->Xapply);
        const float4 r0 = xp[0], r1 = xp[1], r2 = xp[2];
        h.X.m[0] = r0.x; h.X.m[1] = r0.y; h.X.m[2] = r0.z; h.X.m[3] = r0.w;
        h.X.m[4] = r1.x; h.X.m[5] = r1.y; h.X.m[6] = r1.z; h.X.m[7] = r1.w;
        h.X.m[8] = r2.x; h.X.m[9] = r2.y; h.X.m[10] = r2.z; h.X.m[11] = r2.w;
        h.X.nrm_w = reinterpret_cast<const float *>(xp)[12];
    } else if (ACC) {
#pragma unroll
        for (int k = 0; k < 12; k++) h.X.m[k] = in_vgpr(a.X.m[k]);
        h.X.nrm_w = in_vgpr(a.X.nrm_w);
    } else {
        // (the search-only form runs at 5 waves per SIMD, 96 registers: no room for resident copies)
        h.X = a.X;
        if (a.loop) { stop = a.loop->stop; h.X = a.loop->Xapply; }
    }
    const Affine &X = h.X;
#pragma unroll
    for (int k = 0; k < 3; k++) h.pivot[k] = ACC ? in_vgpr(a.pivot[k]) : a.pivot[k];
    h.max_d2 = ACC ? in_vgpr(a.max_d2) : a.max_d2; h.min_ndot = ACC ? in_vgpr(a.min_ndot) : a.min_ndot;
    h.p2p = a.p2p;
#ifdef RS_STAMPS2
    const unsigned long long fs0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long fs1 = 0;
#endif
    const uint32_t shard = blockIdx.x & (kShards - 1);

    // the listed queries (their single certificate has failed): (ACC) first the neighbourhood certificate -- the winner is a member of the set
    // the last scan kept: decided among 8 gathers, accumulated on the spot -- then the exact scan of the others 256 at a time, then (ACC) those
    // that now have a pair are accumulated from their fresh record copy.  (The neighbourhood test used to sit in the streaming loop; with 4
    // points per thread there it is 4 inlined copies and the kernel spills.)
    auto flush = [&]() {
        const uint32_t cnt = s_cnt;
        if (ACC && a.certk) {
            for (uint32_t base = 0; base < cnt; base += kPassThreads) {
                const uint32_t e = base + threadIdx.x;
                if (e >= cnt) continue;
                const uint32_t i = s_list[e];
                const float4 ce = a.cert[i];
                if (!(__float_as_uint(ce.w) & 1u)) continue;
                const float4 q = a.pairrec[2 * (size_t)i], nq = a.pairrec[2 * (size_t)i + 1];
                const int32_t pk = a.pos_prev[i];
                if (nq.w != 0.0f || (uint32_t)pk >= ix.n) continue;
                const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
                const float px = xf_row(X.m + 0, x, y, z, 1.0f), py = xf_row(X.m + 4, x, y, z, 1.0f), pz = xf_row(X.m + 8, x, y, z, 1.0f);
                const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
                if (!(d2 <= __int_as_float(0x7f800000))) continue;
                int32_t pw; float d2w;
                if (hood_test(a, ix, i, px, py, pz, dist2(px, py, pz, ce.x, ce.y, ce.z), pk, d2, __float_as_int(q.w), pw, d2w)) {
                    float4 qw = q, nqw = nq;
                    if (pw != pk) { qw = ix.tn[2 * (size_t)pw]; nqw = ix.tn[2 * (size_t)pw + 1]; }
                    const float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
                    fused_accumulate(acc, h, nx, ny, nz, px, py, pz, qw, nqw, d2w);
                    s_list[e] = 0xFFFFFFFFu;                       // settled (an idle lane of the scan below)
                }
            }
            __syncthreads();
        }
        for (uint32_t base = 0; base < cnt; base += kPassThreads) {
            const uint32_t e = base + threadIdx.x;
            cells_tile<true>(a, ix, wl, X, e < cnt ? s_list[e] : 0xFFFFFFFFu, shard, true);
            __syncthreads();
        }
        if (ACC) {
            for (uint32_t base = 0; base < cnt; base += kPassThreads) {      // (each thread meets the entries it scanned itself)
                const uint32_t e = base + threadIdx.x;
                if (e >= cnt) continue;
                const uint32_t i = s_list[e];
                if (i == 0xFFFFFFFFu) continue;                        // settled by its neighbourhood above
                const float4 q = a.pairrec[2 * (size_t)i], nq = a.pairrec[2 * (size_t)i + 1];
                if (nq.w != 0.0f) continue;                            // no target at all, or handed to the walk
                const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
                const float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
                const float px = xf_row(X.m + 0, x, y, z, 1.0f), py = xf_row(X.m + 4, x, y, z, 1.0f), pz = xf_row(X.m + 8, x, y, z, 1.0f);
                fused_accumulate(acc, h, nx, ny, nz, px, py, pz, q, nq, dist2(px, py, pz, q.x, q.y, q.z));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_total += cnt; s_cnt = 0; }
        __syncthreads();
    };

    // A tile = kPpt x 256 consecutive points; a thread takes the points tid, tid + 256, ... of its tile (every load coalesced).  kPpt = 1:
    // 2 and 4 points per thread with everything requested up front (144 / 288 B per lane in flight instead of 72) were measured in rounds 2
    // and 3 and are no faster at any size: the loop is bound by its arithmetic and one round trip per tile, not by bytes in flight
    constexpr uint32_t kPpt = ACC ? FUSED_PPT : 1u;
    constexpr uint32_t kTilePts = kPpt * kPassThreads;
    const uint32_t tiles = (a.n + kTilePts - 1) / kTilePts;
    // ACC: one contiguous eighth of the tiles per XCD (uniform work, best locality); search only: chunks of 16 tiles (the regions of a
    // cloud differ in cost: xcd_remap_chunked)
    constexpr uint32_t kTileChunk = 16;
    const uint32_t tiles_p = ACC ? ((tiles + 7u) / 8u) * 8u : ((tiles + 8u * kTileChunk - 1u) / (8u * kTileChunk)) * (8u * kTileChunk);
    constexpr int kCheck = ACC ? 1 : 1;
    int since = 0;
    // ---- stream (tile numbers are dealt so that each XCD works on one contiguous part of the sorted source)
    for (uint32_t t = blockIdx.x; t < tiles_p; t += gridDim.x) {
        const uint32_t i0 = (ACC ? xcd_remap(t, tiles_p) : xcd_remap_chunked(t, kTileChunk)) * kTilePts + threadIdx.x;
        if (ACC) asm volatile("" ::: "memory");        // (the column table is read here, per tile: hoisted out of the loop it would be 24 more live registers)
        // one round trip: everything the common case (certified pair) needs.  (Measured and dropped: letting the last block to finish
        // reduce and solve in place of k_reduce_solve with device-scope fences -- every block then waits for an L2 write-back, 79 us
        // per pass instead of 27 + 12.)
        float x[kPpt], y[kPpt], z[kPpt], nx[kPpt], ny[kPpt], nz[kPpt];
        float4 q[kPpt], nq[kPpt], ce[kPpt];
        int32_t pa[kPpt];
#pragma unroll
        for (uint32_t k = 0; k < kPpt; k++) {
            const uint32_t i = i0 + k * kPassThreads;
            nq[k] = make_float4(0.f, 0.f, 0.f, 2.0f);
            pa[k] = -1;
            if (i < a.n) {
                if (ACC) {
                    const size_t o = (size_t)i;
                    x[k] = gcol<float>(s_col[0])[o]; y[k] = gcol<float>(s_col[1])[o]; z[k] = gcol<float>(s_col[2])[o];
                    nx[k] = gcol<float>(s_col[3])[o]; ny[k] = gcol<float>(s_col[4])[o]; nz[k] = gcol<float>(s_col[5])[o];
                    q[k] = as_float4(gcol<f32x4_t>(s_col[6])[2 * o]); nq[k] = as_float4(gcol<f32x4_t>(s_col[6])[2 * o + 1]);      // a fresh copy (w = 0) has the bits of tq[prev]
                    ce[k] = as_float4(gcol<f32x4_t>(s_col[7])[o]);
                } else {
                    x[k] = a.in.x[i]; y[k] = a.in.y[i]; z[k] = a.in.z[i];
                    pa[k] = a.pos_prev[i];
                    if ((uint32_t)pa[k] < ix.n) { q[k] = ix.tq[pa[k]]; nq[k].w = 0.0f; }
                    ce[k] = a.cert[i];
                }
            }
        }
        if (stop) return;                 // (uniform; nothing has been written yet)
#ifdef RS_STAMPS2
        if (fs1 == 0) fs1 = __builtin_amdgcn_s_memrealtime();       // loop state and the first tile's data are here
#endif
#pragma unroll
        for (uint32_t k = 0; k < kPpt; k++) {
            const uint32_t i = i0 + k * kPassThreads;
            if (i < a.n) {
                const float px = xf_row(X.m + 0, x[k], y[k], z[k], 1.0f), py = xf_row(X.m + 4, x[k], y[k], z[k], 1.0f), pz = xf_row(X.m + 8, x[k], y[k], z[k], 1.0f);
                bool certified = false;
                float d2 = 0.0f;
                if (nq[k].w == 0.0f) {
                    d2 = dist2(px, py, pz, q[k].x, q[k].y, q[k].z);
                    const float m2 = dist2(px, py, pz, ce[k].x, ce[k].y, ce[k].z);
                    certified = (__builtin_amdgcn_sqrtf(d2) + __builtin_amdgcn_sqrtf(m2)) * 1.000002f < ce[k].w;          // cells_tile, phase 1
                    if (!ACC && !certified && d2 <= __int_as_float(0x7f800000) && (__float_as_uint(ce[k].w) & 1u) && a.certk) {
                        // the neighbourhood certificate: the winner is a member of the set the last scan kept (ACC: in the flush)
                        const int32_t pk = pa[k];
                        int32_t pw; float d2w;
                        if ((uint32_t)pk < ix.n && hood_test(a, ix, i, px, py, pz, m2, pk, d2, __float_as_int(q[k].w), pw, d2w)) {
                            certified = true;
                            d2 = d2w;
                        }
                    }
                }
                if (certified) {
                    // (the refreshed distance is not stored: 4 of the pass's 76 bytes per point; symmicp_get_correspondences evaluates it)
                    if (ACC) fused_accumulate(acc, h, nx[k], ny[k], nz[k], px, py, pz, q[k], nq[k], d2);
                } else {
                    s_list[atomicAdd(&s_cnt, 1u)] = i;                 // (room for a whole tile: see the flush below)
                }
            }
        }
        // room for another tile?  (uniform: everyone reads s_cnt after the barrier)
        if (++since == kCheck) {
            since = 0;
            __syncthreads();
            if (s_cnt > (uint32_t)(kList - kCheck * (int)kTilePts)) flush();
        }
    }
    __syncthreads();
    flush();
    // pairs that had to be searched (see cells_tile)
    if (threadIdx.x == 0 && s_total) atomicAdd(wl.work.counts + shard * kShardStride + 1, s_total);
#ifdef RS_STAMPS2
    const unsigned long long fs2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (ACC) acc_block_reduce_store(acc, a.partials, a.partial_cols ? a.partial_cols : gridDim.x);
#ifdef RS_STAMPS2
    if (ACC && blockIdx.x == 0 && threadIdx.x == 0) {
        double *dbg = a.partials + (size_t)8191 * kNSum;
        dbg[0] = (double)(fs1 - fs0); dbg[1] = (double)(fs2 - fs1); dbg[2] = (double)(__builtin_amdgcn_s_memrealtime() - fs2);
    }
#endif
}

// ---------------------------------------------------------------------------
// Stragglers.  In a converged pass only a few thousand queries are left on the work list (source
// points outside the overlap: far from the target, but with a TIGHT bound from their previous pair).
// One thread per query would serialise ~100 dependent loads each (hundreds of microseconds for a
// handful of waves), so when the list is short each query gets a whole WAVE: a level-synchronous
// branch-and-bound over the same box tree.  The frontier of one level lives in LDS; a batch of 8
// frontier nodes x 8 children is tested by the 64 lanes at once; survivors are compacted with
// __ballot + popcount.  The pruning radius is min(best real candidate, smallest max-distance of any
// box seen) -- a non-empty box guarantees a point within its farthest corner.  Leaves: 8 leaves x 8
// points per batch, wave-wide argmin on (d2 bits << 32 | row).  If a frontier outgrows its LDS slot
// lane 0 finishes the query with the per-thread octree walk instead (exact either way).
// ---------------------------------------------------------------------------

__device__ __forceinline__ float boxmaxdist2(float px, float py, float pz, const float4 &lo, const float4 &hi)
{
    float dx = fmaxf(fabsf(px - lo.x), fabsf(px - hi.x));
    float dy = fmaxf(fabsf(py - lo.y), fabsf(py - hi.y));
    float dz = fmaxf(fabsf(pz - lo.z), fabsf(pz - hi.z));
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

// k_search_walk: ONE launch for both regimes of the work list (single-wave blocks: walk lengths differ several-fold
// between regions, and single-wave blocks let the dispatcher backfill CUs at wave granularity).
//   long list  (first pass: every query)      -> one THREAD per entry, near-first walk of the sparse octree
//   short list (stragglers of later passes)   -> one WAVE per entry (the scheme described above), first 8192 blocks
constexpr int kWalkThreads = 64;
constexpr uint32_t kWaveWorkers = 8192;

// BUDGETED instantiation (sharded runs, first pass of a small share): `list` is the list to drain (the work list, or the
// retry list of the previous launch); in the one-thread-per-query regime a walk gets `budget` node visits, and what it
// has not settled by then goes to wl.retry with the best point seen so far as its bound, for a second launch in the
// wave-per-query regime.  Why: a wave is as slow as its slowest lane (471 visits against a mean of 81 on the 1M-point
// surface pair), and with ~100k queries per rank that tail is the whole kernel.
template <bool BUDGETED>
__global__ __launch_bounds__(kWalkThreads, 6) void k_search_walk(PassArgs a, TargetIndex ix, WorkLists wl, ShardList list, uint32_t kWaveModeMax,
                                                                 uint32_t budget)
{
    if (a.loop) {                                   // straggler stage of a device-driven loop
        if (a.loop->stop) return;
        a.X = a.loop->Xapply;
    }
    __shared__ uint32_t fr[1][2][kWaveFrontier];
    __shared__ uint32_t pre[kShards + 1];
    sl_prefix(list, pre);
    const float inf = __int_as_float(0x7f800000);
    if (pre[kShards] > kWaveModeMax) {
        // ---- long list: one thread per entry (the launcher normally sizes the grid so that this loop runs once) ----
        for (uint32_t g = blockIdx.x * kWalkThreads + threadIdx.x; ; g += gridDim.x * kWalkThreads) {
            uint32_t i;
            if (!sl_locate(list, pre, g, i)) break;
            const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
            const float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
            Best b;
            b.pos = a.pos_out[i]; b.d2 = a.d2_out[i]; b.row = 0x7fffffff;
            if (b.pos >= 0) b.row = __float_as_int(ix.tq[b.pos].w);
            const uint32_t visits = oct_walk<BUDGETED>(ix, px, py, pz, b, budget);
            a.pos_out[i] = b.pos;
            a.d2_out[i] = b.d2;
            if (BUDGETED && visits > budget) sl_push(wl.retry, blockIdx.x & (kShards - 1), i);      // unfinished: the wave regime takes over
            if (ix.dbg) {
                uint32_t mx = visits;
                for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
                atomicAdd(ix.dbg + 4, (unsigned long long)visits);
                if ((threadIdx.x & 63) == 0) atomicAdd(ix.dbg + 5, (unsigned long long)mx * 64ull);
            }
        }
        return;
    }
    // ---- short list: one wave per entry ----
    if (blockIdx.x >= kWaveWorkers) return;
    const int lane = threadIdx.x, wave = 0;
    const uint32_t nwaves = min(gridDim.x, kWaveWorkers);
    for (uint32_t w = blockIdx.x; ; w += nwaves) {
        uint32_t i;
        if (!sl_locate(list, pre, w, i)) break;
        const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        const float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
        // wave-uniform best: key = (d2 bits << 32 | row), plus the sorted position of that row
        int32_t bpos = a.pos_out[i];
        float bd2 = a.d2_out[i];
        unsigned long long bkey = ~0ull;
        if (bpos >= 0) bkey = ((unsigned long long)__float_as_uint(bd2) << 32) | (unsigned long long)(uint32_t)__float_as_int(ix.tq[bpos].w);
        else bd2 = inf;
        float radius2 = bd2;                    // pruning radius: never below the true nearest distance
        const float pad = kSlackFrac * ix.h;    // scan D beyond it, so the result carries a pair certificate (k_search_cells)
        bool overflowed = false;
        // top level: <= 8 nodes, lanes 0..7
        uint32_t nf = 0;
        {
            float mind = inf, maxd = inf;
            if (lane < (int)ix.ntop) {
                const float4 *bx = ix.boxes + 2 * ((size_t)ix.level_off[ix.top] + lane);
                const float4 lo = bx[0], hi = bx[1];
                if (lo.x <= hi.x) { mind = boxdist2(px, py, pz, lo, hi); maxd = boxmaxdist2(px, py, pz, lo, hi); }
            }
            radius2 = fminf(radius2, wave_min_f32(maxd));
            const float rp = __builtin_amdgcn_sqrtf(radius2) * 1.00001f + pad;
            const bool keep = mind <= rp * rp * 1.00001f && mind < inf;
            const unsigned long long m = __ballot(keep);
            if (keep) fr[wave][0][__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)lane;
            nf = (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        int cur = 0;
        for (int L = ix.top; L >= 1 && !overflowed; L--) {
            // expand the frontier of level L into level L-1
            uint32_t nn = 0;
            for (uint32_t f0 = 0; f0 < nf; f0 += 8) {
                const uint32_t f = f0 + (uint32_t)(lane >> 3);
                float mind = inf, maxd = inf;
                uint32_t child = 0;
                if (f < nf) {
                    child = (fr[wave][cur][f] << 3) + (uint32_t)(lane & 7);
                    const float4 *bx = ix.boxes + 2 * ((size_t)ix.level_off[L - 1] + child);
                    const float4 lo = bx[0], hi = bx[1];
                    if (lo.x <= hi.x) { mind = boxdist2(px, py, pz, lo, hi); maxd = boxmaxdist2(px, py, pz, lo, hi); }
                }
                radius2 = fminf(radius2, wave_min_f32(maxd));
                const float rp = __builtin_amdgcn_sqrtf(radius2) * 1.00001f + pad;
                const bool keep = mind <= rp * rp * 1.00001f && mind < inf;
                const unsigned long long m = __ballot(keep);
                const uint32_t slot = nn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (keep && slot < (uint32_t)kWaveFrontier) fr[wave][cur ^ 1][slot] = child;
                nn += (uint32_t)__popcll(m);
            }
            if (nn > (uint32_t)kWaveFrontier) { overflowed = true; break; }
            nf = nn;
            cur ^= 1;
            __builtin_amdgcn_wave_barrier();
        }
        if (overflowed) {
            // a frontier outgrew its LDS slot (loose bound): lane 0 finishes this one with the per-thread octree walk
            if (lane == 0) {
                Best b;
                b.pos = bpos; b.d2 = bd2; b.row = (bkey == ~0ull) ? 0x7fffffff : (int32_t)(uint32_t)(bkey & 0xFFFFFFFFull);
                oct_walk(ix, px, py, pz, b);
                a.pos_out[i] = b.pos;
                a.d2_out[i] = b.d2;
            }
            continue;
        }
        // frontier = leaves: 8 leaves x 8 points per batch; keep the two smallest keys seen
        uint32_t second = 0x7f800000u;          // d2 bits of the nearest point that is not the winner
        for (uint32_t f0 = 0; f0 < nf; f0 += 8) {
            const uint32_t f = f0 + (uint32_t)(lane >> 3);
            unsigned long long key = ~0ull;
            uint32_t j = 0;
            if (f < nf) {
                j = fr[wave][cur][f] * kLeaf + (uint32_t)(lane & 7);
                if (j < ix.n) {
                    const float4 q = ix.tq[j];
                    const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
                    if (d2 == d2) key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(uint32_t)__float_as_int(q.w);
                }
            }
            const unsigned long long best = wave_min_u64(key);
            const unsigned long long runner = wave_min_u64(key == best ? ~0ull : key);
            if (best < bkey) {
                second = min(second, min((uint32_t)(bkey >> 32), (uint32_t)(runner >> 32)));
                bkey = best;
                const unsigned long long who = __ballot(key == best);
                bpos = (int32_t)__shfl((int)j, __ffsll((long long)who) - 1, 64);
            } else if (best == bkey) {
                second = min(second, (uint32_t)(runner >> 32));           // the previous pair found again
            } else {
                second = min(second, (uint32_t)(best >> 32));
            }
        }
        if (lane == 0 && bkey != ~0ull) {
            // everything not scanned was pruned beyond (nearest + pad): same certificate as in k_search_cells
            const float d1 = sqrtf(__uint_as_float((uint32_t)(bkey >> 32)));
            const float L = fminf(sqrtf(__uint_as_float(second)) * 0.999999f, (d1 + pad) * 0.99999f);
            a.cert[i] = make_float4(px, py, pz, cert_word((L > d1 * 1.000001f) ? L : 0.0f, false));      // (an even word: no neighbourhood)
        }
        if (lane == 0) {
            a.pos_out[i] = bpos;
            a.d2_out[i] = (bkey == ~0ull) ? inf : __uint_as_float((uint32_t)(bkey >> 32));
        }
    }
}

// The pair's distance is recomputed from the gathered q (bit-identical to the stored one) instead of being read.
// (A 4-points-per-thread variant with 16-byte column loads was measured and is no faster: the two 16-byte gathers per
// pair bound this kernel, not the column loads.  Few blocks are: each one ends in a 40-value block reduction.)
__global__ __launch_bounds__(kPassThreads) void k_accumulate(PassArgs a, const float4 *__restrict__ tn)
{
    Acc acc; acc_zero(acc);
    const uint32_t nbp = gridDim.x;
    // grid-stride over blocks of 256 points, XCD-contiguous
    const uint32_t total_blocks = (a.n + kPassThreads - 1) / kPassThreads;
    for (uint32_t lb = xcd_remap(blockIdx.x, nbp); lb < total_blocks; lb += nbp) {
        const uint32_t i = lb * kPassThreads + threadIdx.x;
        if (i >= a.n) continue;
        const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
        const float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
        float4 q = a.pairrec[2 * (size_t)i], nq = a.pairrec[2 * (size_t)i + 1];            // the pair's own copy: coalesced
        if (nq.w == 2.0f) {
            // stale copy: the pair went through the tree walk this pass
            const int32_t pos = a.pos_out[i];
            if (pos >= 0) { q = tn[2 * (size_t)pos]; nq = tn[2 * (size_t)pos + 1]; }
            else nq = make_float4(0.f, 0.f, 0.f, 1.f);
            if (a.refresh_records) { a.pairrec[2 * (size_t)i] = q; a.pairrec[2 * (size_t)i + 1] = nq; }
        }
        const float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
        const float npx = xf_row(a.X.m + 0, nx, ny, nz, a.X.nrm_w), npy = xf_row(a.X.m + 4, nx, ny, nz, a.X.nrm_w),
                    npz = xf_row(a.X.m + 8, nx, ny, nz, a.X.nrm_w);
        if (a.writeback) {
            a.out.x[i] = px; a.out.y[i] = py; a.out.z[i] = pz;
            a.out.nx[i] = npx; a.out.ny[i] = npy; a.out.nz[i] = npz;
        }
        if (nq.w != 0.0f) continue;                       // no target for this point
        const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
        if (a.max_d2 > 0.0f && d2 > a.max_d2) continue;
        if (a.min_ndot > -1.0f && (npx * nq.x + npy * nq.y) + npz * nq.z < a.min_ndot) continue;
        acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, d2, a.pivot, a.p2p);
    }
    acc_block_reduce_store(acc, a.partials, gridDim.x);
}

// The append lists' bookkeeping words, read by the 64 lanes of one wave at once (three independent loads per lane):
//   word 0 of shard t   entries in shard t of the work list          -> len      (summed over the shards)
//   word 1 of shard t   pairs searched this pass, counted per shard  -> searched
//   word 2 of a list    appends dropped because a shard was full     -> dropped  (work + retry list; must be 0)
// Results are valid in every lane.
// Straggler stage of a device-driven loop (after k_pass_fused and k_search_walk): the pairs of the work list's queries -- the fused pass
// left them out and marked their record copies stale -- are gathered, their copies refreshed, their rows summed into `gridDim.x` partial
// columns behind the fused pass's.  Launched whether or not the list is empty (the columns must be written).
__global__ __launch_bounds__(kPassThreads) void k_accumulate_list(PassArgs a, const float4 *__restrict__ tn, ShardList list)
{
    if (a.loop) {
        if (a.loop->stop) return;
        a.X = a.loop->Xapply;
    }
    Acc acc; acc_zero(acc);
    // block b: shards b, b + gridDim.x, ...; its waves take them in turn (a list is a handful of entries: what counts is that the
    // counters and entries of all shards are requested side by side, not one shard after the other)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t shard = blockIdx.x + gridDim.x * wave; shard < (uint32_t)kShards; shard += gridDim.x * (kPassThreads / 64)) {
        const uint32_t cnt = min(list.counts[shard * kShardStride], list.cap);
        for (uint32_t e = lane; e < cnt; e += 64) {
            const uint32_t i = list.items[(size_t)shard * list.cap + e];
            const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
            const float nx = a.in.nx[i], ny = a.in.ny[i], nz = a.in.nz[i];
            const int32_t pos = a.pos_out[i];
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f), nq = make_float4(0.f, 0.f, 0.f, 1.f);
            if (pos >= 0) { q = tn[2 * (size_t)pos]; nq = tn[2 * (size_t)pos + 1]; }
            a.pairrec[2 * (size_t)i] = q; a.pairrec[2 * (size_t)i + 1] = nq;
            if (nq.w != 0.0f) continue;                   // no target for this point
            const float px = xf_row(a.X.m + 0, x, y, z, 1.0f), py = xf_row(a.X.m + 4, x, y, z, 1.0f), pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
            const float npx = xf_row(a.X.m + 0, nx, ny, nz, a.X.nrm_w), npy = xf_row(a.X.m + 4, nx, ny, nz, a.X.nrm_w),
                        npz = xf_row(a.X.m + 8, nx, ny, nz, a.X.nrm_w);
            const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
            if (a.max_d2 > 0.0f && d2 > a.max_d2) continue;
            if (a.min_ndot > -1.0f && (npx * nq.x + npy * nq.y) + npz * nq.z < a.min_ndot) continue;
            acc_pair(acc, px, py, pz, npx, npy, npz, q.x, q.y, q.z, nq.x, nq.y, nq.z, d2, a.pivot, a.p2p);
        }
    }
    acc_block_reduce_store(acc, a.partials, a.partial_cols, a.partial_col0 + blockIdx.x);
}

__device__ __forceinline__ void read_list_words(const uint32_t *cnt, int t, uint32_t &len, uint32_t &searched, uint32_t &dropped)
{
    uint32_t v = cnt[t * kShardStride], u = cnt[t * kShardStride + 1];
    uint32_t d = (t < 2) ? cnt[t * kShards * kShardStride + 2] : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += (uint32_t)__shfl_xor((int)v, off, 64);
        u += (uint32_t)__shfl_xor((int)u, off, 64);
        d += (uint32_t)__shfl_xor((int)d, off, 64);
    }
    len = v; searched = u; dropped = d;
}

// ---------------------------------------------------------------------------
// final reduce: partials[nblocks][40] -> 40 doubles.  One 256-thread block per sum: independent loads
// (8 bytes of every block's record: latency-bound, not bandwidth-bound), then a fixed pairwise tree in LDS (deterministic).  Writes the device record
// and, when given, a host-mapped copy (single-GPU read-back without a memcpy).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_final_reduce(const double *__restrict__ partials, int nblocks,
                                                      double *out_dev, double *out_host, uint32_t *ticket,
                                                      unsigned long long seq, uint32_t *counters_to_clear, int keep_nonempty)
{
    __shared__ double red[256];
    const int k = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int j = t; j < nblocks; j += 256) s += partials[(size_t)j * kNSum + k];
    red[t] = s;
    __syncthreads();
#pragma unroll
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    __shared__ int last;
    if (t == 0) {
        out_dev[k] = red[0];
        if (out_host) out_host[k] = red[0];
        // the block that takes the last ticket publishes the sequence number the host spins on
        __threadfence_system();
        last = (atomicAdd(ticket, 1u) == gridDim.x - 1);
    }
    __syncthreads();
    if (last) {
        // Shard counters of the pass's append lists.  The work list's length travels in the record's last slot (unused
        // by the sums; with several GPUs the all-reduce turns it into the total over ranks): it sizes the next pass's
        // walk grid and tells the host whether a pass that skipped the walk has to be repaired.  The counters are
        // zeroed for the next pass -- unless this pass skipped the walk and the list turned out non-empty.
        if (counters_to_clear) {
            __shared__ uint32_t s_len;
            if (t < 64) {
                uint32_t len, searched, dropped;
                read_list_words(counters_to_clear, t, len, searched, dropped);
                if (t == 0) {
                    s_len = len;
                    // the record's spare slots: list length, pairs searched this pass (cells_tile), dropped appends (sl_push: must be 0)
                    out_dev[kNSum - 1] = (double)len; out_dev[kNSum - 2] = (double)searched; out_dev[kNSum - 3] = (double)dropped;
                    if (out_host) { out_host[kNSum - 1] = (double)len; out_host[kNSum - 2] = (double)searched; out_host[kNSum - 3] = (double)dropped; }
                }
            }
            __syncthreads();
            if (t < 2) counters_to_clear[t * kShards * kShardStride + 2] = 0;
            if (t < kShards) counters_to_clear[t * kShardStride + 1] = 0;
            if (!(keep_nonempty && s_len > 0u))
                for (int c = t; c < 2 * kShards; c += 256) counters_to_clear[c * kShardStride] = 0;      // work list and retry list
        }
        if (t == 0) {
            *ticket = 0;
            if (out_host) {
                __threadfence_system();
                reinterpret_cast<volatile unsigned long long *>(out_host)[kNSum] = seq;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_reduce_solve: the end of one pass and the start of the next WITHOUT the host (device-driven runs of passes).
//   reduce   partials[40][nblocks] -> the pass's record (one 512-thread block: wave w sums rows w, w+8, ...; fixed order).
//            Sharded runs reduce with k_final_reduce, all-reduce the record over the ranks and call this kernel with
//            REDUCE = false: the record is then read from out_dev.
//   check    a non-empty work list means the fused pass left queries to the tree walk: the pass has to be redone through
//            the separate kernels (LOOP_REDO_PASS, the host takes over)
//   loop     the reference's test `diff > threshold && iters++ < max_iters` (myicp.cpp:123) on the record's slot 33
//   solve    estimateTransformSymm (func.cpp:76-102) from the record: the SAME source as the host solve (solve_core.h),
//            run by one thread, with the pivot-ratio conditioning estimate; anything but a clean solve is handed back to
//            the host's exact form (LOOP_HOST_SOLVE).  transform = increment * transform (myicp.cpp:138); the next pass
//            reads it from the loop state.
// Every pass leaves a LoopRecord in host-mapped memory (sums, increment, transform); the host reads them after the batch.
// ---------------------------------------------------------------------------
template <bool REDUCE>
__global__ __launch_bounds__(512) void k_reduce_solve(const double *__restrict__ partials, int nblocks, double *out_dev, int solve_only, LoopState *loop,
                                                       LoopConfig cfg, LoopRecord *ring, int ring_len, uint32_t *counters_to_clear)
{
    __shared__ double s_sum[kNSum];
    __shared__ uint32_t s_len, s_unc;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef RS_STAMPS
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
#endif
    // This block is one chain of dependent steps behind a kernel boundary: everything it reads from memory is requested up front, in one
    // round trip (the loop state, the work list's counters, the partial records), instead of one round trip per step (12 -> ~9 us).
    const LoopState ls = *loop;                       // (wave-uniform: scalar loads)
    uint32_t cw0 = 0, cw1 = 0, cw2 = 0;               // wave 0: the shard counters (see read_list_words)
    const bool want_counters = !solve_only && REDUCE && counters_to_clear && t < 64;
    if (want_counters) {
        cw0 = counters_to_clear[t * kShardStride]; cw1 = counters_to_clear[t * kShardStride + 1];
        cw2 = (t < 2) ? counters_to_clear[t * kShards * kShardStride + 2] : 0u;
    }
    // lane k < 40 of wave w sums slot k of the records of blocks w, w + 8, w + 16, ...: every load is one contiguous 320-byte record,
    // nothing has to cross lanes, the order is fixed.  All loads of a thread are issued before its first sum.
    __shared__ double s_part[8][kNSum];
    double accw = 0.0;
    if (!solve_only && REDUCE && lane < kNSum) {
        if (nblocks <= 512) {
            double c[64];
#pragma unroll
            for (int j = 0; j < 64; j++) { const int b = wave + 8 * j; c[j] = (b < nblocks) ? partials[(size_t)b * kNSum + lane] : 0.0; }
#pragma unroll
            for (int j = 0; j < 64; j++) accw += c[j];
        } else {
            for (int b = wave; b < nblocks; b += 8) accw += partials[(size_t)b * kNSum + lane];
        }
    }
    double rec = 0.0;
    if ((solve_only || !REDUCE) && t < kNSum) rec = out_dev[t];
    if (ls.stop) return;
#ifdef RS_STAMPS
    const unsigned long long tsA = __builtin_amdgcn_s_memrealtime();
#endif
    if (t == 0) { s_len = 0; s_unc = 0; }
    if (!solve_only && REDUCE) {
        if (lane < kNSum) s_part[wave][lane] = accw;
        __syncthreads();
        if (t < kNSum) {
            double x = 0.0;
#pragma unroll
            for (int w = 0; w < 8; w++) x += s_part[w][t];
            s_sum[t] = (t < kNAcc) ? x : 0.0;
        }
    } else if (t < kNSum) s_sum[t] = rec;
    __syncthreads();
    if (!solve_only && counters_to_clear) {
        if (REDUCE) {
            if (t < 64) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    cw0 += (uint32_t)__shfl_xor((int)cw0, off, 64);
                    cw1 += (uint32_t)__shfl_xor((int)cw1, off, 64);
                    cw2 += (uint32_t)__shfl_xor((int)cw2, off, 64);
                }
                if (t == 0) { s_len = cw2 ? 0xFFFFFFFFu : cw0; s_unc = cw1; }      // dropped appends: the host redoes the pass and reports
            }
            if (t < kShards) counters_to_clear[t * kShardStride + 1] = 0;
            for (int c = t; c < 2 * kShards; c += 512) counters_to_clear[c * kShardStride] = 0;
        } else if (t == 0) {
            // summed over the ranks by the all-reduce (k_final_reduce put them in the record)
            s_len = (s_sum[kNSum - 3] != 0.0) ? 0xFFFFFFFFu : (uint32_t)s_sum[kNSum - 1];
            s_unc = (uint32_t)s_sum[kNSum - 2];
        }
    }
    __syncthreads();
#ifdef RS_STAMPS
    const unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
#endif
    // (every thread takes the same decisions from the same words; the record is written by 40 lanes at once, not by one lane 40 times)
    int it = ls.iters;
    if (!solve_only) {
        const uint32_t len = s_len, unc = s_unc;
        // a non-empty work list: handled by the straggler stage of this very pass, or the pass has to be redone by the host; dropped
        // appends (0xFFFFFFFF) always go back
        if (cfg.tree && len > 0u && (!cfg.walk_in_loop || len == 0xFFFFFFFFu)) { if (t == 0) { loop->stop = 1; loop->reason = LOOP_REDO_PASS; } return; }
        it += 1;                                                        // this pass is complete
        LoopRecord &r = ring[it % ring_len];
        if (t < kNSum) { const double v = (t >= kNAcc) ? 0.0 : s_sum[t]; r.sums[t] = v; if (REDUCE) out_dev[t] = v; }
        if (t == 0) {
            loop->iters = it;
            r.solved = 0;
            r.pad = (int32_t)unc;
            r.list_len = (int32_t)len; r.reserved = 0;
        }
        // many pairs had to be searched again (the cloud moved): the separate kernels do that faster; this pass is complete
        if (cfg.tree && (unc > cfg.uncertified_limit || (cfg.walk_in_loop && len > cfg.list_limit))) { if (t == 0) { loop->stop = 1; loop->reason = LOOP_SLOW; } return; }
    }
    if (t != 0) return;
    // ---- myicp.cpp:123
    const float diff = (float)s_sum[33];
    if (ls.small_step || !((cfg.fixed_iters || diff > cfg.diff_threshold) && it < cfg.max_iters)) {
        loop->stop = 1; loop->reason = LOOP_DONE;
        return;
    }
    // ---- func.cpp:76-102
    symmicp_sums S;
    for (int k = 0; k < kNSum; k++) S.s[k] = (k >= kNAcc) ? 0.0 : s_sum[k];
    float pbar[3], qbar[3], av[3], tv[3], rc = 0.f, Xi[16];
    const int st = (cfg.mode == SYMMICP_MODE_QUIRKS) ? solve::solve_quirks(S, pbar, qbar, av, tv, &rc, Xi, false)
                                                      : solve::solve_paper(S, cfg.pivot, pbar, qbar, av, tv, &rc, Xi, false);
    if (st != SYMMICP_OK || !(rc > 1e-6f)) { loop->stop = 1; loop->reason = LOOP_HOST_SOLVE; return; }
#ifdef RS_STAMPS
    const unsigned long long ts2 = __builtin_amdgcn_s_memrealtime();
#endif
    float Xn[16];
    solve::mat4_mul(Xi, ls.X, Xn);                                   // myicp.cpp:138
    for (int k = 0; k < 16; k++) loop->X[k] = Xn[k];
    const float *ap = cfg.incremental ? Xi : Xn;
    for (int k = 0; k < 12; k++) loop->Xapply.m[k] = ap[k];
    loop->Xapply.nrm_w = cfg.nrm_w;
    LoopRecord &r = ring[it % ring_len];
    for (int k = 0; k < 16; k++) { r.increment[k] = Xi[k]; r.X[k] = Xn[k]; }
    r.rcond = rc; r.status = st; r.solved = 1;
#ifdef RS_STAMPS2
    r.sums[37] = partials[(size_t)8191 * kNSum]; r.sums[38] = partials[(size_t)8191 * kNSum + 1]; r.sums[39] = partials[(size_t)8191 * kNSum + 2];
#elif defined(RS_STAMPS)
    r.sums[37] = (double)(ts1 - ts0) + 1e-4 * (double)(tsA - ts0); r.sums[38] = (double)(ts2 - ts1); r.sums[39] = (double)(__builtin_amdgcn_s_memrealtime() - ts2);      // 10 ns ticks: load + reduce, bookkeeping + solve, publish
#endif
    if (cfg.eps_rotation > 0.f && cfg.eps_translation > 0.f && !cfg.fixed_iters) {
        // convergence on the increment (engine.cpp, symmicp_align): the pass that applies this increment still runs
        const double tr = ((double)Xi[0] + Xi[5] + Xi[10] - 1.0) * 0.5;
        const double ang = acos(tr > 1.0 ? 1.0 : (tr < -1.0 ? -1.0 : tr));
        const double tn = sqrt((double)Xi[3] * Xi[3] + (double)Xi[7] * Xi[7] + (double)Xi[11] * Xi[11]);
        if (ang < cfg.eps_rotation && tn < cfg.eps_translation) loop->small_step = 1;
    }
}

// last kernel of a batch: copy the loop state where the host can read it and publish the batch's sequence number
__global__ __launch_bounds__(64) void k_loop_end(const LoopState *loop, LoopState *host_copy, unsigned long long *done_flag, unsigned long long seq)
{
    if (threadIdx.x == 0) {
        *host_copy = *loop;
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(done_flag) = seq;
    }
}

// multi-GPU: after the RCCL all-reduce, copy the record to host-mapped memory and publish the sequence number
__global__ __launch_bounds__(64) void k_publish(const double *__restrict__ sums_dev, double *out_host, unsigned long long seq)
{
    if (threadIdx.x < kNSum) out_host[threadIdx.x] = sums_dev[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        reinterpret_cast<volatile unsigned long long *>(out_host)[kNSum] = seq;
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

// per-pair squared distances of the identity pairing, on request (the streaming pass does not store them)
__global__ __launch_bounds__(256) void k_identity_d2(CloudSoA in, Affine X, CloudSoA tgt, uint32_t tgt_offset, uint32_t n, float *d2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in.x[i], y = in.y[i], z = in.z[i];
    const float px = xf_row(X.m + 0, x, y, z, 1.0f), py = xf_row(X.m + 4, x, y, z, 1.0f), pz = xf_row(X.m + 8, x, y, z, 1.0f);
    const uint32_t j = tgt_offset + i;
    d2[i] = dist2(px, py, pz, tgt.x[j], tgt.y[j], tgt.z[j]);
}

// TREE: the distance of every current pair, from the positions the last pass gave the queries.  The passes store a distance only where
// they searched; a certified pair keeps its target, and the expression below is the one every kernel evaluates for it (same operations,
// same order: the same bits).  No pair: +inf, as the searches store it.
__global__ __launch_bounds__(256) void k_pairs_d2(CloudSoA in, Affine X, const int32_t *__restrict__ pos, const float4 *__restrict__ tq, uint32_t n_t, uint32_t n, float *d2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t p = pos[i];
    float v = __int_as_float(0x7f800000);
    if (p >= 0 && (uint32_t)p < n_t) {
        const float x = in.x[i], y = in.y[i], z = in.z[i];
        const float px = xf_row(X.m + 0, x, y, z, 1.0f), py = xf_row(X.m + 4, x, y, z, 1.0f), pz = xf_row(X.m + 8, x, y, z, 1.0f);
        const float4 q = tq[p];
        v = dist2(px, py, pz, q.x, q.y, q.z);
    }
    d2[i] = v;
}

void launch_pairs_d2(const CloudSoA &in, const Affine &X, const int32_t *pos, const float4 *tq, uint32_t n_t, uint32_t n, float *d2, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_pairs_d2, dim3((n + 255) / 256), dim3(256), 0, s, in, X, pos, tq, n_t, n, d2);
}

void launch_identity_d2(const CloudSoA &in, const Affine &X, const CloudSoA &tgt, uint32_t tgt_offset, uint32_t n, float *d2, hipStream_t s)
{
    hipLaunchKernelGGL(k_identity_d2, dim3((n + 255) / 256), dim3(256), 0, s, in, X, tgt, tgt_offset, n, d2);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_pass_identity(const PassArgs &a, CloudSoA tgt, int blocks, bool vec4_ok, hipStream_t s)
{
    if (vec4_ok) hipLaunchKernelGGL(k_pass_identity<4>, dim3(blocks), dim3(kPassThreads), 0, s, a, tgt);
    else hipLaunchKernelGGL(k_pass_identity<1>, dim3(blocks), dim3(kPassThreads), 0, s, a, tgt);
}

void launch_pass_indexed(const PassArgs &a, const float4 *tn, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_pass_indexed, dim3(blocks), dim3(kPassThreads), 0, s, a, tn);
}

uint32_t shard_capacity(uint32_t n_points)
{
    // a shard only receives appends from producer blocks with (blockIdx & 63) == shard
    const uint32_t nb = (n_points + kPassThreads - 1) / kPassThreads;
    const uint32_t nbp = ((nb + 7u) / 8u) * 8u;
    return ((nbp + kShards - 1) / kShards) * kPassThreads;
}

uint32_t walk_blocks_full(const WorkLists &wl)
{
    return kShards * (wl.work.cap / kWalkThreads);      // one thread per possible list entry
}

// stage: 0 = whole pass (cells, walk, accumulate); 1 = cells and accumulate only (the host expects an empty work list
// and repairs the pass otherwise); 2 = the repair: walk and accumulate
// compact_blocks > 0: the search runs as k_pass_fused<false> on that many blocks (sparse scans), else as k_search_cells
void launch_pass_tree_split(const PassArgs &a_in, const TargetIndex &ix, const WorkLists &wl, int acc_blocks, uint32_t walk_blocks,
                            int stage, int compact_blocks, const PassTuning &tune, hipStream_t s, hipEvent_t *ev)
{
    PassArgs a = a_in;
    a.refresh_records = (stage != 1) ? 1 : 0;      // a stage-1 pass may still be repaired: its accumulate must not settle stale copies
    const uint32_t wave_mode_max = tune.wave_mode_max;
    // all shard counters are zero here: cleared by the previous pass's final reduce
    const uint32_t nb = (a.n + kPassThreads - 1) / kPassThreads;
    const uint32_t nbp = ((nb + 7u) / 8u) * 8u;
    if (ev) hipEventRecord(ev[0], s);
    if (stage != 2 && nbp) {      // (nbp == 0: a rank whose share is empty)
        if (compact_blocks > 0) hipLaunchKernelGGL(k_pass_fused<false>, dim3(min((uint32_t)compact_blocks, nbp)), dim3(kPassThreads), 0, s, a, ix, wl);
        else {
            const uint32_t chunk = tune.cells_chunk ? tune.cells_chunk : 16u;      // tiles per chunk
            // queries per tile: 256 while that fills the chip a few times over (256 CUs x 6-7 workgroups), else 128 or 64 (k_search_cells)
            uint32_t qshift = nb >= 3072u ? 8u : (nb >= 1536u ? 7u : 6u);
            if (tune.cells_queries == 64u) qshift = 6u; else if (tune.cells_queries == 128u) qshift = 7u; else if (tune.cells_queries == 256u) qshift = 8u;
            const uint32_t nq = (a.n + (1u << qshift) - 1u) >> qshift;
            const uint32_t nbc = ((nq + 8u * chunk - 1u) / (8u * chunk)) * (8u * chunk);
            if (a.make_hood) hipLaunchKernelGGL(k_search_cells<true>, dim3(nbc), dim3(kPassThreads), 0, s, a, ix, wl, chunk, qshift);
            else hipLaunchKernelGGL(k_search_cells<false>, dim3(nbc), dim3(kPassThreads), 0, s, a, ix, wl, chunk, qshift);
        }
    }
    if (ev) hipEventRecord(ev[1], s);
    if (ev) hipEventRecord(ev[2], s);
    if (stage != 1 && nbp) {
        if (walk_blocks == 0 || walk_blocks > walk_blocks_full(wl)) walk_blocks = walk_blocks_full(wl);
        // first pass of an alignment: no query has a bound yet, which the wave-per-query walk needs (its frontier would
        // overflow and fall back to one lane): one thread per query whatever the list length (matters for shares or
        // clouds below the threshold, e.g. 1M points over 8 ranks)
        const uint32_t wmm = a.pos_prev ? wave_mode_max : 0u;
        if (a.budget_walk) {
            const uint32_t budget = tune.walk_budget;
            hipLaunchKernelGGL(k_search_walk<true>, dim3(walk_blocks), dim3(kWalkThreads), 0, s, a, ix, wl, wl.work, wmm, budget);
            // the retry list: wave regime forced (every entry carries a bound now), no budget
            hipLaunchKernelGGL(k_search_walk<false>, dim3(8192), dim3(kWalkThreads), 0, s, a, ix, wl, wl.retry, 0xFFFFFFFFu, 0xFFFFFFFFu);
        } else {
            hipLaunchKernelGGL(k_search_walk<false>, dim3(walk_blocks), dim3(kWalkThreads), 0, s, a, ix, wl, wl.work, wmm, 0xFFFFFFFFu);
        }
    }
    if (ev) hipEventRecord(ev[3], s);
    hipLaunchKernelGGL(k_accumulate, dim3(acc_blocks), dim3(kPassThreads), 0, s, a, ix.tn);
    if (ev) hipEventRecord(ev[4], s);
}

void launch_accumulate(const PassArgs &a, const float4 *tn, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_accumulate, dim3(blocks), dim3(kPassThreads), 0, s, a, tn);
}

void launch_final_reduce(const double *partials, int blocks, double *out_dev, double *out_host_mapped, uint32_t *ticket,
                         unsigned long long seq, uint32_t *counters_to_clear, int keep_nonempty, hipStream_t s)
{
    hipLaunchKernelGGL(k_final_reduce, dim3(kNSum), dim3(256), 0, s, partials, blocks, out_dev, out_host_mapped, ticket, seq,
                       counters_to_clear, keep_nonempty);
}

void launch_pass_fused(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int blocks, hipStream_t s)
{
    hipLaunchKernelGGL(k_pass_fused<true>, dim3(blocks), dim3(kPassThreads), 0, s, a, ix, wl);
}

void launch_loop_stragglers(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int list_blocks, const PassTuning &tune, hipStream_t s)
{
    // (short lists: the wave-per-entry regime; anything above the threshold strides one thread per entry over this grid)
    const uint32_t wave_mode_max = tune.wave_mode_max;
    hipLaunchKernelGGL(k_search_walk<false>, dim3(512), dim3(kWalkThreads), 0, s, a, ix, wl, wl.work, wave_mode_max, 0xFFFFFFFFu);
    hipLaunchKernelGGL(k_accumulate_list, dim3(list_blocks), dim3(kPassThreads), 0, s, a, ix.tn, wl.work);
}

void launch_reduce_solve(const double *partials, int blocks, double *out_dev, int mode, LoopState *loop, LoopConfig cfg, LoopRecord *ring, int ring_len,
                         uint32_t *counters_to_clear, hipStream_t s)
{
    // mode 0: reduce + check + solve (single GPU); 1: record already in out_dev (after the all-reduce); 2: solve only (start of a batch)
    if (mode == 0) hipLaunchKernelGGL(k_reduce_solve<true>, dim3(1), dim3(512), 0, s, partials, blocks, out_dev, 0, loop, cfg, ring, ring_len, counters_to_clear);
    else hipLaunchKernelGGL(k_reduce_solve<false>, dim3(1), dim3(512), 0, s, partials, blocks, out_dev, mode == 2 ? 1 : 0, loop, cfg, ring, ring_len, counters_to_clear);
}

void launch_loop_end(const LoopState *loop, LoopState *host_copy, unsigned long long *done_flag, unsigned long long seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_loop_end, dim3(1), dim3(64), 0, s, loop, host_copy, done_flag, seq);
}

void launch_publish(const double *sums_dev, double *out_host_mapped, unsigned long long seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, sums_dev, out_host_mapped, seq);
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
