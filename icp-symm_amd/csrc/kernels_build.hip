// kernels_build.hip -- one-time index build over a cloud (gfx950):
//   k_bbox        bounding box (wave64 shuffle min/max + ordered-uint atomics)
//   k_morton      30-bit Morton keys (10 bits per axis, cubic cells)
//   radix sort    stable LSD, 8 bits per pass: k_rs_hist / k_rs_scan / k_rs_scatter
//   k_gather_*    reorder points + normals into Morton order (float4 for the target)
//   k_level_hist  occupied-cell counts per octree level (picks the grid level)
//   k_cell_table  dense (first,last) table of the Morton cells at the grid level
//   k_leaf_boxes / k_node_boxes   tight boxes of the implicit 8-ary tree (linear BVH)
// These replace nothing in the reference (it has no search structure: myicp.cpp:128-131
// is a todo); they are the SURVEY 8(a) k_morton / k_radix_sort / k_cell_ranges rows.
#include <cstdlib>
#include "symmicp_internal.h"
#pragma clang fp contract(off)

namespace symmicp {

// ---- float <-> order-preserving uint --------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_bbox(const float *__restrict__ x, const float *__restrict__ y,
                                              const float *__restrict__ z, uint32_t n, uint32_t *bbox)
{
    const float inf = __int_as_float(0x7f800000);
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            // fminf/fmaxf drop NaN: poison the box instead so the host sees a non-finite extent
            if (!(fabsf(v[k]) < inf)) { lo[k] = -inf; hi[k] = inf; }
            lo[k] = fminf(lo[k], v[k]); hi[k] = fmaxf(hi[k], v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64));
        }
    // one atomic per value per BLOCK: returning/contended atomics on a single word top out near 88 per microsecond
    __shared__ float s_lo[4][3], s_hi[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s_lo[wave][k] = lo[k]; s_hi[wave][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]));
        const float h = fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]));
        atomicMin(bbox + k, f2ord(l));
        atomicMax(bbox + 3 + k, f2ord(h));
    }
}

__device__ __forceinline__ uint32_t spread3_b(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void k_morton(const float *__restrict__ x, const float *__restrict__ y,
                                                const float *__restrict__ z, uint32_t n, float ox, float oy, float oz,
                                                float inv_h0, uint32_t *keys, uint32_t *vals)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int top = (1 << kMortonBits) - 1;
    int cx = min(max((int)floorf((x[i] - ox) * inv_h0), 0), top);
    int cy = min(max((int)floorf((y[i] - oy) * inv_h0), 0), top);
    int cz = min(max((int)floorf((z[i] - oz) * inv_h0), 0), top);
    keys[i] = (spread3_b((uint32_t)cz) << 2) | (spread3_b((uint32_t)cy) << 1) | spread3_b((uint32_t)cx);
    vals[i] = i;
}

// ---------------------------------------------------------------------------
// LSD radix sort, 8 bits per pass, stable.
// A block owns a tile of kRsTile consecutive elements in every pass.
// ---------------------------------------------------------------------------
constexpr int kRsThreads = 256;
constexpr int kRsItems = 16;
constexpr int kRsTile = kRsThreads * kRsItems;   // 4096
constexpr int kRsBins = 256;

__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const uint32_t *__restrict__ keys, uint32_t n, int shift,
                                                        uint32_t nblocks, uint32_t *hist /* [bin][block] */)
{
    __shared__ uint32_t h[kRsBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile;
#pragma unroll
    for (int k = 0; k < kRsItems; k++) {
        uint32_t i = base + k * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of hist[0 .. total) in place (single block; total = 256 * nblocks)
__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t *hist, uint32_t total)
{
    __shared__ uint32_t sums[1024];
    const uint32_t per = (total + 1023) / 1024;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, total);
    uint32_t s = 0;
    for (uint32_t i = b0; i < b1; i++) s += hist[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partial sums
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= (uint32_t)off) ? sums[threadIdx.x - off] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = (threadIdx.x == 0) ? 0 : sums[threadIdx.x - 1];
    for (uint32_t i = b0; i < b1; i++) { uint32_t v = hist[i]; hist[i] = run; run += v; }
}

__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                           const uint32_t *__restrict__ vals_in, uint32_t n, int shift,
                                                           uint32_t nblocks, const uint32_t *__restrict__ hist,
                                                           uint32_t *keys_out, uint32_t *vals_out)
{
    __shared__ uint32_t base[kRsBins];                    // next free output slot per digit for this block
    __shared__ uint32_t wcount[(kRsThreads / 64)][kRsBins];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    base[threadIdx.x] = hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kRsThreads / 64; w++) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * kRsTile;
    for (int k = 0; k < kRsItems; k++) {
        const uint32_t i = tile0 + k * kRsThreads + threadIdx.x;
        const bool valid = i < n;
        uint32_t key = 0, val = 0, digit = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; digit = (key >> shift) & 0xffu; }
        // lanes of this wave holding the same digit (8 ballots), invalid lanes excluded
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        const uint32_t rank = (uint32_t)__popcll(below);
        if (valid && below == 0ull) wcount[wave][digit] = (uint32_t)__popcll(peers);   // wave leader of this digit
        __syncthreads();
        if (valid) {
            uint32_t off = base[digit] + rank;
            for (int w = 0; w < wave; w++) off += wcount[w][digit];
            keys_out[off] = key; vals_out[off] = val;
        }
        __syncthreads();
        {
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < kRsThreads / 64; w++) { add += wcount[w][threadIdx.x]; wcount[w][threadIdx.x] = 0; }
            base[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

size_t radix_sort_ws_elems(uint32_t n)
{
    size_t nblocks = ((size_t)n + kRsTile - 1) / kRsTile;
    return nblocks * kRsBins;
}

void radix_sort_pairs(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t n,
                      int key_bits, uint32_t *hist_ws, size_t hist_ws_elems, hipStream_t s)
{
    if (n == 0) return;
    const uint32_t nblocks = (n + kRsTile - 1) / kRsTile;
    (void)hist_ws_elems;
    int passes = (key_bits + 7) / 8;
    if (passes & 1) passes++;            // even pass count: result lands back in keys/vals
    uint32_t *kin = keys, *vin = vals, *kout = keys_tmp, *vout = vals_tmp;
    for (int p = 0; p < passes; p++) {
        const int shift = p * 8;
        hipLaunchKernelGGL(k_rs_hist, dim3(nblocks), dim3(kRsThreads), 0, s, kin, n, shift, nblocks, hist_ws);
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, s, hist_ws, nblocks * kRsBins);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblocks), dim3(kRsThreads), 0, s, kin, vin, n, shift, nblocks, hist_ws, kout, vout);
        uint32_t *t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
}

// ---------------------------------------------------------------------------
// gathers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_f4(const float *__restrict__ x, const float *__restrict__ y,
                                                   const float *__restrict__ z, const float *__restrict__ nx,
                                                   const float *__restrict__ ny, const float *__restrict__ nz,
                                                   const uint32_t *__restrict__ order, uint32_t n, float4 *tq, float4 *tn)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = order ? order[i] : i;
    const float4 q = make_float4(x[j], y[j], z[j], __int_as_float((int)j));
    tq[i] = q;
    // tn holds 32-byte pair records (point, normal): the accumulating kernels gather both halves from one half-line
    tn[2 * (size_t)i] = q;
    tn[2 * (size_t)i + 1] = make_float4(nx[j], ny[j], nz[j], 0.0f);
}

// host rows (x y z at float offset `offset` of every `row_stride`-float record) -> three planar columns
__global__ __launch_bounds__(256) void k_deinterleave3(const float *__restrict__ raw, size_t row_stride, size_t offset, uint32_t n,
                                                       float *__restrict__ x, float *__restrict__ y, float *__restrict__ z)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = raw + (size_t)i * row_stride + offset;
    x[i] = r[0]; y[i] = r[1]; z[i] = r[2];
}

void launch_deinterleave3(const float *raw, size_t row_stride, size_t offset, uint32_t n, float *x, float *y, float *z, hipStream_t s)
{
    hipLaunchKernelGGL(k_deinterleave3, dim3((n + 255) / 256), dim3(256), 0, s, raw, row_stride, offset, n, x, y, z);
}

__global__ __launch_bounds__(256) void k_gather_soa(CloudSoA src, const uint32_t *__restrict__ order, uint32_t n, CloudSoA dst)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t j = order[i];
    dst.x[i] = src.x[j]; dst.y[i] = src.y[j]; dst.z[i] = src.z[j];
    dst.nx[i] = src.nx[j]; dst.ny[i] = src.ny[j]; dst.nz[i] = src.nz[j];
}

// current source share back to the caller's row order, packed AoS
__global__ __launch_bounds__(256) void k_unpermute(CloudSoA cur, const uint32_t *__restrict__ order, uint32_t n,
                                                   float *xyz, float *nrm)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = order ? order[i] : i;
    if (xyz) { xyz[3 * (size_t)r] = cur.x[i]; xyz[3 * (size_t)r + 1] = cur.y[i]; xyz[3 * (size_t)r + 2] = cur.z[i]; }
    if (nrm) { nrm[3 * (size_t)r] = cur.nx[i]; nrm[3 * (size_t)r + 1] = cur.ny[i]; nrm[3 * (size_t)r + 2] = cur.nz[i]; }
}

// correspondences in original numbering.  mode 0: identity, 1: best64 (row in low word), 2: sorted position -> tq[pos].w
__global__ __launch_bounds__(256) void k_corr_out(const int32_t *__restrict__ pos, const unsigned long long *__restrict__ best64,
                                                  const float *__restrict__ d2, const float4 *__restrict__ tq,
                                                  const uint32_t *__restrict__ src_order, uint32_t n, int mode,
                                                  uint32_t tgt_offset, int32_t *idx_out, float *d2_out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = src_order ? src_order[i] : i;
    int32_t row = -1;
    if (mode == 0) row = (int32_t)(tgt_offset + i);
    else if (mode == 1) { unsigned long long b = best64[i]; row = (b == ~0ull) ? -1 : (int32_t)(uint32_t)(b & 0xFFFFFFFFull); }
    else { int32_t p = pos[i]; row = (p < 0) ? -1 : __float_as_int(tq[p].w); }
    if (idx_out) idx_out[r] = row;
    if (d2_out) d2_out[r] = d2 ? d2[i] : 0.0f;
}

// ---------------------------------------------------------------------------
// grid level statistics + dense cell table
// ---------------------------------------------------------------------------
// hist[l] (l = 1..10) = number of adjacent sorted pairs whose keys first differ at octree level l
__global__ __launch_bounds__(256) void k_level_hist(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *hist16)
{
    __shared__ uint32_t h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += gridDim.x * blockDim.x) {
        uint32_t x = keys[i] ^ keys[i - 1];
        if (x) {
            int b = 31 - __clz((int)x);          // highest differing bit, 0..29
            int lvl = kMortonBits - b / 3;       // 1..10
            atomicAdd(&h[lvl], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(hist16 + threadIdx.x, h[threadIdx.x]);
}

// nid_top[i] = exclusive count of super-cell (level glevel-3) starts before point i
__global__ __launch_bounds__(256) void k_cell_table(const uint32_t *__restrict__ keys, uint32_t n, int shift, int top_shift,
                                                    const uint32_t *__restrict__ nid_top, uint32_t *ctop, uint2 *cells)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = keys[i] >> shift;
    const bool top_start = i == 0 || (top_shift < 30 && (keys[i - 1] >> top_shift) != (keys[i] >> top_shift));
    const uint32_t block = nid_top[i] - (top_start ? 0u : 1u);
    if (top_start) ctop[top_shift < 30 ? (keys[i] >> top_shift) : 0u] = block;
    uint2 *slot = cells + (size_t)block * 512u + (c & 511u);
    if (i == 0 || (keys[i - 1] >> shift) != c) slot->x = i;
    if (i == n - 1 || (keys[i + 1] >> shift) != c) slot->y = i + 1;
}

// ---------------------------------------------------------------------------
// implicit 8-ary tree of tight boxes over the Morton order
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_leaf_boxes(const float4 *__restrict__ tq, uint32_t n, float4 *boxes, uint32_t nleaf_padded)
{
    uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nleaf_padded) return;
    const float inf = __int_as_float(0x7f800000);
    float4 lo = make_float4(inf, inf, inf, 0.f), hi = make_float4(-inf, -inf, -inf, 0.f);
    uint32_t j0 = l * kLeaf;
    for (uint32_t j = j0; j < j0 + kLeaf && j < n; j++) {
        float4 q = tq[j];
        lo.x = fminf(lo.x, q.x); lo.y = fminf(lo.y, q.y); lo.z = fminf(lo.z, q.z);
        hi.x = fmaxf(hi.x, q.x); hi.y = fmaxf(hi.y, q.y); hi.z = fmaxf(hi.z, q.z);
    }
    boxes[2 * (size_t)l] = lo; boxes[2 * (size_t)l + 1] = hi;
}

__global__ __launch_bounds__(256) void k_node_boxes(const float4 *__restrict__ child, uint32_t nchild_padded, float4 *parent,
                                                    uint32_t nparent_padded)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nparent_padded) return;
    const float inf = __int_as_float(0x7f800000);
    float4 lo = make_float4(inf, inf, inf, 0.f), hi = make_float4(-inf, -inf, -inf, 0.f);
    for (uint32_t c = p * kFan; c < p * kFan + kFan && c < nchild_padded; c++) {
        float4 a = child[2 * (size_t)c], b = child[2 * (size_t)c + 1];
        lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
        hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
    }
    parent[2 * (size_t)p] = lo; parent[2 * (size_t)p + 1] = hi;
}

// ---------------------------------------------------------------------------
// sparse octree over the sorted keys.  Level l groups points by the top 3*l key bits.
// ---------------------------------------------------------------------------

__device__ __forceinline__ bool oct_is_start(const uint32_t *__restrict__ keys, uint32_t i, int shift)
{
    // shift == 30 (level 0) puts every point in the root
    return i == 0 || (shift < 30 && (keys[i] >> shift) != (keys[i - 1] >> shift));
}

__global__ __launch_bounds__(256) void k_oct_flags(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t *nid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nid[i] = oct_is_start(keys, i, shift) ? 1u : 0u;
}

// after the exclusive scan nid[i] = number of node starts before i = id of the node that starts at i
__global__ __launch_bounds__(256) void k_oct_first(const uint32_t *__restrict__ keys, uint32_t n, int shift,
                                                   const uint32_t *__restrict__ nid, uint32_t *first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && oct_is_start(keys, i, shift)) first[nid[i]] = i;
}

__global__ __launch_bounds__(256) void k_oct_nodes(int level, const float4 *__restrict__ tq, uint32_t n,
                                                   const uint32_t *__restrict__ first, uint32_t n_nodes,
                                                   const uint32_t *__restrict__ nid_next, uint32_t n_nodes_next,
                                                   const float4 *__restrict__ nodes_next, float4 *nodes, uint32_t kOctLeafMax)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_nodes) return;
    const float inf = __int_as_float(0x7f800000);
    const uint32_t p0 = first[id], p1 = (id + 1 < n_nodes) ? first[id + 1] : n;
    const uint32_t npts = p1 - p0;
    float4 lo = make_float4(inf, inf, inf, 0.f), hi = make_float4(-inf, -inf, -inf, 0.f);
    uint32_t packed;
    if (level == kMortonBits || npts <= kOctLeafMax) {
        for (uint32_t j = p0; j < p1; j++) {
            const float4 q = tq[j];
            lo.x = fminf(lo.x, q.x); lo.y = fminf(lo.y, q.y); lo.z = fminf(lo.z, q.z);
            hi.x = fmaxf(hi.x, q.x); hi.y = fmaxf(hi.y, q.y); hi.z = fmaxf(hi.z, q.z);
        }
        packed = npts & kOctCfMask;                 // leaf: nchild = 0
    } else {
        const uint32_t c0 = nid_next[p0];
        const uint32_t c1 = (id + 1 < n_nodes) ? nid_next[p1] : n_nodes_next;
        for (uint32_t c = c0; c < c1; c++) {
            const float4 a = nodes_next[2 * (size_t)c], b = nodes_next[2 * (size_t)c + 1];
            lo.x = fminf(lo.x, a.x); lo.y = fminf(lo.y, a.y); lo.z = fminf(lo.z, a.z);
            hi.x = fmaxf(hi.x, b.x); hi.y = fmaxf(hi.y, b.y); hi.z = fmaxf(hi.z, b.z);
        }
        packed = (c0 & kOctCfMask) | ((c1 - c0) << 28);
    }
    lo.w = __int_as_float((int)p0);
    hi.w = __int_as_float((int)packed);
    nodes[2 * (size_t)id] = lo;
    nodes[2 * (size_t)id + 1] = hi;
}

void launch_oct_flags(const uint32_t *keys, uint32_t n, int level, uint32_t *nid, hipStream_t s)
{
    hipLaunchKernelGGL(k_oct_flags, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, 3 * (kMortonBits - level), nid);
}

// multi-block exclusive scan: per-block scan of 2048 elements + block totals, scan of the totals, add-back
constexpr int kScanPer = 8;
constexpr int kScanTile = 256 * kScanPer;

__global__ __launch_bounds__(256) void k_scan_tiles(uint32_t *data, uint32_t n, uint32_t *tile_sums)
{
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    uint32_t v[kScanPer], t = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) { v[k] = (base + k < n) ? data[base + k] : 0u; t += v[k]; }
    // inclusive scan of the per-thread totals across the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < wave) wbase += wsum[w];
    uint32_t run = wbase + incl - t;
#pragma unroll
    for (int k = 0; k < kScanPer; k++) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = wbase + incl;
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *data, uint32_t n, const uint32_t *__restrict__ tile_offsets)
{
    const uint32_t off = tile_offsets[blockIdx.x];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
#pragma unroll
    for (int k = 0; k < kScanPer; k++)
        if (base + k < n) data[base + k] += off;
}

// tile_ws: at least ceil(n / 2048) words of scratch
void launch_exclusive_scan(uint32_t *data, uint32_t n, uint32_t *tile_ws, hipStream_t s)
{
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, s, data, n, tile_ws);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, s, tile_ws, tiles);      // exclusive scan of the tile totals
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(256), 0, s, data, n, (const uint32_t *)tile_ws);
}

void launch_oct_first(const uint32_t *keys, uint32_t n, int level, const uint32_t *nid, uint32_t *first, hipStream_t s)
{
    hipLaunchKernelGGL(k_oct_first, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, 3 * (kMortonBits - level), nid, first);
}

void launch_oct_nodes(int level, const float4 *tq, uint32_t n, const uint32_t *first, uint32_t n_nodes, const uint32_t *nid_next,
                      uint32_t n_nodes_next, const float4 *nodes_next, float4 *nodes, uint32_t leaf_max, hipStream_t s)
{
    // leaf_max: a cell with at most this many points is a leaf
    hipLaunchKernelGGL(k_oct_nodes, dim3((n_nodes + 255) / 256), dim3(256), 0, s, level, tq, n, first, n_nodes, nid_next, n_nodes_next,
                       nodes_next, nodes, leaf_max);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline uint32_t nblk(uint32_t n, uint32_t t) { return (n + t - 1) / t; }

void launch_bbox(const float *x, const float *y, const float *z, uint32_t n, uint32_t *bbox_ord6, hipStream_t s)
{
    // lo slots start at 0xFFFFFFFF, hi slots at 0
    hipMemsetAsync(bbox_ord6, 0xFF, 3 * sizeof(uint32_t), s);
    hipMemsetAsync(bbox_ord6 + 3, 0x00, 3 * sizeof(uint32_t), s);
    uint32_t blocks = nblk(n, 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, s, x, y, z, n, bbox_ord6);
}

void launch_morton(const float *x, const float *y, const float *z, uint32_t n, float ox, float oy, float oz,
                   float inv_h0, uint32_t *keys, uint32_t *vals, hipStream_t s)
{
    hipLaunchKernelGGL(k_morton, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, z, n, ox, oy, oz, inv_h0, keys, vals);
}

void launch_gather_f4(const float *x, const float *y, const float *z, const float *nx, const float *ny, const float *nz,
                      const uint32_t *order, uint32_t n, float4 *tq, float4 *tn, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_f4, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, z, nx, ny, nz, order, n, tq, tn);
}

void launch_iota_f4(const float *x, const float *y, const float *z, const float *nx, const float *ny, const float *nz,
                    uint32_t n, float4 *tq, float4 *tn, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_f4, dim3(nblk(n, 256)), dim3(256), 0, s, x, y, z, nx, ny, nz, (const uint32_t *)nullptr, n, tq, tn);
}

__global__ __launch_bounds__(256) void k_offset_u32(const uint32_t *__restrict__ in, uint32_t off, uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + off;
}

void launch_offset_u32(const uint32_t *in, uint32_t off, uint32_t n, uint32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_offset_u32, dim3(nblk(n, 256)), dim3(256), 0, s, in, off, n, out);
}

void launch_gather_soa(const CloudSoA &src, const uint32_t *order, uint32_t n, CloudSoA dst, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_soa, dim3(nblk(n, 256)), dim3(256), 0, s, src, order, n, dst);
}

void launch_unpermute(const CloudSoA &cur, const uint32_t *order, uint32_t n, float *xyz_aos, float *nrm_aos, hipStream_t s)
{
    hipLaunchKernelGGL(k_unpermute, dim3(nblk(n, 256)), dim3(256), 0, s, cur, order, n, xyz_aos, nrm_aos);
}

void launch_corr_out(const int32_t *pos, const unsigned long long *best64, const float *d2, const float4 *tq,
                     const uint32_t *src_order, uint32_t n, int mode, uint32_t tgt_offset, int32_t *idx_out, float *d2_out,
                     hipStream_t s)
{
    hipLaunchKernelGGL(k_corr_out, dim3(nblk(n, 256)), dim3(256), 0, s, pos, best64, d2, tq, src_order, n, mode, tgt_offset,
                       idx_out, d2_out);
}

void launch_level_hist(const uint32_t *keys, uint32_t n, uint32_t *hist16, hipStream_t s)
{
    hipMemsetAsync(hist16, 0, 16 * sizeof(uint32_t), s);
    uint32_t blocks = nblk(n, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_level_hist, dim3(blocks), dim3(256), 0, s, keys, n, hist16);
}

void launch_cell_table(const uint32_t *keys, uint32_t n, int glevel, const uint32_t *nid_top, uint32_t *ctop, uint2 *cells, hipStream_t s)
{
    const int shift = 3 * (kMortonBits - glevel);
    const int ltop = glevel > 3 ? glevel - 3 : 0;
    hipLaunchKernelGGL(k_cell_table, dim3(nblk(n, 256)), dim3(256), 0, s, keys, n, shift, 3 * (kMortonBits - ltop), nid_top, ctop, cells);
}

void launch_leaf_boxes(const float4 *tq, uint32_t n, float4 *boxes, uint32_t nleaf_padded, hipStream_t s)
{
    hipLaunchKernelGGL(k_leaf_boxes, dim3(nblk(nleaf_padded, 256)), dim3(256), 0, s, tq, n, boxes, nleaf_padded);
}

void launch_node_boxes(const float4 *child, uint32_t nchild_padded, float4 *parent, uint32_t nparent_padded, hipStream_t s)
{
    hipLaunchKernelGGL(k_node_boxes, dim3(nblk(nparent_padded, 256)), dim3(256), 0, s, child, nchild_padded, parent, nparent_padded);
}

}  // namespace symmicp
