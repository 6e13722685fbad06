// device_common.h -- small device helpers shared by the pass kernels (kernels_pass.hip, kernels_packet.hip).
//
// fp32 expressions here must stay UNFUSED (the including file is compiled with -ffp-contract=off and carries
// `#pragma clang fp contract(off)`) and keep the association written: the CPU oracle uses the same expressions, so
// nearest-neighbour choices compare bit for bit.
#pragma once
#include "symmicp_internal.h"
#pragma clang fp contract(off)

namespace symmicp {

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

// A shard's capacity covers every producer -> shard mapping the kernels use (shard_capacity); an append beyond it would run
// into the next shard, so it is dropped and counted instead (word 2 of the list's counter block): the final reduce reports the
// count and the host fails the pass rather than return pairs that were never searched.
__device__ __forceinline__ void sl_push(const ShardList &L, uint32_t shard, uint32_t v)
{
    const uint32_t k = atomicAdd(L.counts + shard * kShardStride, 1u);
    if (k < L.cap) L.items[(size_t)shard * L.cap + k] = v;
    else atomicAdd(L.counts + 2, 1u);
}

// A list is consumed as the concatenation of its shards (balances consumers when shards are unevenly filled).
// sl_prefix: every thread of the block calls it once; pre[0..64] = exclusive prefix sums of the shard counts.
__device__ __forceinline__ void sl_prefix(const ShardList &L, uint32_t *pre /* __shared__ [kShards + 1] */)
{
    if (threadIdx.x < kShards) {
        uint32_t v = min(L.counts[threadIdx.x * kShardStride], L.cap);
#pragma unroll
        for (int off = 1; off < kShards; off <<= 1) {
            const uint32_t u = __shfl_up(v, off, 64);
            if ((int)threadIdx.x >= off) v += u;
        }
        pre[threadIdx.x + 1] = v;
        if (threadIdx.x == 0) pre[0] = 0;
    }
    __syncthreads();
}

// entry number g of the concatenated list; false when g is past the end
__device__ __forceinline__ bool sl_locate(const ShardList &L, const uint32_t *pre, uint32_t g, uint32_t &item)
{
    if (g >= pre[kShards]) return false;
    int lo = 0;
#pragma unroll
    for (int step = kShards / 2; step > 0; step >>= 1)
        if (pre[lo + step] <= g) lo += step;
    item = L.items[(size_t)lo * L.cap + (g - pre[lo])];
    return true;
}

__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nb_padded)
{
    return (b & 7u) * (nb_padded >> 3) + (b >> 3);
}

// The same dealing in chunks: XCD x works on chunks x, x + 8, x + 16, ... of `chunk` consecutive units each.  One contiguous eighth
// per XCD is the best locality but leaves the XCDs as unbalanced as the regions of the cloud are (first pass of the 1M surface
// pair: 313 .. 657 ms of packet time per eighth, the launch ends with one XCD working); chunks keep a compact region per XCD at
// any moment and even the work out.  The grid has to be a multiple of 8 * chunk; units beyond the last one are skipped by the caller.
__device__ __forceinline__ uint32_t xcd_remap_chunked(uint32_t b, uint32_t chunk)
{
    const uint32_t x = b & 7u, j = b >> 3;
    return ((j / chunk) * 8u + x) * chunk + (j % chunk);
}

__device__ __forceinline__ void store_pair_record(const PassArgs &a, const TargetIndex &ix, uint32_t i, int32_t pos)
{
    a.pairrec[2 * (size_t)i] = ix.tn[2 * (size_t)pos];
    a.pairrec[2 * (size_t)i + 1] = ix.tn[2 * (size_t)pos + 1];
}


}  // namespace symmicp
