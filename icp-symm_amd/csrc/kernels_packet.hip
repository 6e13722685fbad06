// kernels_packet.hip -- exact nearest neighbours for 64 queries at a time (gfx950).
//
// Fills the correspondence step the reference leaves as a todo (ICP/myicp.cpp:128-131) for the passes in which most
// queries have to be searched (the first pass of an alignment, the passes right after a large motion).
//
// One wave = one PACKET of 64 consecutive queries of the Morton-sorted source, i.e. a compact patch of the cloud.  A
// per-thread tree walk pays for every node visit with 64 lanes' worth of divergent 16-byte loads and a few hundred vector
// instructions, and a wave is as slow as its slowest lane; a depth-first traversal shared by the wave (the first form of
// this kernel, kept below as the overflow fallback) cuts the instructions but is one chain of ~300 dependent LDS and
// memory round trips per packet: 1.1 ms for the first pass of the 1M-point surface pair against 1.7 ms for the
// per-thread walk, latency-bound (occupancy 32 -> 16 -> 8 waves per CU: 1.08 -> 1.28 -> 1.60 ms) with a tail of packets
// whose queries sit in several clusters (up to 2300 pops against a mean of 215).  So the traversal is BREADTH-first:
//   dive      root to leaf along the children nearest to a query in the middle of the packet (no siblings kept); every
//             lane tests that leaf's points: all 64 queries now have a finite bound (~11 dependent loads)
//   levels    the frontier of a level sits in LDS; 8 frontier nodes x 8 child slots are handled per step, ONE LANE PER
//             CHILD: the lane loads the child's 32-byte record (all loads of a step are independent) and tests its box
//             against the bounding boxes and loosest bounds of the packet's four 16-query sub-groups (wave-uniform SGPR
//             operands).  Surviving internal nodes are compacted into the next frontier (ballot + mbcnt)
//   leaves    surviving leaves are taken one at a time: the box is broadcast with v_readlane (SGPR operands), every lane tests it
//             against its own bound (no lane needs it -> skip), then the leaf's points arrive through the SCALAR cache (s_load,
//             8 points per round trip) and are tested as SGPR operands: ~12 vector instructions per point for all 64 queries;
//             bounds tighten as leaves are scanned
// A packet is ~25 dependent round trips and ~12 k vector instructions instead of ~300 and ~21 k.
//
// Exactness.  A box is skipped only when boxdist2 (the same monotone fp32 expression as dist2, so boxdist2 <= dist2 to
// every point inside) exceeds the lane's bound for every lane; the sub-group test is the same expression on interval
// gaps (a lower bound of boxdist2 for every lane of the sub-group).  The running best of a lane is the lexicographic
// minimum of (d2, original row) over the points seen so far, as in brute force: the fast path keeps the first point
// that reaches a strictly smaller d2 and flags equal distances; a flagged group of points is scanned again with the row
// comparison (rare).
// CERT: scan everything within (nearest + pad) and keep the second-nearest distance, so that the result carries a pair
// certificate (see k_search_cells in kernels_pass.hip).
#include <cstdlib>
#include "symmicp_internal.h"
#include "device_common.h"
#pragma clang fp contract(off)

namespace symmicp {

constexpr int kPktStack = 128;            // >= 7 * octree levels + 1
static_assert(kPktStack >= 7 * (kMortonBits + 1) + 1, "packet stack too small for the octree depth");

typedef float f32x16 __attribute__((ext_vector_type(16)));

// two 64-byte scalar loads from a wave-uniform address (hipcc does not count asm loads: the wait is part of the statement)
__device__ __forceinline__ void sload_2x16(const void *p, f32x16 &a, f32x16 &b)
{
    // (provably uniform for the compiler: inside a non-inlined function the pointer arrives in vector registers)
    const unsigned long long u = (unsigned long long)p;
    p = (const void *)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}

// ---- cross-lane reductions on the DPP network -------------------------------------------------------------
// unsigned max / min: used on the bit patterns of non-negative floats (same order), identity as `old` so the
// compiler can fold the move into the ALU instruction
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umax(uint32_t x)
{
    return max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_umin(uint32_t x)
{
    return min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_fmin(float x)
{
    return fminf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false)));
}
template <int CTRL>
__device__ __forceinline__ float dpp_fmax(float x)
{
    return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false)));
}
// reductions over the 16 lanes of a row; the result sits in the row's last lane (15, 31, 47, 63)
__device__ __forceinline__ float row_fmin(float x)
{
    x = dpp_fmin<0x111>(x); x = dpp_fmin<0x112>(x); x = dpp_fmin<0x114>(x); x = dpp_fmin<0x118>(x);
    return x;
}
__device__ __forceinline__ float row_fmax(float x)
{
    x = dpp_fmax<0x111>(x); x = dpp_fmax<0x112>(x); x = dpp_fmax<0x114>(x); x = dpp_fmax<0x118>(x);
    return x;
}
__device__ __forceinline__ uint32_t row_umax(uint32_t x)
{
    x = dpp_umax<0x111, 0xf>(x); x = dpp_umax<0x112, 0xf>(x); x = dpp_umax<0x114, 0xf>(x); x = dpp_umax<0x118, 0xf>(x);
    return x;
}

__device__ __forceinline__ float pkt_threshold(float d2, float pad)
{
    if (pad <= 0.0f) return d2;
    const float rp = __builtin_amdgcn_sqrtf(d2) * 1.00001f + pad;
    return rp * rp * 1.00001f;
}

struct PktBest {
    float d2;           // nearest so far
    uint32_t second;    // CERT: bits of the second-nearest d2 seen
    int32_t pos;        // sorted position of the nearest
};

// fast path for one point: strict improvement only; equal distances are flagged in `tie`
template <bool CERT, int K>
__device__ __forceinline__ void pkt_point(PktBest &b, int32_t &slot, unsigned long long &tie, float px, float py, float pz, float qx, float qy, float qz)
{
    const float d2 = dist2(px, py, pz, qx, qy, qz);
    const bool lt = d2 < b.d2;
    tie |= __ballot(d2 == b.d2);
    if (CERT) b.second = min(b.second, max(__float_as_uint(d2), __float_as_uint(b.d2)));      // the loser of every comparison
    b.d2 = lt ? d2 : b.d2;
    slot = lt ? K : slot;
}

// ---- scanning one leaf ------------------------------------------------------------------------------------------
// Tie rule shared by both forms: after the fast path b.d2 is the minimum over everything seen; if some lane met a point
// exactly as far as its best at that moment, the lowest original row among the points AT that distance has to win, so
// the rows of this group's points at that distance are compared with the row of the current holder.

// (first, cnt) wave-uniform; the points come through the scalar cache, 8 at a time (tq is padded by 8 entries)
template <bool CERT>
__device__ __forceinline__ void pkt_leaf_scalar(const float4 *__restrict__ tq, uint32_t first, uint32_t cnt, float px, float py, float pz,
                                                PktBest &b, unsigned long long &c_tie)
{
    for (uint32_t e0 = 0; e0 < cnt; e0 += 8) {
        const uint32_t m = min(cnt - e0, 8u);
        f32x16 qa, qb;
        sload_2x16(tq + first + e0, qa, qb);
        int32_t slot = 8;
        unsigned long long tie = 0;
        pkt_point<CERT, 0>(b, slot, tie, px, py, pz, qa[0], qa[1], qa[2]);
        if (m > 1u) pkt_point<CERT, 1>(b, slot, tie, px, py, pz, qa[4], qa[5], qa[6]);
        if (m > 2u) pkt_point<CERT, 2>(b, slot, tie, px, py, pz, qa[8], qa[9], qa[10]);
        if (m > 3u) pkt_point<CERT, 3>(b, slot, tie, px, py, pz, qa[12], qa[13], qa[14]);
        if (m > 4u) pkt_point<CERT, 4>(b, slot, tie, px, py, pz, qb[0], qb[1], qb[2]);
        if (m > 5u) pkt_point<CERT, 5>(b, slot, tie, px, py, pz, qb[4], qb[5], qb[6]);
        if (m > 6u) pkt_point<CERT, 6>(b, slot, tie, px, py, pz, qb[8], qb[9], qb[10]);
        if (m > 7u) pkt_point<CERT, 7>(b, slot, tie, px, py, pz, qb[12], qb[13], qb[14]);
        if (slot != 8) b.pos = (int32_t)(first + e0) + slot;
        if (tie != 0ull) {
            c_tie++;
            int32_t brow = (b.pos >= 0) ? __float_as_int(tq[b.pos].w) : 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if ((uint32_t)k < m) {
                    const float qx = k < 4 ? qa[4 * k] : qb[4 * (k - 4)], qy = k < 4 ? qa[4 * k + 1] : qb[4 * (k - 4) + 1],
                                qz = k < 4 ? qa[4 * k + 2] : qb[4 * (k - 4) + 2];
                    const int32_t row = __float_as_int(k < 4 ? qa[4 * k + 3] : qb[4 * (k - 4) + 3]);
                    const float d2 = dist2(px, py, pz, qx, qy, qz);
                    if (d2 == b.d2 && row < brow) { brow = row; b.pos = (int32_t)(first + e0) + k; }
                }
            }
        }
    }
}

__device__ __forceinline__ float bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// ---- depth-first traversal shared by the wave (fallback when a frontier outgrows its LDS slot) -------------------------
// pop a node (its 32-byte record travels on the stack); every lane tests the node's box against its own bound; no lane
// wants it -> next pop; leaf -> scan; internal -> lane 4c+g tests child c against sub-group g's box and bound, the
// children some sub-group may need are pushed, the nearest on top.  Exact for any starting bounds.
struct PktCounters {
    unsigned long long nodes, leaves, rejected, points, ties, steps, overflow, wantlanes;
};

template <bool CERT>
__device__ __noinline__ void pkt_dfs(const float4 *__restrict__ onodes, const float4 *__restrict__ tq, const uint32_t *s_off, float4 (*stk)[2],
                                     const float (*s_sb)[8], uint8_t *s_lvl, const int lane, float px, float py, float pz, PktBest &b, float &thr, float pad, PktCounters &cn)
{
    const int sub = lane & 3, chl = lane >> 2;                          // lane 4c+g = (child c, sub-group g)
    const int bperm = (16 * sub + 15) * 4;                              // ds_bpermute address of sub-group g's last lane
    const float4 sblo = *reinterpret_cast<const float4 *>(&s_sb[sub][0]), sbhi = *reinterpret_cast<const float4 *>(&s_sb[sub][4]);
    if (lane == 0) {
        const float4 rA = onodes[0], rB = onodes[1];
        stk[0][0] = rA; stk[0][1] = rB;
        s_lvl[0] = 0;                                                   // the level of every stacked node rides beside it
    }
    int sp = 1;
    __builtin_amdgcn_wave_barrier();
    while (sp > 0) {
        sp--;
        const float4 nA = stk[sp][0], nB = stk[sp][1];
        const bool want = boxdist2(px, py, pz, nA, nB) <= thr;
        if (__ballot(want) == 0ull) { cn.rejected++; continue; }
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(nA.w));
        const uint32_t packed = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(nB.w));
        const uint32_t level = s_lvl[sp], nch = oct_nch(packed), cf = oct_cf(packed);
        if (nch == 0) {
            cn.leaves++; cn.points += cf;
            pkt_leaf_scalar<CERT>(tq, first, cf, px, py, pz, b, cn.ties);
            thr = fminf(thr, pkt_threshold(b.d2, pad));
        } else {
            cn.nodes++;
            const uint32_t tb = (thr >= 0.0f) ? __float_as_uint(thr) + 1u : 0u;      // 0: the lane wants nothing; else bits + 1 (a bound of 0 still admits gap 0)
            const uint32_t thr_g = (uint32_t)__builtin_amdgcn_ds_bpermute(bperm, (int)row_umax(tb));
            uint32_t bbd = 0x7f800000u;
            float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
            const bool mine = (lane < 32) && ((uint32_t)chl < nch);
            if (mine) {
                const float4 *__restrict__ ch = onodes + 2 * ((size_t)s_off[level + 1] + cf + (uint32_t)chl);
                cA = ch[0]; cB = ch[1];
                const float gx = fmaxf(fmaxf(cA.x - sbhi.x, sblo.x - cB.x), 0.0f);
                const float gy = fmaxf(fmaxf(cA.y - sbhi.y, sblo.y - cB.y), 0.0f);
                const float gz = fmaxf(fmaxf(cA.z - sbhi.z, sblo.z - cB.z), 0.0f);
                bbd = __float_as_uint((gx * gx + gy * gy) + gz * gz);
            }
            const bool keep = mine && bbd < thr_g;
            const uint32_t m32 = (uint32_t)__ballot(keep);
            if (m32) {
                uint32_t mc = m32 | (m32 >> 1);
                mc = (mc | (mc >> 2)) & 0x11111111u;                     // bit 4c: child c is needed by some sub-group
                uint32_t key = keep ? ((bbd & ~7u) | (uint32_t)chl) : 0xFFFFFFFFu;
                key = dpp_umin<0xB1, 0xf>(key);                          // quad_perm [1,0,3,2]
                key = dpp_umin<0x4E, 0xf>(key);                          // quad_perm [2,3,0,1]: every lane of the quad holds the child's minimum
                key = dpp_umin<0x114, 0xf>(key);                         // row_shr:4
                key = dpp_umin<0x118, 0xf>(key);                         // row_shr:8: lanes 15 / 31 hold children 0..3 / 4..7
                const uint32_t kmin = min((uint32_t)__builtin_amdgcn_readlane((int)key, 15), (uint32_t)__builtin_amdgcn_readlane((int)key, 31));
                const uint32_t cmin = kmin & 7u;
                const uint32_t nk = (uint32_t)__popc(mc);
                const uint32_t rank = (uint32_t)__popc(mc & ((1u << (4 * (chl & 7))) - 1u));
                const uint32_t rmin = (uint32_t)__popc(mc & ((1u << (4 * cmin)) - 1u));
                if (mine && sub == 0 && ((mc >> (4 * (chl & 7))) & 1u)) {
                    const uint32_t slot = ((uint32_t)chl == cmin) ? nk - 1u : (rank > rmin ? rank - 1u : rank);
                    stk[sp + slot][0] = cA; stk[sp + slot][1] = cB;
                    s_lvl[sp + slot] = (uint8_t)(level + 1u);
                }
                sp += (int)nk;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// One step of the breadth-first sweep: 8 frontier nodes (fr_cur[f0 .. f0 + 8)) x 8 child slots, one lane per child.  The child's box is tested
// against the four sub-group boxes and loosest bounds; surviving internal nodes are appended to fr_next (the counter is an LDS word the
// waves of a packet share); surviving leaves are scanned one after the other against every lane's own bound.  Returns whether a leaf
// was scanned (the lane's b / thr may have improved).  ND leaves named in `dive` have been scanned already and are skipped.
template <int ND, bool DBG>
__device__ __forceinline__ bool pkt_step(const uint32_t *fr_cur, uint32_t *fr_next, uint32_t nf, uint32_t f0, uint32_t off_next, uint32_t *cnt_next, uint32_t front_cap, int lane,
                                         const float4 *__restrict__ onodes, const float4 *__restrict__ tq, float px, float py, float pz, bool wants,
                                         const float (&gl)[4][3], const float (&gh)[4][3], const uint32_t (&dive)[ND], PktBest &b, float &thr, PktCounters &cn)
{
    constexpr bool CERT = false;
    const float pad = 0.0f;
    uint32_t tg[4];
    {
        const uint32_t tb = wants ? __float_as_uint(thr) + 1u : 0u;   // 0: the lane wants nothing; else bits + 1 (a bound of 0 still admits gap 0)
        const uint32_t r = row_umax(tb);
#pragma unroll
        for (int g = 0; g < 4; g++) tg[g] = (uint32_t)__builtin_amdgcn_readlane((int)r, 16 * g + 15);
    }
    const uint32_t f = f0 + (uint32_t)(lane >> 3), c = (uint32_t)(lane & 7);
    const uint32_t pkd = (f < nf) ? fr_cur[f] : 0u;
    const bool valid = c < oct_nch(pkd);
    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
    bool keep = false;
    if (valid) {
        const float4 *__restrict__ ch = onodes + 2 * ((size_t)off_next + oct_cf(pkd) + c);
        cA = ch[0]; cB = ch[1];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            // gap between the child's box and the sub-group's box: a lower bound of boxdist2 for each of its lanes
            const float gx = fmaxf(fmaxf(cA.x - gh[g][0], gl[g][0] - cB.x), 0.0f);
            const float gy = fmaxf(fmaxf(cA.y - gh[g][1], gl[g][1] - cB.y), 0.0f);
            const float gz = fmaxf(fmaxf(cA.z - gh[g][2], gl[g][2] - cB.z), 0.0f);
            keep |= __float_as_uint((gx * gx + gy * gy) + gz * gz) < tg[g];
        }
    }
    const uint32_t cpk = (uint32_t)__float_as_int(cB.w);
    const bool isleaf = oct_nch(cpk) == 0u;
    const bool keepL = keep && isleaf, keepI = keep && !isleaf;
    if (DBG) cn.nodes += (unsigned long long)__popcll(__ballot(valid));
    // internal survivors -> next frontier
    const unsigned long long mI = __ballot(keepI);
    if (mI) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cnt_next, (uint32_t)__popcll(mI));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t dst = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mI >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mI, 0u));
        if (keepI && dst < front_cap) fr_next[dst] = cpk;
    }
    // leaf survivors: one leaf at a time
    unsigned long long mL = __ballot(keepL);
    bool scanned = false;
    if (mL) {
        while (mL) {
            const int src = (int)__ffsll((long long)mL) - 1;
            mL &= mL - 1ull;
            if (DBG) cn.leaves++;
            const uint32_t lfirst = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cA.w), src);
            bool dived = false;
#pragma unroll
            for (int w = 0; w < ND; w++) dived |= (lfirst == dive[w]);
            if (dived) continue;                                                 // scanned by a dive already
            float4 lo, hi;
            lo.x = bcast(cA.x, src); lo.y = bcast(cA.y, src); lo.z = bcast(cA.z, src);
            hi.x = bcast(cB.x, src); hi.y = bcast(cB.y, src); hi.z = bcast(cB.z, src);
            const unsigned long long mw = __ballot(boxdist2(px, py, pz, lo, hi) <= thr);
            if (mw == 0ull) { if (DBG) cn.rejected++; continue; }
            if (DBG) cn.wantlanes += (unsigned long long)__popcll(mw);
            const uint32_t cnt = oct_cf((uint32_t)__builtin_amdgcn_readlane((int)cpk, src));
            if (DBG) cn.points += cnt;
            // the leaf's points come through the scalar cache, 8 at a time (one dependent round trip per 8 points, hidden by the other waves of the
            // SIMD: the kernel is bound by vector-instruction issue).  Round 2 had the leaf's lane load them into 32 vector registers and
            // broadcast them with three v_readlane per point -- 4 cycles of vector issue each: 0.483 against 0.471 ms at 1M, 0.58 against 0.53 at 2M
            pkt_leaf_scalar<CERT>(tq, lfirst, cnt, px, py, pz, b, cn.ties);
            thr = fminf(thr, pkt_threshold(b.d2, pad));
            scanned = true;
        }
    }
    return scanned;
}

// The packets are runs of up to 64 consecutive queries of the Morton-sorted share (PassArgs::pkt_tab, or 64 as they lie): the first pass
// of an alignment (no previous pairs, every query is searched from scratch).  Later passes have bounds from their previous pairs and
// mostly certificates; packets were measured there too (previous pair as the bound, certificate test per lane) and lose to the cell
// scans of k_search_cells: 0.55 ms against 0.35 ms for pass 2 of the 1M-point surface pair, 0.56 against 0.17 for pass 3 -- with tight
// bounds a packet is ten nearly empty levels of round trips.
//
// W waves per packet (one workgroup of 64 W threads).  A launch lasts at least as long as its slowest packet, and a rank's share of a
// sharded run is a few thousand packets on 4096 wave slots: with one wave per packet the first pass did not divide by the number of
// GPUs at all (0.57 ms at 1 GPU, 0.53 ms per rank at 8).  So the waves of a workgroup SHARE one packet: every wave holds all 64 queries
// (one per lane) and its own running bests; the frontier of a level sits in LDS once, and the waves draw its steps (8 frontier nodes x 8
// children) from a counter in LDS, scan the leaves their own steps turn up, and exchange bounds through 64 words of LDS (atomic min of
// the d2 bits: a lane prunes with the best distance ANY wave has reached for its query).  At the end the per-wave bests are merged by
// an LDS 64-bit atomic min on (d2 bits << 32 | original row): the lexicographic minimum brute force takes, whatever wave saw the point.
// Every wave dives for a different query of the packet (quantiles of the wanting lanes), so the W dives leave every lane a bound from
// a leaf near it.  W = 1 is the same code with wave-local barriers.
constexpr int kFrontCap = 512;            // frontier nodes per level (two buffers of uint32 in LDS = the DFS stack's 4 KB); 256 / 384: 1.08 ms (more packets fall back to the depth-first walk), 1024: 0.90 ms (the widest packets sweep 1000-node levels: slower than falling back) against 0.66
static_assert(2 * kFrontCap * sizeof(uint32_t) <= kPktStack * 2 * sizeof(float4), "frontier buffers alias the DFS stack");
constexpr int kPktLevels = kMortonBits + 3;

template <int W>
__device__ __forceinline__ void pkt_sync()
{
    if (W == 1) __builtin_amdgcn_wave_barrier();      // (one wave: its LDS accesses are in order)
    else __syncthreads();
}

// waves per SIMD the kernels are compiled for (87 / 79 VGPRs; one wave per packet would also fit 6: 0.478 ms against 0.470 at 5 and 0.494 at 4)
#ifndef PKT_WAVES1
#define PKT_WAVES1 5
#endif
#ifndef PKT_WAVESN
#define PKT_WAVESN 5
#endif
template <int W, bool DBG>
__global__ __launch_bounds__(64 * W, W == 1 ? PKT_WAVES1 : PKT_WAVESN) void k_search_packet(PassArgs a, TargetIndex ix, uint32_t chunk)
{
    __shared__ float4 s_buf[kPktStack][2];         // BFS: two frontier buffers; DFS fallback: the stack
    __shared__ float s_sb[4][8];                   // bounding boxes of the four 16-query sub-groups
    __shared__ uint8_t s_lvl[kPktStack];           // DFS fallback: level of each stacked node
    __shared__ uint32_t s_off[kMortonBits + 2];
    __shared__ uint32_t s_cnt[kPktLevels];         // frontier size per level
    __shared__ uint32_t s_stp[kPktLevels];         // step dispenser per level (W > 1)
    __shared__ uint32_t s_thr[64];                 // per query: d2 bits of the best distance any wave has reached (W > 1)
    __shared__ unsigned long long s_key[64];       // per query: final merge (W > 1)
    __shared__ uint32_t s_dive[W];                 // first point of the leaf each wave's dive scanned
    uint32_t (*fr)[kFrontCap] = reinterpret_cast<uint32_t (*)[kFrontCap]>(&s_buf[0][0]);
    const int lane = threadIdx.x & 63;
    const int wave = (W == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float inf = __int_as_float(0x7f800000);
    if (threadIdx.x < kMortonBits + 2) s_off[threadIdx.x] = ix.olevel_off[threadIdx.x];
    __syncthreads();
    const uint32_t npk = a.pkt_tab ? a.pkt_count : (a.n + 63u) >> 6;
    const uint32_t front_cap = min(a.pkt_front_cap ? a.pkt_front_cap : (uint32_t)kFrontCap, (uint32_t)kFrontCap);
    constexpr bool CERT = false;              // the first pass leaves no certificates (the cloud is about to move by its whole misalignment)
    const float pad = 0.0f;
    const float4 *__restrict__ tq = ix.tq;
    const float4 *__restrict__ onodes = ix.onodes;
    PktCounters cn = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t slot = xcd_remap_chunked(blockIdx.x, chunk); slot < npk; slot += gridDim.x) {
        // a packet: `count` consecutive queries from `first` (the table: widest first), or 64 as they lie
        uint32_t first = slot * 64u, count = 64u;
        if (a.pkt_tab) { const uint2 e = a.pkt_tab[slot]; first = e.x; count = e.y; }
        const uint32_t pk = slot;
        unsigned long long t_start = 0;
        if (DBG || ix.dbg_trace) t_start = __builtin_amdgcn_s_memrealtime();
        const PktCounters cn0 = cn;
        if (wave == 0) {
            if (lane < kPktLevels) { s_cnt[lane] = 0; s_stp[lane] = 0; }
            if (W > 1) { s_thr[lane] = 0x7f800000u; s_key[lane] = ~0ull; }
            if (lane < W) s_dive[lane] = 0xFFFFFFFFu;
        }
        pkt_sync<W>();
        const uint32_t i = first + (uint32_t)lane;
        const bool active = (uint32_t)lane < count && i < a.n;
        float px = 0.f, py = 0.f, pz = 0.f;
        PktBest b;
        b.d2 = inf; b.second = 0x7f800000u; b.pos = -1;
        float thr = -1.0f;                 // prune bound on d2; -1: this lane wants nothing
        if (active) {
            const float x = a.in.x[i], y = a.in.y[i], z = a.in.z[i];
            px = xf_row(a.X.m + 0, x, y, z, 1.0f); py = xf_row(a.X.m + 4, x, y, z, 1.0f); pz = xf_row(a.X.m + 8, x, y, z, 1.0f);
            thr = ((px == px) & (py == py) & (pz == pz)) ? inf : -1.0f;      // NaN query: no pair
        }
        const bool wants = thr >= 0.0f;     // (never changes: bounds only shrink towards 0)
        const unsigned long long wants0 = __ballot(wants);
        uint32_t steps_pk = 0;
        bool ovf_pk = false;
        if (wants0 != 0ull) {               // (the same in every wave of the workgroup: the barriers below are uniform)
            // ---- bounding boxes of the four sub-groups (lanes that want nothing do not stretch them): SGPRs, and LDS for the fallback
            float gl[4][3], gh[4][3];
            {
                const float lx = row_fmin(wants ? px : inf), ly = row_fmin(wants ? py : inf), lz = row_fmin(wants ? pz : inf);
                const float hx = row_fmax(wants ? px : -inf), hy = row_fmax(wants ? py : -inf), hz = row_fmax(wants ? pz : -inf);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    gl[g][0] = bcast(lx, 16 * g + 15); gl[g][1] = bcast(ly, 16 * g + 15); gl[g][2] = bcast(lz, 16 * g + 15);
                    gh[g][0] = bcast(hx, 16 * g + 15); gh[g][1] = bcast(hy, 16 * g + 15); gh[g][2] = bcast(hz, 16 * g + 15);
                }
                if (wave == 0 && (lane & 15) == 15) {
                    float *sb = const_cast<float *>(s_sb[lane >> 4]);
                    sb[0] = lx; sb[1] = ly; sb[2] = lz; sb[3] = 0.f; sb[4] = hx; sb[5] = hy; sb[6] = hz; sb[7] = 0.f;
                }
            }
            const float4 rootA = onodes[0], rootB = onodes[1];                            // (same address in every lane)
            const uint32_t root_first = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(rootA.w));
            const uint32_t root_packed = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(rootB.w));
            bool done = false;
            // ---- dive: a finite bound for every lane before the frontier is built.  Wave w dives for the query at quantile
            // (2w + 1) / 2W of the wanting lanes, so that every lane gets a bound from a leaf near it
            {
                const int nw = __popcll(wants0);
                int kth = ((2 * wave + 1) * nw) / (2 * W);            // 0-based rank among the wanting lanes
                unsigned long long m = wants0;
                while (kth-- > 0) m &= m - 1ull;
                const int mid = (int)__ffsll((long long)m) - 1;
                const float cx = bcast(px, mid), cy = bcast(py, mid), cz = bcast(pz, mid);
                uint32_t packed = root_packed, dfirst = root_first, level = 0;
                while (oct_nch(packed) != 0u) {
                    const uint32_t nch = oct_nch(packed), cf = oct_cf(packed);
                    uint32_t key = 0xFFFFFFFFu;
                    float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
                    if ((uint32_t)lane < nch) {
                        const float4 *__restrict__ ch = onodes + 2 * ((size_t)s_off[level + 1] + cf + (uint32_t)lane);
                        cA = ch[0]; cB = ch[1];
                        key = (__float_as_uint(boxdist2(cx, cy, cz, cA, cB)) & ~7u) | (uint32_t)lane;
                    }
                    key = dpp_umin<0x111, 0xf>(key); key = dpp_umin<0x112, 0xf>(key); key = dpp_umin<0x114, 0xf>(key);      // lane 7: min of lanes 0..7
                    const int cmin = __builtin_amdgcn_readlane((int)key, 7) & 7;
                    packed = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cB.w), cmin);
                    dfirst = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cA.w), cmin);
                    level++;
                    if (DBG) cn.steps++;
                }
                const uint32_t cnt = oct_cf(packed);
                if (DBG) cn.points += cnt;
                pkt_leaf_scalar<CERT>(tq, dfirst, cnt, px, py, pz, b, cn.ties);
                thr = fminf(thr, pkt_threshold(b.d2, pad));
                done = (level == 0);               // the root itself is a leaf: everything has been scanned
                if (lane == 0) s_dive[wave] = dfirst;   // the sweep skips these leaves (a second scan would count their points as ties)
                if (W > 1 && wants) atomicMin(&s_thr[lane], __float_as_uint(thr));
            }
            // ---- breadth-first sweep
            if (!done && wave == 0 && lane == 0) { fr[0][0] = root_packed; s_cnt[0] = 1u; }
            pkt_sync<W>();
            uint32_t dive_w[W];
#pragma unroll
            for (int w = 0; w < W; w++) dive_w[w] = s_dive[w];
            int cur = 0;
            bool overflow = false;
            for (uint32_t level = 0; !done; level++) {
                const uint32_t nf = s_cnt[level];                       // (written before the barrier: the same in every wave)
                if (nf == 0u) break;
                if (nf > front_cap) { overflow = true; break; }
                const uint32_t off_next = s_off[level + 1];
                uint32_t my_step = 0;
                while (true) {
                    uint32_t st = my_step++;
                    if (W > 1) {
                        if (lane == 0) st = atomicAdd(&s_stp[level], 1u);
                        st = (uint32_t)__builtin_amdgcn_readfirstlane((int)st);
                    }
                    const uint32_t f0 = st * 8u;
                    if (f0 >= nf) break;
                    if (DBG) { cn.steps++; steps_pk++; }
                    if (W > 1 && wants) thr = __uint_as_float(min(__float_as_uint(thr), s_thr[lane]));      // what the other waves have reached
                    const bool scanned = pkt_step<W, DBG>(fr[cur], fr[cur ^ 1], nf, f0, off_next, &s_cnt[level + 1], front_cap, lane, onodes, tq, px, py, pz, wants, gl, gh, dive_w, b, thr, cn);
                    if (W > 1 && scanned && wants) thr = __uint_as_float(min(__float_as_uint(thr), atomicMin(&s_thr[lane], __float_as_uint(thr))));
                }
                cur ^= 1;
                pkt_sync<W>();
            }
            if (overflow) {
                // a frontier outgrew its LDS slot: wave 0 finishes depth-first from the root with the bounds reached so far (exact
                // for any starting bounds; the other waves' bests join in the merge below)
                ovf_pk = true;
                if (wave == 0) {
                    cn.overflow++;
                    if (lane == 0 && a.pkt_fallbacks) atomicAdd(a.pkt_fallbacks, 1u);
                    if (W > 1 && wants) thr = __uint_as_float(min(__float_as_uint(thr), s_thr[lane]));
                    // (copies: the call is not inlined, and state whose address escapes would live in scratch memory for the whole kernel --
                    // every bound update of the sweep above a scratch store; that was most of the 217 MB the round-2 kernel wrote per launch)
                    PktBest b2 = b;
                    float thr2 = thr;
                    PktCounters cn2 = {0, 0, 0, 0, 0, 0, 0, 0};
                    pkt_dfs<CERT>(onodes, tq, s_off, s_buf, s_sb, s_lvl, lane, px, py, pz, b2, thr2, pad, cn2);
                    b = b2; thr = thr2;
                    if (DBG) { cn.nodes += cn2.nodes; cn.leaves += cn2.leaves; cn.rejected += cn2.rejected; cn.points += cn2.points; cn.ties += cn2.ties; }
                }
            }
        }
        // ---- the answer: lexicographic minimum of (d2, original row) over the waves' bests
        bool writer = (wave == 0);
        if (W > 1) {
            unsigned long long key = ~0ull;
            if (active && b.pos >= 0) key = ((unsigned long long)__float_as_uint(b.d2) << 32) | (unsigned long long)(uint32_t)__float_as_int(tq[b.pos].w);
            if (key != ~0ull) atomicMin(&s_key[lane], key);
            __syncthreads();
            const unsigned long long kmin = s_key[lane];
            writer = (kmin == ~0ull) ? (wave == 0) : (key == kmin);      // (two waves holding the same point write the same values)
            if (kmin == ~0ull) b.pos = -1;
        }
        if (active && writer) {
            a.pos_out[i] = b.pos;
            a.d2_out[i] = b.d2;
            a.cert[i] = make_float4(px, py, pz, 0.0f);
            if (b.pos >= 0) store_pair_record(a, ix, i, b.pos);
            else a.pairrec[2 * (size_t)i + 1] = make_float4(0.f, 0.f, 0.f, 1.f);
        }
        if (!DBG && ix.dbg_trace && wave == 0 && lane == 0 && (size_t)pk < ((size_t)1 << 21)) {
            // timing-only trace of the production kernel (SYMMICP_DEBUG_TRACE without SYMMICP_DEBUG_COUNTERS): two stores per packet
            const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_start;
            ix.dbg_trace[2 * (size_t)pk] = t_start;
            ix.dbg_trace[2 * (size_t)pk + 1] = (dt << 32) | (unsigned long long)((first << 1) | (ovf_pk ? 1u : 0u));      // (first query, fallback flag)
        }
        if (DBG && lane == 0) {
            const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_start;      // 100 MHz ticks
            if (wave == 0) { atomicMax(ix.dbg + 6, dt); atomicAdd(ix.dbg + 7, dt); }
            if (ix.dbg_trace && (size_t)pk < ((size_t)1 << 18)) {
                // instrumented build: 8 words per packet -- start, (ticks, first query), then per wave (steps, leaves scanned, points, tie rescans) since the packet began
                unsigned long long *tr = ix.dbg_trace + 8 * (size_t)pk;
                if (wave == 0) { tr[0] = t_start; tr[1] = (dt << 32) | (unsigned long long)((first << 1) | (ovf_pk ? 1u : 0u)); }
                tr[2 + wave] = ((unsigned long long)(steps_pk & 0xFFFFu) << 48) | ((unsigned long long)((cn.leaves - cn.rejected - cn0.leaves + cn0.rejected) & 0xFFFFu) << 32) |
                               ((unsigned long long)((cn.points - cn0.points) & 0xFFFFFFu) << 8) | (unsigned long long)((cn.ties - cn0.ties) & 0xFFu);
                if (wave == 0) tr[6] = ((unsigned long long)(cn.nodes - cn0.nodes) << 32) | (unsigned long long)(cn.leaves - cn0.leaves);
            }
        }
        if (W > 1) __syncthreads();          // (the next packet resets what this one's merge has just read)
    }
    if (DBG && lane == 0) {
        atomicAdd(ix.dbg + 0, cn.steps);
        atomicAdd(ix.dbg + 1, cn.overflow);
        atomicAdd(ix.dbg + 2, cn.ties);
        atomicAdd(ix.dbg + 3, cn.points);
        atomicAdd(ix.dbg + 4, cn.nodes + cn.leaves);
        atomicAdd(ix.dbg + 5, cn.rejected);
        atomicAdd(ix.dbg + 8, cn.wantlanes);
    }
}

// The first pass's packets and their start order (engine.cpp, set_source).  A packet is a run of up to 64 consecutive queries of the
// Morton-sorted share.  Its cost is predicted by the radius of its queries around their centroid (invariant under the rigid transform
// the alignment applies): packets are started longest-first.  A block of 64 queries that straddles a jump of the Morton curve would be
// one packet 6-20 times wider than the rest -- 100-190 sweep steps instead of 19, 300-600 us: the slowest packets of every launch, and
// what a launch with few packets per GPU (a rank's share) lasts as long as -- so a block is cut into runs at its (up to 3) jumps: steps
// between consecutive queries longer than kJumpFactor x the block's own scale (the smallest radius of its four groups of 16).
// Per run: (first, count) into `runs`, key = descending squared radius (top key_bits bits), value = its index in `runs`.
constexpr float kJumpFactor = 4.5f;        // (round 2: 3: 0.60 ms, 5: 0.55, 6: 0.56, 8-10: 0.59, never: 0.655 on the 1M surface pair; end of round 3: 3 / 4 / 4.5 / 5 / 5.5 / never: 0.446 / 0.395 / 0.394 / 0.390 / 0.492 / 0.479 -- flat between 4 and 5 with a cliff right behind 5, hence 4.5)
// A block whose four groups of 16 ALL straddle jumps has no small group to take its scale from, and with three cuts at most a block with
// more jumps kept runs hundreds of point spacings wide: the first blocks of a rank's share (a thin strip of the cloud, crossed by the
// Morton curve again and again) were such -- two packets per share of the 8-way split 1M pair overflowed their frontier and finished
// depth-first in 450-500 us, which was the whole first pass of those ranks.  So the scale is also bounded by the block's MEDIAN step
// between consecutive queries (kMedianSteps of them: a compact group of 16 is 4-8 median steps across), and a block may be cut 7 times.
constexpr float kMedianSteps = 6.0f;
constexpr int kMaxRunsPerBlock = 8;

__global__ __launch_bounds__(64) void k_packet_runs(CloudSoA src, uint32_t n, float jump_factor, uint2 *__restrict__ runs, uint32_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals, uint32_t *__restrict__ count, int key_bits)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const int lane = threadIdx.x;
    const bool active = i < n;
    const float x = active ? src.x[i] : 0.f, y = active ? src.y[i] : 0.f, z = active ? src.z[i] : 0.f;
    // scale of the block: smallest squared radius of its (non-empty) groups of 16
    float sx = x, sy = y, sz = z, cnt = active ? 1.f : 0.f;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        sx += __shfl_xor(sx, off, 64); sy += __shfl_xor(sy, off, 64); sz += __shfl_xor(sz, off, 64); cnt += __shfl_xor(cnt, off, 64);
    }
    float r16 = active ? dist2(x, y, z, sx / cnt, sy / cnt, sz / cnt) : 0.f;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) r16 = fmaxf(r16, __shfl_xor(r16, off, 64));
    float scale2 = (cnt >= 8.f) ? r16 : __int_as_float(0x7f800000);        // (a nearly empty tail group says nothing)
    scale2 = fminf(scale2, __shfl_xor(scale2, 16, 64));
    scale2 = fminf(scale2, __shfl_xor(scale2, 32, 64));
    // steps from the previous query of the block, and their median (rank counting over the wave; ties by lane)
    const float px = __shfl_up(x, 1, 64), py = __shfl_up(y, 1, 64), pz = __shfl_up(z, 1, 64);
    const bool has_step = active && lane > 0;
    const float step2 = has_step ? dist2(x, y, z, px, py, pz) : __int_as_float(0x7f800000);
    const int nsteps = __popcll(__ballot(has_step));
    int rank = 0;                          // steps of the block shorter than this lane's
    for (int j = 1; j < 64; j++) {
        const float o = __shfl(step2, j, 64);
        rank += (o < step2 || (o == step2 && j < lane)) ? 1 : 0;
    }
    {
        const unsigned long long mm = __ballot(has_step && rank == (nsteps - 1) / 2);
        if (nsteps >= 3 && mm) {
            const float med2 = __shfl(step2, (int)__ffsll((long long)mm) - 1, 64);
            scale2 = fminf(scale2, kMedianSteps * kMedianSteps * med2);
        }
    }
    // jumps: steps longer than the factor allows; of more than kMaxRunsPerBlock - 1 the longest are cut
    const bool jump = has_step && jump_factor > 0.f && step2 > jump_factor * jump_factor * scale2 && rank >= nsteps - (kMaxRunsPerBlock - 1);
    const unsigned long long cuts = __ballot(jump) | 1ull;                    // bit j set: a run starts at lane j
    const unsigned long long act = __ballot(active);
    // this lane's run: from the last cut at or below it to the next cut (or the end of the active lanes)
    const unsigned long long below = cuts & ((2ull << lane) - 1ull);
    const int start = 63 - __clzll((long long)below);
    const unsigned long long above = cuts & ~((2ull << lane) - 1ull);
    const int nact = __popcll(act);
    const int end = above ? (int)__ffsll((long long)above) - 1 : nact;       // exclusive
    // radius of the run: centroid, then the farthest member (segmented by run: lanes exchange only within [start, end))
    float cx = 0.f, cy = 0.f, cz = 0.f, r2 = 0.f;
    for (unsigned long long m = cuts; m; m &= m - 1) {                       // (<= kMaxRunsPerBlock runs)
        const int s0 = (int)__ffsll((long long)m) - 1;
        const bool mine = active && start == s0;
        float ax = mine ? x : 0.f, ay = mine ? y : 0.f, az = mine ? z : 0.f, ac = mine ? 1.f : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ax += __shfl_xor(ax, off, 64); ay += __shfl_xor(ay, off, 64); az += __shfl_xor(az, off, 64); ac += __shfl_xor(ac, off, 64);
        }
        float d = mine ? dist2(x, y, z, ax / ac, ay / ac, az / ac) : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d = fmaxf(d, __shfl_xor(d, off, 64));
        if (mine) { cx = ax; cy = ay; cz = az; r2 = d; }
    }
    (void)cx; (void)cy; (void)cz;
    if (active && lane == start) {
        const uint32_t k = atomicAdd(count, 1u);
        runs[k] = make_uint2(i, (uint32_t)(end - start));
        // (key_bits <= 0: Morton order, the key is the run's first query)
        keys[k] = key_bits <= 0 ? i : (key_bits >= 32 ? ~__float_as_uint(r2) : ((1u << key_bits) - 1u) - (__float_as_uint(r2) >> (32 - key_bits)));
        vals[k] = k;
    }
}

// A better start key when the target is already known (set_target before set_source): what a packet costs is set by how far its queries are from
// the target, which its radius does not see -- the packets that end a launch are the ones 3 to 5 times the median distance away (their balls touch
// ten times the leaves).  One wave per run: centroid, greedy descent of the target's octree to the leaf nearest to it (as the search's own dive),
// distance d to that leaf's nearest point; the ball of radius d + r around the centroid cuts a disc of area ~ 2 d r + r^2 out of a surface: that
// is the key.  Evaluated under the IDENTITY transform (the usual initial guess): another guess only changes the start order, never a result.
__global__ __launch_bounds__(64) void k_packet_cost(CloudSoA src, const uint2 *__restrict__ runs, uint32_t npk, TargetIndex ix, uint32_t *__restrict__ keys, int key_bits)
{
    const uint32_t k = blockIdx.x;
    if (k >= npk) return;
    const int lane = threadIdx.x;
    const uint2 run = runs[k];
    const bool active = (uint32_t)lane < run.y;
    const uint32_t i = run.x + (uint32_t)lane;
    const float x = active ? src.x[i] : 0.f, y = active ? src.y[i] : 0.f, z = active ? src.z[i] : 0.f;
    float ax = x, ay = y, az = z, ac = active ? 1.f : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ax += __shfl_xor(ax, off, 64); ay += __shfl_xor(ay, off, 64); az += __shfl_xor(az, off, 64); ac += __shfl_xor(ac, off, 64); }
    const float cx = ax / ac, cy = ay / ac, cz = az / ac;
    float r2 = active ? dist2(x, y, z, cx, cy, cz) : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off, 64));
    // greedy descent: at every level the child whose box is nearest to the centroid
    const float4 rootA = ix.onodes[0], rootB = ix.onodes[1];
    uint32_t first = (uint32_t)__float_as_int(rootA.w), packed = (uint32_t)__float_as_int(rootB.w), level = 0;
    while (oct_nch(packed) != 0u) {
        const uint32_t nch = oct_nch(packed), cf = oct_cf(packed);
        uint32_t key = 0xFFFFFFFFu;
        float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cB = cA;
        if ((uint32_t)lane < nch) {
            const float4 *__restrict__ ch = ix.onodes + 2 * ((size_t)ix.olevel_off[level + 1] + cf + (uint32_t)lane);
            cA = ch[0]; cB = ch[1];
            key = (__float_as_uint(boxdist2(cx, cy, cz, cA, cB)) & ~7u) | (uint32_t)lane;
        }
        key = dpp_umin<0x111, 0xf>(key); key = dpp_umin<0x112, 0xf>(key); key = dpp_umin<0x114, 0xf>(key);
        const int cmin = __builtin_amdgcn_readlane((int)key, 7) & 7;
        packed = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cB.w), cmin);
        first = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cA.w), cmin);
        level++;
    }
    const uint32_t cnt = oct_cf(packed);
    float d2 = __int_as_float(0x7f800000);
    if ((uint32_t)lane < cnt) { const float4 q = ix.tq[first + (uint32_t)lane]; d2 = dist2(cx, cy, cz, q.x, q.y, q.z); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d2 = fminf(d2, __shfl_xor(d2, off, 64));
    if (lane == 0) {
        const float cost = 2.0f * sqrtf(d2) * sqrtf(r2) + r2;
        const float c2 = (cost == cost && cost < 3.0e38f) ? cost : r2;
        keys[k] = key_bits >= 32 ? ~__float_as_uint(c2) : ((1u << key_bits) - 1u) - (__float_as_uint(c2) >> (32 - key_bits));
    }
}

// the packet table in start order
__global__ __launch_bounds__(256) void k_packet_table(const uint32_t *__restrict__ order, const uint2 *__restrict__ runs, uint32_t npk, uint2 *__restrict__ tab)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < npk) tab[k] = runs[order[k]];
}

void launch_packet_runs(const CloudSoA &src, uint32_t n, float jump_factor, uint2 *runs, uint32_t *keys, uint32_t *vals, uint32_t *count, int key_bits, hipStream_t s)
{
    const uint32_t nblk = (n + 63u) / 64u;
    if (nblk) hipLaunchKernelGGL(k_packet_runs, dim3(nblk), dim3(64), 0, s, src, n, jump_factor < 0.f ? kJumpFactor : jump_factor, runs, keys, vals, count, key_bits);
}

void launch_packet_cost(const CloudSoA &src, const uint2 *runs, uint32_t npk, const TargetIndex &ix, uint32_t *keys, int key_bits, hipStream_t s)
{
    if (npk) hipLaunchKernelGGL(k_packet_cost, dim3(npk), dim3(64), 0, s, src, runs, npk, ix, keys, key_bits);
}

void launch_packet_table(const uint32_t *order, const uint2 *runs, uint32_t npk, uint2 *tab, hipStream_t s)
{
    if (npk) hipLaunchKernelGGL(k_packet_table, dim3((npk + 255u) / 256u), dim3(256), 0, s, order, runs, npk, tab);
}

// first pass of an alignment: no previous pairs, every query is searched -- packets over the whole (sorted) share
void launch_accumulate(const PassArgs &a, const float4 *tn, int blocks, hipStream_t s);

// Waves per packet.  One wave per packet has the best throughput (no barriers, no redundant dives) as long as the launch has several
// rounds of packets per wave slot to balance its tail with (longest first); a launch with few packets per slot -- a rank's share of a
// sharded run, a small cloud -- lasts as long as its slowest packets, and those run W times faster on W waves.
static uint32_t pick_packet_waves(uint32_t npk)
{
    constexpr uint32_t kWaveSlots = 256u * 4u * PKT_WAVESN;     // CUs x SIMDs x waves per SIMD of the shared-packet kernels
    if (npk >= 4u * kWaveSlots) return 1u;
    if (npk >= 2u * kWaveSlots) return 2u;
    return 4u;
}

template <int W>
static void launch_packets(const PassArgs &a, const TargetIndex &ix, uint32_t nbp, uint32_t chunk, uint32_t lds_pad, hipStream_t s)
{
    if (ix.dbg) hipLaunchKernelGGL((k_search_packet<W, true>), dim3(nbp), dim3(64 * W), lds_pad, s, a, ix, chunk);
    else hipLaunchKernelGGL((k_search_packet<W, false>), dim3(nbp), dim3(64 * W), lds_pad, s, a, ix, chunk);
}

void launch_pass_tree_first(const PassArgs &a_in, const TargetIndex &ix, const WorkLists &, int acc_blocks, hipStream_t s, hipEvent_t *ev)
{
    PassArgs a = a_in;
    a.refresh_records = 1;
    const uint32_t npk = a.pkt_tab ? a.pkt_count : (a.n + 63u) / 64u;
    const uint32_t chunk = a.pkt_chunk ? a.pkt_chunk : 64u;      // packets per chunk (xcd_remap_chunked)
    const uint32_t nbp = ((npk + 8u * chunk - 1u) / (8u * chunk)) * (8u * chunk);
    if (ev) { hipEventRecord(ev[0], s); hipEventRecord(ev[1], s); hipEventRecord(ev[2], s); }
    const uint32_t w = a.pkt_waves ? a.pkt_waves : pick_packet_waves(npk);
    if (!npk) { /* empty share */ }
    else if (w >= 4u) launch_packets<4>(a, ix, nbp, chunk, a.pkt_lds_pad, s);
    else if (w >= 2u) launch_packets<2>(a, ix, nbp, chunk, a.pkt_lds_pad, s);
    else launch_packets<1>(a, ix, nbp, chunk, a.pkt_lds_pad, s);
    if (ev) hipEventRecord(ev[3], s);
    launch_accumulate(a, ix.tn, acc_blocks, s);
    if (ev) hipEventRecord(ev[4], s);
}

}  // namespace symmicp
