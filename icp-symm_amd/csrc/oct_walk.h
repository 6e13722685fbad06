// oct_walk.h -- exact per-thread walk of the sparse octree (TargetIndex::onodes), shared by k_search_walk
// (kernels_pass.hip) and the packet search's per-lane regime (kernels_packet.hip).
#pragma once
#include "symmicp_internal.h"
#include "device_common.h"
#pragma clang fp contract(off)

namespace symmicp {

struct Best {
    float d2;
    int32_t pos;      // sorted position
    int32_t row;      // original target row (tie-break key)
};

__device__ __forceinline__ void test_point(Best &b, const float4 &q, int32_t pos, float px, float py, float pz)
{
    const float d2 = dist2(px, py, pz, q.x, q.y, q.z);
    if (d2 <= b.d2) {                                  // rarely taken: most candidates are not better
        const int32_t row = __float_as_int(q.w);
        if (d2 < b.d2 || row < b.row) { b.d2 = d2; b.pos = pos; b.row = row; }
    }
}


// ---- exact walk of the sparse octree (TargetIndex::onodes) ------------------------------------------
// Nodes are octree cells that hold points; siblings are disjoint, so the box distance separates them at every
// level and the first root-to-leaf descent already lands next to the answer (the run tree above needs ~30 more
// expansions for that on a surface cloud: runs of the sorted order straddle the jumps of the Z curve and their
// boxes overlap).  Visiting a node: a leaf scans its (<= 8, unless duplicates pile up in one finest cell) points;
// an internal node loads its <= 8 child boxes (contiguous in the next level), enters the nearest one that can still
// win and remembers the others as (child_first << 8 | 8-bit mask) in a register stack, one word per level.  Coming
// back to a sibling, its box is tested again against the (now smaller) best.
// one word per level: (low 24 bits of child_first) << 8 | 8-bit mask of the siblings still to visit; the top 4 bits of the
// 28-bit child_first ride in a 64-bit shift register of nibbles (two registers instead of one more word per level: the walk
// kernel is register-bound)
struct OctStack {
    uint32_t s[kMortonBits];
    unsigned long long hi;
    __device__ __forceinline__ void push(uint32_t cf, uint32_t mask)
    {
#pragma unroll
        for (int k = kMortonBits - 1; k > 0; k--) s[k] = s[k - 1];
        s[0] = ((cf & 0xFFFFFFu) << 8) | mask;
        hi = (hi << 4) | (unsigned long long)(cf >> 24);
    }
    __device__ __forceinline__ void pop()
    {
#pragma unroll
        for (int k = 0; k < kMortonBits - 1; k++) s[k] = s[k + 1];
        s[kMortonBits - 1] = 0;
        hi >>= 4;
    }
    __device__ __forceinline__ uint32_t top_cf() const { return (s[0] >> 8) | ((uint32_t)(hi & 0xFull) << 24); }
};
static_assert(kMortonBits * 4 <= 64, "nibble register too short for the octree depth");

// One loop iteration = one node visit, and a leaf is visited with the SAME code as an internal node: its points
// are read as degenerate boxes (lo = hi = point), for which boxdist2 is bit-for-bit dist2.  Lanes of a wave sit at
// different nodes of different kinds, but now they all execute one common body (load <= 8 entities, 8 distances)
// instead of serialising a leaf path, an internal path and a sibling path each trip.
// BUDGETED: give up after `budget` node visits (return value > budget); b then holds the best point seen so far, a valid
// bound for whoever finishes the query.  The plain instantiation carries no trace of it (the walk kernel is
// register-bound: the extra state costs the unbudgeted walk 4 %).
template <bool BUDGETED = false>
__device__ __forceinline__ uint32_t oct_walk(const TargetIndex &ix, float px, float py, float pz, Best &b, uint32_t budget = 0xFFFFFFFFu)
{
    const float inf = __int_as_float(0x7f800000);
    OctStack st;
#pragma unroll
    for (int k = 0; k < kMortonBits; k++) st.s[k] = 0;
    st.hi = 0;
    int L = 0;
    uint32_t idx = 0;
    uint32_t visits = 0;
    while (true) {
        visits++;
        if (BUDGETED && visits > budget) return visits;
        const float4 *__restrict__ nd = ix.onodes + 2 * ((size_t)ix.olevel_off[L] + idx);
        const float4 hA = nd[0], hB = nd[1];
        const uint32_t first = (uint32_t)__float_as_int(hA.w);
        const uint32_t packed = (uint32_t)__float_as_int(hB.w);
        // the header is the node's own box: a sibling taken from the pending list is re-tested here against the
        // current best, so the list pop below needs no loads of its own
        const bool alive = boxdist2(px, py, pz, hA, hB) <= b.d2;
        const uint32_t nch = oct_nch(packed);
        const bool leaf = (nch == 0);
        const uint32_t cf = oct_cf(packed);                     // internal: first child; leaf: number of points
        const uint32_t total = alive ? (leaf ? cf : nch) : 0u;
        // entity e of this node: leaf -> point tq[first + e] ; internal -> child box onodes[level L+1][cf + e]
        const float4 *__restrict__ ent = leaf ? (ix.tq + first) : (ix.onodes + 2 * ((size_t)ix.olevel_off[L + 1] + cf));
        const int stride = leaf ? 1 : 2;
        bool descended = false;
        for (uint32_t e0 = 0; e0 < total; e0 += 8) {           // more than one trip only for > 8 duplicates in a finest cell
            // All lanes run the same straight-line body over 8 entity slots; slots past the end re-read the last
            // entity (harmless duplicate) instead of being predicated off.  One running arg-min serves both kinds:
            // for a leaf it is the candidate point (ties -> lowest row), for an internal node the child to enter.
            const uint32_t last = total - 1 - e0;               // index of the last valid slot in this group (may be >= 8)
            uint32_t mask = 0;
            int ec = 0, erow = 0x7fffffff;
            float ed = inf;
#pragma unroll
            for (int h = 0; h < 8; h += 4) {
                float4 lo[4], hi[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t e = e0 + min((uint32_t)(h + c), last);
                    lo[c] = ent[e * stride];
                    hi[c] = ent[e * stride + (stride - 1)];
                }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // branch-free on purpose (selects, not jumps): eight tiny divergent branches per visit cost more
                    // scalar/exec bookkeeping than the work they skip
                    const float d = boxdist2(px, py, pz, lo[c], hi[c]);
                    const int row = __float_as_int(lo[c].w);
                    mask |= (d <= b.d2) ? (1u << (h + c)) : 0u;
                    const bool better = (d < ed) | ((d == ed) & (row < erow));
                    ed = better ? d : ed;
                    erow = better ? row : erow;
                    ec = better ? (h + c) : ec;
                }
            }
            if (last < 7u) mask &= (2u << last) - 1u;           // drop the duplicate slots
            if (leaf) {
                if (ed <= b.d2 && (ed < b.d2 || erow < b.row)) { b.d2 = ed; b.row = erow; b.pos = (int32_t)(first + e0 + min((uint32_t)ec, last)); }
            } else if (mask) {
                st.push(cf, mask & ~(1u << ec));
                L++;
                idx = cf + (uint32_t)ec;
                descended = true;
            }
        }
        if (descended) continue;
        // node done: next pending sibling, climbing as levels run out
        bool found = false;
        while (L > 0) {
            const uint32_t w = st.s[0];
            const uint32_t mask = w & 0xFFu;
            if (mask == 0) { st.pop(); L--; continue; }
            const int c = __ffs((int)mask) - 1;
            st.s[0] = w & ~(1u << c);
            idx = st.top_cf() + (uint32_t)c;
            found = true;
            break;
        }
        if (!found) break;
    }
    return visits;
}


}  // namespace symmicp
