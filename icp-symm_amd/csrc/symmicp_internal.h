// symmicp_internal.h -- shared between the HIP kernels and the host engine.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, 256 CUs in 8 XCDs, 160 KB LDS/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "symmicp.h"

namespace symmicp {

constexpr int kNSum = SYMMICP_NSUM;   // 40 doubles per reduction record
constexpr int kNAcc = 37;             // live accumulators (rest of the record is 0)
constexpr int kLeaf = 8;              // target points per leaf (8 x 16 B = one 128-B line)
constexpr int kFan = 8;               // children per tree node
constexpr int kMaxTreeLevels = 12;
constexpr int kMortonBits = 10;       // per axis -> 30-bit keys
constexpr int kPassThreads = 256;
constexpr int kBruteTile = 1024;      // target points staged in LDS per tile (16 KB)
constexpr int kBruteQ = 4;            // queries per thread in the brute-force kernel

// 3x4 affine passed by value in kernel arguments (row-major), plus the weight
// the translation column gets when it is applied to normals
// (1 = reference quirk myicp.cpp:137, 0 = rotate only).
struct Affine {
    float m[12];
    float nrm_w;
};

// planar (structure-of-arrays) cloud: the layout of the reference's column-major
// Eigen::MatrixXf N x 3 blocks (func.cpp:5-15).
struct CloudSoA {
    float *x, *y, *z, *nx, *ny, *nz;
};

// Search index over the target: Morton-sorted points, dense cell table at one
// octree level (the "grid"), and an implicit 8-ary tree of tight boxes over
// runs of the sorted order (the "linear BVH").
struct TargetIndex {
    const float4 *tq;        // sorted: xyz + original row (int bits) in w
    const float4 *tn;        // sorted, 2 per point: (xyz + original row, normal xyz + 0) -- the 32-byte pair record of the accumulating kernels
    uint32_t n;
    // tree: 2 float4 per node (lo, hi); level l starts at node level_off[l]
    const float4 *boxes;
    uint32_t level_off[kMaxTreeLevels];
    int32_t top;             // index of the top level
    uint32_t ntop;           // nodes in the top level (<= kFan)
    // grid
    // two-level cell table: ctop[morton >> 9] = block number (or ~0 when the 8x8x8 super-cell is empty),
    // cells[block * 512 + (morton & 511)] = (first, last+1) of the Morton cell, 0/0 when empty.
    // Memory follows the occupied super-cells (a surface touches few), so the grid can go to level 10.
    const uint32_t *ctop;
    const uint2 *cells;
    int32_t glevel;          // 0 = no grid
    int32_t gdim;            // 1 << glevel
    float ox, oy, oz;        // grid origin (target bbox min)
    float h, inv_h;          // cell edge at glevel
    unsigned long long *dbg; // optional debug counters (SYMMICP_DEBUG_COUNTERS=1), else null
    unsigned long long *dbg_trace;   // with dbg: per packet (start tick, ticks), (pops, hw id) -- 2 x u64 x 2 stages
    // sparse octree over the same sorted points: node = one Morton prefix (an octree cell that holds points),
    // levels 0 (root) .. kMortonBits.  Two float4 per node:
    //   A = (lo.x, lo.y, lo.z, first point as int bits)
    //   B = (hi.x, hi.y, hi.z, packed)   packed = child_first (28 bits, index within the next level) | nchild << 28 ;
    //                                     nchild == 0 marks a leaf, whose low 28 bits hold its point count
    //                                     (oct_nch / oct_cf below; 2^28 nodes per level: targets of ~250M points)
    // Sibling cells are disjoint, so box distances discriminate at every level (unlike runs of the sorted order,
    // whose boxes straddle the big jumps of the Z curve).
    const float4 *onodes;
    uint32_t olevel_off[kMortonBits + 2];
};

constexpr uint32_t kOctCfMask = 0x0FFFFFFFu;
__host__ __device__ inline uint32_t oct_nch(uint32_t packed) { return packed >> 28; }
__host__ __device__ inline uint32_t oct_cf(uint32_t packed) { return packed & kOctCfMask; }

// Append lists for queries a kernel hands to a later kernel of the same pass.  One returning atomic on a single
// word tops out near 88 per microsecond chip-wide, so a list is 64 shards (counter words 64 B apart); a producer
// block appends to shard (blockIdx & 63) and consumer block b drains shard (b & 63): same XCD on both sides.
constexpr int kShards = 64;
constexpr int kShardStride = 16;          // uint32 words between shard counters
struct ShardList {
    uint32_t *items;                      // [kShards][cap]
    uint32_t *counts;                     // [kShards * kShardStride]
    uint32_t cap;
};
struct WorkLists {
    ShardList work;                       // queries k_search_cells hands to k_search_walk
    ShardList retry;                      // queries the budgeted thread-per-query walk gave up on
};                                        // the two counter blocks are contiguous and cleared by the final reduce

// Device-resident state of a device-driven run of passes (symmicp_align's batches): written by k_reduce_solve, read by the
// pass kernels.  Lives in device memory; `ring` entries live in host-mapped memory and are read by the host afterwards.
struct LoopState {
    float X[16];             // cumulative transform (myicp.cpp:138)
    Affine Xapply;           // what the next pass applies to its input (cumulative apply: X; incremental: the last increment)
    int32_t stop;            // != 0: every kernel of the batch returns at once
    int32_t reason;          // LOOP_* below
    int32_t iters;           // passes of this alignment completed so far (the reference's `iters`)
    int32_t small_step;      // the last increment was below eps_rotation / eps_translation: stop after the pass that applies it
};
enum { LOOP_RUNNING = 0, LOOP_DONE = 1, LOOP_REDO_PASS = 2, LOOP_HOST_SOLVE = 3, LOOP_SLOW = 4 };
struct LoopConfig {
    int32_t mode, fixed_iters, max_iters, incremental, tree;
    float diff_threshold, eps_rotation, eps_translation, nrm_w;
    float pivot[3];
    uint32_t uncertified_limit;   // TREE: more pairs than this searched in a pass -> back to the separate kernels (LOOP_SLOW)
    int32_t walk_in_loop;         // TREE: the batch runs k_search_walk + k_accumulate_list behind every fused pass: a non-empty work list is
                                  // handled there (it stops the loop otherwise: LOOP_REDO_PASS)
    uint32_t list_limit;          // ... unless it is longer than this (LOOP_SLOW: the host sizes the walk for long lists)
};
struct LoopRecord {          // one per pass, host-mapped
    double sums[SYMMICP_NSUM];       // the pass's record (after the exchange over ranks)
    float increment[16], X[16];      // the increment solved FROM this record and the cumulative transform after it
    float rcond;
    int32_t status;                  // status of that solve
    int32_t solved;                  // 1: increment / X are valid (the loop went on)
    int32_t pad;                     // (pairs searched in this pass)
    int32_t list_len, reserved;      // work-list length of this pass (queries handed to the tree walk)
};

struct PassArgs {
    // source share (planar).  `in` is read; if writeback, `out` receives the transformed points/normals
    CloudSoA in, out;
    uint32_t n;              // points in this share
    uint32_t tgt_offset;     // identity pairing: target row = tgt_offset + i
    Affine X;                // transform applied to `in` on the fly
    float pivot[3];          // subtracted from p and q before forming rows (0 in QUIRKS)
    float max_d2;            // <= 0: keep all pairs
    float min_ndot;          // <= -1: keep all pairs; else drop pairs with n_p . n_q below it
    int32_t p2p;             // point-to-point mode: slots 0..8 accumulate p q^T instead of the 6x6 Gram / rhs
    int32_t writeback;
    // correspondences
    const unsigned long long *best64;   // BRUTE: (d2 bits << 32 | target row), ~0 = none
    int32_t *pos_prev;                  // TREE: sorted position found by the previous pass (-1 = none); updated
    float *d2_out;                      // optional per-point squared distance (may be null)
    int32_t *pos_out;                   // sorted position / target row chosen this pass (may alias pos_prev)
    double *partials;                   // [kNSum][blocks] (transposed)
    // pair certificates (TREE): position of the query when its pair was last searched, and the radius around it known
    // to hold no other target point (see k_search_cells)
    float4 *pairrec;                    // TREE: per pair, its own copy of the target's (point, normal) record (2 float4)
    int32_t budget_walk;                // this pass's thread-per-query walk runs with a visit budget + retry launch (see k_search_walk)
    int32_t refresh_records;            // k_accumulate may replace stale copies (0 in a pass that may still be repaired)
    float4 *cert;                       // per pair: (ref.xyz, clear radius L around ref; L <= 0: no single certificate); bit 0 of L's word: certk[i] is valid
    uint4 *certk;                       // per pair, 2 x uint4: the neighbourhood certificate's members (8 sorted positions; see hood_test in kernels_pass.hip)
    float2 *hoodr;                      // ... its radius T and the radius hint for the pair's next scan
    int32_t make_hood;                  // k_search_cells keeps neighbourhoods in this pass (existing ones are honoured either way)
    int32_t use_slack;                  // 0 on the first pass of an alignment (certificates not valid yet)
    const LoopState *loop;              // device-driven loop: the transform comes from here (null: from X above)
    uint32_t partial_cols, partial_col0;   // partial records: columns in all (0: the grid's size) and this launch's first column
    const uint2 *pkt_tab;               // first pass (packets): (first query, count <= 64) per packet in start order (null: 64 as they lie)
    uint32_t pkt_count;
    uint32_t pkt_waves;                 // first pass: waves that share one packet (1, 2 or 4; 0: the launcher decides from the packet count)
    uint32_t pkt_front_cap;             // ... frontier capacity per level (0: kFrontCap; smaller values force the depth-first fallback: tests)
    uint32_t *pkt_fallbacks;            // ... counts the packets that finished depth-first (may be null)
    uint32_t pkt_chunk, pkt_lds_pad;    // ... packets per XCD chunk (0: 64); extra dynamic LDS per workgroup (occupancy experiments)
};

// host-side launch tuning of the tree passes (environment switches, read once by the engine)
struct PassTuning {
    uint32_t wave_mode_max = 20000u;    // work lists longer than this use one thread per query (the 48k-entry list after the first move of the 1M surface pair: 115 -> 85 us)
    uint32_t cells_chunk = 16u;         // tiles per XCD chunk of k_search_cells
    uint32_t cells_queries = 0u;        // queries per tile of k_search_cells (64, 128, 256; 0: from the share's size)
    uint32_t walk_budget = 160u;        // node visits of the budgeted walk (sharded first passes on volume-like targets)
};

// ---- kernel launchers (kernels.hip) ---------------------------------------
void launch_pass_identity(const PassArgs &a, CloudSoA tgt, int blocks, bool vec4_ok, hipStream_t s);
void launch_pass_indexed(const PassArgs &a, const float4 *tn /* pair records */, int blocks, hipStream_t s);
// ev: null, or 5 events recorded before cells / after cells / (same again) / after walk / after accumulate
// walk_blocks: grid of k_search_walk (any size is correct; 0 = one thread per possible list entry)
// stage: 0 = cells, walk, accumulate; 1 = cells, accumulate (walk skipped); 2 = walk, accumulate (repair of a stage-1 pass)
// compact_blocks > 0: the cell search runs as the streaming, compacting kernel (k_pass_fused<false>) on that many blocks
void launch_pass_tree_split(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int acc_blocks, uint32_t walk_blocks,
                            int stage, int compact_blocks, const PassTuning &tune, hipStream_t s, hipEvent_t *ev);
void launch_pass_tree_first(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int acc_blocks, hipStream_t s, hipEvent_t *ev);
uint32_t walk_blocks_full(const WorkLists &wl);
uint32_t shard_capacity(uint32_t n_points);
// keep_nonempty: leave the append-list counters alone when the work list is not empty (stage-1 passes)
void launch_final_reduce(const double *partials, int blocks, double *out_dev, double *out_host_mapped, uint32_t *ticket,
                         unsigned long long seq, uint32_t *counters_to_clear, int keep_nonempty, hipStream_t s);
// fused pass of a converged alignment (k_pass_fused) and the device-side end of a pass (k_reduce_solve): see kernels_pass.hip
void launch_pass_fused(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int blocks, hipStream_t s);
// device-driven loop, straggler stage behind a fused pass: the work list through the tree walk, then its pairs into `list_blocks` more partial columns
void launch_loop_stragglers(const PassArgs &a, const TargetIndex &ix, const WorkLists &wl, int list_blocks, const PassTuning &tune, hipStream_t s);
// mode 0: reduce + check + solve (single GPU); 1: the record is already in out_dev (after the all-reduce); 2: solve only (start of a batch)
void launch_reduce_solve(const double *partials, int blocks, double *out_dev, int mode, LoopState *loop, LoopConfig cfg, LoopRecord *ring, int ring_len,
                         uint32_t *counters_to_clear, hipStream_t s);
void launch_loop_end(const LoopState *loop, LoopState *host_copy, unsigned long long *done_flag, unsigned long long seq, hipStream_t s);
void launch_publish(const double *sums_dev, double *out_host_mapped, unsigned long long seq, hipStream_t s);
void launch_nn_brute(const CloudSoA &src, uint32_t n_s, const Affine &X, const float4 *tq, uint32_t n_t,
                     unsigned long long *best64, hipStream_t s);

// index build
void launch_bbox(const float *x, const float *y, const float *z, uint32_t n, uint32_t *bbox_ord6, hipStream_t s);
void launch_morton(const float *x, const float *y, const float *z, uint32_t n, float ox, float oy, float oz,
                   float inv_h0, uint32_t *keys, uint32_t *vals, hipStream_t s);
// sorts (keys, vals) ascending, stable; tmp arrays same size; result lands back in keys/vals
void radix_sort_pairs(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp, uint32_t n,
                      int key_bits, uint32_t *hist_ws, size_t hist_ws_elems, hipStream_t s);
size_t radix_sort_ws_elems(uint32_t n);
void launch_gather_f4(const float *x, const float *y, const float *z, const float *nx, const float *ny, const float *nz,
                      const uint32_t *order, uint32_t n, float4 *tq, float4 *tn, hipStream_t s);
void launch_gather_soa(const CloudSoA &src, const uint32_t *order, uint32_t n, CloudSoA dst, hipStream_t s);
void launch_offset_u32(const uint32_t *in, uint32_t off, uint32_t n, uint32_t *out, hipStream_t s);
// first pass: the packets (runs of <= 64 queries, blocks cut at jumps of the Morton curve; jump_factor < 0: the default) with radius keys, and the table in start order
void launch_packet_runs(const CloudSoA &src, uint32_t n, float jump_factor, uint2 *runs, uint32_t *keys, uint32_t *vals, uint32_t *count, int key_bits, hipStream_t s);
// start keys from the packets' distance to the target (identity transform): replaces the radius keys of launch_packet_runs when the target index exists
void launch_packet_cost(const CloudSoA &src, const uint2 *runs, uint32_t npk, const TargetIndex &ix, uint32_t *keys, int key_bits, hipStream_t s);
void launch_packet_table(const uint32_t *order, const uint2 *runs, uint32_t npk, uint2 *tab, hipStream_t s);
void launch_deinterleave3(const float *raw, size_t row_stride, size_t offset, uint32_t n, float *x, float *y, float *z, hipStream_t s);
void launch_level_hist(const uint32_t *keys, uint32_t n, uint32_t *hist16, hipStream_t s);
void launch_cell_table(const uint32_t *keys, uint32_t n, int glevel, const uint32_t *nid_top, uint32_t *ctop, uint2 *cells, hipStream_t s);
// sparse octree build (levels bottom-up); see TargetIndex::onodes
void launch_oct_flags(const uint32_t *keys, uint32_t n, int level, uint32_t *nid, hipStream_t s);
void launch_exclusive_scan(uint32_t *data, uint32_t n, uint32_t *tile_ws /* >= ceil(n/2048) words */, hipStream_t s);
void launch_oct_first(const uint32_t *keys, uint32_t n, int level, const uint32_t *nid, uint32_t *first, hipStream_t s);
void launch_oct_nodes(int level, const float4 *tq, uint32_t n, const uint32_t *first, uint32_t n_nodes, const uint32_t *nid_next,
                      uint32_t n_nodes_next, const float4 *nodes_next, float4 *nodes, uint32_t leaf_max, hipStream_t s);
void launch_leaf_boxes(const float4 *tq, uint32_t n, float4 *boxes, uint32_t nleaf_padded, hipStream_t s);
void launch_node_boxes(const float4 *child, uint32_t nchild_padded, float4 *parent, uint32_t nparent_padded, hipStream_t s);
void launch_iota_f4(const float *x, const float *y, const float *z, const float *nx, const float *ny, const float *nz,
                    uint32_t n, float4 *tq, float4 *tn, hipStream_t s);
void launch_unpermute(const CloudSoA &cur, const uint32_t *order, uint32_t n, float *xyz_aos, float *nrm_aos, hipStream_t s);
void launch_corr_out(const int32_t *pos, const unsigned long long *best64, const float *d2, const float4 *tq,
                     const uint32_t *src_order, uint32_t n, int mode, uint32_t tgt_offset, int32_t *idx_out, float *d2_out,
                     hipStream_t s);

void launch_identity_d2(const CloudSoA &in, const Affine &X, const CloudSoA &tgt, uint32_t tgt_offset, uint32_t n, float *d2, hipStream_t s);
void launch_pairs_d2(const CloudSoA &in, const Affine &X, const int32_t *pos, const float4 *tq, uint32_t n_t, uint32_t n, float *d2, hipStream_t s);
void launch_normals_knn(const TargetIndex &ix, int k, const float vp[3], float *nrm_out, float *curv_out, hipStream_t s);

}  // namespace symmicp
