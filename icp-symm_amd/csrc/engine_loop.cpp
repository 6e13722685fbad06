// engine_loop.cpp -- the iteration loop of libsymmicp (reference ICP/myicp.cpp:117-150): one pass over the source share (run_pass), device-driven
// runs of passes (run_batch), symmicp_begin / step / align, the host solve entry point and the reference's result block.
#include "engine_internal.h"

extern "C" {

// ---- one pass over the source share ------------------------------------------------------------
static void fill_pass_args(symmicp_ctx *c, PassArgs &a, const float Xapply[16], bool from_cur, bool writeback, bool first)
{
    a.in = from_cur ? c->cur : c->src0;
    a.out = c->cur;
    a.n = c->n_loc;
    a.tgt_offset = c->src_off;
    for (int k = 0; k < 12; k++) a.X.m[k] = Xapply[k];
    a.X.nrm_w = (c->cfg.mode == SYMMICP_MODE_QUIRKS) ? 1.0f : 0.0f;   // myicp.cpp:137 quirk
    const bool paper = c->cfg.mode != SYMMICP_MODE_QUIRKS;          // PAPER and P2P take their sums about the pivot
    for (int k = 0; k < 3; k++) a.pivot[k] = paper ? c->pivot[k] : 0.0f;
    a.p2p = c->cfg.mode == SYMMICP_MODE_P2P ? 1 : 0;
    a.max_d2 = c->cfg.max_corr_dist > 0.f ? c->cfg.max_corr_dist * c->cfg.max_corr_dist : 0.f;
    a.min_ndot = c->cfg.min_normal_dot;
    a.writeback = writeback ? 1 : 0;
    a.best64 = c->best64;
    a.pos_prev = first ? nullptr : c->pos;
    a.pos_out = c->pos;
    // identity pairing streams at the HBM roof: the per-pair distances (4 B/point) are only written on request
    a.d2_out = (c->cfg.corr == SYMMICP_CORR_IDENTITY) ? nullptr : c->d2;
    a.partials = c->partials;
    a.cert = reinterpret_cast<float4 *>(c->cert);
    a.certk = c->sw.no_hood ? nullptr : reinterpret_cast<uint4 *>(c->certk);
    a.hoodr = reinterpret_cast<float2 *>(c->hoodr);
    // neighbourhoods are worth their stores once the alignment is settling (the previous pass searched under 80 % of the pairs: 8M scan pair
    // 2 680 iter/s at 50 %, 2 776 at 80 %, 2 742 always; 1M surface pair 16 050 / 15 960 / 15 700)
    a.make_hood = (a.certk && !first && c->last_uncertified >= 0 && (double)c->last_uncertified < c->sw.hood_frac * (double)c->n_s_total) ? 1 : 0;
    a.pairrec = c->pairrec;
    // sharded runs: the first pass over a small share is bound by its slowest walks, not by throughput
    a.budget_walk = c->sw.budget_walk >= 0 ? c->sw.budget_walk : (first && c->nranks > 1 && c->n_loc < 400000u);      // (SYMMICP_BUDGET_WALK: "0" never, "1" always)
    a.use_slack = (!first && c->cert && !c->sw.no_cert) ? 1 : 0;
    a.loop = nullptr;
    a.pkt_tab = reinterpret_cast<const uint2 *>(c->pkt_tab);
    a.pkt_count = c->pkt_count;
    a.pkt_waves = c->sw.packet_waves;
    a.pkt_front_cap = c->sw.packet_front_cap;
    a.pkt_fallbacks = c->pkt_fallbacks;
    a.pkt_chunk = c->sw.packet_chunk;
    a.pkt_lds_pad = c->sw.packet_lds_pad;
}

static int run_pass(symmicp_ctx *c, const float Xapply[16], bool from_cur, bool writeback, bool first)
{
    PassArgs a{};
    fill_pass_args(c, a, Xapply, from_cur, writeback, first);
    int blocks = (int)((c->n_loc + kPassThreads - 1) / kPassThreads);
    int cap = c->sw.pass_blocks;
    if (cap > 8192) cap = 8192;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    c->pass_blocks = blocks;
    const double t_l0 = now_s();
    if (c->t_last_done > 0) c->t_between += t_l0 - c->t_last_done;
    hipEvent_t *ev = nullptr;
    uint32_t walk_blocks = 0;
    bool optimistic = false;
    if (c->timing) {
        if (c->ev_used == symmicp_ctx::kEvRing) flush_events(c);
        ev = c->ev + c->ev_used * symmicp_ctx::kEvPer;
        c->ev_split[c->ev_used] = (c->timing == 1 || c->timing == 3) ? 1 : 0;
        c->ev_weight[c->ev_used] = 1;
    }
    // (in per-kernel mode the split launcher records ev[0] itself)
    if (ev && !(c->timing == 2 && c->cfg.corr == SYMMICP_CORR_TREE)) hipEventRecord(ev[0], c->stream);
    switch (c->cfg.corr) {
    case SYMMICP_CORR_IDENTITY: {
        // 16-byte column loads need every planar column (length n_loc / n_t) and the shard offset to keep 16-B alignment
        const bool vec4 = (c->n_loc % 4 == 0) && (c->n_t % 4 == 0) && (c->src_off % 4 == 0);
        if (vec4) {
            const int id_cap = c->sw.id_blocks;
            const int nb4 = (int)((c->n_loc / 4 + kPassThreads - 1) / kPassThreads);
            blocks = nb4 < id_cap ? (nb4 > 0 ? nb4 : 1) : id_cap;
            c->pass_blocks = blocks;
        }
        launch_pass_identity(a, c->tgt, blocks, vec4, c->stream);
        break;
    }
    case SYMMICP_CORR_BRUTE:
        launch_nn_brute(a.in, c->n_loc, a.X, c->tq, c->n_t, c->best64, c->stream);
        launch_pass_indexed(a, c->tn, blocks, c->stream);
        break;
    default:
        {
            // the accumulate kernel is streaming with a 40-value block reduction at the end of every block: 2 blocks per
            // CU measured best (18 us at 512 blocks, 23 us at 2048, 1M points); a multiple of 8 for the XCD remap
            const int acc_cap = c->sw.acc_blocks;
            const int nb_all = (int)((c->n_loc + kPassThreads - 1) / kPassThreads);
            int ab = nb_all < acc_cap ? ((nb_all + 7) / 8) * 8 : acc_cap;
            if (ab < 8) ab = 8;                     // (a rank whose share is empty still writes its zero record)
            c->pass_blocks = blocks = ab;
            // Walk grid: any size is correct (the kernel strides over the list); sized from the previous pass's list
            // length, which only shrinks while an alignment converges.  Unknown or long lists get the full grid.
            if (!first && c->last_list_len >= 0 && c->last_list_len <= 50000) {
                walk_blocks = (uint32_t)(2 * c->last_list_len);
                if (walk_blocks < 256u) walk_blocks = 256u;
                if (walk_blocks > 8192u) walk_blocks = 8192u;
            }
            if (!first && c->last_list_len == -2) walk_blocks = 16384u;
            if (c->sw.walk_full_grid) walk_blocks = 0;
            // Once an alignment has converged the work list stays empty (every pair is certified or settled by the cell
            // scan), and an empty walk launch still costs ~6 us of a ~50 us pass.  So after a pass with an empty list the
            // walk is skipped; the final reduce reports the list's length, and in the rare case that it is not empty
            // after all the pass is repaired below (walk, accumulate and reduce again).
            optimistic = c->sw.optimistic >= 0 ? c->sw.optimistic == 1 : (!first && c->last_list_len == 0);      // (SYMMICP_OPTIMISTIC: "0" never, "1" always)
            if (writeback) optimistic = false;      // in-place write-back: a repair would transform the cloud twice
            // per-kernel events only in timing mode 2; mode 1 brackets the pass (events 0 and 4)
            if (first && c->target_surface_like) launch_pass_tree_first(a, c->ix, c->wl, ab, c->stream, c->timing == 2 ? ev : nullptr);
            else
            {
                // Sparse scans (the previous pass searched under a tenth of the pairs): the streaming kernel compacts the
                // failures of several tiles into full scan rounds; blocks enough to fill the chip at 5 waves per SIMD
                const int cp_blocks = c->sw.compact_blocks;
                const bool compact = a.use_slack && (c->sw.compact >= 0 ? c->sw.compact == 1 : (c->last_uncertified >= 0 && c->last_uncertified < (long long)(c->n_s_total / 10)));
                launch_pass_tree_split(a, c->ix, c->wl, ab, walk_blocks, optimistic ? 1 : 0, compact ? cp_blocks : 0, c->sw.tune, c->stream, c->timing == 2 ? ev : nullptr);
            }
            if (ev && c->timing == 2) c->ev_split[c->ev_used] = 2;
        }
        break;
    }
    if (ev && c->ev_split[c->ev_used] != 2) hipEventRecord(ev[4], c->stream);
    c->t_launch += now_s() - t_l0;
    // Final reduce (+ all-reduce over ranks), then wait for the record.  It arrives in host-mapped memory followed by its
    // sequence number: spin on that word instead of paying a stream-synchronise wake-up per iteration.  A stuck stream
    // (kernel fault) is caught by the fallback.
    auto reduce_and_wait = [&](int keep_nonempty) -> int {
        const unsigned long long seq = ++c->seq;
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(c->h_sums + kNSum);
        launch_final_reduce(c->partials, blocks, c->d_sums, c->comm ? nullptr : c->h_sums_dev, c->ticket, seq,
                            c->cfg.corr == SYMMICP_CORR_TREE ? c->wl_count : nullptr, keep_nonempty, c->stream);
        if (c->comm) {
            if (ev) { hipEventRecord(ev[6], c->stream); c->ev_coll[c->ev_used] = 1; }
            int r = g_rccl.AllReduce(c->d_sums, c->d_sums, kNSum, kNcclFloat64, kNcclSum, c->comm, c->stream);
            if (r != 0) return fail(c, SYMMICP_ERR_COMM, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
            if (ev) hipEventRecord(ev[7], c->stream);
            launch_publish(c->d_sums, c->h_sums_dev, seq, c->stream);
        }
        // the final-reduce time is only separated out in per-kernel mode; otherwise event 5 is event 4 again
        if (ev && c->ev_split[c->ev_used] != 1) hipEventRecord(ev[5], c->stream);
        const double t_spin = now_s();
        unsigned spins = 0;
        bool got = false;
        while (!(got = (*flag == seq))) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFu) == 0) {
                if (hipStreamQuery(c->stream) != hipErrorNotReady) { got = (*flag == seq); break; }
                if (now_s() - t_spin > 30.0) break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        c->t_spin += now_s() - t_spin;
        if (!got) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipGetLastError());
            if (*flag != seq) return fail(c, SYMMICP_ERR_HIP, "pass finished without publishing its record");
        }
        return SYMMICP_OK;
    };
    if (int st = reduce_and_wait(optimistic ? 1 : 0)) return st;
    if (c->shm.slots) { if (int st = shm_exchange(c, c->h_sums)) return st; }
    // length of the work list (summed over ranks by the exchange, so every rank takes the same decision)
    if (c->cfg.corr == SYMMICP_CORR_TREE && c->h_sums[kNSum - 3] != 0.0)
        return fail(c, SYMMICP_ERR_HIP, "a work-list shard overflowed: appends were dropped (internal capacity error)");
    long long list_len = (c->cfg.corr == SYMMICP_CORR_TREE) ? (long long)c->h_sums[kNSum - 1] : -1;
    c->last_uncertified = (c->cfg.corr == SYMMICP_CORR_TREE && !first) ? (long long)c->h_sums[kNSum - 2] : -1;
    if (optimistic && list_len > 0) {
        // the walk was skipped but some queries needed it: their pairs are provisional, so are the sums
        c->st.kernel_launches[7]++;
        launch_pass_tree_split(a, c->ix, c->wl, blocks, list_len <= 50000 ? 8192u : 0u, 2, 0, c->sw.tune, c->stream, nullptr);
        if (ev && c->ev_split[c->ev_used] != 2) hipEventRecord(ev[4], c->stream);
        if (int st = reduce_and_wait(0)) return st;
        if (c->shm.slots) { if (int st = shm_exchange(c, c->h_sums)) return st; }
    }
    if (ev) c->ev_used++;
    c->t_last_done = now_s(); c->n_pass_timed++;
    if (c->dbg_trace && first) dump_packet_trace(c);
    if (c->ix.dbg) print_pass_counters(c, first, list_len);
    std::memcpy(c->last.s, c->h_sums, sizeof(double) * kNSum);
    if (c->cfg.corr == SYMMICP_CORR_TREE) c->last.s[kNSum - 1] = c->last.s[kNSum - 2] = c->last.s[kNSum - 3] = 0.0;      // those slots carried the list length and the number of searched pairs, not sums
    c->last_list_len = list_len;
    // the packet first pass has no work list; the pass after it (the cloud has just moved by its whole misalignment) always has one:
    // -2 = "expect a list" (no optimistic skip of the walk, which would only be repaired; a mid-sized walk grid)
    if (first && c->cfg.corr == SYMMICP_CORR_TREE && c->target_surface_like) c->last_list_len = -2;
    c->st.passes++;
    return SYMMICP_OK;
}

// ---- device-driven runs of passes -------------------------------------------------------------------
// Once an alignment has converged a pass is ~30 us of kernels, and the host's share of an iteration (read the record back,
// solve, launch) is as long as a kernel.  So symmicp_align hands runs of passes to the device: the solve of func.cpp:76-102
// runs at the end of the reduce (k_reduce_solve, same source as the host solve: solve_core.h), the next pass reads its
// transform from device memory, and the loop test of myicp.cpp:123 sets a stop flag every later kernel of the batch
// honours.  The host solve stays the reference: anything but a clean, well-conditioned solve, and any pass that needs
// the tree walk, stops the batch and the host loop takes that iteration (LOOP_HOST_SOLVE / LOOP_REDO_PASS).
// work lists up to this length are walked inside a device-driven loop (straggler stage); longer ones go back to the host, which sizes the walk
static constexpr uint32_t kLoopListLimit = 8192;
// Scans are what the fused pass is slow at (two waves per SIMD: a dense cell's dependent loads are not hidden; ~6 ns per scanned pair
// against 0.3 in k_search_cells), so a run starts once the last pass searched under 0.4 % of the pairs and under 8192 of them -- the
// separate kernels win above that whatever the cloud's size (8M scan pair: 328 us fused against 150 + 80 at 30 k scans) -- and is left
// (LOOP_SLOW) at four times as many.
static uint32_t loop_scan_limit(uint32_t n) { const uint32_t f = n / 256; return f < 8192u ? f : 8192u; }
static bool batch_eligible(const symmicp_ctx *c)
{
    if (c->sw.host_loop) return false;                                    // SYMMICP_HOST_LOOP=1: never batch (A/B runs, tests)
    if (c->cfg.host_loop) return false;
    if (c->external_exchange || c->shm.slots) return false;               // those exchanges run on the host
    if (c->timing == 2 || c->ix.dbg) return false;                        // per-kernel tables and debug counters: host loop
    if (c->cfg.mode == SYMMICP_MODE_P2P) return false;                    // (3x3 SVD by Jacobi sweeps: host)
    if (c->n_loc == 0 && !c->comm) return false;                          // (an empty share of an RCCL run takes part: every input below is global,
                                                                          // and a rank that stayed in the host loop would issue a different number of all-reduces)
    if (c->host_passes_since_bailout < 2) return false;
    const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
    if (c->cfg.corr == SYMMICP_CORR_IDENTITY) return true;
    // TREE: the fused pass is for converged alignments: under 0.4 % of the pairs searched again, a short work list at most (stragglers
    // outside the overlap: real scans always have some)
    if (c->cfg.corr == SYMMICP_CORR_TREE)
        return !incr && c->last_list_len >= 0 && c->last_list_len <= (c->sw.no_loop_stragglers ? 0ll : (long long)kLoopListLimit) && c->last_uncertified >= 0 &&
               c->last_uncertified <= (long long)(loop_scan_limit(c->n_s_total) / (c->last_list_len > 0 ? 2 : 1)) && c->cert && !c->sw.no_cert;      // (the straggler stage costs two launches per pass)
    return false;
}

// Runs up to `want` iterations on the device.  On return c->iters, c->X and c->last describe the last COMPLETE pass, exactly as
// if symmicp_step had been called (c->iters - iters_before) times; diffs_before[k] = the diff the reference prints before
// iteration iters_before + 1 + k.  *small_step: the increment rule ended the alignment.
static constexpr int kListBlocks = 8;      // partial columns of the straggler stage
static constexpr int kEvSampleStride = 4;  // timing mode 3: passes of a device-driven run that carry events
static int run_batch(symmicp_ctx *c, int want, float *diffs_before, int *n_done, bool *small_step)
{
    *n_done = 0;
    *small_step = false;
    if (want > symmicp_ctx::kRing - 1) want = symmicp_ctx::kRing - 1;
    if (want <= 0) return SYMMICP_OK;
    const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
    const bool tree = c->cfg.corr == SYMMICP_CORR_TREE;
    LoopConfig lc{};
    lc.mode = c->cfg.mode; lc.fixed_iters = c->cfg.fixed_iters; lc.max_iters = c->iters + want; lc.incremental = incr ? 1 : 0; lc.tree = tree ? 1 : 0;
    lc.diff_threshold = c->cfg.diff_threshold; lc.eps_rotation = c->cfg.eps_rotation; lc.eps_translation = c->cfg.eps_translation;
    lc.nrm_w = (c->cfg.mode == SYMMICP_MODE_QUIRKS) ? 1.0f : 0.0f;
    for (int k = 0; k < 3; k++) lc.pivot[k] = c->pivot[k];
    lc.uncertified_limit = 4 * loop_scan_limit(c->n_s_total);
    // stragglers: while passes leave a work list, every fused pass is followed by the walk over the list and the accumulation of its
    // pairs (two more launches per pass, ~10 us); decided per chunk of passes from the lists the previous chunk left.  Without the stage
    // a list that turns up stops the loop: the host redoes that pass and the next batch starts with the stage.
    const bool no_stage = c->sw.no_loop_stragglers;      // A/B runs
    bool stragglers = tree && c->last_list_len > 0 && !no_stage;
    lc.list_limit = kLoopListLimit;
    if (lc.max_iters > c->cfg.max_iters) lc.max_iters = c->cfg.max_iters;
    const int it0 = c->iters;
    float X0[16];
    std::memcpy(X0, c->X, sizeof(X0));
    // loop state as the host loop left it
    LoopState ls{};
    std::memcpy(ls.X, c->X, sizeof(ls.X));
    for (int k = 0; k < 12; k++) ls.Xapply.m[k] = c->X[k];
    ls.Xapply.nrm_w = lc.nrm_w;
    ls.iters = it0;
    *c->h_loop = ls;
    HIP_TRY(c, hipMemcpyAsync(c->d_loop, c->h_loop, sizeof(LoopState), hipMemcpyHostToDevice, c->stream));
    PassArgs a{};
    fill_pass_args(c, a, c->X, /*from_cur=*/incr, /*writeback=*/incr, /*first=*/false);
    a.loop = c->d_loop;
    int blocks;
    bool vec4 = false;
    if (tree) {
        const int fb = c->sw.fused_blocks;
        const int tiles = (int)((c->n_loc + kPassThreads - 1) / kPassThreads);
        blocks = tiles < fb ? tiles : fb;
        if (blocks < 1) blocks = 1;             // (an empty share still writes its zero record)
    } else {
        vec4 = (c->n_loc % 4 == 0) && (c->n_t % 4 == 0) && (c->src_off % 4 == 0);
        const int id_cap = c->sw.id_blocks;
        const int nb = (int)(((vec4 ? c->n_loc / 4 : c->n_loc) + kPassThreads - 1) / kPassThreads);
        blocks = nb < id_cap ? (nb > 0 ? nb : 1) : id_cap;
    }
    const int fused_blocks_full = blocks;
    c->pass_blocks = blocks;
    uint32_t *counters = tree ? c->wl_count : nullptr;
    // The record of the last complete pass is in d_sums: solve from it, then (pass, reduce, [all-reduce,] solve) per iteration.
    // Passes are enqueued in chunks of 4, 8, 16, ... with one look at the loop state between chunks: a loop that stops early
    // (convergence, a pass that has to be redone) leaves at most one chunk of no-op launches behind.
    const bool timed_run = c->timing == 1 || c->timing == 3;
    const int ev_stride = c->timing == 3 ? kEvSampleStride : 1;      // mode 3: every 4th pass of the run
    if (timed_run && c->ev_used + want > symmicp_ctx::kEvRing) flush_events(c);
    // (a run that only a bail-out can stop -- fixed iteration count, no increment criterion, no work list at entry -- is enqueued in one
    // piece: every chunk boundary is a host look at the loop state, ~12 us of idle GPU)
    const bool unstoppable = lc.fixed_iters && !(lc.eps_rotation > 0.f && lc.eps_translation > 0.f) && !stragglers;
    int enq = 0, chunk = unstoppable ? want : 4, n_stage = 0;
    bool first_chunk = true;
    while (enq < want) {
        const int nq = (want - enq < chunk) ? want - enq : chunk;
        const int it_before = first_chunk ? it0 : c->h_loop->iters;
        const int ev_used0 = c->ev_used;
        // this chunk's variant: partial columns, what a non-empty list means to the solve
        // (with the stage: 512 partial columns in all, k_reduce_solve's fast path)
        const int fused_blocks = (stragglers && fused_blocks_full + kListBlocks > 512) ? 512 - kListBlocks : fused_blocks_full;
        blocks = fused_blocks + (stragglers ? kListBlocks : 0);
        a.partial_cols = stragglers ? (uint32_t)blocks : 0u;
        a.partial_col0 = (uint32_t)fused_blocks;
        lc.walk_in_loop = stragglers ? 1 : 0;
        if (first_chunk) launch_reduce_solve(c->partials, blocks, c->d_sums, 2, c->d_loop, lc, c->h_ring_dev, symmicp_ctx::kRing, counters, c->stream);
        for (int p = 0; p < nq; p++) {
            hipEvent_t *ev = nullptr;
            if (timed_run && (enq + p) % ev_stride == 0 && c->ev_used < symmicp_ctx::kEvRing) {
                ev = c->ev + c->ev_used * symmicp_ctx::kEvPer;
                c->ev_split[c->ev_used] = 1;
                c->ev_weight[c->ev_used] = 1;
                hipEventRecord(ev[0], c->stream);
            }
            if (tree) {
                launch_pass_fused(a, c->ix, c->wl, fused_blocks, c->stream);
                if (stragglers) launch_loop_stragglers(a, c->ix, c->wl, kListBlocks, c->sw.tune, c->stream);
            }
            else launch_pass_identity(a, c->tgt, blocks, vec4, c->stream);
            if (ev) { hipEventRecord(ev[4], c->stream); c->ev_used++; }
            if (c->comm || blocks > 512) {
                // sharded: the record is summed over the ranks before the solve; many partial records (an 8M-point identity pass):
                // the 40-block reduce is faster than one block's
                launch_final_reduce(c->partials, blocks, c->d_sums, nullptr, c->ticket, 0ull, counters, 0, c->stream);
                if (c->comm) {
                    if (ev) { hipEventRecord(ev[6], c->stream); c->ev_coll[c->ev_used - 1] = 1; }
                    int r = g_rccl.AllReduce(c->d_sums, c->d_sums, kNSum, kNcclFloat64, kNcclSum, c->comm, c->stream);
                    if (r != 0) {
                        // passes are already queued behind this point and (incremental mode) write into `cur`: drain the stream and make the
                        // context ask for a fresh symmicp_begin -- its loop state no longer describes device memory
                        (void)hipStreamSynchronize(c->stream);
                        c->begun = false;
                        c->ev_used = 0;
                        return fail(c, SYMMICP_ERR_COMM, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
                    }
                    if (ev) hipEventRecord(ev[7], c->stream);
                }
                launch_reduce_solve(c->partials, blocks, c->d_sums, 1, c->d_loop, lc, c->h_ring_dev, symmicp_ctx::kRing, counters, c->stream);
            } else {
                launch_reduce_solve(c->partials, blocks, c->d_sums, 0, c->d_loop, lc, c->h_ring_dev, symmicp_ctx::kRing, counters, c->stream);
            }
        }
        const unsigned long long seq = ++c->batch_seq;
        launch_loop_end(c->d_loop, c->h_loop_dev, c->h_done_dev, seq, c->stream);
        volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(c->h_done);
        const double t_spin = now_s();
        unsigned spins = 0;
        bool got = false;
        while (!(got = (*flag == seq))) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFu) == 0) {
                if (hipStreamQuery(c->stream) != hipErrorNotReady) { got = (*flag == seq); break; }
                if (now_s() - t_spin > 60.0) break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (!got) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipGetLastError());
            if (*flag != seq) return fail(c, SYMMICP_ERR_HIP, "batch finished without publishing its end flag");
        }
        enq += nq;
        first_chunk = false;
        // passes of this chunk that did run (a pass that has to be redone ran too); the events of the no-op launches behind a
        // stop are dropped
        const int ran = (c->h_loop->iters - it_before) + (c->h_loop->reason == LOOP_REDO_PASS ? 1 : 0);
        if (timed_run && ran >= 0) {
            int sampled_ran = 0;                      // event pairs of the passes that ran; each stands for the passes up to the next pair
            for (int p = 0; p < ran && p < nq; p++) {
                if ((enq - nq + p) % ev_stride != 0) continue;
                if (ev_used0 + sampled_ran < symmicp_ctx::kEvRing) {
                    const int left = (ran < nq ? ran : nq) - p;
                    c->ev_weight[ev_used0 + sampled_ran] = left < ev_stride ? left : ev_stride;
                }
                sampled_ran++;
            }
            if (ev_used0 + sampled_ran < c->ev_used) c->ev_used = ev_used0 + sampled_ran;
        }
        if (stragglers) n_stage += c->h_loop->iters - it_before;
        if (c->h_loop->stop) break;
        if (tree && !no_stage) {
            // the stage stays (or comes) on while any pass of this chunk left a list
            stragglers = false;
            for (int k = it_before + 1; k <= c->h_loop->iters; k++) stragglers = stragglers || c->h_ring[k % symmicp_ctx::kRing].list_len > 0;
        }
        // with the stage on, chunks stay short: the stage (two more launches per pass) can only be dropped between chunks, and lists die out
        // within a few passes of a run's start (8M scan pair: 8 8 6 2 | 2 0 1 0 | 0 ...); once it is off and only a bail-out can stop the run,
        // the rest goes out in one piece
        if (stragglers) chunk = 4;
        else if (lc.fixed_iters && !(lc.eps_rotation > 0.f && lc.eps_translation > 0.f)) chunk = want;
        else chunk *= 2;
    }
    const LoopState &hl = *c->h_loop;
    const int it1 = hl.iters;                         // passes complete
    if (it1 < it0 || it1 > it0 + want) return fail(c, SYMMICP_ERR_HIP, "device loop state out of range");
    // the diffs the reference prints before iterations it0+1 .. it1 (+1 when the loop went on into a pass that was not completed)
    for (int k = it0; k < it1; k++) diffs_before[k - it0] = (k == it0) ? (float)c->last.s[33] : (float)c->h_ring[k % symmicp_ctx::kRing].sums[33];
    if (it1 > it0) {
        std::memcpy(c->last.s, c->h_ring[it1 % symmicp_ctx::kRing].sums, sizeof(double) * kNSum);
        std::memcpy(c->X, c->h_ring[(it1 - 1) % symmicp_ctx::kRing].X, sizeof(float) * 16);      // transform the last complete pass applied
        c->st.passes += it1 - it0;
        c->st.loop_passes += it1 - it0;
        c->st.loop_straggler_passes += n_stage;
    }
    (void)X0;
    c->iters = it1;
    c->last_list_len = tree ? (it1 > it0 ? (long long)c->h_ring[it1 % symmicp_ctx::kRing].list_len : c->last_list_len) : -1;
    *n_done = it1 - it0;
    *small_step = hl.reason == LOOP_DONE && hl.small_step != 0;
    if (tree && it1 > it0) c->last_uncertified = c->h_ring[it1 % symmicp_ctx::kRing].pad;
    if (c->sw.debug_host && it1 - it0 >= 2) {      // (a library built with -DRS_STAMPS leaves k_reduce_solve's phase durations in the spare slots of each record)
        const double *q = c->h_ring[(it1 - 1) % symmicp_ctx::kRing].sums;
        if (q[37] != 0.0) std::fprintf(stderr, "[symmicp host] k_reduce_solve stamps of pass %d: load + reduce %.2f us (loop state there after %.2f us), bookkeeping + solve %.2f us, publish %.2f us\n", it1 - 1, std::floor(q[37]) * 0.01, (q[37] - std::floor(q[37])) * 1e4 * 0.01, q[38] * 0.01, q[39] * 0.01);
    }
    if (c->sw.debug_host && tree) {
        std::fprintf(stderr, "[symmicp host] batch: work list / searched pairs after each device pass:");
        for (int k = it0 + 1; k <= it1; k++) std::fprintf(stderr, " %d/%d", c->h_ring[k % symmicp_ctx::kRing].list_len, c->h_ring[k % symmicp_ctx::kRing].pad);
        std::fprintf(stderr, "\n");
    }
    if (c->sw.debug_host) std::fprintf(stderr, "[symmicp host] batch: %d of %d passes on the device (%d enqueued), reason %d, %d with the straggler stage, list after %lld\n", it1 - it0, want, enq, hl.reason, n_stage, (long long)c->last_list_len);
    if (hl.reason == LOOP_SLOW) c->host_passes_since_bailout = 1;      // one host pass, then look again
    if (hl.reason == LOOP_REDO_PASS || hl.reason == LOOP_HOST_SOLVE) {
        c->host_passes_since_bailout = 0;
        c->last_list_len = -1;                        // the host's next pass runs the full search
        c->st.kernel_launches[7]++;                   // counted with the repairs
    }
    return SYMMICP_OK;
}

static int check_ready(symmicp_ctx *c)
{
    if (!c->tgt_block || !c->src0_block) return fail(c, SYMMICP_ERR_STATE, "source and target must be set first (myicp.cpp:102)");
    if (c->cfg.corr == SYMMICP_CORR_IDENTITY && c->n_s_total != c->n_t)
        return fail(c, SYMMICP_ERR_SIZE, "identity pairing needs N_s == N_t (func.cpp:21)");
    if (c->cfg.corr == SYMMICP_CORR_TREE && !c->have_index) return fail(c, SYMMICP_ERR_STATE, "target index missing");
    if (c->cfg.corr != SYMMICP_CORR_IDENTITY && !c->tq) return fail(c, SYMMICP_ERR_STATE, "target was set under a different corr mode");
    return SYMMICP_OK;
}

static void fill_iter(symmicp_ctx *c, symmicp_iter_result *out, int status, float rcond, const float *incr)
{
    if (!out) return;
    out->status = status;
    out->iter = c->iters;
    out->diff = (float)c->last.s[33];
    out->rcond = rcond;
    out->pairs = c->last.s[34];
    if (incr) std::memcpy(out->increment, incr, sizeof(float) * 16); else identity16(out->increment);
    out->sums = c->last;
}

int symmicp_begin(symmicp_ctx *c, const float *guess16, symmicp_iter_result *out)
{
    if (!c) return SYMMICP_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int st = check_ready(c);
    if (st != SYMMICP_OK) return st;
    if (guess16) std::memcpy(c->X, guess16, sizeof(float) * 16); else identity16(c->X);
    c->iters = 0;
    const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
    // incremental: cur <- X * src0 (write-back); cumulative: read src0 through X every pass
    st = run_pass(c, c->X, /*from_cur=*/false, /*writeback=*/incr, /*first=*/true);
    if (st != SYMMICP_OK) return st;
    c->begun = true;
    c->sums_exchanged = false;
    fill_iter(c, out, SYMMICP_OK, 1.0f, nullptr);
    return SYMMICP_OK;
}

int symmicp_step(symmicp_ctx *c, symmicp_iter_result *out)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!c->begun) return fail(c, SYMMICP_ERR_STATE, "symmicp_step before symmicp_begin");
    if (c->external_exchange && !c->sums_exchanged)
        return fail(c, SYMMICP_ERR_STATE, "external exchange: symmicp_set_sums(total over ranks) must follow every pass");
    c->sums_exchanged = false;
    HIP_TRY(c, hipSetDevice(c->device));
    float pbar[3], qbar[3], a[3], t[3], rc = 0.f, Xi[16];
    int st = (c->cfg.mode == SYMMICP_MODE_QUIRKS) ? solve_quirks(c->last, pbar, qbar, a, t, &rc, Xi)
             : (c->cfg.mode == SYMMICP_MODE_PAPER) ? solve_paper(c->last, c->pivot, pbar, qbar, a, t, &rc, Xi)
                                                   : solve_p2p(c->last, c->pivot, &rc, Xi);
    if (st != SYMMICP_OK) {
        c->err = "degenerate system (rank-deficient normal equations or non-finite transform; func.cpp:70,96)";
        fill_iter(c, out, st, rc, nullptr);
        return st;
    }
    mat4_mul(Xi, c->X, c->X);                     // myicp.cpp:138
    const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
    st = incr ? run_pass(c, Xi, true, true, false) : run_pass(c, c->X, false, false, false);
    if (st != SYMMICP_OK) return st;
    c->iters++;
    fill_iter(c, out, SYMMICP_OK, rc, Xi);
    return SYMMICP_OK;
}

// ---- the reference's result block (myicp.cpp:146-149) --------------------------------------------------------------
// `std::cout << matrix` with Eigen's default IOFormat: every coefficient through the stream's default float formatting (precision 6:
// what "%g" prints), right-aligned to the widest coefficient OF THAT MATRIX, columns separated by one space, rows by a newline
// (Eigen/src/Core/IO.h, print_matrix; published behaviour -- Eigen is not under /root/reference).
static size_t append_eigen(std::string &out, const float *m, int rows, int cols, int stride)
{
    char cell[16][32];
    size_t width = 0;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) {
            std::snprintf(cell[r * cols + c], sizeof(cell[0]), "%g", (double)m[r * stride + c]);
            width = std::max(width, std::strlen(cell[r * cols + c]));
        }
    for (int r = 0; r < rows; r++) {
        for (int c = 0; c < cols; c++) {
            if (c) out += ' ';
            const size_t len = std::strlen(cell[r * cols + c]);
            out.append(width - len, ' ');
            out += cell[r * cols + c];
        }
        out += '\n';                 // (rows are separated by "\n"; the reference ends each matrix with std::endl)
    }
    return width;
}

// Affine3f::rotation() is the orthogonal polar factor of the linear part (Eigen computes it as U V^T of a JacobiSVD in fp32): here by
// Newton's iteration R <- (R + R^-T) / 2 in fp64, which converges to the same matrix; for the rigid transforms of this path it differs
// from the linear part in the last bit or two.  A singular linear part is printed as it is.
static void polar_rotation(const float X[16], float R[9])
{
    double A[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = X[4 * r + c];
    for (int it = 0; it < 100; it++) {
        const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) break;
        double inv_t[3][3];          // A^-T = cofactor matrix / det
        inv_t[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; inv_t[0][1] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; inv_t[0][2] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
        inv_t[1][0] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; inv_t[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; inv_t[1][2] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        inv_t[2][0] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det; inv_t[2][1] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det; inv_t[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
        double change = 0.0;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) { const double v = 0.5 * (A[r][c] + inv_t[r][c]); change = std::max(change, std::fabs(v - A[r][c])); A[r][c] = v; }
        if (change < 1e-15) break;
    }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = std::isfinite(A[r][c]) ? (float)A[r][c] : X[4 * r + c];
}

size_t symmicp_format_result(const float transform16[16], char *buf, size_t cap)
{
    if (!transform16) return 0;
    std::string out = "Result transform:\n";
    append_eigen(out, transform16, 4, 4, 4);
    out += "  rotation:\n";
    float R[9];
    polar_rotation(transform16, R);
    append_eigen(out, R, 3, 3, 3);
    out += "  translation:\n";
    append_eigen(out, transform16 + 3, 3, 1, 4);
    if (buf && cap) {
        const size_t n = std::min(out.size(), cap - 1);
        std::memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return out.size();
}

int symmicp_align(symmicp_ctx *c, const float *guess16, symmicp_result *out)
{
    if (!c || !out) return SYMMICP_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    if (c->external_exchange) { out->status = SYMMICP_ERR_STATE; return fail(c, SYMMICP_ERR_STATE, "symmicp_align is not available with external exchange: drive begin/set_sums/step"); }
    const double t0 = now_s();
    symmicp_iter_result it;
    int st = symmicp_begin(c, guess16, &it);
    if (st != SYMMICP_OK) { out->status = st; return st; }
    float diff = it.diff;                                           // myicp.cpp:122
    out->diff_initial = diff;
    int iters = 0;
    // myicp.cpp:123: while (diff > diff_threshold && iters++ < max_iters)
    while ((c->cfg.fixed_iters || diff > c->cfg.diff_threshold) && iters < c->cfg.max_iters) {
        if (batch_eligible(c)) {
            // the same loop, run by the device for the iterations that are left (run_batch); afterwards this loop goes on from the
            // last complete pass -- and ends by its own test where the device loop ended by the same test
            float db[symmicp_ctx::kRing];
            int done = 0;
            bool small = false;
            st = run_batch(c, c->cfg.max_iters - iters, db, &done, &small);
            if (st != SYMMICP_OK) break;
            for (int k = 0; k < done; k++) {
                iters++;
                if (c->cfg.verbose) std::printf("iters#%d\ndiff: %g\n", iters, db[k]);
                if (iters <= 64) out->diffs[iters - 1] = db[k];
            }
            diff = (float)c->last.s[33];
            if (small) break;
            if (done > 0 || c->host_passes_since_bailout == 0) continue;
            // (the device did nothing and asked for nothing: take one step here)
        }
        iters++;
        if (c->cfg.verbose) std::printf("iters#%d\ndiff: %g\n", iters, diff);                       // myicp.cpp:125-126
        if (iters <= 64) out->diffs[iters - 1] = diff;
        st = symmicp_step(c, &it);
        if (st != SYMMICP_OK) { iters--; break; }
        c->host_passes_since_bailout++;
        diff = it.diff;                                             // myicp.cpp:141
        if (c->cfg.eps_rotation > 0.f && c->cfg.eps_translation > 0.f && !c->cfg.fixed_iters) {
            // convergence on the increment (the reference only has the diff threshold, myicp.cpp:123)
            const float *Xi = it.increment;
            const double tr = ((double)Xi[0] + Xi[5] + Xi[10] - 1.0) * 0.5;
            const double ang = std::acos(tr > 1.0 ? 1.0 : (tr < -1.0 ? -1.0 : tr));
            const double tn = std::sqrt((double)Xi[3] * Xi[3] + (double)Xi[7] * Xi[7] + (double)Xi[11] * Xi[11]);
            if (ang < c->cfg.eps_rotation && tn < c->cfg.eps_translation) break;
        }
    }
    out->status = st;
    out->iters = iters;
    out->diff_final = diff;
    std::memcpy(out->transform, c->X, sizeof(float) * 16);
    out->seconds_total = now_s() - t0;
    if (c->cfg.verbose) {                                           // myicp.cpp:146-149
        char buf[1024];
        symmicp_format_result(c->X, buf, sizeof(buf));
        std::fputs(buf, stdout);
    }
    return st;
}

int symmicp_solve(int mode, const symmicp_sums *sums, const float pivot[3], float pbar[3], float qbar[3], float a[3],
                  float t[3], float *rcond, float out16[16])
{
    if (!sums || !pbar || !qbar || !a || !t || !out16) return SYMMICP_ERR_ARG;
    if (mode == SYMMICP_MODE_QUIRKS) return solve_quirks(*sums, pbar, qbar, a, t, rcond, out16);
    if (mode == SYMMICP_MODE_PAPER) return solve_paper(*sums, pivot, pbar, qbar, a, t, rcond, out16);
    if (mode == SYMMICP_MODE_P2P) {
        for (int k = 0; k < 3; k++) pbar[k] = qbar[k] = a[k] = t[k] = 0.f;
        return solve_p2p(*sums, pivot, rcond, out16);
    }
    return SYMMICP_ERR_ARG;
}

}  // extern "C"
