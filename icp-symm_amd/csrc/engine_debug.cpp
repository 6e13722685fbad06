// engine_debug.cpp -- diagnostics of libsymmicp, off unless asked for: SYMMICP_DEBUG_COUNTERS=1 (per-pass counters of the search kernels on
// stderr) and SYMMICP_DEBUG_TRACE=file (per-packet trace of the first pass, read by scratch/pkt_timeline.py and friends).
#include "engine_internal.h"

// SYMMICP_DEBUG_TRACE=file: the per-packet trace of the first pass that has just run (k_search_packet), then cleared for the next one
static constexpr size_t kTraceWords = (size_t)1 << 22;
void dump_packet_trace(symmicp_ctx *c)
{
    if (FILE *f = std::fopen(c->sw.debug_trace.c_str(), "wb")) {
        std::vector<unsigned long long> t(kTraceWords);
        hipMemcpy(t.data(), c->dbg_trace, t.size() * 8, hipMemcpyDeviceToHost);
        std::fwrite(t.data(), 8, t.size(), f);
        std::fclose(f);
    }
    hipMemset(c->dbg_trace, 0, kTraceWords * 8);
}


// the device-side counters of the pass that has just run (k_search_packet / cells_tile / k_search_walk), then cleared
void print_pass_counters(symmicp_ctx *c, bool first, long long list_len)
{
        unsigned long long h[12];
        hipMemcpy(h, c->ix.dbg, sizeof(h), hipMemcpyDeviceToHost);
        hipMemset(c->ix.dbg, 0, sizeof(h));
        if (first && c->cfg.corr == SYMMICP_CORR_TREE && c->target_surface_like)
            std::fprintf(stderr, "[symmicp dbg] pass %lld (packets): steps=%llu nodes+leaf candidates=%llu leaf candidates rejected=%llu points=%llu tie rescans=%llu overflows=%llu lanes wanting a scanned leaf (sum)=%llu | packet ticks (10 ns): sum=%llu max=%llu\n",
                         (long long)c->st.passes, h[0], h[4], h[5], h[3], h[2], h[1], h[8], h[7], h[6]);
        else
        std::fprintf(stderr, "[symmicp dbg] pass %lld: cells: certified=%llu (by neighbourhood %llu) scans=%llu (neighbourhoods kept %llu, not kept %llu) probes=%llu to-walk=%llu items=%llu points=%llu | walk list=%lld visits=%llu wave-max*64=%llu\n",
                     (long long)c->st.passes, h[1], h[8], h[2], h[9], h[10], h[6], h[7], h[0], h[3], list_len, h[4], h[5]);
}
