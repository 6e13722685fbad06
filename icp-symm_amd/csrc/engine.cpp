// engine.cpp -- host side of libsymmicp: the C-ABI of include/symmicp.h.
//
// Host C++ owns control flow and the 6x6 / 3x3 solve (host_solve.cpp); the GPU owns
// every O(N) pass (kernels_pass.hip) and the one-time index build (kernels_build.hip).
// Per iteration of the reference loop (ICP/myicp.cpp:123-142) the host does:
//   solve(last sums) -> 4x4 increment -> launch the pass (one streaming kernel for identity pairing; search + accumulate
//   kernels for the nearest-neighbour modes) + the final reduce -> [RCCL all-reduce of 40 doubles when sharded] ->
//   spin on the record's sequence word in host-mapped memory -> repeat.
// There is no CPU fallback: without a HIP device every entry point fails loudly.
#include "engine_internal.h"

void read_switches(Switches &w)
{
    auto flag = [](const char *n) { return std::getenv(n) != nullptr; };
    auto num = [](const char *n, long def) { const char *e = std::getenv(n); return e ? std::atol(e) : def; };
    auto tri = [](const char *n) { const char *e = std::getenv(n); return e ? (e[0] == '1' ? 1 : 0) : -1; };
    w.allow_any_arch = flag("SYMMICP_ALLOW_ANY_ARCH");
    w.debug_host = flag("SYMMICP_DEBUG_HOST");
    w.debug_counters = flag("SYMMICP_DEBUG_COUNTERS");
    if (const char *e = std::getenv("SYMMICP_DEBUG_TRACE")) w.debug_trace = e;
    if (const char *e = std::getenv("SYMMICP_GRID_PPC")) w.grid_ppc = std::atof(e);
    w.grid_maxlevel = (int)num("SYMMICP_GRID_MAXLEVEL", kMortonBits);
    w.grid_level = (int)num("SYMMICP_GRID_LEVEL", -1);
    if (const char *e = std::getenv("SYMMICP_FIRST_PASS")) w.first_pass = (e[0] == 'p') ? 1 : 0;      // "packet" / "walk"
    w.oct_leaf = (int)num("SYMMICP_OCT_LEAF", 0);
    if (const char *e = std::getenv("SYMMICP_PACKET_ORDER")) w.packet_order = e[0] != '0';
    if (const char *e = std::getenv("SYMMICP_PACKET_JUMP")) w.packet_jump = (float)std::atof(e);
    w.packet_key_bits = (int)num("SYMMICP_PACKET_KEY_BITS", 16);
    w.packet_chunk = (uint32_t)num("SYMMICP_PACKET_CHUNK", 0);
    w.packet_lds_pad = (uint32_t)num("SYMMICP_PACKET_LDS_PAD", 0);
    w.packet_waves = (uint32_t)num("SYMMICP_PACKET_WAVES", 0);
    w.packet_front_cap = (uint32_t)num("SYMMICP_PACKET_FRONT_CAP", 0);
    if (const char *e = std::getenv("SYMMICP_HOOD_FRAC")) w.hood_frac = std::atof(e);
    w.no_hood = flag("SYMMICP_NO_NEIGHBOURHOOD");
    w.no_cert = flag("SYMMICP_NO_CERT");
    w.walk_full_grid = flag("SYMMICP_WALK_FULL_GRID");
    { const char *e = std::getenv("SYMMICP_HOST_LOOP"); w.host_loop = e && e[0] == '1'; }
    w.no_loop_stragglers = flag("SYMMICP_NO_LOOP_STRAGGLERS");
    w.force_comm = flag("SYMMICP_FORCE_COMM");
    if (const char *e = std::getenv("SYMMICP_PACKET_COST_KEY")) w.packet_cost_key = e[0] != '0';
    w.budget_walk = tri("SYMMICP_BUDGET_WALK");
    w.optimistic = tri("SYMMICP_OPTIMISTIC");
    w.compact = tri("SYMMICP_COMPACT");
    w.pass_blocks = (int)num("SYMMICP_PASS_BLOCKS", 2048);
    w.id_blocks = (int)num("SYMMICP_ID_BLOCKS", 2048);
    w.acc_blocks = (int)num("SYMMICP_ACC_BLOCKS", 512);
    w.fused_blocks = (int)num("SYMMICP_FUSED_BLOCKS", 512);
    w.compact_blocks = (int)num("SYMMICP_COMPACT_BLOCKS", 1280);
    w.tune.wave_mode_max = (uint32_t)num("SYMMICP_WAVE_MODE_MAX", 20000);
    w.tune.cells_chunk = (uint32_t)num("SYMMICP_CELLS_CHUNK", 16);
    w.tune.cells_queries = (uint32_t)num("SYMMICP_CELLS_QUERIES", 0);
    w.tune.walk_budget = (uint32_t)num("SYMMICP_WALK_BUDGET", 160);
}

// rewind the arena and make sure it holds `want` bytes (contents are dead: called at the start of a public call)
void arena_begin(Arena &a, size_t want)
{
    a.off = 0;
    if (a.cap >= want) return;
    if (a.base) hipFree(a.base);
    a.base = nullptr; a.cap = 0;
    const size_t cap = want + want / 4;
    if (hipMalloc((void **)&a.base, cap) == hipSuccess) a.cap = cap;      // on failure every alloc_temp falls back to hipMalloc
    else (void)hipGetLastError();
}


extern "C" {

int symmicp_version(void) { return 100; }

void symmicp_config_default(symmicp_config *cfg)
{
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(*cfg);
    cfg->device = -1;
    cfg->mode = SYMMICP_MODE_QUIRKS;
    cfg->corr = SYMMICP_CORR_IDENTITY;
    cfg->apply = SYMMICP_APPLY_DEFAULT;
    cfg->max_iters = 10;            // myicp.cpp:6
    cfg->diff_threshold = 1.0f;     // myicp.cpp:6
    cfg->max_corr_dist = 0.f;
    cfg->fixed_iters = 0;
    cfg->sort_source = 1;
    cfg->verbose = 0;
    cfg->min_normal_dot = -2.0f;
    cfg->eps_rotation = 0.f;
    cfg->eps_translation = 0.f;
}

static int check_cfg(const symmicp_config *cfg)
{
    if (!cfg || cfg->struct_size != (int32_t)sizeof(symmicp_config)) return SYMMICP_ERR_ARG;
    if (cfg->mode < SYMMICP_MODE_QUIRKS || cfg->mode > SYMMICP_MODE_P2P) return SYMMICP_ERR_ARG;
    if (cfg->corr < SYMMICP_CORR_IDENTITY || cfg->corr > SYMMICP_CORR_TREE) return SYMMICP_ERR_ARG;
    if (cfg->apply < SYMMICP_APPLY_DEFAULT || cfg->apply > SYMMICP_APPLY_CUMULATIVE) return SYMMICP_ERR_ARG;
    if (cfg->max_iters < 0) return SYMMICP_ERR_ARG;
    return SYMMICP_OK;
}

int symmicp_create(const symmicp_config *cfg, symmicp_ctx **out)
{
    if (!out) return SYMMICP_ERR_ARG;
    *out = nullptr;
    if (check_cfg(cfg) != SYMMICP_OK) return SYMMICP_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SYMMICP_ERR_HIP;   // no CPU fallback
    symmicp_ctx *c = new symmicp_ctx();
    read_switches(c->sw);
    c->cfg = *cfg;
    if (cfg->device >= 0) {
        if (cfg->device >= ndev || hipSetDevice(cfg->device) != hipSuccess) { delete c; return SYMMICP_ERR_HIP; }
        c->device = cfg->device;
    } else if (hipGetDevice(&c->device) != hipSuccess) { delete c; return SYMMICP_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) { delete c; return SYMMICP_ERR_HIP; }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !c->sw.allow_any_arch) {
        // the code object only carries gfx950 ISA
        delete c;
        return SYMMICP_ERR_HIP;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SYMMICP_ERR_HIP; }
    bool ok = hipMalloc((void **)&c->partials, sizeof(double) * kNSum * 8192) == hipSuccess &&
              hipMalloc((void **)&c->d_sums, sizeof(double) * kNSum) == hipSuccess &&
              hipHostMalloc((void **)&c->h_sums, sizeof(double) * (kNSum + 8), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipMalloc((void **)&c->ticket, 2 * sizeof(uint32_t)) == hipSuccess && hipMemset(c->ticket, 0, 2 * sizeof(uint32_t)) == hipSuccess &&
              hipHostGetDevicePointer((void **)&c->h_sums_dev, c->h_sums, 0) == hipSuccess &&
              std::memset(c->h_sums, 0, sizeof(double) * (kNSum + 8)) != nullptr &&
              hipMalloc((void **)&c->d_loop, sizeof(LoopState)) == hipSuccess &&
              hipHostMalloc((void **)&c->h_loop, sizeof(LoopState), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **)&c->h_loop_dev, c->h_loop, 0) == hipSuccess &&
              hipHostMalloc((void **)&c->h_ring, sizeof(LoopRecord) * symmicp_ctx::kRing, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **)&c->h_ring_dev, c->h_ring, 0) == hipSuccess &&
              hipHostMalloc((void **)&c->h_done, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **)&c->h_done_dev, c->h_done, 0) == hipSuccess &&
              std::memset(c->h_done, 0, 64) != nullptr &&
              true;
    for (int k = 0; ok && k < symmicp_ctx::kEvRing * symmicp_ctx::kEvPer; k++) ok = hipEventCreateWithFlags(&c->ev[k], hipEventDisableSystemFence) == hipSuccess;      // timing only: no system-scope cache flush per record
    if (!ok) { symmicp_destroy(c); return SYMMICP_ERR_HIP; }
    c->pkt_fallbacks = c->ticket + 1;
    identity16(c->X);
    *out = c;
    return SYMMICP_OK;
}

extern "C++" hipError_t keep_alloc(symmicp_ctx *c, void **out, size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (!bytes) bytes = 256;
    if (c->keep.base && c->keep.off + bytes <= c->keep.cap) { *out = c->keep.base + c->keep.off; c->keep.off += bytes; return hipSuccess; }
    void *p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) { c->keep_extra.push_back(p); *out = p; }
    return e;
}

extern "C++" void free_target(symmicp_ctx *c)
{
    // (the arrays live in the keep-arena, which stays allocated for the next target)
    for (void *p : c->keep_extra) hipFree(p);
    c->keep_extra.clear();
    c->keep.off = 0;
    hipFree(c->dbg); hipFree(c->dbg_trace); c->dbg_trace = nullptr;
    c->ctop = nullptr; c->dbg = nullptr; c->ix = TargetIndex{}; c->tgt_block = nullptr; c->tq = nullptr; c->tn = nullptr; c->boxes = nullptr; c->cells = nullptr; c->onodes = nullptr;
    c->have_index = false; c->n_t = 0;
}

extern "C++" void forget_source(symmicp_ctx *c)
{
    // (the arrays live in one block, c->src_all, which is kept for the next source of the same or a smaller size)
    c->worklist = c->wl_count = nullptr; c->cert = nullptr; c->certk = nullptr; c->hoodr = nullptr; c->pkt_tab = nullptr; c->pkt_count = 0; c->pairrec = nullptr;
    c->src0_block = c->cur_block = nullptr; c->src_order = nullptr; c->pos = nullptr; c->d2 = nullptr; c->best64 = nullptr;
    c->n_loc = c->n_s_total = c->src_off = 0;
}

static void free_source(symmicp_ctx *c)
{
    forget_source(c);
    hipFree(c->src_all);
    c->src_all = nullptr; c->src_all_cap = 0;
}

void symmicp_destroy(symmicp_ctx *c)
{
    if (!c) return;
    if (c->sw.debug_host && c->n_pass_timed)
        std::fprintf(stderr, "[symmicp host] passes %ld: launch %.1f us, spin %.1f us, between passes %.1f us (per pass)\n", c->n_pass_timed, 1e6 * c->t_launch / c->n_pass_timed, 1e6 * c->t_spin / c->n_pass_timed, 1e6 * c->t_between / c->n_pass_timed);
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    shm_close(c);
    free_target(c);
    free_source(c);
    hipFree(c->partials); hipFree(c->d_sums); hipFree(c->ticket); hipFree(c->arena.base); hipFree(c->keep.base);
    if (c->h_sums) hipHostFree(c->h_sums);
    hipFree(c->d_loop);
    if (c->h_loop) hipHostFree(c->h_loop);
    if (c->h_ring) hipHostFree(c->h_ring);
    if (c->h_done) hipHostFree(c->h_done);
    for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char *symmicp_last_error(const symmicp_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int symmicp_set_config(symmicp_ctx *c, const symmicp_config *cfg)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (check_cfg(cfg) != SYMMICP_OK) return fail(c, SYMMICP_ERR_ARG, "bad config");
    // correspondence kind decides device layouts: it can only change before clouds are set
    if ((c->n_t || c->n_loc) && cfg->corr != c->cfg.corr) return fail(c, SYMMICP_ERR_STATE, "corr cannot change after clouds are set");
    if ((c->n_t || c->n_loc) && cfg->sort_source != c->cfg.sort_source) return fail(c, SYMMICP_ERR_STATE, "sort_source cannot change after clouds are set");
    int dev = c->cfg.device;
    c->cfg = *cfg;
    c->cfg.device = dev;
    c->begun = false;
    return SYMMICP_OK;
}

// host strided cloud -> device block of 6 planar arrays.  The usual layouts never touch a host staging loop: records with
// contiguous x y z (packed AoS, PointXYZ, PointNormal) are copied as they are and split into columns on the device;
// column-major matrices (Eigen) are copied column by column.  Anything else goes through a host transpose.
static int upload_planar(symmicp_ctx *c, const float *xyz, size_t xr, size_t xc, const float *nrm, size_t nr, size_t nc,
                         size_t n, DevBuf<float> &block, bool temp, double centroid[3])
{
    if (centroid) {
        // fp64, in row order (the oracle's order: the pivot has to come out bit-identical)
        double s[3] = {0, 0, 0};
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) s[k] += (double)xyz[i * xr + k * xc];
        for (int k = 0; k < 3; k++) centroid[k] = s[k] / (double)n;
    }
    if (temp) HIP_TRY(c, block.alloc_temp(c->arena, 6 * n));
    else { HIP_TRY(c, keep_alloc(c, (void **)&block.p, sizeof(float) * 6 * n)); block.owned = false; }
    float *col[6];
    for (int k = 0; k < 6; k++) col[k] = block.p + (size_t)k * n;
    struct Part { const float *base; size_t rs, cs; int first_col; } parts[2] = {{xyz, xr, xc, 0}, {nrm, nr, nc, 3}};
    for (const Part &p : parts) {
        if (p.cs == 1 && p.rs >= 3) {
            const size_t fl = (n - 1) * p.rs + 3;                  // floats from the first x to the last z
            DevBuf<float> raw;
            HIP_TRY(c, raw.alloc_temp(c->arena, fl));
            HIP_TRY(c, hipMemcpyAsync(raw.p, p.base, sizeof(float) * fl, hipMemcpyHostToDevice, c->stream));
            launch_deinterleave3(raw.p, p.rs, 0, (uint32_t)n, col[p.first_col], col[p.first_col + 1], col[p.first_col + 2], c->stream);
            HIP_TRY(c, hipStreamSynchronize(c->stream));           // raw is freed on scope exit
        } else if (p.rs == 1 && p.cs >= n) {
            for (int k = 0; k < 3; k++)
                HIP_TRY(c, hipMemcpyAsync(col[p.first_col + k], p.base + (size_t)k * p.cs, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        } else {
            std::vector<float> stage(3 * n);
            for (size_t i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) stage[(size_t)k * n + i] = p.base[i * p.rs + k * p.cs];
            HIP_TRY(c, hipMemcpyAsync(col[p.first_col], stage.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    HIP_TRY(c, hipGetLastError());
    return SYMMICP_OK;
}

static float ord2f(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Morton order of a planar cloud: fills order[n] (sorted position -> row) and, optionally, keeps the sorted keys.
static int morton_order(symmicp_ctx *c, const CloudSoA &cl, uint32_t n, DevBuf<uint32_t> &vals, DevBuf<uint32_t> *keys_out,
                        float origin[3], float *h0_out)
{
    DevBuf<uint32_t> bbox, keys_local, kt, vt, ws;
    DevBuf<uint32_t> &keys = keys_out ? *keys_out : keys_local;
    const size_t wse = radix_sort_ws_elems(n);
    HIP_TRY(c, bbox.alloc_temp(c->arena, 6));
    HIP_TRY(c, keys.alloc_temp(c->arena, n));
    HIP_TRY(c, vals.alloc_temp(c->arena, n));
    HIP_TRY(c, kt.alloc_temp(c->arena, n));
    HIP_TRY(c, vt.alloc_temp(c->arena, n));
    HIP_TRY(c, ws.alloc_temp(c->arena, wse));
    launch_bbox(cl.x, cl.y, cl.z, n, bbox.p, c->stream);
    uint32_t hb[6];
    HIP_TRY(c, hipMemcpyAsync(hb, bbox.p, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) { lo[k] = ord2f(hb[k]); hi[k] = ord2f(hb[3 + k]); }
    float emax = 0.f;
    for (int k = 0; k < 3; k++) {
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) return fail(c, SYMMICP_ERR_ARG, "cloud has non-finite coordinates");
        emax = std::fmax(emax, hi[k] - lo[k]);
    }
    if (!(emax > 0.f)) emax = 1.f;
    const float h0 = emax * 1.00001f / (float)(1 << kMortonBits);
    launch_morton(cl.x, cl.y, cl.z, n, lo[0], lo[1], lo[2], 1.0f / h0, keys.p, vals.p, c->stream);
    radix_sort_pairs(keys.p, vals.p, kt.p, vt.p, n, 3 * kMortonBits, ws.p, wse, c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    for (int k = 0; k < 3; k++) origin[k] = lo[k];
    *h0_out = h0;
    return SYMMICP_OK;
}

// Search index over a planar cloud: Morton sort -> float4 gather -> (optional) dense cell table at the
// chosen octree level -> implicit 8-ary box tree.  tq/tn must already be allocated (n float4 each).
// sparse octree over the sorted keys (levels 0..kMortonBits, built bottom-up); see TargetIndex::onodes
static int build_octree(symmicp_ctx *c, const uint32_t *keys, const float4 *tq, uint32_t n, uint32_t leaf_max, float4 **onodes_out, TargetIndex *ix)
{
    constexpr int NL = kMortonBits + 1;
    DevBuf<uint32_t> nid;                         // [NL][n]: id of the node that starts at point i, per level
    DevBuf<uint32_t> scan_ws, first;
    float4 *nodes = nullptr;
    HIP_TRY(c, nid.alloc_temp(c->arena, (size_t)NL * n));
    HIP_TRY(c, scan_ws.alloc_temp(c->arena, (size_t)n / 2048 + 2));
    for (int l = 0; l < NL; l++) {
        launch_oct_flags(keys, n, l, nid.p + (size_t)l * n, c->stream);
        launch_exclusive_scan(nid.p + (size_t)l * n, n, scan_ws.p, c->stream);
    }
    // node counts: exclusive scan value at the last point, +1 if the last point starts a node (host checks the keys)
    uint32_t last_excl[NL], kl[2] = {0, 0};
    for (int l = 0; l < NL; l++)
        HIP_TRY(c, hipMemcpyAsync(&last_excl[l], nid.p + (size_t)l * n + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (n >= 2) HIP_TRY(c, hipMemcpyAsync(kl, keys + (n - 2), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint32_t cnt[NL];
    size_t total = 0;
    for (int l = 0; l < NL; l++) {
        const int shift = 3 * (kMortonBits - l);
        const bool last_starts = (n == 1) || (shift < 30 && (kl[1] >> shift) != (kl[0] >> shift));
        cnt[l] = last_excl[l] + (last_starts ? 1u : 0u);
        ix->olevel_off[l] = (uint32_t)total;
        total += cnt[l];
    }
    ix->olevel_off[NL] = (uint32_t)total;
    for (int l = 0; l < NL; l++)
        // child_first is a 28-bit field of the node word: targets beyond ~268M distinct finest cells are refused
        if (cnt[l] > kOctCfMask) return fail(c, SYMMICP_ERR_SIZE, "octree level exceeds 2^28 nodes (target cloud too large for SYMMICP_CORR_TREE)");
    HIP_TRY(c, first.alloc_temp(c->arena, total + 1));
    HIP_TRY(c, keep_alloc(c, (void **)&nodes, sizeof(float4) * 2 * total));
    for (int l = 0; l < NL; l++)
        launch_oct_first(keys, n, l, nid.p + (size_t)l * n, first.p + ix->olevel_off[l], c->stream);
    for (int l = NL - 1; l >= 0; l--) {
        const bool bottom = (l == NL - 1);
        launch_oct_nodes(l, tq, n, first.p + ix->olevel_off[l], cnt[l], bottom ? nullptr : nid.p + (size_t)(l + 1) * n,
                         bottom ? 0u : cnt[l + 1], bottom ? nullptr : nodes + 2 * (size_t)ix->olevel_off[l + 1],
                         nodes + 2 * (size_t)ix->olevel_off[l], leaf_max, c->stream);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    *onodes_out = nodes;
    ix->onodes = *onodes_out;
    return SYMMICP_OK;
}

// The outputs (*boxes_out, *cells_out, *onodes_out, *ctop_out) belong to the caller's context as soon as they are set:
// on failure the caller's free_target() / cleanup releases them.
static int build_index(symmicp_ctx *c, const CloudSoA &cl, uint32_t n, bool want_grid, float4 *tq, float4 *tn,
                       float4 **boxes_out, uint2 **cells_out, TargetIndex *ix_out, int32_t *glevel_out, int32_t *nlevels_out,
                       float4 **onodes_out = nullptr, uint32_t **ctop_out = nullptr)
{
    DevBuf<uint32_t> order, keys;
    float origin[3], h0;
    int st = morton_order(c, cl, n, order, &keys, origin, &h0);
    if (st != SYMMICP_OK) return st;
    launch_gather_f4(cl.x, cl.y, cl.z, cl.nx, cl.ny, cl.nz, order.p, n, tq, tn, c->stream);
    TargetIndex ix{};
    ix.tq = tq; ix.tn = tn; ix.n = n;
    int glevel = 0;
    if (want_grid) {
        DevBuf<uint32_t> hist;
        HIP_TRY(c, hist.alloc_temp(c->arena, 16));
        launch_level_hist(keys.p, n, hist.p, c->stream);
        uint32_t hh[16];
        HIP_TRY(c, hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        // finest level whose occupied cells still hold >= ppc points on average
        const double ppc = c->sw.grid_ppc;
        int lcap = c->sw.grid_maxlevel;                  // the table is two-level: memory follows the occupied super-cells
        if (lcap > kMortonBits) lcap = kMortonBits;
        glevel = 1;
        double occ = 1.0;
        for (int l = 1; l <= lcap; l++) {
            occ += (double)hh[l];
            if ((double)n / occ >= ppc) glevel = l;
        }
        if (c->sw.grid_level >= 0) glevel = c->sw.grid_level;   // 0 disables the grid phase
        if (glevel > lcap) glevel = lcap;
        if (glevel < 0) glevel = 0;
        // Is the target a surface or a volume?  Occupied cells grow ~4x per octree level on a surface and ~8x in a volume
        // (measured one level above the grid level, where cells still hold several points).  On surface-like targets the
        // queries of a packet share most of their search (offset surfaces: every ball touches the target in a wide disc), so
        // the first pass runs as packets (kernels_packet.hip) over an octree with larger leaves; in a volume cloud the
        // neighbours are half a spacing away, nothing is shared, and the per-thread walk stays (100k uniform cube, first
        // pass: 0.11 ms per-thread walk, 0.31 ms packets; 1M surface pair: 1.69 ms against 0.87 ms).
        {
            double occ_l[kMortonBits + 1];
            double o = 1.0;
            occ_l[0] = 1.0;
            for (int l = 1; l <= kMortonBits; l++) { o += (double)hh[l]; occ_l[l] = o; }
            const int lg = glevel >= 2 ? glevel - 1 : 1;
            const double growth = occ_l[lg] / occ_l[lg - 1];
            c->target_surface_like = growth < 5.5;
            if (c->sw.first_pass >= 0) c->target_surface_like = c->sw.first_pass == 1;      // SYMMICP_FIRST_PASS=packet|walk: A/B runs
        }
    }
    ix.glevel = glevel;
    if (glevel > 0) {
        const int ltop = glevel > 3 ? glevel - 3 : 0;
        const size_t ntop = (size_t)1 << (3 * ltop);
        // block numbers of the occupied super-cells: exclusive scan of their start flags
        DevBuf<uint32_t> nid_top, scan_ws;
        HIP_TRY(c, nid_top.alloc_temp(c->arena, n));
        HIP_TRY(c, scan_ws.alloc_temp(c->arena, (size_t)n / 2048 + 2));
        launch_oct_flags(keys.p, n, ltop, nid_top.p, c->stream);
        launch_exclusive_scan(nid_top.p, n, scan_ws.p, c->stream);
        uint32_t last_excl = 0, kl[2] = {0, 0};
        HIP_TRY(c, hipMemcpyAsync(&last_excl, nid_top.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        if (n >= 2) HIP_TRY(c, hipMemcpyAsync(kl, keys.p + (n - 2), 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const int tshift = 3 * (kMortonBits - ltop);
        const bool last_starts = (n == 1) || (tshift < 30 && (kl[1] >> tshift) != (kl[0] >> tshift));
        const size_t nblocks = (size_t)last_excl + (last_starts ? 1 : 0);
        HIP_TRY(c, keep_alloc(c, (void **)ctop_out, sizeof(uint32_t) * ntop));
        HIP_TRY(c, hipMemsetAsync(*ctop_out, 0xFF, sizeof(uint32_t) * ntop, c->stream));
        HIP_TRY(c, keep_alloc(c, (void **)cells_out, sizeof(uint2) * nblocks * 512));
        HIP_TRY(c, hipMemsetAsync(*cells_out, 0, sizeof(uint2) * nblocks * 512, c->stream));
        launch_cell_table(keys.p, n, glevel, nid_top.p, *ctop_out, *cells_out, c->stream);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        ix.ctop = *ctop_out;
        ix.cells = *cells_out;
        ix.gdim = 1 << glevel;
        ix.ox = origin[0]; ix.oy = origin[1]; ix.oz = origin[2];
        ix.h = h0 * (float)(1 << (kMortonBits - glevel));
        ix.inv_h = (1.0f / h0) / (float)(1 << (kMortonBits - glevel));
    }
    // tree levels: level 0 = leaves of kLeaf points, each level padded to a multiple of kFan
    uint32_t cnt[kMaxTreeLevels], pad[kMaxTreeLevels];
    int nl = 0;
    uint32_t m = (n + kLeaf - 1) / kLeaf;
    size_t total = 0;
    while (true) {
        cnt[nl] = m;
        pad[nl] = (m <= (uint32_t)kFan) ? m : ((m + kFan - 1) / kFan) * kFan;
        ix.level_off[nl] = (uint32_t)total;
        total += pad[nl];
        nl++;
        if (m <= (uint32_t)kFan) break;
        m = pad[nl - 1] / kFan;
        if (nl >= kMaxTreeLevels) return fail(c, SYMMICP_ERR_SIZE, "tree too deep");
    }
    ix.top = nl - 1;
    ix.ntop = cnt[nl - 1];
    HIP_TRY(c, keep_alloc(c, (void **)boxes_out, sizeof(float4) * 2 * total));
    float4 *boxes = *boxes_out;
    launch_leaf_boxes(tq, n, boxes + 2 * (size_t)ix.level_off[0], pad[0], c->stream);
    for (int l = 1; l < nl; l++)
        launch_node_boxes(boxes + 2 * (size_t)ix.level_off[l - 1], pad[l - 1], boxes + 2 * (size_t)ix.level_off[l], pad[l], c->stream);
    ix.boxes = boxes;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (onodes_out) {
        uint32_t leaf_max = c->target_surface_like ? 24u : 8u;      // (packets: 12 / 16 / 20 / 24 / 28 points per leaf: 0.425 / 0.411 / 0.394 / 0.389 / 0.394 ms first pass of the 1M surface pair)
        if (c->sw.oct_leaf > 0) leaf_max = (uint32_t)c->sw.oct_leaf;
        st = build_octree(c, keys.p, tq, n, leaf_max, onodes_out, &ix);
        if (st != SYMMICP_OK) return st;
    }
    *ix_out = ix;
    if (glevel_out) *glevel_out = glevel;
    if (nlevels_out) *nlevels_out = nl;
    return SYMMICP_OK;
}

int symmicp_set_target(symmicp_ctx *c, const float *xyz, size_t xr, size_t xc, const float *nrm, size_t nr, size_t nc, size_t n)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!xyz || !nrm) return fail(c, SYMMICP_ERR_ARG, "null target cloud (myicp.cpp:102 assert)");
    if (n == 0 || n > 0x7fffffffull) return fail(c, SYMMICP_ERR_SIZE, "target size out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    const double t0 = now_s();
    free_target(c);
    c->begun = false;
    {
        // the target's persistent arrays: planar cloud 24 B, tq 16, pair records 32, octree ~64, run tree ~5, cell table
        const size_t want = n * 160 + ((size_t)16 << 20);
        if (c->keep.cap < want) {
            hipFree(c->keep.base);
            c->keep = Arena{};
            if (hipMalloc((void **)&c->keep.base, want + want / 8) == hipSuccess) c->keep.cap = want + want / 8;
            else (void)hipGetLastError();      // every keep_alloc then falls back to its own hipMalloc
        }
    }
    // temporaries of the upload and of the index build: ~80 B per point (11 octree-level id arrays, sort buffers, raw rows)
    arena_begin(c->arena, n * (96 + 4 * (xr + nr)) + ((size_t)1 << 20));
    double cen[3];
    DevBuf<float> tblock;
    int st = upload_planar(c, xyz, xr, xc, nrm, nr, nc, n, tblock, /*temp=*/false, cen);
    if (st != SYMMICP_OK) return st;
    c->tgt_block = tblock.release();
    soa_from_block(c->tgt_block, n, c->tgt);
    c->n_t = (uint32_t)n;
    for (int k = 0; k < 3; k++) c->pivot[k] = (float)cen[k];
    c->st.upload_ms += (now_s() - t0) * 1e3;
    if (c->cfg.corr == SYMMICP_CORR_IDENTITY) return SYMMICP_OK;

    const double t1 = now_s();
    HIP_TRY(c, keep_alloc(c, (void **)&c->tq, sizeof(float4) * (n + 8)));     // + 8: the packet search reads a leaf's points in groups of 8
    HIP_TRY(c, hipMemsetAsync(c->tq + n, 0, sizeof(float4) * 8, c->stream));
    HIP_TRY(c, keep_alloc(c, (void **)&c->tn, sizeof(float4) * 2 * n));      // (point, normal) pair records
    if (c->cfg.corr == SYMMICP_CORR_BRUTE) {
        launch_iota_f4(c->tgt.x, c->tgt.y, c->tgt.z, c->tgt.nx, c->tgt.ny, c->tgt.nz, c->n_t, c->tq, c->tn, c->stream);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->st.build_ms = (now_s() - t1) * 1e3;
        return SYMMICP_OK;
    }
    st = build_index(c, c->tgt, c->n_t, /*want_grid=*/true, c->tq, c->tn, &c->boxes, &c->cells, &c->ix, &c->st.grid_level, &c->st.tree_levels,
                     &c->onodes, &c->ctop);
    if (st != SYMMICP_OK) { free_target(c); return st; }
    if (c->sw.debug_counters) {
        HIP_TRY(c, hipMalloc((void **)&c->dbg, 12 * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->dbg, 0, 12 * sizeof(unsigned long long)));
        c->ix.dbg = c->dbg;
    }
    if (!c->sw.debug_trace.empty()) {          // (alone: the production kernel with a timing-only trace)
        HIP_TRY(c, hipMalloc((void **)&c->dbg_trace, ((size_t)1 << 22) * 8));      // (kTraceWords)
        HIP_TRY(c, hipMemset(c->dbg_trace, 0, ((size_t)1 << 22) * 8));
        c->ix.dbg_trace = c->dbg_trace;
    }
    c->have_index = true;
    c->st.build_ms = (now_s() - t1) * 1e3;
    return SYMMICP_OK;
}

int symmicp_set_source(symmicp_ctx *c, const float *xyz, size_t xr, size_t xc, const float *nrm, size_t nr, size_t nc, size_t n)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!xyz || !nrm) return fail(c, SYMMICP_ERR_ARG, "null source cloud (myicp.cpp:102 assert)");
    if (n == 0 || n > 0x7fffffffull) return fail(c, SYMMICP_ERR_SIZE, "source size out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    const double t0 = now_s();
    forget_source(c);
    c->begun = false;
    // This rank's share is a contiguous block of the CALLER's rows, and only those rows are uploaded and sorted: set_source costs
    // 1/nranks of the single-GPU call on every rank (round 1 uploaded and sorted the whole cloud on every rank and kept a
    // slice of the global Morton order).  Any partition of the source is exact -- queries are independent given the
    // transform -- and the share's own Morton sort below gives the waves their locality.
    size_t b0 = 0, bc = 0;
    symmicp_shard_range(n, c->nranks, c->rank, &b0, &bc);
    const size_t nu = bc > 0 ? bc : 1;                       // rows uploaded (an empty share still stages one row)
    const size_t r0 = bc > 0 ? b0 : 0;
    arena_begin(c->arena, nu * (56 + 4 * (xr + nr)) + ((size_t)1 << 20));      // (+8 B per point: the packet table's temporaries)
    DevBuf<float> full;
    int st = upload_planar(c, xyz + r0 * xr, xr, xc, nrm + r0 * nr, nr, nc, nu, full, /*temp=*/true, nullptr);
    if (st != SYMMICP_OK) return st;
    CloudSoA fs;
    soa_from_block(full.p, nu, fs);
    const uint32_t nl = bc > 0 ? (uint32_t)bc : 1;
    const bool sorted = c->cfg.corr != SYMMICP_CORR_IDENTITY && c->cfg.sort_source;
    const bool tree = c->cfg.corr == SYMMICP_CORR_TREE, brute = c->cfg.corr == SYMMICP_CORR_BRUTE;
    // every per-source array in ONE allocation (a hipFree costs ~100 us, and a tracker calls this once per frame)
    const uint32_t cap = shard_capacity(nl);
    const size_t per_list = (size_t)kShards * cap, ncount = (size_t)kShards * kShardStride;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_src0 = take(sizeof(float) * 6 * nl), o_cur = take(sizeof(float) * 6 * nl);
    const size_t o_order = sorted ? take(sizeof(uint32_t) * nl) : 0;
    const size_t o_pos = take(sizeof(int32_t) * nl), o_d2 = take(sizeof(float) * nl);
    const size_t o_best = brute ? take(sizeof(unsigned long long) * nl) : 0;
    const size_t o_cert = tree ? take(sizeof(float) * 4 * nl) : 0;
    const size_t o_certk = tree ? take(sizeof(uint32_t) * 8 * nl) : 0, o_hoodr = tree ? take(sizeof(float) * 2 * nl) : 0;
    const size_t o_pkt = tree ? take(sizeof(uint32_t) * 2 * 8 * ((nl + 63) / 64)) : 0;        // (a block of 64 queries may be cut into 8 packets: kMaxRunsPerBlock)
    const size_t o_prec = tree ? take(sizeof(float4) * 2 * nl) : 0;
    const size_t o_wl = tree ? take(sizeof(uint32_t) * 2 * per_list) : 0, o_cnt = tree ? take(sizeof(uint32_t) * 2 * ncount) : 0;      // work + retry lists
    if (off > c->src_all_cap) {
        hipFree(c->src_all);
        c->src_all = nullptr; c->src_all_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->src_all, off));
        c->src_all_cap = off;
    }
    c->n_s_total = (uint32_t)n;
    c->src_off = (uint32_t)b0;
    c->n_loc = (uint32_t)bc;
    c->src0_block = reinterpret_cast<float *>(c->src_all + o_src0);
    c->cur_block = reinterpret_cast<float *>(c->src_all + o_cur);
    soa_from_block(c->src0_block, nl, c->src0);
    soa_from_block(c->cur_block, nl, c->cur);
    c->pos = reinterpret_cast<int32_t *>(c->src_all + o_pos);
    c->d2 = reinterpret_cast<float *>(c->src_all + o_d2);
    if (brute) c->best64 = reinterpret_cast<unsigned long long *>(c->src_all + o_best);
    if (sorted) {
        DevBuf<uint32_t> order;
        float origin[3], h0;
        st = morton_order(c, fs, (uint32_t)nu, order, nullptr, origin, &h0);
        if (st != SYMMICP_OK) { forget_source(c); return st; }
        c->src_order = reinterpret_cast<uint32_t *>(c->src_all + o_order);
        if (c->n_loc) {
            launch_gather_soa(fs, order.p, c->n_loc, c->src0, c->stream);               // share position -> row of the uploaded block
            launch_offset_u32(order.p, (uint32_t)b0, c->n_loc, c->src_order, c->stream); // ... -> row of the caller's cloud
        }
    } else if (c->n_loc) {
        HIP_TRY(c, hipMemcpyAsync(c->src0_block, full.p, sizeof(float) * 6 * c->n_loc, hipMemcpyDeviceToDevice, c->stream));
    }
    if (tree) {
        c->cert = reinterpret_cast<float *>(c->src_all + o_cert);
        c->certk = reinterpret_cast<uint32_t *>(c->src_all + o_certk);      // (validity lives in bit 0 of the certificate word: no clearing needed)
        c->hoodr = reinterpret_cast<float *>(c->src_all + o_hoodr);
        HIP_TRY(c, hipMemsetAsync(c->hoodr, 0, sizeof(float) * 2 * nl, c->stream));                // no radius hints yet
        c->pairrec = reinterpret_cast<float4 *>(c->src_all + o_prec);
        c->worklist = reinterpret_cast<uint32_t *>(c->src_all + o_wl);
        c->wl_count = reinterpret_cast<uint32_t *>(c->src_all + o_cnt);
        HIP_TRY(c, hipMemsetAsync(c->cert, 0, sizeof(float) * 4 * nl, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->wl_count, 0, sizeof(uint32_t) * 2 * ncount, c->stream));
        c->wl.work = ShardList{c->worklist, c->wl_count, cap};
        c->wl.retry = ShardList{c->worklist + per_list, c->wl_count + ncount, cap};
    }
    if (tree && c->n_loc) {
        // The first pass's packets, in start order: by decreasing radius.  A packet that straddles a jump of the Morton curve takes several
        // times as many sweep steps as a compact one (median 210 us, 1 % above 830 us in the traced build), and a launch that meets such
        // packets last ends with a handful of waves running: longest-first is the classic remedy.  Radius keys, the index build's radix
        // sort (n / 64 keys) and the table, all on the device (+0.1 ms of set_source at 1M points; a host sort cost 0.85 ms).
        // Measured on the 1M surface pair: 0.81 -> 0.67 ms (only the widest third first: 0.70 -- the Morton order of the rest, i.e. XCD
        // locality, is worth less than the balance; splitting the widest packets into halves / quarters on top: no gain).
        const uint32_t nblk = (c->n_loc + 63u) / 64u;
        const bool ordered = c->sw.packet_order;      // (SYMMICP_PACKET_ORDER=0: packets as they lie)
        c->pkt_tab = nullptr; c->pkt_count = 0;
        if (ordered) {
            // (a block of 64 queries is cut into runs at its jumps of the Morton curve: k_packet_runs)
            const float jump = c->sw.packet_jump;      // x the block's scale; 0: never cut
            const int key_bits = c->sw.packet_key_bits;      // (with the blocks cut at their jumps the order needs no more: 10 / 12 / 16 / 32 bits all 0.56-0.58 ms; two radix passes instead of four)
            const uint32_t cap = 8u * nblk;            // (kMaxRunsPerBlock)
            DevBuf<uint32_t> keys, vals, kt, vt, ws, cnt;
            DevBuf<uint2> runs;
            const size_t wse = radix_sort_ws_elems(cap);
            HIP_TRY(c, keys.alloc_temp(c->arena, cap));
            HIP_TRY(c, vals.alloc_temp(c->arena, cap));
            HIP_TRY(c, kt.alloc_temp(c->arena, cap));
            HIP_TRY(c, vt.alloc_temp(c->arena, cap));
            HIP_TRY(c, runs.alloc_temp(c->arena, cap));
            HIP_TRY(c, ws.alloc_temp(c->arena, wse));
            HIP_TRY(c, cnt.alloc_temp(c->arena, 1));
            HIP_TRY(c, hipMemsetAsync(cnt.p, 0, sizeof(uint32_t), c->stream));
            launch_packet_runs(c->src0, c->n_loc, jump, runs.p, keys.p, vals.p, cnt.p, key_bits, c->stream);
            uint32_t npk = 0;
            HIP_TRY(c, hipMemcpyAsync(&npk, cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (npk < nblk || npk > cap) { forget_source(c); return fail(c, SYMMICP_ERR_HIP, "packet table: count out of range"); }
            // the target is known already: key the start order by what a packet will cost (its distance to the target), not by its extent alone
            if (c->have_index && c->ix.onodes && key_bits > 0 && c->sw.packet_cost_key) launch_packet_cost(c->src0, runs.p, npk, c->ix, keys.p, key_bits, c->stream);
            radix_sort_pairs(keys.p, vals.p, kt.p, vt.p, npk, key_bits > 0 ? key_bits : 32, ws.p, wse, c->stream);
            c->pkt_tab = reinterpret_cast<uint32_t *>(c->src_all + o_pkt);
            c->pkt_count = npk;
            launch_packet_table(vals.p, runs.p, npk, reinterpret_cast<uint2 *>(c->pkt_tab), c->stream);
            HIP_TRY(c, hipStreamSynchronize(c->stream));       // (the temporaries are about to go out of scope)
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));       // (the arena-backed upload is read by the gather above)
    HIP_TRY(c, hipGetLastError());
    c->last_list_len = -1;
    c->st.upload_ms += (now_s() - t0) * 1e3;
    return SYMMICP_OK;
}

// resolve the recorded event ring into per-kernel totals (called lazily: ring full, or stats requested)
extern "C++" void flush_events(symmicp_ctx *c)
{
    for (int p = 0; p < c->ev_used; p++) {
        hipEvent_t *e = c->ev + p * symmicp_ctx::kEvPer;
        if (hipEventSynchronize(c->ev_split[p] == 1 ? e[4] : e[5]) != hipSuccess) continue;
        float ms = 0.f;
        double pass_ms = 0.0;
        if (c->ev_split[p] == 2) {
            for (int k = 0; k < 5; k++)
                if (hipEventElapsedTime(&ms, e[k], e[k + 1]) == hipSuccess) { c->st.kernel_ms[k] += ms; c->st.kernel_launches[k]++; if (k < 4) pass_ms += ms; }
        } else if (c->ev_split[p] == 1) {
            if (hipEventElapsedTime(&ms, e[0], e[4]) == hipSuccess) { pass_ms += ms; c->st.kernel_ms[6] += ms; c->st.kernel_launches[6]++; }
        } else {
            if (hipEventElapsedTime(&ms, e[0], e[4]) == hipSuccess) { c->st.kernel_ms[5] += ms; c->st.kernel_launches[5]++; pass_ms += ms; }
            if (hipEventElapsedTime(&ms, e[4], e[5]) == hipSuccess) { c->st.kernel_ms[4] += ms; c->st.kernel_launches[4]++; }
        }
        if (c->ev_coll[p] && hipEventElapsedTime(&ms, e[6], e[7]) == hipSuccess) { c->st.allreduce_ms += ms; c->st.allreduce_timed++; }
        c->ev_coll[p] = 0;
        const int w = c->ev_weight[p] > 0 ? c->ev_weight[p] : 1;
        c->st.last_pass_ms = pass_ms;
        c->st.sum_pass_ms += pass_ms * w;
        for (int k = 0; k < w && c->st.passes_timed + k < 8; k++) c->st.pass_ms_head[c->st.passes_timed + k] = pass_ms;      // (a sampled pass stands for w passes)
        c->st.passes_timed += w;
    }
    c->ev_used = 0;
}

int symmicp_get_transform(const symmicp_ctx *c, float out16[16])
{
    if (!c || !out16) return SYMMICP_ERR_ARG;
    std::memcpy(out16, c->X, sizeof(float) * 16);
    return SYMMICP_OK;
}

int symmicp_get_pivot(const symmicp_ctx *c, float out3[3])
{
    if (!c || !out3) return SYMMICP_ERR_ARG;
    const bool paper = c->cfg.mode != SYMMICP_MODE_QUIRKS;
    for (int k = 0; k < 3; k++) out3[k] = paper ? c->pivot[k] : 0.f;
    return SYMMICP_OK;
}

size_t symmicp_local_source_count(const symmicp_ctx *c) { return c ? c->n_loc : 0; }
size_t symmicp_local_source_offset(const symmicp_ctx *c) { return c ? c->src_off : 0; }

// rows are reported relative to this rank's share; with a sorted source the share is a set of
// caller rows (not a contiguous range), so indices are written at [row] of a full-size array.
int symmicp_get_correspondences(symmicp_ctx *c, int32_t *idx, float *d2, size_t cap)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!c->begun) return fail(c, SYMMICP_ERR_STATE, "no pass has run yet");
    const size_t need = c->src_order ? c->n_s_total : c->n_loc;
    if (cap < need) return fail(c, SYMMICP_ERR_SIZE, "output too small");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<int32_t> d_idx;
    DevBuf<float> d_d2;
    arena_begin(c->arena, need * 8 + 4096);
    HIP_TRY(c, d_idx.alloc_temp(c->arena, need));
    HIP_TRY(c, d_d2.alloc_temp(c->arena, need));
    HIP_TRY(c, hipMemsetAsync(d_idx.p, 0xFF, sizeof(int32_t) * need, c->stream));
    HIP_TRY(c, hipMemsetAsync(d_d2.p, 0, sizeof(float) * need, c->stream));
    const int mode = c->cfg.corr == SYMMICP_CORR_IDENTITY ? 0 : (c->cfg.corr == SYMMICP_CORR_BRUTE ? 1 : 2);
    if (mode == 0 && d2) {
        // the identity pass streams without storing distances: evaluate them now from the current source positions
        const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
        Affine X{};
        float I[16];
        identity16(I);
        const float *m = incr ? I : c->X;
        for (int k = 0; k < 12; k++) X.m[k] = m[k];
        X.nrm_w = 0.f;
        launch_identity_d2(incr ? c->cur : c->src0, X, c->tgt, c->src_off, c->n_loc, c->d2, c->stream);
    }
    if (mode == 2 && d2) {
        // the tree passes store a distance only for the pairs they searched: evaluate all of them at the positions of the last pass
        const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
        Affine X{};
        float I[16];
        identity16(I);
        const float *m = incr ? I : c->X;
        for (int k = 0; k < 12; k++) X.m[k] = m[k];
        X.nrm_w = 0.f;
        launch_pairs_d2(incr ? c->cur : c->src0, X, c->pos, c->tq, c->n_t, c->n_loc, c->d2, c->stream);
    }
    launch_corr_out(c->pos, c->best64, c->d2, c->tq, c->src_order, c->n_loc, mode, c->src_off, d_idx.p, d_d2.p, c->stream);
    if (idx) HIP_TRY(c, hipMemcpyAsync(idx, d_idx.p, sizeof(int32_t) * need, hipMemcpyDeviceToHost, c->stream));
    if (d2) HIP_TRY(c, hipMemcpyAsync(d2, d_d2.p, sizeof(float) * need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SYMMICP_OK;
}

int symmicp_get_source(symmicp_ctx *c, float *xyz, float *nrm, size_t cap)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!c->begun) return fail(c, SYMMICP_ERR_STATE, "no pass has run yet");
    if (resolved_apply(c->cfg) != SYMMICP_APPLY_INCREMENTAL)
        return fail(c, SYMMICP_ERR_STATE, "the transformed source is only materialised with SYMMICP_APPLY_INCREMENTAL");
    const size_t need = c->src_order ? c->n_s_total : c->n_loc;
    if (cap < need) return fail(c, SYMMICP_ERR_SIZE, "output too small");
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<float> dx, dn;
    arena_begin(c->arena, need * 24 + 4096);
    HIP_TRY(c, dx.alloc_temp(c->arena, 3 * need));
    HIP_TRY(c, dn.alloc_temp(c->arena, 3 * need));
    HIP_TRY(c, hipMemsetAsync(dx.p, 0, sizeof(float) * 3 * need, c->stream));
    HIP_TRY(c, hipMemsetAsync(dn.p, 0, sizeof(float) * 3 * need, c->stream));
    launch_unpermute(c->cur, c->src_order, c->n_loc, dx.p, dn.p, c->stream);
    if (xyz) HIP_TRY(c, hipMemcpyAsync(xyz, dx.p, sizeof(float) * 3 * need, hipMemcpyDeviceToHost, c->stream));
    if (nrm) HIP_TRY(c, hipMemcpyAsync(nrm, dn.p, sizeof(float) * 3 * need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SYMMICP_OK;
}

// diagnostic: the pair certificates as they stand, in the share's sorted order: (ref.xyz, L), the neighbourhood's 8 members and radius, the
// current winner, target points named by their ROW in the caller's target cloud (the device holds sorted positions)
int symmicp_get_certificates(symmicp_ctx *c, float *cert4, uint32_t *hood8, float *hood_radius, int32_t *winner_row, size_t cap)
{
    if (!c || !cert4) return SYMMICP_ERR_ARG;
    if (!c->begun || !c->cert) return fail(c, SYMMICP_ERR_STATE, "no tree pass has run yet");
    if (cap < c->n_loc) return fail(c, SYMMICP_ERR_SIZE, "output too small");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(cert4, c->cert, sizeof(float) * 4 * c->n_loc, hipMemcpyDeviceToHost, c->stream));
    std::vector<int32_t> pos(winner_row ? c->n_loc : 0);
    std::vector<float> tq((hood8 || winner_row) ? (size_t)c->n_t * 4 : 0);
    if (hood8) HIP_TRY(c, hipMemcpyAsync(hood8, c->certk, sizeof(uint32_t) * 8 * c->n_loc, hipMemcpyDeviceToHost, c->stream));
    std::vector<float> tr(hood_radius ? (size_t)c->n_loc * 2 : 0);
    if (hood_radius) HIP_TRY(c, hipMemcpyAsync(tr.data(), c->hoodr, sizeof(float) * 2 * c->n_loc, hipMemcpyDeviceToHost, c->stream));
    if (winner_row) HIP_TRY(c, hipMemcpyAsync(pos.data(), c->pos, sizeof(int32_t) * c->n_loc, hipMemcpyDeviceToHost, c->stream));
    if (!tq.empty()) HIP_TRY(c, hipMemcpyAsync(tq.data(), c->tq, sizeof(float) * 4 * c->n_t, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    auto row_of = [&](uint32_t p) -> uint32_t {
        if (p >= c->n_t) return 0xFFFFFFFFu;
        uint32_t r; std::memcpy(&r, &tq[(size_t)p * 4 + 3], 4); return r;      // the row rides in the w slot of the sorted point
    };
    for (size_t i = 0; i < c->n_loc; i++) {
        if (hood8) for (int k = 0; k < 8; k++) hood8[i * 8 + k] = row_of(hood8[i * 8 + k]);
        if (winner_row) winner_row[i] = pos[i] < 0 ? -1 : (int32_t)row_of((uint32_t)pos[i]);
        if (hood_radius) hood_radius[i] = tr[i * 2];
    }
    return SYMMICP_OK;
}

// ---- normals pre-step (MyICP::estimateNormals, myicp.cpp:152-172) --------------------------------
// Runs on a context the caller owns (its stream and arenas are reused: a tracker that estimates normals per scan pays no
// context set-up); the context's target and source stay as they are -- what the estimate allocates behind the target's arrays
// in the keep-arena is released again.
int symmicp_ctx_estimate_normals(symmicp_ctx *c, const float *xyz, size_t row_stride, size_t col_stride, size_t n, int k,
                                 const float viewpoint[3], float *nrm_out, float *curv_out)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (!xyz || !nrm_out || n == 0 || n > 0x7fffffffull) return fail(c, SYMMICP_ERR_ARG, "bad cloud");
    if (k < 3 || k > 16 || (size_t)k > n) return fail(c, SYMMICP_ERR_ARG, "k out of range (3..16, <= n)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->keep.cap == 0) {
        // no target yet: give the keep-arena the size this cloud needs (a later set_target reuses it)
        const size_t want = n * 96 + ((size_t)4 << 20);
        if (hipMalloc((void **)&c->keep.base, want) == hipSuccess) { c->keep.cap = want; c->keep.off = 0; }
        else (void)hipGetLastError();
    }
    const size_t keep_off0 = c->keep.off, extra0 = c->keep_extra.size();
    const bool surf0 = c->target_surface_like;
    float *block = nullptr, *d_nrm = nullptr, *d_curv = nullptr;
    float4 *tq = nullptr, *tn = nullptr, *boxes = nullptr;
    uint2 *cells = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(c->stream);
        while (c->keep_extra.size() > extra0) { hipFree(c->keep_extra.back()); c->keep_extra.pop_back(); }
        c->keep.off = keep_off0;
        c->target_surface_like = surf0;
    };
    // the cloud has no normals yet: stage xyz twice (the normal slots are ignored)
    arena_begin(c->arena, n * (96 + 8 * row_stride + 16) + ((size_t)1 << 20));
    int st;
    {
        DevBuf<float> b;
        st = upload_planar(c, xyz, row_stride, col_stride, xyz, row_stride, col_stride, n, b, /*temp=*/false, nullptr);
        if (st != SYMMICP_OK) { cleanup(); return st; }
        block = b.release();
    }
    CloudSoA cl;
    soa_from_block(block, n, cl);
    TargetIndex ix{};
    DevBuf<float> b_nrm, b_curv;
    bool ok = keep_alloc(c, (void **)&tq, sizeof(float4) * (n + 8)) == hipSuccess && keep_alloc(c, (void **)&tn, sizeof(float4) * 2 * n) == hipSuccess &&
              b_nrm.alloc_temp(c->arena, 3 * n) == hipSuccess && b_curv.alloc_temp(c->arena, n) == hipSuccess;
    if (!ok) { cleanup(); return fail(c, SYMMICP_ERR_HIP, "out of device memory"); }
    d_nrm = b_nrm.p; d_curv = b_curv.p;
    st = build_index(c, cl, (uint32_t)n, /*want_grid=*/false, tq, tn, &boxes, &cells, &ix, nullptr, nullptr);
    if (st != SYMMICP_OK) { cleanup(); return st; }
    const float vp0[3] = {0.f, 0.f, 0.f};
    launch_normals_knn(ix, k, viewpoint ? viewpoint : vp0, d_nrm, d_curv, c->stream);
    hipError_t e = hipMemcpyAsync(nrm_out, d_nrm, sizeof(float) * 3 * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && curv_out) e = hipMemcpyAsync(curv_out, d_curv, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    cleanup();
    if (e != hipSuccess) return fail(c, SYMMICP_ERR_HIP, std::string("normals: ") + hipGetErrorString(e));
    return SYMMICP_OK;
}

int symmicp_estimate_normals(int device, const float *xyz, size_t row_stride, size_t col_stride, size_t n, int k,
                             const float viewpoint[3], float *nrm_out, float *curv_out)
{
    if (!xyz || !nrm_out || n == 0 || n > 0x7fffffffull) return SYMMICP_ERR_ARG;
    if (k < 3 || k > 16 || (size_t)k > n) return SYMMICP_ERR_ARG;
    symmicp_config cfg;
    symmicp_config_default(&cfg);
    cfg.device = device;
    symmicp_ctx *c = nullptr;
    int st = symmicp_create(&cfg, &c);
    if (st != SYMMICP_OK) return st;
    st = symmicp_ctx_estimate_normals(c, xyz, row_stride, col_stride, n, k, viewpoint, nrm_out, curv_out);
    symmicp_destroy(c);
    return st;
}

// ---- stats -------------------------------------------------------------------------------------
int symmicp_enable_timing(symmicp_ctx *c, int on)
{
    if (!c) return SYMMICP_ERR_ARG;
    c->timing = on < 0 ? 0 : (on > 3 ? 3 : on);
    return SYMMICP_OK;
}

int symmicp_reset_stats(symmicp_ctx *c)
{
    if (!c) return SYMMICP_ERR_ARG;
    flush_events(c);
    c->st.last_pass_ms = c->st.sum_pass_ms = 0.0;
    c->st.passes = 0;
    c->st.passes_timed = 0;
    c->st.loop_passes = c->st.loop_straggler_passes = 0;
    c->st.allreduce_ms = 0.0; c->st.allreduce_timed = 0;
    for (int k = 0; k < 8; k++) { c->st.kernel_ms[k] = 0.0; c->st.kernel_launches[k] = 0; c->st.pass_ms_head[k] = 0.0; }
    return SYMMICP_OK;
}

int symmicp_get_stats(symmicp_ctx *c, symmicp_stats *out)
{
    if (!c || !out) return SYMMICP_ERR_ARG;
    flush_events(c);
    c->st.pass_blocks = c->pass_blocks;
    const bool incr = resolved_apply(c->cfg) == SYMMICP_APPLY_INCREMENTAL;
    int64_t b = 0;
    if (c->cfg.corr == SYMMICP_CORR_IDENTITY) b = (int64_t)c->n_loc * 48;
    else b = (int64_t)c->n_loc * (48 + 4 + 4) + (int64_t)c->n_t * 12;
    if (incr) b += (int64_t)c->n_loc * 24;
    c->st.bytes_algorithmic_per_pass = b;
    {
        uint32_t fb = 0;
        if (c->pkt_fallbacks && hipMemcpy(&fb, c->pkt_fallbacks, sizeof(fb), hipMemcpyDeviceToHost) == hipSuccess) c->st.packet_fallbacks = (int64_t)fb;
    }
    *out = c->st;
    return SYMMICP_OK;
}

}  // extern "C"
