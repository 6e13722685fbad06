// engine.cpp -- host side of libsymmicp: the C-ABI of include/symmicp.h.
//
// Host C++ owns control flow and the 6x6 / 3x3 solve (host_solve.cpp); the GPU owns
// every O(N) pass (kernels_pass.hip) and the one-time index build (kernels_build.hip).
// Per iteration of the reference loop (ICP/myicp.cpp:123-142) the host does:
//   solve(last sums) -> 4x4 increment -> launch the pass (one streaming kernel for identity pairing; search + accumulate
//   kernels for the nearest-neighbour modes) + the final reduce -> [RCCL all-reduce of 40 doubles when sharded] ->
//   spin on the record's sequence word in host-mapped memory -> repeat.
// There is no CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "symmicp.h"
#include "symmicp_internal.h"
#include "host_solve.h"

using namespace symmicp;

// ---- RCCL, loaded lazily (single-GPU users never touch it) ---------------------
namespace {
typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
typedef int (*fn_ncclGetUniqueId)(rcclUniqueId *);
typedef int (*fn_ncclCommInitRank)(rcclComm_t *, int, rcclUniqueId, int);
typedef int (*fn_ncclCommDestroy)(rcclComm_t);
typedef int (*fn_ncclAllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t);
typedef const char *(*fn_ncclGetErrorString)(int);
constexpr int kNcclFloat64 = 8;   // ncclDouble (rccl.h ncclDataType_t)
constexpr int kNcclSum = 0;       // ncclSum

struct Rccl {
    void *handle = nullptr;
    fn_ncclGetUniqueId GetUniqueId = nullptr;
    fn_ncclCommInitRank CommInitRank = nullptr;
    fn_ncclCommDestroy CommDestroy = nullptr;
    fn_ncclAllReduce AllReduce = nullptr;
    fn_ncclGetErrorString GetErrorString = nullptr;
    std::string err;
    std::mutex mu;
    bool load()
    {
        std::lock_guard<std::mutex> lock(mu);       // contexts of different host threads may attach communicators concurrently
        if (handle && AllReduce) return true;
        // prefer an RCCL already mapped into the process (e.g. the copy torch links against)
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *nm : names) { handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (handle) break; }
        if (!handle)
            for (const char *nm : names) { handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (handle) break; }
        if (!handle) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        GetUniqueId = (fn_ncclGetUniqueId)dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (fn_ncclCommInitRank)dlsym(handle, "ncclCommInitRank");
        CommDestroy = (fn_ncclCommDestroy)dlsym(handle, "ncclCommDestroy");
        AllReduce = (fn_ncclAllReduce)dlsym(handle, "ncclAllReduce");
        GetErrorString = (fn_ncclGetErrorString)dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce) { err = "librccl misses nccl* symbols"; return false; }
        return true;
    }
};
Rccl g_rccl;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// Every SYMMICP_* environment switch, read ONCE in symmicp_create (A/B runs and tests; none is needed in production -- DESIGN.md 6 lists
// them).  Nothing on the pass loop calls getenv.
struct Switches {
    bool allow_any_arch = false, debug_host = false, debug_counters = false;
    std::string debug_trace;               // per-packet trace file of the first pass ("" = off)
    double grid_ppc = 2.0;                 // points per occupied cell the grid level is chosen for
    int grid_maxlevel = kMortonBits, grid_level = -1;      // -1: chosen from the cloud; 0 disables the grid phase
    int first_pass = -1;                   // -1: decided per target (build_index); 0: per-thread walk; 1: packets
    int oct_leaf = 0;                      // octree leaf size (0: 16 on surface-like targets, 8 otherwise)
    bool packet_order = true;              // packets started longest-first
    float packet_jump = -1.0f;             // cut factor of k_packet_runs (< 0: the default, 0: never cut)
    int packet_key_bits = 16;
    uint32_t packet_chunk = 0, packet_lds_pad = 0, packet_waves = 0, packet_front_cap = 0;
    bool no_hood = false, no_cert = false, walk_full_grid = false, host_loop = false, no_loop_stragglers = false, force_comm = false;
    int budget_walk = -1, optimistic = -1, compact = -1;      // -1 auto, 0 never, 1 always
    int pass_blocks = 2048, id_blocks = 2048, acc_blocks = 512, fused_blocks = 512, compact_blocks = 1280;
    PassTuning tune;                       // wave_mode_max, cells_chunk, walk_budget
};

static void read_switches(Switches &w)
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
    w.no_hood = flag("SYMMICP_NO_NEIGHBOURHOOD");
    w.no_cert = flag("SYMMICP_NO_CERT");
    w.walk_full_grid = flag("SYMMICP_WALK_FULL_GRID");
    { const char *e = std::getenv("SYMMICP_HOST_LOOP"); w.host_loop = e && e[0] == '1'; }
    w.no_loop_stragglers = flag("SYMMICP_NO_LOOP_STRAGGLERS");
    w.force_comm = flag("SYMMICP_FORCE_COMM");
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
    w.tune.walk_budget = (uint32_t)num("SYMMICP_WALK_BUDGET", 160);
}

// Scratch arena of a context: the builds need dozens of temporaries, and every hipFree costs ~100 us (it synchronises the
// device) -- half of a set_target + set_source at 1M points.  Temporaries are bump-allocated from one block that is
// rewound at the start of each public call and only ever grows; persistent results are hipMalloc'ed as before.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
};

// Intra-node exchange of the 40-double record through POSIX shared memory (symmicp_comm_init_shm): every rank spins on
// its own GPU's record as in the single-GPU path, publishes it in its slot, waits for the other ranks' slots and adds
// them up in rank order (so every rank gets bit-identical sums).  For a 320-byte latency-bound exchange this beats a
// collective kernel launch; slots are double-buffered by exchange parity (a rank cannot be more than one exchange ahead).
struct ShmSlot {
    volatile unsigned long long seq;
    double s[SYMMICP_NSUM];
    char pad[512 - 8 - 8 * SYMMICP_NSUM];
};
static_assert(sizeof(ShmSlot) == 512, "slot = 4 cache lines");
struct ShmExchange {
    ShmSlot *slots = nullptr;        // [2][nranks]
    size_t bytes = 0;
    unsigned long long count = 0;    // exchanges done
    std::string name;
    bool owner = false;
};

struct symmicp_ctx {
    Switches sw;                     // environment switches as they stood at symmicp_create
    Arena arena;                     // temporaries of one public call
    Arena keep;                      // the target's persistent arrays (reused by the next set_target)
    std::vector<void *> keep_extra;  // ... and those that did not fit
    ShmExchange shm;
    symmicp_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // communicator
    int nranks = 1, rank = 0;
    rcclComm_t comm = nullptr;
    bool external_exchange = false;  // sharded, but the application sums the records (symmicp_set_sums)
    bool sums_exchanged = false;     // ... and has done so for the last pass
    // target
    uint32_t n_t = 0;
    float *tgt_block = nullptr;      // 6 planar arrays
    CloudSoA tgt{};
    float4 *tq = nullptr, *tn = nullptr;
    float4 *boxes = nullptr;
    float4 *onodes = nullptr;
    uint2 *cells = nullptr;
    uint32_t *ctop = nullptr;
    unsigned long long *dbg = nullptr;   // debug counters (SYMMICP_DEBUG_COUNTERS)
    unsigned long long *dbg_trace = nullptr;   // per-packet trace of the first pass (SYMMICP_DEBUG_TRACE=file)
    TargetIndex ix{};
    bool have_index = false;
    bool target_surface_like = false;   // decides the first-pass regime (build_index)
    float pivot[3] = {0, 0, 0};
    // source share
    uint32_t n_s_total = 0, n_loc = 0, src_off = 0;
    char *src_all = nullptr;         // one allocation behind all per-source arrays below (reused by the next set_source)
    size_t src_all_cap = 0;
    float *src0_block = nullptr, *cur_block = nullptr;
    CloudSoA src0{}, cur{};
    uint32_t *src_order = nullptr;   // share position -> row in the caller's cloud (null = identity)
    int32_t *pos = nullptr;
    float *d2 = nullptr;
    float4 *pairrec = nullptr;       // TREE: per pair, its own copy of the target's (point, normal) record
    float *cert = nullptr;           // TREE pair certificates: one float4 (ref.xyz, clear radius) per source point
    uint32_t *certk = nullptr;       // ... and their neighbourhood certificates: 8 member words per source point
    float *hoodr = nullptr;          // ... (T, radius hint) per source point
    uint32_t *pkt_tab = nullptr;     // TREE: the first pass's packets, (first query, count) in start order (widest first)
    uint32_t pkt_count = 0;
    uint32_t *pkt_fallbacks = nullptr;  // device counter: packets of first passes that finished depth-first (k_search_packet)
    unsigned long long *best64 = nullptr;
    uint32_t *worklist = nullptr, *wl_count = nullptr;   // the sharded work list + its counters
    WorkLists wl{};
    // reduction
    int pass_blocks = 0;
    double *partials = nullptr, *d_sums = nullptr, *h_sums = nullptr, *h_sums_dev = nullptr;   // h_sums: 40 doubles + sequence word
    uint32_t *ticket = nullptr;          // ticket of the final reduce
    long long last_list_len = -1;        // work-list length of the last pass (all ranks); -1 = unknown (full walk grid)
    long long last_uncertified = -1;     // TREE: pairs the last pass had to search again (all ranks); -1 = unknown
    // device-driven runs of passes (run_batch): loop state in device memory, per-pass records + end flag in host-mapped memory
    static constexpr int kRing = 65;
    LoopState *d_loop = nullptr, *h_loop = nullptr, *h_loop_dev = nullptr;
    LoopRecord *h_ring = nullptr, *h_ring_dev = nullptr;
    unsigned long long *h_done = nullptr, *h_done_dev = nullptr;
    unsigned long long batch_seq = 0;
    int host_passes_since_bailout = 1000;   // batches resume after two clean host-driven passes
    unsigned long long seq = 0;
    // loop state
    bool begun = false;
    int iters = 0;
    float X[16];
    symmicp_sums last{};
    // stats
    int timing = 0;                  // 0 off, 1 two events per pass, 2 events around every kernel of a pass
    // timing mode: 6 events per pass in a ring of kEvRing passes, resolved lazily (no sync inside the loop)
    static constexpr int kEvRing = 64, kEvPer = 8;      // per pass: 0..4 the kernels, 5 the reduce, 6..7 around the collective (sharded runs)
    hipEvent_t ev[kEvRing * kEvPer] = {};
    int ev_split[kEvRing] = {};      // 1 = split TREE pass (5 kernels), 0 = single pass kernel
    int ev_weight[kEvRing] = {};     // passes this entry stands for (timing mode 3 samples the passes of a device-driven run)
    int ev_coll[kEvRing] = {};       // 1: events 6 and 7 bracket this pass's all-reduce
    int ev_used = 0;
    symmicp_stats st{};
    // host-side timing of the pass loop, printed by symmicp_destroy under SYMMICP_DEBUG_HOST
    double t_launch = 0, t_spin = 0, t_between = 0, t_last_done = 0;
    long n_pass_timed = 0;
};

// device buffer freed on every exit path.  alloc(): its own hipMalloc (release() hands the pointer to the context);
// alloc_temp(): from the context's arena when it fits (nothing to free), else its own hipMalloc.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p && owned) hipFree(p); }
    hipError_t alloc(size_t count) { owned = true; return hipMalloc((void **)&p, sizeof(T) * (count ? count : 1)); }
    hipError_t alloc_temp(Arena &a, size_t count)
    {
        const size_t bytes = ((sizeof(T) * (count ? count : 1)) + 255) & ~(size_t)255;
        if (a.base && a.off + bytes <= a.cap) { p = reinterpret_cast<T *>(a.base + a.off); a.off += bytes; owned = false; return hipSuccess; }
        return alloc(count);
    }
    T *release() { T *q = p; p = nullptr; return q; }           // (owned buffers only)
};

// persistent allocation for the target: from the context's keep-arena when it fits, else its own hipMalloc (tracked)
static hipError_t keep_alloc(symmicp_ctx *c, void **out, size_t bytes);

// rewind the arena and make sure it holds `want` bytes (contents are dead: called at the start of a public call)
static void arena_begin(Arena &a, size_t want)
{
    a.off = 0;
    if (a.cap >= want) return;
    if (a.base) hipFree(a.base);
    a.base = nullptr; a.cap = 0;
    const size_t cap = want + want / 4;
    if (hipMalloc((void **)&a.base, cap) == hipSuccess) a.cap = cap;      // on failure every alloc_temp falls back to hipMalloc
    else (void)hipGetLastError();
}

#define HIP_TRY(ctx, call)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                \
            return SYMMICP_ERR_HIP;                                                                         \
        }                                                                                                   \
    } while (0)


static void shm_close(symmicp_ctx *c);

static int fail(symmicp_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

static void soa_from_block(float *block, size_t n, CloudSoA &s)
{
    s.x = block; s.y = block + n; s.z = block + 2 * n; s.nx = block + 3 * n; s.ny = block + 4 * n; s.nz = block + 5 * n;
}

static void identity16(float X[16])
{
    for (int k = 0; k < 16; k++) X[k] = (k % 5 == 0) ? 1.f : 0.f;
}

static int resolved_apply(const symmicp_config &c)
{
    if (c.apply == SYMMICP_APPLY_INCREMENTAL || c.apply == SYMMICP_APPLY_CUMULATIVE) return c.apply;
    return c.mode == SYMMICP_MODE_QUIRKS ? SYMMICP_APPLY_INCREMENTAL : SYMMICP_APPLY_CUMULATIVE;
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

static hipError_t keep_alloc(symmicp_ctx *c, void **out, size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (!bytes) bytes = 256;
    if (c->keep.base && c->keep.off + bytes <= c->keep.cap) { *out = c->keep.base + c->keep.off; c->keep.off += bytes; return hipSuccess; }
    void *p = nullptr;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) { c->keep_extra.push_back(p); *out = p; }
    return e;
}

static void free_target(symmicp_ctx *c)
{
    // (the arrays live in the keep-arena, which stays allocated for the next target)
    for (void *p : c->keep_extra) hipFree(p);
    c->keep_extra.clear();
    c->keep.off = 0;
    hipFree(c->dbg); hipFree(c->dbg_trace); c->dbg_trace = nullptr;
    c->ctop = nullptr; c->dbg = nullptr; c->ix = TargetIndex{}; c->tgt_block = nullptr; c->tq = nullptr; c->tn = nullptr; c->boxes = nullptr; c->cells = nullptr; c->onodes = nullptr;
    c->have_index = false; c->n_t = 0;
}

static void forget_source(symmicp_ctx *c)
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
        uint32_t leaf_max = c->target_surface_like ? 16u : 8u;
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
static void flush_events(symmicp_ctx *c)
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

static void shm_close(symmicp_ctx *c)
{
    if (c->shm.slots) munmap((void *)c->shm.slots, c->shm.bytes);
    if (c->shm.owner && !c->shm.name.empty()) shm_unlink(c->shm.name.c_str());
    c->shm = ShmExchange{};
}

// rec[40]: this rank's record in, the sum over ranks out
static int shm_exchange(symmicp_ctx *c, double *rec)
{
    ShmExchange &x = c->shm;
    const unsigned long long k = ++x.count;
    ShmSlot *buf = x.slots + (size_t)(k & 1ull) * c->nranks;
    ShmSlot &mine = buf[c->rank];
    for (int j = 0; j < kNSum; j++) mine.s[j] = rec[j];
    __atomic_store_n(&mine.seq, k, __ATOMIC_RELEASE);
    double tot[kNSum];
    for (int j = 0; j < kNSum; j++) tot[j] = 0.0;
    const double t0 = now_s();
    for (int r = 0; r < c->nranks; r++) {
        unsigned spins = 0;
        while (__atomic_load_n(&buf[r].seq, __ATOMIC_ACQUIRE) != k) {
            __builtin_ia32_pause();
            if ((++spins & 0xFFFFu) == 0 && now_s() - t0 > 60.0) return fail(c, SYMMICP_ERR_COMM, "shared-memory exchange: a rank did not arrive within 60 s");
        }
        for (int j = 0; j < kNSum; j++) tot[j] += buf[r].s[j];
    }
    for (int j = 0; j < kNSum; j++) rec[j] = tot[j];
    return SYMMICP_OK;
}

// SYMMICP_DEBUG_TRACE=file: the per-packet trace of the first pass that has just run (k_search_packet), then cleared for the next one
static constexpr size_t kTraceWords = (size_t)1 << 22;
static void dump_packet_trace(symmicp_ctx *c)
{
    if (FILE *f = std::fopen(c->sw.debug_trace.c_str(), "wb")) {
        std::vector<unsigned long long> t(kTraceWords);
        hipMemcpy(t.data(), c->dbg_trace, t.size() * 8, hipMemcpyDeviceToHost);
        std::fwrite(t.data(), 8, t.size(), f);
        std::fclose(f);
    }
    hipMemset(c->dbg_trace, 0, kTraceWords * 8);
}

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
    // neighbourhoods are worth their stores once the alignment is settling (the previous pass searched under half of the pairs)
    a.make_hood = (a.certk && !first && c->last_uncertified >= 0 && c->last_uncertified < (long long)(c->n_s_total / 2)) ? 1 : 0;
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
    if (c->ix.dbg) {
        unsigned long long h[12];
        hipMemcpy(h, c->ix.dbg, sizeof(h), hipMemcpyDeviceToHost);
        hipMemset(c->ix.dbg, 0, sizeof(h));
        if (first && c->cfg.corr == SYMMICP_CORR_TREE && c->target_surface_like)
            std::fprintf(stderr, "[symmicp dbg] pass %lld (packets): steps=%llu nodes+leaf candidates=%llu leaf candidates rejected=%llu points=%llu tie rescans=%llu overflows=%llu | packet ticks (10 ns): sum=%llu max=%llu\n",
                         (long long)c->st.passes, h[0], h[4], h[5], h[3], h[2], h[1], h[7], h[6]);
        else
        std::fprintf(stderr, "[symmicp dbg] pass %lld: cells: certified=%llu (by neighbourhood %llu) scans=%llu (neighbourhoods kept %llu, not kept %llu) probes=%llu to-walk=%llu items=%llu points=%llu | walk list=%lld visits=%llu wave-max*64=%llu\n",
                     (long long)c->st.passes, h[1], h[8], h[2], h[9], h[10], h[6], h[7], h[0], h[3], list_len, h[4], h[5]);
    }
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
        chunk *= 2;
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

// ---- multi-GPU ---------------------------------------------------------------------------------
int symmicp_comm_get_unique_id(void *out128)
{
    if (!out128) return SYMMICP_ERR_ARG;
    if (!g_rccl.load()) return SYMMICP_ERR_COMM;
    rcclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != 0) return SYMMICP_ERR_COMM;
    std::memcpy(out128, &id, SYMMICP_UNIQUE_ID_BYTES);
    return SYMMICP_OK;
}

int symmicp_shard_range(size_t n, int nranks, int rank, size_t *begin, size_t *count)
{
    if (!begin || !count || nranks < 1 || rank < 0 || rank >= nranks) return SYMMICP_ERR_ARG;
    const uint64_t b0 = (uint64_t)n * (uint64_t)rank / (uint64_t)nranks;
    const uint64_t b1 = (uint64_t)n * (uint64_t)(rank + 1) / (uint64_t)nranks;
    *begin = (size_t)b0;
    *count = (size_t)(b1 - b0);
    return SYMMICP_OK;
}

int symmicp_comm_init_rank(symmicp_ctx *c, int nranks, int rank, const void *uid)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, SYMMICP_ERR_ARG, "bad rank/nranks");
    if (c->src0_block) return fail(c, SYMMICP_ERR_STATE, "attach the communicator before symmicp_set_source");
    // a 1-rank communicator is legal RCCL; it is only built on request (exercises the RCCL path on one GPU)
    if (nranks == 1 && !c->sw.force_comm) { c->nranks = 1; c->rank = 0; return SYMMICP_OK; }
    if (!uid) {
        // external exchange: shard like a rank of `nranks`, publish local records, the application sums them (symmicp_set_sums)
        if (nranks == 1) return fail(c, SYMMICP_ERR_ARG, "null unique id");
        c->nranks = nranks; c->rank = rank; c->external_exchange = true;
        return SYMMICP_OK;
    }
    if (!g_rccl.load()) return fail(c, SYMMICP_ERR_COMM, g_rccl.err);
    HIP_TRY(c, hipSetDevice(c->device));
    rcclUniqueId id;
    std::memcpy(&id, uid, SYMMICP_UNIQUE_ID_BYTES);
    int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) return fail(c, SYMMICP_ERR_COMM, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"));
    c->nranks = nranks;
    c->rank = rank;
    c->external_exchange = false;
    return SYMMICP_OK;
}

int symmicp_comm_init_shm(symmicp_ctx *c, int nranks, int rank, const char *job_name)
{
    if (!c) return SYMMICP_ERR_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks || !job_name || !*job_name) return fail(c, SYMMICP_ERR_ARG, "bad rank/nranks/name");
    if (c->src0_block) return fail(c, SYMMICP_ERR_STATE, "attach the exchange before symmicp_set_source");
    if (c->comm || c->shm.slots) return fail(c, SYMMICP_ERR_STATE, "a communicator is already attached");
    std::string name = std::string("/symmicp_") + job_name;
    for (char &ch : name) if (ch == '/' && &ch != &name[0]) ch = '_';
    const size_t bytes = sizeof(ShmSlot) * 2 * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name.c_str());                                  // a leftover of a crashed job with the same name
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name.c_str()); fd = -1; }
    } else {
        // wait for rank 0 to create and size the segment
        const double t0 = now_s();
        while (now_s() - t0 < 60.0) {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
                close(fd); fd = -1;
            }
            usleep(1000);
        }
    }
    if (fd < 0) return fail(c, SYMMICP_ERR_COMM, "cannot open shared-memory segment " + name);
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(c, SYMMICP_ERR_COMM, "mmap of " + name + " failed");
    c->shm.slots = static_cast<ShmSlot *>(p);
    c->shm.bytes = bytes;
    c->shm.name = name;
    c->shm.owner = (rank == 0);
    c->shm.count = 0;
    c->nranks = nranks;
    c->rank = rank;
    c->external_exchange = false;
    return SYMMICP_OK;
}

int symmicp_set_sums(symmicp_ctx *c, const symmicp_sums *total)
{
    if (!c || !total) return SYMMICP_ERR_ARG;
    if (!c->external_exchange) return fail(c, SYMMICP_ERR_STATE, "symmicp_set_sums needs external exchange (comm_init_rank with a null id)");
    if (!c->begun) return fail(c, SYMMICP_ERR_STATE, "no pass has run yet");
    c->last = *total;
    c->sums_exchanged = true;
    return SYMMICP_OK;
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
