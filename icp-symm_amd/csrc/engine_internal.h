// engine_internal.h -- what the host-side translation units of libsymmicp share: the context, its scratch arenas, the environment
// switches, the RCCL loader and the helpers every entry point uses.
//   engine.cpp           lifetime, configuration, uploads, index build, set_target / set_source, getters, normals pre-step, statistics
//   engine_loop.cpp      the iteration loop: one pass (run_pass), device-driven runs of passes (run_batch), begin / step / align, result block
//   engine_exchange.cpp  multi-GPU: RCCL (loaded lazily), shared-memory exchange, communicator entry points
//   engine_debug.cpp     SYMMICP_DEBUG_COUNTERS / SYMMICP_DEBUG_TRACE dumps
#pragma once
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


namespace symmicp {}
using namespace symmicp;

// ---- RCCL, loaded lazily (single-GPU users never touch it) ---------------------
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
extern Rccl g_rccl;      // (engine_exchange.cpp)

inline double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Every SYMMICP_* environment switch, read ONCE in symmicp_create (A/B runs and tests; none is needed in production -- DESIGN.md 6 lists
// them).  Nothing on the pass loop calls getenv.
struct Switches {
    bool allow_any_arch = false, debug_host = false, debug_counters = false;
    std::string debug_trace;               // per-packet trace file of the first pass ("" = off)
    double grid_ppc = 3.0;                 // points per occupied cell the grid level is chosen for (2.0 until the end of round 3: one level finer on the surface pairs -- 15 (query, cell) items of ~1 point per probe instead of 8 of ~4; 1M pair 12 740 -> 13 230 iter/s at 20 iterations, 250k / 4M +3 %, scan pairs and the 100k cube unchanged)
    int grid_maxlevel = kMortonBits, grid_level = -1;      // -1: chosen from the cloud; 0 disables the grid phase
    int first_pass = -1;                   // -1: decided per target (build_index); 0: per-thread walk; 1: packets
    int oct_leaf = 0;                      // octree leaf size (0: 16 on surface-like targets, 8 otherwise)
    bool packet_order = true;              // packets started longest-first
    bool packet_cost_key = true;           // ... keyed by their distance to the target when it is known at set_source (0: by radius alone)
    float packet_jump = -1.0f;             // cut factor of k_packet_runs (< 0: the default, 0: never cut)
    int packet_key_bits = 16;
    uint32_t packet_chunk = 0, packet_lds_pad = 0, packet_waves = 0, packet_front_cap = 0;
    double hood_frac = 0.8;                // scans keep neighbourhoods once the previous pass searched under this fraction of the pairs
    bool no_hood = false, no_cert = false, walk_full_grid = false, host_loop = false, no_loop_stragglers = false, force_comm = false;
    int budget_walk = -1, optimistic = -1, compact = -1;      // -1 auto, 0 never, 1 always
    int pass_blocks = 2048, id_blocks = 2048, acc_blocks = 512, fused_blocks = 512, compact_blocks = 1280;
    PassTuning tune;                       // wave_mode_max, cells_chunk, walk_budget
};

void read_switches(Switches &w);

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
hipError_t keep_alloc(symmicp_ctx *c, void **out, size_t bytes);
// rewind the arena and make sure it holds `want` bytes (contents are dead: called at the start of a public call)
void arena_begin(Arena &a, size_t want);

#define HIP_TRY(ctx, call)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                \
            return SYMMICP_ERR_HIP;                                                                         \
        }                                                                                                   \
    } while (0)



inline int fail(symmicp_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

inline void soa_from_block(float *block, size_t n, CloudSoA &s)
{
    s.x = block; s.y = block + n; s.z = block + 2 * n; s.nx = block + 3 * n; s.ny = block + 4 * n; s.nz = block + 5 * n;
}

inline void identity16(float X[16])
{
    for (int k = 0; k < 16; k++) X[k] = (k % 5 == 0) ? 1.f : 0.f;
}

inline int resolved_apply(const symmicp_config &c)
{
    if (c.apply == SYMMICP_APPLY_INCREMENTAL || c.apply == SYMMICP_APPLY_CUMULATIVE) return c.apply;
    return c.mode == SYMMICP_MODE_QUIRKS ? SYMMICP_APPLY_INCREMENTAL : SYMMICP_APPLY_CUMULATIVE;
}


// engine.cpp
void free_target(symmicp_ctx *c);
void forget_source(symmicp_ctx *c);
void flush_events(symmicp_ctx *c);
// engine_exchange.cpp
void shm_close(symmicp_ctx *c);
int shm_exchange(symmicp_ctx *c, double *rec);      // rec[40]: this rank's record in, the sum over ranks out
// engine_debug.cpp
void dump_packet_trace(symmicp_ctx *c);
void print_pass_counters(symmicp_ctx *c, bool first, long long list_len);
