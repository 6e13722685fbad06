// engine_exchange.cpp -- multi-GPU side of libsymmicp (new: the reference is single-threaded): the RCCL loader, the shared-memory
// exchange of the 40-double record between ranks of one node, and the communicator entry points.
#include "engine_internal.h"

Rccl g_rccl;

void shm_close(symmicp_ctx *c)
{
    if (c->shm.slots) munmap((void *)c->shm.slots, c->shm.bytes);
    if (c->shm.owner && !c->shm.name.empty()) shm_unlink(c->shm.name.c_str());
    c->shm = ShmExchange{};
}

int shm_exchange(symmicp_ctx *c, double *rec)
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


extern "C" {

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

}  // extern "C"
