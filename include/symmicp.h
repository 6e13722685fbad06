/*
 * symmicp.h -- C-ABI of libsymmicp, the MI355X (gfx950) symmetric-ICP engine.
 *
 * This is the drop-in boundary for the ONE hot path of StephenNG59/ICP-symm:
 * the MyICP::RegisterSymm() iteration loop (reference ICP/myicp.cpp:100-150)
 * and the free functions it calls (reference ICP/func.cpp:19-121).  Every
 * entry point below names the reference interface it replaces.  Plain
 * pointers and sizes only; no C++ or torch types.  The C++ class of the
 * reference (ICP/myicp.h:7-36) is mirrored on top of this ABI by
 * include/myicp.h.
 *
 * Threading: one ctx = one host thread = one GPU (one rank).  Multi-GPU runs
 * are one process (or thread) per GPU, joined by symmicp_comm_init_rank (RCCL)
 * or symmicp_comm_init_shm (ranks of one node).
 * Errors: every call returns SYMMICP_OK or an error code; the message is
 * available from symmicp_last_error().  The library never falls back to a
 * CPU path: without a usable HIP device symmicp_create fails.
 */
#ifndef SYMMICP_H
#define SYMMICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYMMICP_NSUM 40            /* doubles per reduction record, see symmicp_sums layout */
#define SYMMICP_UNIQUE_ID_BYTES 128

typedef enum {
    SYMMICP_OK = 0,
    SYMMICP_ERR_ARG = 1,           /* null / out-of-range argument */
    SYMMICP_ERR_SIZE = 2,          /* empty cloud, or N_s != N_t with identity pairing (func.cpp:21 assert) */
    SYMMICP_ERR_DEGENERATE = 3,    /* rank-deficient system / non-finite transform (func.cpp:70,96 produce NaN) */
    SYMMICP_ERR_IO = 4,
    SYMMICP_ERR_HIP = 5,           /* HIP runtime error or no gfx950-capable device */
    SYMMICP_ERR_STATE = 6,         /* call out of order (e.g. step before begin) */
    SYMMICP_ERR_COMM = 7           /* RCCL or shared-memory exchange error */
} symmicp_status;

/* Arithmetic mode.  QUIRKS reproduces the reference exactly as written:
 * un-centred rows (func.cpp:51-58), alternating a-then-t 3x3 solves seeded
 * with qbar-pbar (func.cpp:86-88), reversed composition (func.cpp:95-99) and
 * the full affine applied to the normals (myicp.cpp:137).  PAPER is the
 * formulation the reference's own comments intend (func.cpp:84,94;
 * Rusinkiewicz 2019): centred rows, joint 6x6 solve, T(q)RT(t)RT(-p),
 * normals rotated only.  P2P is the closed-form point-to-point fit of the reference's
 * regist.h:8-72 (registrateNPoint: centroids, 3x3 cross-covariance, SVD, reflection fix) run as an
 * ICP loop -- what the RegisterP2P stub (myicp.cpp:43-59) was heading for; record slots 0..8 then hold
 * sum p q^T instead of the symmetric-objective Gram matrix. */
typedef enum { SYMMICP_MODE_QUIRKS = 0, SYMMICP_MODE_PAPER = 1, SYMMICP_MODE_P2P = 2 } symmicp_mode;

/* Correspondence.  IDENTITY is what the reference does (myicp.cpp:130, the
 * search is a todo at :128-131).  BRUTE and TREE are exact nearest neighbour
 * (ties -> lowest target index) by LDS-tiled brute force, or by the index
 * built over the Morton-sorted target (cell table + sparse octree; once an
 * alignment has converged, per-pair certificates prove the pair unchanged and
 * the search is skipped). */
typedef enum { SYMMICP_CORR_IDENTITY = 0, SYMMICP_CORR_BRUTE = 1, SYMMICP_CORR_TREE = 2 } symmicp_corr;

/* How the source is advanced.  INCREMENTAL rewrites source points and normals
 * with each incremental transform, as applyTransform does (func.cpp:104-121,
 * myicp.cpp:136-137), so the fp32 rounding history matches the reference.
 * CUMULATIVE never rewrites the source: each pass applies the accumulated 4x4
 * to the original points (<= ~1e-6 relative drift, 24 B/point less traffic). */
typedef enum { SYMMICP_APPLY_DEFAULT = 0, SYMMICP_APPLY_INCREMENTAL = 1, SYMMICP_APPLY_CUMULATIVE = 2 } symmicp_apply;

typedef struct {
    int32_t struct_size;       /* = sizeof(symmicp_config), checked */
    int32_t device;            /* HIP device ordinal; -1 = current */
    int32_t mode;              /* symmicp_mode */
    int32_t corr;              /* symmicp_corr */
    int32_t apply;             /* symmicp_apply; DEFAULT = INCREMENTAL for QUIRKS, CUMULATIVE for PAPER */
    int32_t max_iters;         /* myicp.cpp:6   default 10 */
    float diff_threshold;      /* myicp.cpp:6   default 1.0, loop runs while diff > threshold (myicp.cpp:123) */
    float max_corr_dist;       /* <= 0: keep every pair; else pairs farther than this are dropped */
    int32_t fixed_iters;       /* != 0: ignore the threshold, run exactly max_iters iterations */
    int32_t sort_source;       /* != 0 (default for BRUTE/TREE): Morton-sort the source shard for locality */
    int32_t verbose;           /* != 0: print the reference's stdout lines (myicp.cpp:125-126,146-149) */
    /* robustness options of the paper-style loop; all off by default = reference behaviour */
    float min_normal_dot;      /* > -1: drop pairs whose (transformed) source normal . target normal is below this */
    float eps_rotation;        /* > 0 (radians) together with eps_translation > 0: also stop once an increment */
    float eps_translation;     /*   rotates by less than eps_rotation and translates by less than eps_translation */
    int32_t host_loop;         /* != 0: symmicp_align never hands runs of iterations to the device (every solve on the host, as symmicp_step does) */
    int32_t reserved[1];
} symmicp_config;

/* One reduction record = everything the host needs from one pass over the
 * source.  v_i = (M_i, N_i) is the 6-vector of
 * func.cpp:54,56 and c_i the scalar of func.cpp:58.
 *   [0..20]  upper triangle, row-major, of sum_i v_i v_i^T
 *   [21..26] sum_i v_i c_i
 *   [27..29] sum_i p_i        [30..32] sum_i q_i   (about `pivot`, see symmicp_get_pivot)
 *   [33] sum_i |p_i - q_i|   (evalDiff, func.cpp:19-32, over the current pairs)
 *   [34] number of pairs     [35] sum_i c_i^2      [36] sum_i |p_i - q_i|^2
 *   [37..39] reserved (0) */
typedef struct { double s[SYMMICP_NSUM]; } symmicp_sums;

typedef struct {
    int32_t status;            /* symmicp_status of this iteration */
    int32_t iter;              /* iterations completed so far */
    float diff;                /* sum |p-q| after this iteration's update (myicp.cpp:141) */
    float rcond;               /* smallest/largest eigenvalue of the solved system(s) */
    double pairs;              /* correspondences that entered the sums */
    float increment[16];       /* row-major 4x4 of this iteration (func.cpp:91-101) */
    symmicp_sums sums;         /* record the NEXT solve will use (already all-reduced) */
} symmicp_iter_result;

typedef struct {
    int32_t status;
    int32_t iters;             /* iterations run */
    float diff_initial;        /* myicp.cpp:122 */
    float diff_final;
    float transform[16];       /* row-major 4x4, original source -> target (myicp.cpp:138,147) */
    float diffs[64];           /* diff printed at the top of each of the first 64 iterations (myicp.cpp:126) */
    double seconds_total;      /* wall time of the loop */
} symmicp_result;

typedef struct symmicp_ctx symmicp_ctx;

/* ---- lifetime ---------------------------------------------------------- */
void symmicp_config_default(symmicp_config *cfg);                 /* MyICP::MyICP, myicp.cpp:6 */
int symmicp_create(const symmicp_config *cfg, symmicp_ctx **out); /* MyICP::MyICP, myicp.cpp:6-14 */
void symmicp_destroy(symmicp_ctx *ctx);                           /* MyICP::~MyICP, myicp.cpp:16-18 */
const char *symmicp_last_error(const symmicp_ctx *ctx);           /* (reference has none: asserts / silent NaN) */
int symmicp_set_config(symmicp_ctx *ctx, const symmicp_config *cfg);   /* myicp.h:19 "todo add params" */
int symmicp_version(void);

/* ---- clouds (replaces pasteInMatrix, func.cpp:5-15, myicp.cpp:110-111) --
 * Host arrays, element (i,k) at base[i*row_stride + k*col_stride] (floats):
 *   packed xyz AoS        row_stride=3, col_stride=1
 *   pcl::PointXYZ (16 B)  row_stride=4, col_stride=1
 *   pcl::PointNormal      xyz: base=&pt[0].x, 12,1 ; normals: base=&pt[0].normal_x, 12,1
 *   Eigen::MatrixXf Nx3   row_stride=1, col_stride=N   (column-major)
 * Data is copied to the device; the caller keeps ownership.  With a
 * communicator attached every rank passes the FULL cloud (same pointer arithmetic on every rank) and uploads only its own
 * share: rows [begin, begin + count) of symmicp_shard_range, Morton-sorted on the device; the target is replicated.
 * set_target also builds the search index when corr != IDENTITY. */
int symmicp_set_source(symmicp_ctx *ctx, const float *xyz, size_t xyz_row_stride, size_t xyz_col_stride,
                       const float *nrm, size_t nrm_row_stride, size_t nrm_col_stride, size_t n);
int symmicp_set_target(symmicp_ctx *ctx, const float *xyz, size_t xyz_row_stride, size_t xyz_col_stride,
                       const float *nrm, size_t nrm_row_stride, size_t nrm_col_stride, size_t n);

/* ---- the loop (replaces MyICP::RegisterSymm, myicp.cpp:117-142) -------- */
/* align = begin + step until the stop rule of myicp.cpp:123. guess16 may be NULL (identity). */
int symmicp_align(symmicp_ctx *ctx, const float *guess16, symmicp_result *out);
/* begin: evaluate the initial pairs/sums/diff (myicp.cpp:122); no update yet. */
int symmicp_begin(symmicp_ctx *ctx, const float *guess16, symmicp_iter_result *out);
/* step: one trip of the loop body: solve (func.cpp:76-102) -> compose -> apply
 * (func.cpp:104-121) -> new pairs + sums + diff (myicp.cpp:128-141). */
int symmicp_step(symmicp_ctx *ctx, symmicp_iter_result *out);
int symmicp_get_transform(const symmicp_ctx *ctx, float out16[16]);           /* myicp.cpp:147 */
/* The result block exactly as the reference prints it (myicp.cpp:146-149: "Result transform:" + transform.matrix(), "  rotation:" +
 * transform.rotation(), "  translation:" + transform.translation(), each through Eigen's default IOFormat: precision 6, every
 * coefficient right-aligned to the widest one of its matrix, one space between columns).  Writes at most cap - 1 characters and a
 * terminating 0; returns the length of the whole text (buf may be NULL to query it).  symmicp_align prints this when cfg.verbose. */
size_t symmicp_format_result(const float transform16[16], char *buf, size_t cap);
int symmicp_get_pivot(const symmicp_ctx *ctx, float out3[3]);
/* current pairs in ORIGINAL numbering: idx[i] = target row paired with source row i
 * (rows of this rank's share; -1 = rejected), d2[i] = squared distance. Either may be NULL.  (SYMMICP_CORR_IDENTITY and _TREE evaluate
 * the distances here, at the positions the last pass gave the points: a pass stores them only for the pairs it searched.) */
int symmicp_get_correspondences(symmicp_ctx *ctx, int32_t *idx, float *d2, size_t cap);
/* current (transformed) source points / normals of this rank's share, original row order, packed AoS. */
int symmicp_get_source(symmicp_ctx *ctx, float *xyz, float *nrm, size_t cap);
/* diagnostic (SYMMICP_CORR_TREE): the pair certificates of this rank's share in its sorted order: cert4 [n_loc][4] = position of the
 * query when its pair was last searched + the clear radius L (<= 0: no single certificate; bit 0 of the word: the neighbourhood is
 * valid); hood8 (may be NULL) [n_loc][8]: the neighbourhood's members as target rows (0xFFFFFFFF: empty); hood_radius (may be NULL)
 * [n_loc]: its radius T; winner_row (may be NULL) [n_loc]: the pair's current target row (-1: none) */
int symmicp_get_certificates(symmicp_ctx *ctx, float *cert4, uint32_t *hood8, float *hood_radius, int32_t *winner_row, size_t cap);
size_t symmicp_local_source_count(const symmicp_ctx *ctx);
size_t symmicp_local_source_offset(const symmicp_ctx *ctx);

/* ---- host-side pieces of func.cpp:76-102, exposed for parity tests ----- */
int symmicp_solve(int mode, const symmicp_sums *sums, const float pivot[3],
                  float pbar[3], float qbar[3], float a[3], float t[3], float *rcond, float out16[16]);

/* ---- normals pre-step (replaces MyICP::estimateNormals, myicp.cpp:152-172: PCL NormalEstimation,
 * setKSearch(10), viewpoint (0,0,0)).  Exact k-NN (the point itself included) + PCA on the GPU.
 * xyz strided as in set_source; nrm_out packed AoS [n][3]; curv_out (lambda_min / trace) may be NULL;
 * viewpoint may be NULL (origin); 3 <= k <= 16. */
int symmicp_estimate_normals(int device, const float *xyz, size_t row_stride, size_t col_stride, size_t n, int k,
                             const float viewpoint[3], float *nrm_out, float *curv_out);
/* the same on a context the caller already owns (stream, arenas and code objects are reused; the context's own clouds are
 * left alone): what MyICP::estimateNormals calls for both clouds before every alignment (myicp.cpp:105) */
int symmicp_ctx_estimate_normals(symmicp_ctx *ctx, const float *xyz, size_t row_stride, size_t col_stride, size_t n, int k,
                                 const float viewpoint[3], float *nrm_out, float *curv_out);

/* ---- multi-GPU (new: the reference is single-threaded) ----- */
/* rank 0 creates an id, the application ships it to the other ranks (any channel),
 * then every rank calls comm_init_rank BEFORE set_source.  One RCCL all-reduce of
 * SYMMICP_NSUM doubles per pass. */
int symmicp_comm_get_unique_id(void *out128);
/* the share of the source `rank` of `nranks` owns: rows [begin, begin+count) of the caller's cloud;
 * pure host arithmetic, the same partition symmicp_set_source applies. */
int symmicp_shard_range(size_t n, int nranks, int rank, size_t *begin, size_t *count);
int symmicp_comm_init_rank(symmicp_ctx *ctx, int nranks, int rank, const void *unique_id128);
/* External exchange: with unique_id128 == NULL and nranks > 1 the context is sharded as above but owns no
 * communicator; every pass then ends with THIS RANK's record (symmicp_iter_result.sums).  The application sums the
 * records over the ranks however it likes (MPI, gloo, shared memory, ...) and hands the total back before the next
 * step; all ranks must pass the same 40 doubles so that their solves agree.  (symmicp_align is not available in this
 * mode: it would solve from the local record.) */
int symmicp_set_sums(symmicp_ctx *ctx, const symmicp_sums *total_over_ranks);
/* Ranks of ONE node can also exchange the record through POSIX shared memory instead of RCCL (a 320-byte, purely
 * latency-bound exchange: no collective kernel, no extra launch).  job_name must be the same on all ranks and unique
 * per job (it names the segment /symmicp_<job_name>; rank 0 creates and, on destroy, removes it).  Call it instead of
 * symmicp_comm_init_rank, before symmicp_set_source; everything else (align, step, ...) works as with RCCL. */
int symmicp_comm_init_shm(symmicp_ctx *ctx, int nranks, int rank, const char *job_name);

/* ---- measurement helpers ------------------------------------------------ */
typedef struct {
    double last_pass_ms;       /* HIP-event time of the most recent pass kernel(s), on the ctx stream */
    double sum_pass_ms;        /* accumulated since the last reset */
    int64_t passes;
    double build_ms;           /* target index build (sort + grid + tree) */
    double upload_ms;
    int32_t grid_level;        /* cells per axis = 2^grid_level */
    int32_t tree_levels;
    int64_t pass_blocks;
    int64_t bytes_algorithmic_per_pass;  /* N_s*(48+4+4)+N_t*12 (NN) or N_s*48 (identity) [+24 N_s write-back] */
    /* per-kernel HIP-event time since the last reset (timing mode), slots:
     * 0 k_search_cells, 1 idle gap between cells and walk, 2 k_search_walk, 3 k_accumulate, 4 k_final_reduce,
     * 5 the single pass kernel of the IDENTITY / BRUTE modes (k_pass_identity, or k_nn_brute + k_pass_indexed),
     * 6 the whole pass bracketed by two events (timing mode 1);
     * kernel_launches[7] counts passes that skipped the tree walk and had to be repaired */
    double kernel_ms[8];
    int64_t kernel_launches[8];
    /* timing on: duration of each of the first 8 passes since symmicp_reset_stats (pass 0 = the correspondence pass of
     * symmicp_begin, myicp.cpp:122), and how many passes were timed in all (timing mode 3: a pass that carried events counts, in
     * sum_pass_ms / passes_timed / pass_ms_head, for itself and the up to three passes behind it that carried none) */
    double pass_ms_head[8];
    int64_t passes_timed;
    /* passes that ran inside device-driven runs of iterations (symmicp_align), and those of them that carried the straggler stage
     * (tree walk over a non-empty work list inside the run) */
    int64_t loop_passes, loop_straggler_passes;
    /* first passes run as 64-query packets (kernels_packet.hip): packets whose breadth-first frontier outgrew its LDS slot and that
     * finished depth-first instead (exact either way; 0 on the BASELINE workloads), since symmicp_create */
    int64_t packet_fallbacks;
    /* sharded runs with the library's own RCCL communicator, timing on: HIP-event time spent in the per-pass all-reduce of the record
     * (events on the ctx stream around the collective) and the number of all-reduces that carried events */
    double allreduce_ms;
    int64_t allreduce_timed;
} symmicp_stats;
int symmicp_get_stats(symmicp_ctx *ctx, symmicp_stats *out);
int symmicp_reset_stats(symmicp_ctx *ctx);
/* HIP events on the ctx stream: 0 off, 1 two events bracketing each pass, 2 events around every kernel of a pass, 3 as 1 but only
 * every 4th pass of a device-driven run of iterations carries events and stands for its neighbours in the sums (each record costs ~2.5 us of GPU timeline: two per pass are 13 % of a converged
 * 1M-point iteration; mode 2 is for profiling, not for throughput runs) */
int symmicp_enable_timing(symmicp_ctx *ctx, int on);

/* ---- PCD I/O (replaces pcl::PCDReader use in MyICP::LoadCloud, myicp.cpp:20-31) */
/* returns point count (>=0) or -symmicp_status.  xyz/nrm packed AoS (3 floats per point); pass
 * xyz==NULL to query the count.  nrm may be NULL.  *has_normals reports normal_x/y/z fields. */
long symmicp_pcd_read(const char *path, float *xyz, float *nrm, size_t cap, int *has_normals);
int symmicp_pcd_write(const char *path, const float *xyz, const float *nrm, size_t n, int binary);

#ifdef __cplusplus
}
#endif
#endif /* SYMMICP_H */
