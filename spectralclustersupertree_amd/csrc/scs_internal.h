// Internal declarations shared by the translation units of libscs_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/scs_hip.h"

// ---- error plumbing --------------------------------------------------------
void scs_set_error(const char *fmt, ...);

#define SCS_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            scs_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                          __LINE__);                                                         \
            return (e_ == hipErrorOutOfMemory) ? SCS_ENOMEM : SCS_EHIP;                      \
        }                                                                                    \
    } while (0)

#define SCS_REQUIRE(cond, ...)                                                               \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            scs_set_error(__VA_ARGS__);                                                      \
            return SCS_EINVAL;                                                               \
        }                                                                                    \
    } while (0)

#define SCS_TRY(expr)                                                                        \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != SCS_OK) return rc_;                                                       \
    } while (0)

// ---- environment switches ---------------------------------------------------
// Two kinds (DESIGN.md section 12 has the table).  SUPPORTED switches select a documented behaviour and are read
// with getenv where they apply (SCS_LOWP, SCS_LOWP_MAX_BYTES, SCS_WS_LIMIT_MB, SCS_HOST_THREADS here; SCS_DEVICE,
// SCS_TEAM, SCS_MULTI_MODE, SCS_SHARD_MIN_VERTICES, SCS_CHILD_RNG, SCS_AHEAD, SCS_AHEAD_WORKERS,
// SCS_SPEC_MAX_TAXA, SCS_KMEANS, SCS_MALLOC_TUNE, SCS_RDZV_PORT in the Python host).  PROBE switches -- the A/B
// paths of measurements that are committed under profiles/, thresholds, traces -- exist for tools/ and tests/ only
// and are read through scs_dbg, which answers "unset" unless SCS_DEBUG=1 was in the environment when the
// library was loaded: in a production process no probe path can be reached, whatever else is exported.
#include <cstdlib>
static inline const char *scs_dbg(const char *name) {
    static const bool on = [] {
        const char *e = getenv("SCS_DEBUG");
        return e && atoi(e) != 0;
    }();
    return on ? getenv(name) : nullptr;
}

// ---- device memory ------------------------------------------------------------
// Every hipMalloc / hipFree of the library goes through these two (scs_ctx.hip): one place to trace them
// (SCS_ALLOC_TRACE, a probe switch: calls of more than 5 ms with the calling thread and the time since start).
// With a context they take from / give back to the device's ARENA (scs_arena.h: slabs kept from the driver,
// requests carved out of them); without one (the matrix-free comparison of tools/) straight from the driver.
struct scs_ctx;
hipError_t scs_dev_malloc_impl(scs_ctx *ctx, void **p, size_t bytes);
hipError_t scs_dev_free(void *p);
template <typename T>
static inline hipError_t scs_dev_malloc(scs_ctx *ctx, T **p, size_t bytes) {
    return scs_dev_malloc_impl(ctx, (void **)p, bytes);
}
// free bytes the arena of a device holds (beside what hipMemGetInfo reports as free)
size_t scs_arena_free_bytes(int device);

// ---- tile geometry of the accumulate kernel --------------------------------
constexpr int SCS_TR = 64;    // rows of W per tile (one block record)
constexpr int SCS_TCW = 256;  // threads per workgroup = columns per column group
constexpr int SCS_NPAD = 512; // position tables are padded to a multiple of this
constexpr int SCS_LD_ALIGN = 512;  // leading dimension of W (doubles), see scs_symm.h

// a scratch slot of the build above this size is released when the call ends (a batch's range-minimum
// tables are 0.3 GB at 10 000 leaves and 0.6 GB from 50 000 on: kept -- a hipFree / hipMalloc pair of that
// size cost 0.4 ms of every 21 ms step; the workspace of a memory-bound job, tens of GB, is not)
constexpr size_t SCS_SCRATCH_KEEP = (size_t)1 << 30;
constexpr size_t SCS_PINNED_KEEP = (size_t)2 << 30;  // free page-locked host blocks kept per context
// (until round 6 a context kept free device blocks of its own, at most SCS_BLOCK_KEEP bytes, and one W buffer;
// both are gone: blocks and W buffers of every size are chunks of the device's arena, scs_arena.h)

struct scs_ctx;
// device blocks of a context, carved out of the device's arena (scs_ctx.hip)
int scs_block_alloc(scs_ctx *ctx, size_t bytes, void **out);
void scs_block_release(scs_ctx *ctx, void *p);
// every free cached block back to the runtime (an allocation outside the cache has failed)
void scs_block_drop_free(scs_ctx *ctx);
// cached page-locked host blocks of a context (scs_ctx.hip): the download side of scs_forest_split
int scs_pinned_get(scs_ctx *ctx, size_t bytes, void **out);
void scs_pinned_release(scs_ctx *ctx, void *p);

// ---- communicator -----------------------------------------------------------
struct scs_local_group;  // in-process barrier + exchange slots

struct scs_comm {
    int rank = 0;
    int world = 1;
    int kind = 0;  // 0 none, 1 rccl, 2 local
    void *rccl_comm = nullptr;
    scs_local_group *group = nullptr;
};

// all-gather of equal-sized fp64 chunks: sendbuf has `count` doubles, recvbuf
// world*count doubles (device pointers), enqueued on `stream`.
int scs_comm_allgather_f64(scs_comm *comm, const double *sendbuf, double *recvbuf, size_t count,
                           hipStream_t stream);
// all-to-all-v of fp64 (host offset arrays of world + 1, in doubles): see scs_ctx.hip
int scs_comm_alltoallv_f64(scs_comm *comm, const double *sendbuf, const int64_t *send_off,
                           double *recvbuf, const int64_t *recv_off, hipStream_t stream);
int scs_comm_init_rccl(scs_comm *comm, int rank, int world, const void *uid);
int scs_comm_destroy(scs_comm *comm);

// ---- context ----------------------------------------------------------------
struct scs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // late chunks of a table upload (created on first use)
    scs_comm comm;
    int n_cu = 256;
    int max_lds_bytes = 65536;  // dynamic LDS one workgroup may ask for (hipDeviceAttributeMaxSharedMemoryPerBlock)
    size_t ws_limit = 0;  // bytes of build scratch allowed per tree batch
    // 64 doubles of mapped, coherent host memory the eigensolver's small kernel reports
    // residual norms into (allocated on first use)
    double *h_report = nullptr;
    double *d_report = nullptr;
    unsigned long long report_seq = 0;  // sequence number of the last report requested
    // small scratch buffers of scs_pcg_build kept between calls (the recursion makes
    // thousands of tiny builds: hipMalloc/hipFree dominated them); anything above
    // SCS_SCRATCH_KEEP bytes is released at the end of the call that needed it
    struct scratch_slot {
        void *p = nullptr;
        size_t cap = 0;
    } scratch[8];
    hipEvent_t build_events[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t solve_events[2] = {nullptr, nullptr};
    // the W buffer of the graph freed last: the next graph of about that size takes it over
    // instead of a hipFree / hipMalloc pair (a step of the benchmark re-builds the same 20 GB
    // matrix; the recursion makes thousands of small ones).  One buffer at most.
    double *w_cache = nullptr;
    size_t w_cache_bytes = 0;
    // the single-precision image of the graph freed last (mixed-precision loop, scs_eig.hip): any later
    // image that fits takes it over; the larger of two is kept.  Large hipMalloc / hipFree pairs are not
    // free on this runtime (tools/malloc_time.py: a free returns in 0.4 ms and a LATER hipMalloc of tens
    // of GB pays 2 - 6 s for it now and then), so the image does not go through the block cache.
    float *w32_cache = nullptr;
    size_t w32_cache_bytes = 0;
    // device blocks of the solver, kept between calls (scs_fiedler needs a dozen buffers per
    // call; the recursion calls it thousands of times): a block is handed out again when it is
    // large enough and at most twice the request.  Free blocks beyond SCS_BLOCK_KEEP bytes in
    // total are released.
    struct cached_block {
        void *p;
        size_t bytes;
        bool in_use;
    };
    // (the two caches below are locked: a forest's last reference may be dropped -- and its blocks handed
    // back -- by another host thread than the one working on this context, e.g. the look-ahead worker)
    std::mutex cache_mu;
    std::vector<cached_block> blocks;
    std::vector<cached_block> pinned;  // page-locked host blocks (scs_pinned_get), at most SCS_PINNED_KEEP free
    std::vector<hipEvent_t> event_pool;  // events of the timed SYMM launches, reused
    // staging of scs_small_solve (one pinned host block, one device block), grown on demand
    unsigned *h_flags = nullptr;  // 64 pinned bytes: results of device-side argument checks
    // one slot per scs_small_solve_begin that has not been ended yet (and the free ones kept for
    // the next): page-locked host block + device block of the same layout, the event that fires
    // when the results have landed in the host block, where they sit in it
    struct small_slot {
        unsigned char *host = nullptr;
        unsigned char *dev = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
        size_t o_maps = 0, maps_bytes = 0, o_lam = 0, lam_bytes = 0, o_w = 0, w_bytes = 0;
        // the batch's device scratch (addends, uncontracted weights): owned by the slot until the
        // ticket is ended -- the batch runs on small_stream, beside whatever the main stream does
        unsigned char *scratch = nullptr;
        size_t scratch_cap = 0;
    };
    std::vector<small_slot> small_slots;
    // Round 5: begun small solves run on a stream of their own.  The walk of the recursion needs the
    // split of the node it stands on at once, while a right sibling's solve -- begun earlier, wanted
    // later -- may still be running: on one stream the split queued behind it.
    hipStream_t small_stream = nullptr;
};

struct scs_tables {
    int32_t n_taxa = 0;
    int32_t n_trees = 0;
    int64_t n_leaves = 0;
    int32_t max_leaves = 0;            // largest tree
    std::vector<int64_t> h_tree_off;   // host copy (batch planning)
    void *d_block = nullptr;           // the one cached device block the five arrays live in
    int64_t *d_tree_off = nullptr;     // [n_trees+1]
    int32_t *d_leaf_taxon = nullptr;   // [L]
    int32_t *d_adj_depth = nullptr;    // [L]
    double *d_adj_val = nullptr;       // [L]
    double *d_tree_w = nullptr;        // [n_trees]
    // Page-locked source arrays: scs_tables_upload returns once the leaf arrays of the first
    // `late_start[0]` trees have arrived and been checked; the rest travels on the context's copy
    // stream in chunks -- chunk c holds the trees [late_start[c], late_start[c + 1]) and is complete
    // (copied and range-checked into d_flags) when late_ev[c] has fired.  scs_pcg_build makes its
    // stream wait for the chunks a tree batch needs and collects the verdict at its end.
    mutable std::vector<int32_t> late_start;  // empty: everything arrived with the upload
    mutable std::vector<hipEvent_t> late_ev;
    unsigned *d_flags = nullptr;
};

// scs_ctx.hip: make `stream` wait until the leaf arrays of the trees [0, t_end) are on the device
int scs_tables_wait(scs_ctx *ctx, const scs_tables *t, int32_t t_end, hipStream_t stream);
// wait (on the host) for everything still on its way; SCS_EINVAL when its range check failed
int scs_tables_finish(scs_ctx *ctx, const scs_tables *t);

// A graph WITHOUT its matrix (scs_graph_matrix_free, scs_matfree.h: the measured comparison of round 5):
// the operator is applied from the tables; everything here is plain device memory owned by the graph.
struct mf_data {
    const scs_tables *tb = nullptr;  // borrowed: must outlive the graph
    int64_t *d_stack_off = nullptr;
    int64_t stack_total = 0;
    int b_cap = 0;                   // widest block the buffers below are sized for
    int chunks = 1;                  // pieces a tree's leaves are cut into per direction (scs_matfree.h)
    double *st_val = nullptr, *st_sum = nullptr;
    int32_t *st_dep = nullptr;
    double *sm_val = nullptr, *sm_sum = nullptr, *cy_pa = nullptr, *cy_ps = nullptr;
    int32_t *sm_dep = nullptr, *cy_dep = nullptr, *sm_cnt = nullptr, *sm_root = nullptr, *cy_cnt = nullptr;
    double *x = nullptr;             // [n][b_cap] operand, row-major
    double *y = nullptr;             // [n][b_cap] result, row-major
    double *slabs = nullptr;         // [2][n_trees][n][b_cap]
    int slabs_b = 0;                 // block width the slabs were last written with (absent taxa stay zero)
    int64_t n_apply = 0;
};

struct scs_graph {
    mf_data *mf = nullptr;  // non-null: no d_w; see above
    int32_t n = 0;          // V: number of vertices (columns)
    int32_t row_begin = 0;  // first row owned by this rank
    int32_t row_end = 0;
    int64_t ld = 0;         // leading dimension (doubles) of d_w
    double *d_w = nullptr;  // (row_end-row_begin) x ld, row-major
    // SCS_BUILD_UPPER graphs: a row holds only the columns [col0, n) -- column c of row r sits at
    // d_w[(r - row_begin) * ld + (c - col0)] -- and of those only the cells of the upper-triangle
    // tiles are defined (everything the symmetric SYMM reads).  col0 = 0, upper = false otherwise.
    int32_t col0 = 0;
    bool upper = false;
    size_t w_bytes = 0;     // size of the d_w allocation (may exceed the need: reused buffer)
    bool w_block = false;   // d_w is a block of the arena (scs_block_alloc): always, since round 6
    // degree data for all V vertices (filled lazily by scs_graph_prepare_degrees)
    bool have_deg = false;
    void *deg_stage = nullptr;  // page-locked copy in flight (scs_graph_prepare_degrees_begin)
    // single-precision image of the tiles the symmetric SYMM streams (written beside the degrees when the
    // eigen-solver asks for it: mixed-precision LOBPCG, scs_eig.hip); leading dimension ld
    float *d_w32 = nullptr;
    size_t w32_bytes = 0;  // size of that allocation (may exceed the need: reused buffer)
    bool have_w32 = false;
    double *d_deg = nullptr;   // [V] row sums
    double *d_dinv = nullptr;  // [V] 1/sqrt(deg) (1 where deg == 0)
    int32_t n_isolated = 0;
    double dd_norm = 0.0;  // ||sqrt(deg)||_2 with isolated rows counted as 1
};

// ---- source forests in HBM (scs_forest.hip; scs_eig.hip packs a small node's tables from one) ----
#include <memory>
// device arrays a split leaves behind for ALL children: one allocation each, children are slices
struct forest_region {
    scs_ctx *ctx = nullptr;
    std::shared_ptr<forest_region> base;  // a slice (scs_forest_slice) keeps the arrays it points into alive
    std::vector<void *> blocks;
    void *host_block = nullptr;  // page-locked copy of the children's tables (scs_pinned_get)
    ~forest_region() {
        for (void *p : blocks) scs_block_release(ctx, p);
        if (host_block) scs_pinned_release(ctx, host_block);
    }
    int alloc(size_t bytes, void **out) {
        SCS_TRY(scs_block_alloc(ctx, bytes, out));
        blocks.push_back(*out);
        return SCS_OK;
    }
};

struct scs_forest {
    int32_t n_taxa = 0, n_trees = 0;
    int64_t n_nodes = 0, n_leaves = 0;
    std::shared_ptr<forest_region> region;  // (owner of every pointer below)
    int64_t *node_off = nullptr;            // [n_trees + 1], relative to this forest's arrays
    int32_t *parent = nullptr, *taxon = nullptr;
    double *length = nullptr, *support = nullptr;
    double *weights = nullptr;  // [n_trees]
    mutable int32_t *tree_id = nullptr;  // [n_nodes] tree of a node (made on first use by the node-parallel split)
    // children of a split also carry their flattened tables
    bool has_tables = false;
    int64_t *tree_off = nullptr;  // [n_trees + 1]
    int32_t *leaf_taxon = nullptr, *adj_depth = nullptr, *tree_index = nullptr;
    double *adj_val = nullptr;
    unsigned char *present = nullptr;  // [n_taxa]
    // ... and a copy of them in page-locked host memory (valid as long as the forest lives)
    const int64_t *h_tree_off = nullptr;
    const int32_t *h_leaf_taxon = nullptr, *h_adj_depth = nullptr, *h_tree_index = nullptr;
    const double *h_adj_val = nullptr, *h_weights = nullptr;
    const unsigned char *h_present = nullptr;
};


// build.hip
int scs_graph_prepare_degrees(scs_ctx *ctx, scs_graph *g);
int scs_graph_prepare_degrees_begin(scs_ctx *ctx, scs_graph *g, bool want_w32 = false);
// scs_eig.hip: y (row-major n x b, unscaled) = W x for a matrix-free graph; zt k-major (b x ldz)
int scs_matfree_apply(scs_ctx *ctx, scs_graph *g, const double *zt, int64_t ldz, int b, double *y_out,
                      hipStream_t stream);
void scs_matfree_release(scs_graph *g);
// row splits of every rank (contiguous, ordered by rank): collective, world + 1 entries
int scs_gather_row_splits(scs_ctx *ctx, int32_t row_begin, int32_t row_end, int32_t n,
                          std::vector<int32_t> &splits);

// eig.hip helpers used by debug entry points are declared in scs_hip.h

static inline int64_t scs_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
