/*
 * scs_hip.h -- C-ABI of libscs_hip.so, the MI355X (gfx950) spectral-clustering
 * core for Spectral Cluster Supertree.
 *
 * The reference (rmcar17/SpectralClusterSupertree) has no FFI seam on this
 * path: the hot path is private Python called in-process
 * (src/sc_supertree/scs.py:110-134) and its only delegate is scikit-learn's
 * estimator call at scs.py:235-252.  Each entry point below names the reference
 * code it replaces.  All pointers are plain host pointers unless stated, all
 * sizes are explicit, no C++ or torch types cross the boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SCS_E* code on failure;
 *     scs_last_error() gives the message of the calling thread's last failure.
 *   - input arrays are caller-owned, read-only for the duration of the call.
 *   - handles are owned by the library; free them with the matching *_free.
 *   - one host thread per context at a time; different contexts are independent.
 *   - no entry point ever computes on the CPU: without a usable HIP device
 *     scs_ctx_create fails and nothing else can be called.
 */
#ifndef SCS_HIP_H
#define SCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCS_OK 0
#define SCS_EINVAL (-1)   /* bad argument (shape, range, null pointer)          */
#define SCS_EHIP (-2)     /* a HIP runtime call failed                          */
#define SCS_ENOMEM (-3)   /* device or host allocation failed                   */
#define SCS_ECOMM (-4)    /* RCCL / communicator failure                        */
#define SCS_ENOCONV (-5)  /* scs_fiedler stopped above tol: maps_out and stats filled   */
#define SCS_EUNSUP (-6)   /* valid request the library does not support (yet)   */

#define SCS_UNIQUE_ID_BYTES 128

typedef struct scs_ctx scs_ctx;       /* device, stream, communicator, scratch        */
typedef struct scs_tables scs_tables; /* flattened source trees, resident in HBM      */
typedef struct scs_graph scs_graph;   /* this rank's row block of the PCG weight matrix */

/* Per-call report of scs_fiedler (all times in milliseconds, device-side
 * hipEvent measurements on the context's stream). */
typedef struct scs_stats {
    int32_t n_vertices;      /* V of the graph the solve ran on                      */
    int32_t block;           /* LOBPCG block width actually used (0: dense path)     */
    int32_t iterations;      /* LOBPCG iterations                                    */
    int32_t n_apply;         /* operator applications (launches of the SYMM kernel)  */
    int32_t converged;       /* 1 if the wanted pair(s) met tol                      */
    int32_t used_constraint; /* 1: trivial eigenvector deflated analytically         */
    double lambda[2];        /* two largest eigenvalues of S = D^-1/2 A D^-1/2       */
    double resid[2];         /* ||S x - lambda x||_2 of the returned unit vectors    */
    double lambda_next;      /* next Ritz value below lambda[1] (gap estimate)       */
    double apply_ms_total;   /* SYMM kernel time: HIP-event-timed launches (every 4th
                              * in the fused loop), scaled to all n_apply launches */
    double apply_ms_min;     /* fastest single SYMM launch                           */
    double solve_ms;         /* whole scs_fiedler call, device time                  */
    double apply_bytes;      /* algorithmic HBM bytes of ONE SYMM launch on this rank */
    double allgather_ms_total; /* multi-rank solves: device time of the per-iteration all-gathers
                                * (timed every 4th, scaled to all n_allgather)            */
    double allgather_bytes;  /* bytes ONE all-gather delivers to this rank (world x chunk x 8) */
    int32_t n_allgather;     /* all-gathers enqueued by the solve                      */
    int32_t n_apply32;       /* of n_apply: launches that streamed the single-precision image of W
                              * (mixed-precision loop; apply_ms_total covers the OTHER launches)   */
    double apply32_ms_total; /* SYMM kernel time of those launches (timed sample scaled to all)   */
    double apply32_bytes;    /* algorithmic HBM bytes of ONE such launch                          */
    int32_t lowp_renewals;   /* times S X and S P were renewed through W inside the loop          */
    int32_t reserved;
    double event_pair_ms;    /* an EMPTY event pair on the solve's stream: what the pair around a timed
                              * launch adds to apply_ms_* (a profiler's kernel time is about that much less) */
} scs_stats;

/* Per-call report of scs_pcg_build. */
typedef struct scs_build_stats {
    int32_t n_taxa;
    int32_t n_trees;
    int32_t row_begin, row_end; /* rows of W this rank owns                          */
    int32_t symmetric;          /* 1: upper tiles computed, lower mirrored; 2: shared */
    int32_t n_tiles;            /* workgroups of the accumulate kernel per batch     */
    int32_t n_batches;          /* tree batches (scratch bounded by ctx workspace)   */
    int32_t spec_batches;       /* of them walked by the producer / consumer tile kernel
                                   (k_accumulate_spec; the others by the 4-wave kernels)   */
    double cell_trees;          /* (matrix cell, tree) evaluations performed         */
    double prep_ms;             /* position / sparse-table / block-record kernels    */
    double accumulate_ms;       /* the tile accumulate kernel(s)                     */
    double exchange_ms;         /* shared build: tile all-gather + unpack            */
    double total_ms;            /* whole call, device time                           */
    double bytes_w;             /* algorithmic bytes: W written once (8 * rows * V)  */
    double bytes_tables;        /* algorithmic bytes: tables read once               */
    double exchange_bytes;      /* shared build: bytes of packed tiles this rank received */
    int32_t tree_parallel_batches; /* batches of a small node built tree-parallel: a workgroup
                                      per (tile, tree), the trees' cells added up in order
                                      afterwards (k_sum_tree_tiles) -- same bits as the walk */
    int32_t spec_trees;            /* trees the spec_batches walked (the rest: 4-wave kernels)       */
    double spec_ms;                /* device time of the k_accumulate_spec launches (of accumulate_ms) */
    int32_t listed_batches;        /* batches walked through per-tile tree lists (partial-coverage forests) */
    int32_t reserved;
} scs_build_stats;

/* ABI version of this header: 105.  104 -> 105: scs_debug_arena_stats and scs_ctx_reserve added; scs_ctx_trim's keep_bytes counts the
 * free bytes of the device's arena.  103 -> 104: scs_forest_split_level, scs_forest_analyze,
 * scs_forest_tables_download_range, scs_tables_from_forest_range, scs_small_solve_begin_level added;
 * scs_forest_upload checks the arrays.  102 -> 103: scs_stats ends with event_pair_ms.  101 -> 102: scs_stats is
 * 24 bytes longer (n_apply32 in the old `reserved` slot, apply32_ms_total, apply32_bytes, lowp_renewals).  100 -> 101: scs_build_stats is 8 bytes longer
 * (tree_parallel_batches; the old `reserved` slot became spec_batches) and again by spec_trees /
 * spec_ms; scs_forest_* added.  Callers compare it with the value they were compiled against before passing
 * structs (the Python binding refuses a library of another version at load). */
int scs_version(void);
const char *scs_last_error(void);

/* Number of HIP devices visible (0 when none); never fails. */
int scs_device_count(void);

/* ---- context ----------------------------------------------------------- */

/* Rank 0 calls this and ships the 128 bytes to the other ranks by any host
 * channel before they all call scs_ctx_create (RCCL bootstrap). */
int scs_comm_unique_id(void *out128);

/* One context per process per GPU.  world == 1: single device, unique_id may be
 * NULL.  world > 1: row-partitioned solve, one RCCL rank per process
 * (SURVEY.md 8e); the only collective on the data path is the all-gather of
 * the Krylov block inside scs_fiedler. */
int scs_ctx_create(int device, int rank, int world, const void *unique_id128, scs_ctx **out);

/* In-process communicator for `world` contexts that live in ONE process and
 * are driven by `world` host threads (used to exercise the row-partitioned
 * code path on a single GPU; the collective is a thread barrier plus device
 * copies instead of RCCL).  `group` comes from scs_local_group_create. */
typedef struct scs_local_group scs_local_group;
int scs_local_group_create(int world, scs_local_group **out);
int scs_local_group_destroy(scs_local_group *group);
int scs_ctx_create_local(int device, int rank, scs_local_group *group, scs_ctx **out);

int scs_ctx_destroy(scs_ctx *ctx);
int scs_ctx_synchronize(scs_ctx *ctx);
/* Device memory: every block of the library is carved out of a per-device ARENA shared by all contexts of the
 * process (csrc/scs_arena.h) -- slabs taken from the driver stay with the process until they are handed back
 * here, or until the driver refuses a request (then whole free slabs go back and the request is tried again).
 * scs_ctx_trim hands back whole free slabs, largest first, until at most keep_bytes of free arena memory remain
 * on the context's device (0: everything that is free), and with keep_bytes == 0 the context's free
 * page-locked host blocks and the staging / scratch blocks of its small-solve slots that no ticket holds.
 * No reference counterpart. */
int scs_ctx_trim(scs_ctx *ctx, int64_t keep_bytes);
/* Make sure the arena of the context's device holds `bytes` of free memory in one piece, so that a request of
 * that size is served without the driver (on this pool hipMalloc of memory another process has used before costs
 * ~25 ms per GB: the driver clears it).  What a warm-up call leaves behind, without the call. */
int scs_ctx_reserve(scs_ctx *ctx, int64_t bytes);
/* The arena of a device, for tests and reports: out8 = {bytes of slabs held, bytes in use, slabs, chunks,
 * released chunks not yet safe for other contexts, driver allocations so far, driver releases so far,
 * requests served so far}. */
int scs_debug_arena_stats(int device, int64_t *out8);
/* The communicator as it sees itself: kind (0 none, 1 RCCL, 2 in-process team), the world / rank
 * it was created with, and what ncclCommCount / ncclCommUserRank report (-1: not available).
 * No reference counterpart (the reference has no parallelism, scs.py:239 n_jobs = 1). */
int scs_ctx_comm_info(scs_ctx *ctx, int32_t *kind, int32_t *world, int32_t *rank,
                      int32_t *reported_world, int32_t *reported_rank);

/* ---- source forests in HBM (round 5) ------------------------------------ */

/* The source trees as preorder node arrays, resident on the device: what the recursion restricts
 * at every node (reference: src/sc_supertree/scs.py:139-171 with :411-455 --
 * `_generate_induced_trees_with_weights`, cogent3's get_sub_tree per tree and part).
 * Arrays as spectralclustersupertree_amd/treearrays.py: tree t owns nodes [node_off[t],
 * node_off[t+1]) in preorder; parent relative to the tree's first node (-1 root); taxon -1 for
 * inner nodes; length / support NaN for None; n_leaves = nodes with taxon >= 0. */
typedef struct scs_forest scs_forest;
typedef struct scs_forest_info {
    int32_t n_trees;  /* trees that keep two or more leaves of the part       */
    int32_t monotone; /* `branch`: 1 unless a merged inner length is negative  */
    int64_t n_nodes;
    int64_t n_leaves;
} scs_forest_info;

int scs_forest_upload(scs_ctx *ctx, int32_t n_taxa, int32_t n_trees, const int64_t *node_off,
                      const int32_t *parent, const int32_t *taxon, const double *length,
                      const double *support, const double *weights, int64_t n_leaves, scs_forest **out);
int scs_forest_free(scs_ctx *ctx, scs_forest *forest);

/* The forests induced on up to 8 disjoint taxon sets, all from ONE sweep of `forest`, each with
 * its flattened tables (scs.py:411-455 + the strategy values of :555-564): part_of[x] = part of
 * taxon x or -1, new_id[x] = its id inside the part (children number their taxa 0 .. k-1),
 * part_taxa[c] = k of part c, strategy 0 one / 1 depth / 2 branch / 3 bootstrap.  Dropped
 * leaves, spliced unary nodes with merged lengths (folded bottom-up, the parent's length in
 * front), collapsed unary roots, trees left with fewer than two leaves of a part dropped --
 * bit for bit what libscs_host.so's scs_host_split_* + scs_host_flatten produce.
 * SCS_EUNSUP: bootstrap weighting met an inner node without support (the reference fails in
 * `length * tree_weight`, scs.py:656). */
int scs_forest_split(scs_ctx *ctx, const scs_forest *forest, const int32_t *part_of,
                     const int32_t *new_id, int32_t n_parts, const int32_t *part_taxa, int32_t strategy,
                     scs_forest **out_forests, scs_forest_info *info);

/* The tables of a child of scs_forest_split to the host (any pointer may be null): tree_off
 * [n_trees + 1], leaf_taxon / adj_depth / adj_val [n_leaves], tree_index [n_trees] (the tree's
 * index in the parent forest), tree_w [n_trees], present [n_taxa] (1: the taxon occurs). */
int scs_forest_tables_download(scs_ctx *ctx, const scs_forest *forest, int64_t *tree_off,
                               int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val,
                               int32_t *tree_index, double *tree_w, uint8_t *present);

/* The same arrays WITHOUT a copy: scs_forest_split leaves the children's tables in page-locked
 * host memory of the context (one device-to-host copy for all parts); the pointers stay valid
 * until the forest is freed. */
int scs_forest_tables_host(scs_ctx *ctx, const scs_forest *forest, const int64_t **tree_off,
                           const int32_t **leaf_taxon, const int32_t **adj_depth, const double **adj_val,
                           const int32_t **tree_index, const double **tree_w, const uint8_t **present);

/* Node arrays of the trees [t_begin, t_end) to the host (node_off [t_end - t_begin + 1], made
 * relative to the first of them; the other pointers may be null). */
int scs_forest_download(scs_ctx *ctx, const scs_forest *forest, int32_t t_begin, int32_t t_end,
                        int64_t *node_off, int32_t *parent, int32_t *taxon, double *length,
                        double *support, double *weights);

/* The tables of a child of scs_forest_split as an scs_tables handle (what scs_tables_upload makes from
 * host arrays) WITHOUT leaving the device: relabel[x] (may be null: identity) = id of the forest's taxon
 * x in the node's numbering, n_taxa = the node's taxon count.  For scs_pcg_build of the recursion's
 * larger nodes; free with scs_tables_free.  `ctx` may be another context on the same GPU than the one
 * the forest was split on (the look-ahead worker's). */
int scs_tables_from_forest(scs_ctx *ctx, const scs_forest *forest, const int32_t *relabel, int32_t n_taxa,
                           scs_tables **out);

/* ---- a whole level of the recursion in one call (round 6) ---------------- */

/* The reference walks the recursion depth-first, one node at a time (scs.py:139-171).  The nodes of one
 * LEVEL below some node hold disjoint taxon sets, so their forests fit in ONE forest -- the trees of node
 * 0, then of node 1, ... -- over one numbering of the taxa (node k owns a consecutive id range), and one
 * call restricts all of them to all of their parts (scs.py:411-455 for every node of the level):
 *   node_tree_end[k]  exclusive end of node k's trees in `forest` (the last entry = its tree count)
 *   part_of[x]        part of taxon x INSIDE ITS NODE (0 .. n_parts - 1 <= 7), -1: in no child
 *   new_id[x]         its id in the children's numbering (unique over all children), < child_taxa
 * The children come back as ONE forest again -- part-major: the first parts of all nodes (node order),
 * then the second parts, ... -- with its tables resident, and of it the host receives only
 *   child_trees / child_leaves [n_parts][n_nodes]   trees (>= 2 leaves kept) and leaves of child (part, node)
 *   present   [child_taxa]      1: the taxon occurs in a kept tree
 *   comp_root [child_taxa]      smallest id of the taxon's connected component in the proper cluster
 *                               graph of its child (scs.py:458-492 `_get_graph_components`; edges = shared
 *                               root sides whatever the weight, :651-652)
 *   sig       [child_taxa][2]   a 128-bit function of the SET of (tree, root side) the taxon occurs in:
 *                               two taxa are contracted (scs.py:302-316) only if their sets -- hence these
 *                               values -- are equal; distinct values prove that nothing contracts, equal
 *                               ones send the node to the exact host routine.
 * info[0] describes the union (monotone: all parts).  Same restriction semantics and bits as
 * scs_forest_split. */
int scs_forest_split_level(scs_ctx *ctx, const scs_forest *forest, const int32_t *part_of,
                           const int32_t *new_id, int32_t n_parts, int32_t child_taxa, int32_t strategy,
                           int32_t n_nodes, const int32_t *node_tree_end, scs_forest **out_union,
                           scs_forest_info *info, int32_t *child_trees, int64_t *child_leaves,
                           uint8_t *present, int32_t *comp_root, uint64_t *sig);

/* The trees [t_begin, t_end) of a forest as a forest of its own WITHOUT copying a node (the slice points
 * into the arrays of `forest` and keeps them alive; taxon ids stay those of `forest`): one node of a level
 * forest whose subtree is taken up again from its own trees when the true draws did not confirm its
 * provisional partition (scs.py:139-171: the labels of record are the stream's).  Free with
 * scs_forest_free. */
int scs_forest_slice(scs_ctx *ctx, const scs_forest *forest, int32_t t_begin, int32_t t_end, scs_forest **out);

/* comp_root [n_taxa] and sig [n_taxa][2] (as above) of a forest that carries tables: the first forest of a
 * level-synchronous walk (a child of scs_forest_split). */
int scs_forest_analyze(scs_ctx *ctx, const scs_forest *forest, int32_t *comp_root, uint64_t *sig);

/* Tables of the trees [t_begin, t_end) of a forest that carries them to the host (tree_off
 * [t_end - t_begin + 1] made relative to the first of them; other pointers may be null): the exact host
 * routine for the contraction groups of a node whose signatures collide (scs.py:302-316). */
int scs_forest_tables_download_range(scs_ctx *ctx, const scs_forest *forest, int32_t t_begin, int32_t t_end,
                                     int64_t *tree_off, int32_t *leaf_taxon, int32_t *adj_depth,
                                     double *adj_val, double *tree_w);

/* scs_tables_from_forest for ONE node of a level forest: the trees [t_begin, t_end), whose taxon ids lie
 * in [u_base, u_base + u_size); relabel[x - u_base] = the node's own id of taxon x (present taxa only,
 * contraction groups consecutive), n_taxa = the node's taxon count. */
int scs_tables_from_forest_range(scs_ctx *ctx, const scs_forest *forest, int32_t t_begin, int32_t t_end,
                                 int32_t u_base, int32_t u_size, const int32_t *relabel, int32_t n_taxa,
                                 scs_tables **out);

/* ---- tables ------------------------------------------------------------ */

/* Page-locked host memory (hipHostMalloc) for callers that want scs_tables_upload to run at
 * the full PCIe rate: tables that live in such a block are copied by DMA straight from it, a
 * pageable source is staged by the runtime at a fraction of that.  Optional -- any host
 * pointer is accepted by scs_tables_upload.  Free with scs_host_free. */
int scs_host_alloc(size_t bytes, void **out);
int scs_host_free(void *p);

/* Host -> HBM copy of the flattened trees (layout: DESIGN.md "Tables").
 * Replaces the reference's in-memory cogent3 trees as the input of
 * _proper_cluster_graph_edges (scs.py:495-583); the weighting strategy
 * (scs.py:555-564) is already folded into adj_val by the host.
 *   tree_off   int64 [n_trees+1]   leaf offsets, tree_off[0] == 0
 *   leaf_taxon int32 [L]           taxon id of each leaf in DFS order
 *   adj_depth  int32 [L]           depth of LCA(leaf p, leaf p+1); last slot of a tree unused
 *   adj_val    fp64  [L]           strategy value at that LCA
 *   tree_w     fp64  [n_trees]     tree weights
 * Shapes are checked on the host, the ranges of leaf_taxon / adj_depth by a kernel on the
 * uploaded copy (SCS_EINVAL, nothing is kept).  The arrays share ONE device block taken from
 * the context's block cache.
 * Page-locked arrays (scs_host_alloc) of a forest that scs_pcg_build walks in several tree
 * batches (more than 2 048 taxa): the call returns once the first 64 trees have arrived and been
 * checked (scs_pcg_build makes them a first, short batch); the rest travels on a copy stream in chunks, and scs_pcg_build makes
 * its stream wait for the chunks each tree batch needs -- the copy overlaps the first batches.
 * The three leaf arrays must then stay valid and unchanged until the first scs_pcg_build on
 * these tables (or scs_tables_free) has returned, and a range error in a late chunk is reported
 * by that call (SCS_EINVAL) instead of this one.  Pageable arrays: copied whole, as before.  */
int scs_tables_upload(scs_ctx *ctx, int32_t n_taxa, int32_t n_trees, const int64_t *tree_off,
                      const int32_t *leaf_taxon, const int32_t *adj_depth, const double *adj_val,
                      const double *tree_w, scs_tables **out);
int scs_tables_free(scs_ctx *ctx, scs_tables *tables);

/* ---- proper cluster graph ---------------------------------------------- */

/* Build rows [row_begin, row_end) of the N x N fp64 weight matrix W on the
 * device: W[a][b] = sum over trees, in tree order, of value(LCA(a,b)) * w_t for
 * every tree in which a and b share a root side.  Bit-identical to the
 * reference's accumulation (scs.py:655-657): one rounded multiply, then rounded
 * adds in tree order.  Replaces _proper_cluster_graph_edges/_dfs_pcg_weights
 * (scs.py:495-663) and the dense fill loop (scs.py:246-250).
 * row_begin = 0, row_end = n_taxa with world == 1 uses the symmetric schedule.
 * flags: SCS_BUILD_MONOTONE promises that adj_val never decreases from an
 * ancestor to a descendant inside any tree (true for `one`, `depth`, and for
 * `branch` when no internal branch length is negative); it selects a cheaper
 * kernel that produces the same bits.  Pass 0 when unsure.
 * SCS_BUILD_SHARED (collective, world > 1, every rank passes it or none does):
 * instead of every rank evaluating all the cells of its own rows -- each
 * off-diagonal cell twice across the job -- the ranks split the upper-triangle
 * tiles of the whole matrix round-robin, all-gather the packed tiles and unpack
 * their own rows (direct cells and mirror images).  Same bits, half the
 * evaluations.  The exchange is point to point: a tile travels only to the ranks whose rows
 * it touches (one grouped round of RCCL send/recv, ~8 V^2 / world bytes received per rank;
 * SCS_EXCHANGE=allgather selects one all-gather of the whole packed triangle instead).  Falls
 * back to the plain row build when the packed triangle would exceed 96 GiB.  stats may be
 * NULL. */
#define SCS_BUILD_MONOTONE 1
#define SCS_BUILD_SHARED 2
/* SCS_BUILD_UPPER (world > 1, every rank passes it or none does; not together with
 * SCS_BUILD_SHARED): W is symmetric, so the job keeps only its upper triangle.  Rank r builds the
 * tiles on and right of the diagonal of ITS rows and nothing else -- every cell is evaluated
 * once across the job, there is NO exchange, and the rank stores (row_end - row_begin) x
 * (V - row_begin) doubles instead of (row_end - row_begin) x V.  row_begin must be a multiple
 * of 256 (the tile width); splits that balance the trapezoids' areas come from the host
 * (partition.row_splits_upper).  scs_fiedler applies the operator from these tiles twice
 * (directly and transposed), all-gathers the ranks' V x b partial products and adds them in
 * rank order -- half the streamed bytes of the row-partitioned solve at every world size, the
 * same bits on every rank.  Such a graph cannot be contracted (scs_graph_contract needs whole
 * rows: SCS_EUNSUP); scs_graph_download_rows returns zeros left of the rank's first column. */
#define SCS_BUILD_UPPER 4
/* SCS_BUILD_SCATTER (with SCS_BUILD_MONOTONE, one rank, the whole matrix): the input-stationary
 * COMPARISON variant -- one workgroup per 64 x 64 block of leaf pairs of a tree, fp64 atomicAdd
 * into W (both triangles).  Same sums up to the order of the additions (<= 1e-12 relative, not
 * bit-exact), two orders of magnitude slower than the ordered tile kernel on random 8-byte
 * read-modify-writes; kept so that the choice is a measured one (profiles/). */
#define SCS_BUILD_SCATTER 8
int scs_pcg_build(scs_ctx *ctx, const scs_tables *tables, int32_t row_begin, int32_t row_end,
                  int32_t flags, scs_graph **out, scs_build_stats *stats);

/* Contract consecutive index ranges into single vertices: vertex g of the new
 * graph is rows/columns [group_start[g], group_start[g+1]) of the old one,
 * W'[g][h] = max over member pairs, diagonal 0 (reference: scs.py:336-387; the
 * host has already relabelled taxa so that every contraction group is a
 * consecutive range that does not straddle a rank's row block).
 * group_start: int32 [n_groups+1], group_start[0] == 0, last == old V.
 * The input graph is consumed (freed) and *out receives the contracted one. */
int scs_graph_contract(scs_ctx *ctx, scs_graph *graph, const int32_t *group_start,
                       int32_t n_groups, scs_graph **out);

/* Shape of this rank's block. */
int scs_graph_shape(const scs_graph *graph, int32_t *n_vertices, int32_t *row_begin,
                    int32_t *row_end);

/* Dense copy of this rank's rows to the host: out is (row_end-row_begin) x V,
 * row-major, caller-owned.  For tests and golden vectors. */
int scs_graph_download(scs_ctx *ctx, const scs_graph *graph, double *out);

/* Rows [first, first+count) of the graph (global row indices, inside this
 * rank's block) to the host: out is count x V row-major. */
int scs_graph_download_rows(scs_ctx *ctx, const scs_graph *graph, int32_t first, int32_t count,
                            double *out);

/* Row sums of this rank's rows (the degree vector d of scipy's normalized
 * Laplacian, scipy/sparse/csgraph/_laplacian.py:550): out has row_end-row_begin
 * entries. */
int scs_graph_degrees(scs_ctx *ctx, scs_graph *graph, double *out);

int scs_graph_free(scs_ctx *ctx, scs_graph *graph);

/* ---- Fiedler solve ----------------------------------------------------- */

/* The V x 2 spectral embedding scikit-learn's SpectralClustering clusters
 * (replaces sklearn/manifold/_spectral_embedding.py:299-467 as reached from
 * scs.py:252): column 0 <-> eigenvalue 1 of S = D^-1/2 A D^-1/2, column 1 the
 * Fiedler vector; unit-norm eigenvectors divided elementwise by sqrt(degree)
 * (1 where the degree is 0), each column sign-flipped so its entry of largest
 * magnitude is positive.
 *   x_init   fp64 [V] or NULL   start vector for the first block column (the
 *                               reference's ARPACK v0 draw, sklearn/utils/_arpack.py:31-33)
 *   tol      residual target: ||S x - lambda x||_2 <= tol (unit x)
 *   max_iter LOBPCG iteration cap
 *   block    LOBPCG block width (<= 16); 0 picks the default
 *   maps_out fp64 [V*2] row-major, caller-owned; every rank receives all V rows
 * Collective over the context's communicator when world > 1.
 * Returns SCS_ENOCONV when the wanted pair(s) stopped above tol (iteration cap, or the
 * residual stopped improving): maps_out and stats then hold the block that was reached
 * and the caller decides (the host retries wider, then warns or raises). */
int scs_fiedler(scs_ctx *ctx, scs_graph *graph, const double *x_init, double tol,
                int32_t max_iter, int32_t block, double *maps_out, scs_stats *stats);

/* A MEASURED COMPARISON, not the product path: a graph WITHOUT its matrix.  scs_fiedler on
 * it applies W straight from the flattened tables (per tree one sweep from the left and one from the right
 * over the leaves, a stack of (depth, value, sum) per sweep; trees added in order) -- scs_pcg_build is not run
 * at all.  One device, more than 128 taxa, block widths 4 and 8 (max_block: the widest the graph's buffers are
 * made for); the tables must outlive the graph; no contraction, no download.  Results agree with the dense
 * path to rounding, not bit for bit.  Free with scs_graph_free. */
int scs_graph_matrix_free(scs_ctx *ctx, const scs_tables *tables, int32_t max_block, scs_graph **out);

/* ---- batched small nodes ----------------------------------------------- */

/* Deep recursion levels (reference: scs.py:110-134 at depth): K
 * independent nodes of at most 128 taxa each (round 4; 64 before), the batch spread over the
 * chip in three launches.  Every node goes from its flattened tables to the V x 2 embedding: proper-cluster-graph weights
 * (bit-identical to scs_pcg_build's W), contraction of consecutive id ranges (max over member
 * pairs, scs.py:336-387), scipy's degree scaling, full Jacobi eigen-decomposition in LDS
 * (two-sided up to 64 vertices, one-sided up to 128),
 * scikit-learn's embedding conventions (as scs_fiedler).  Replaces, per node,
 * scs_tables_upload + scs_pcg_build + scs_graph_contract + scs_fiedler.
 *   n_taxa, n_trees, n_groups  int32 [K]   (2 <= n_groups <= n_taxa <= 128)
 *   tree_off    int32, per node n_trees+1 leaf offsets starting at 0, nodes concatenated
 *   leaf_taxon / adj_depth / adj_val       the nodes' tables, concatenated (taxon ids 0..n_taxa-1,
 *                                          numbered so that every contraction group is a
 *                                          consecutive range)
 *   tree_w      fp64, n_trees per node, concatenated
 *   group_start int32, per node n_groups+1 entries 0 .. n_taxa, concatenated
 *   maps_out    fp64 [sum n_groups][2]     lambda_out fp64 [K][3] (three largest eigenvalues of S)
 *   w_out       fp64, per node n_groups^2 (row-major), concatenated, or NULL */
int scs_small_solve(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa, const int32_t *n_trees,
                    const int32_t *n_groups, const int32_t *tree_off, const int32_t *leaf_taxon,
                    const int32_t *adj_depth, const double *adj_val, const double *tree_w,
                    const int32_t *group_start, double *maps_out, double *lambda_out,
                    double *w_out);

/* The same in two halves.  _begin copies the inputs into a page-locked block of the context,
 * enqueues the upload, the three launches and the download of the results and returns a ticket
 * without waiting; _end waits for that ticket's results and copies them out (NULL outputs: the
 * results are dropped).  Several tickets may be outstanding on a context; they complete in the
 * order they were begun.  The recursion begins a child's solve as soon as its tables are
 * flattened: the next child is flattened on the host meanwhile, and a right sibling's embedding is
 * long there when the walk arrives (reference order of the walk: scs.py:139-171).
 * want_w != 0: the contracted weight matrices are kept for _end's w_out.  The input arrays may be
 * reused as soon as _begin returns. */
int scs_small_solve_begin(scs_ctx *ctx, int32_t n_nodes, const int32_t *n_taxa,
                          const int32_t *n_trees, const int32_t *n_groups, const int32_t *tree_off,
                          const int32_t *leaf_taxon, const int32_t *adj_depth, const double *adj_val,
                          const double *tree_w, const int32_t *group_start, int32_t want_w,
                          int32_t *ticket_out);
/* scs_small_solve_begin for ONE node whose tables are resident -- a child of scs_forest_split
 * (declared above): its leaf arrays are packed on the device, only the group boundaries and the
 * renumbering travel.  relabel[x] (may be null: identity) = id of the forest's taxon x in the node's
 * numbering (present taxa only, contraction groups consecutive); n_taxa = the node's taxon count. */
int scs_small_solve_begin_forest(scs_ctx *ctx, const scs_forest *forest, const int32_t *relabel,
                                 int32_t n_taxa, int32_t n_groups, const int32_t *group_start,
                                 int32_t want_w, int32_t *ticket_out);
/* scs_small_solve_begin for the K small nodes of ONE LEVEL of the recursion, straight from the level
 * forest's resident tables (scs_forest_split_level; reference: scs.py:110-134 for every node of the
 * level): node k owns the trees [t_begin[k], t_begin[k] + n_trees[k]) with n_leaves[k] leaves in all and
 * the taxon ids [u_base[k], u_base[k] + u_size[k]); relabel (all nodes' maps concatenated, u_size[k]
 * entries each) sends id x to relabel[.. + x - u_base[k]], the node's own numbering (present taxa only,
 * contraction groups consecutive); n_taxa / n_groups / group_start as scs_small_solve_begin. */
int scs_small_solve_begin_level(scs_ctx *ctx, const scs_forest *forest, int32_t n_nodes,
                                const int32_t *t_begin, const int32_t *n_trees, const int64_t *n_leaves,
                                const int32_t *u_base, const int32_t *u_size, const int32_t *relabel,
                                const int32_t *n_taxa, const int32_t *n_groups, const int32_t *group_start,
                                int32_t want_w, int32_t *ticket_out);
int scs_small_solve_end(scs_ctx *ctx, int32_t ticket, double *maps_out, double *lambda_out,
                        double *w_out);

/* ---- diagnostics used by the parity tests ------------------------------ */

/* The stop / renew / confirm rules of scs_fiedler's loop (csrc/scs_policy.h; what stands in for ARPACK's
 * tol = 0 behind scs.py:252) on a scripted sequence of residuals, no device involved: actions_out[i] = 0 go
 * on, 1 stop for the confirmation (the next residual of the script is the one measured through W), 2 renew
 * S X / S P through W, 3 / 4 the loop ended converged / not converged, -1 behind the end.  image != 0: the
 * loop starts on the single-precision image of W. */
int scs_debug_loop_policy(double tol, int32_t lowp_mode, double lowp_tol, double lowp_tol2, int32_t image,
                          int32_t n, const double *residuals, int32_t *actions_out);

/* Eigen-decomposition of a dense symmetric n x n matrix (n <= 64) by the
 * device Jacobi kernel that serves the Rayleigh-Ritz step: eigenvalues
 * descending in w[n], eigenvectors in the columns of v (n x n row-major). */
int scs_debug_jacobi(scs_ctx *ctx, const double *a, int32_t n, double *w, double *v);

/* out (ka x kb, row-major) = A^T B for host matrices A (n x ka), B (n x kb),
 * row-major, through the device Gram kernel; use_mfma selects the
 * v_mfma_f64_16x16x4_f64 variant. */
int scs_debug_gram(scs_ctx *ctx, const double *a, const double *b, int32_t n, int32_t ka,
                   int32_t kb, int32_t use_mfma, double *out);

/* y (rows x b) = S[row block] * x for a host x (V x b): one launch of the
 * SYMM kernel with the graph's degree scaling. */
int scs_debug_apply(scs_ctx *ctx, scs_graph *graph, const double *x, int32_t b, double *y);

/* One grouped round of ncclSend + ncclRecv from this rank to itself through the wrapper the
 * shared build's tile exchange uses (ncclGroupStart ... ncclGroupEnd, ncclFloat64, the
 * context's stream): host_out receives host_in (count doubles).  Needs a context created
 * with an RCCL communicator (scs_ctx_create with world >= 1 and a unique id). */
int scs_debug_comm_selftest(scs_ctx *ctx, int32_t count, const double *host_in, double *host_out);

/* The box's own copy rate, to quote beside the nominal 8 TB/s (SURVEY.md 8d "Peak to quote"):
 * `reps` device-to-device copies of `bytes` bytes on the context's stream, timed with HIP
 * events after one untimed copy; *gbs_out = 2 * bytes * reps / time in GB/s (a copy reads
 * and writes every byte once). */
int scs_debug_copy_bandwidth(scs_ctx *ctx, int64_t bytes, int32_t reps, double *gbs_out);

#ifdef __cplusplus
}
#endif
#endif /* SCS_HIP_H */
