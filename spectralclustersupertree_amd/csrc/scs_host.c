/*
 * Host-side tree arrays: restriction and flattening without Python tree objects
 * (reference: src/sc_supertree/scs.py:411-455 for the
 * restriction, :495-663 for what the flattened tables feed).
 *
 * A forest is stored as node arrays in preorder (a node's first child is the next
 * index), all trees concatenated, tree t owning nodes [node_off[t], node_off[t+1]):
 *   parent[i]   int32  index of the parent INSIDE the tree (relative), -1 for the root
 *   taxon[i]    int32  taxon id of a leaf, -1 for an internal node
 *   length[i]   fp64   branch length above the node, NaN = None
 *   support[i]  fp64   support of the node, NaN = None
 *
 * scs_host_restrict keeps the leaves whose taxon is marked, drops nodes left without
 * leaves, splices out nodes left with one child -- bottom-up, the parent's length added
 * in front of the child's accumulated one, exactly the order of
 * spectralclustersupertree_amd/tree.py:get_sub_tree -- lets a root left with one child
 * collapse onto it, and drops trees left with fewer than two leaves.
 *
 * scs_host_flatten produces the leaf_taxon / adj_depth / adj_val tables of
 * include/scs_hip.h from the arrays (the same values flatten.py computes from tree
 * objects, bit for bit: one running value per root path, the same additions in the
 * same order).
 *
 * Plain C, no GPU code; bound with ctypes in spectralclustersupertree_amd/treearrays.py.
 */
#include <malloc.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define SCS_HOST_OK 0
#define SCS_HOST_ENOMEM (-1)
#define SCS_HOST_EINVAL (-2)
#define SCS_HOST_ENOSUPPORT (-3) /* bootstrap strategy met an internal node without support */

/*
 * The trees of a forest are independent in everything below, and a node of the recursion
 * walks the whole forest of its parent (a child of three taxa split off a parent of 30 000
 * still visits every node of every tree): forests of more than a few tens of thousands of
 * nodes are cut over a persistent team of threads, trees handed out one at a time from a shared
 * counter.  SCS_HOST_THREADS sets the team size (default: the online cores, at most 32; 1 =
 * serial).
 */

typedef int (*tree_fn)(int32_t t, void *scratch, void *ctx);
typedef struct {
    int32_t n_trees;
    int32_t next;  /* shared counter (atomic) */
    int32_t chunk; /* trees handed out per visit to the counter (tiny trees: the counter's cache
                      line would otherwise bounce between the threads once per microsecond) */
    tree_fn fn;
    void *ctx;
    size_t scratch_bytes;
    int rc; /* first error (atomic) */
} tree_team;

static void *tree_team_worker(void *arg) {
    tree_team *tm = (tree_team *)arg;
    void *scratch = calloc(tm->scratch_bytes ? tm->scratch_bytes : 1, 1); /* zeroed: split_tree relies on it */
    if (!scratch) {
        __atomic_store_n(&tm->rc, SCS_HOST_ENOMEM, __ATOMIC_RELAXED);
        return 0;
    }
    const int32_t chunk = tm->chunk > 0 ? tm->chunk : 1;
    for (;;) {
        const int32_t t0 = __atomic_fetch_add(&tm->next, chunk, __ATOMIC_RELAXED);
        if (t0 >= tm->n_trees || __atomic_load_n(&tm->rc, __ATOMIC_RELAXED) != SCS_HOST_OK) break;
        const int32_t t1 = t0 + chunk < tm->n_trees ? t0 + chunk : tm->n_trees;
        for (int32_t t = t0; t < t1; ++t) {
            const int rc = tm->fn(t, scratch, tm->ctx);
            if (rc != SCS_HOST_OK) {
                __atomic_store_n(&tm->rc, rc, __ATOMIC_RELAXED);
                break;
            }
        }
    }
    free(scratch);
    return 0;
}

static int host_threads(void) {
    static int cached = 0;
    if (!cached) {
        const char *e = getenv("SCS_HOST_THREADS");
        long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
        if (!e && n > 32) n = 32;
        if (n < 1) n = 1;
        if (n > 64) n = 64;
        cached = (int)n;
    }
    return cached;
}

/*
 * A persistent team.  Creating and joining threads per call (round 2) only paid for forests of
 * hundreds of thousands of nodes; the recursion makes tens of thousands of calls on forests of
 * 10^4..10^5 nodes (a node of a dozen taxa still carries every source tree), each a
 * millisecond of serial work.  The workers are created once, sleep on a condition variable
 * between jobs and are handed a job by generation number; the caller works along and waits
 * for the last worker.  One job at a time: a caller that finds the team busy (another host
 * thread of the process is in here) simply runs its job alone.
 */
typedef struct {
    pthread_mutex_t m;
    pthread_cond_t cv_work, cv_done;
    pthread_mutex_t busy;
    tree_team *job;
    unsigned long generation;
    int want;    /* workers asked to join the current job */
    int joined;  /* ... that have picked it up */
    int active;  /* ... still working */
    int started; /* threads created */
    pthread_t th[64];
} team_pool;

static team_pool g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
                           PTHREAD_MUTEX_INITIALIZER, 0, 0, 0, 0, 0, 0, {0}};

static void *pool_worker(void *arg) {
    (void)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.m);
    for (;;) {
        while (g_pool.generation == seen || g_pool.joined >= g_pool.want) {
            if (g_pool.generation != seen) seen = g_pool.generation; /* job is fully staffed: skip it */
            pthread_cond_wait(&g_pool.cv_work, &g_pool.m);
        }
        seen = g_pool.generation;
        g_pool.joined += 1;
        tree_team *job = g_pool.job;
        pthread_mutex_unlock(&g_pool.m);
        tree_team_worker(job);
        pthread_mutex_lock(&g_pool.m);
        if (--g_pool.active == 0) pthread_cond_signal(&g_pool.cv_done);
    }
    return 0;
}

/* fn(t, scratch, ctx) for every tree t; scratch is a per-thread block of scratch_bytes */
static int for_each_tree(int32_t n_trees, int64_t total_nodes, size_t scratch_bytes, tree_fn fn,
                         void *ctx) {
    tree_team tm = {n_trees, 0, 1, fn, ctx, scratch_bytes, SCS_HOST_OK};
    int n_thr = host_threads();
    /* a few thousand nodes of work per thread before another one is worth waking (a deep node
     * of configs[4] is 5 000 trees of a dozen nodes: 5 ms alone, a fraction of that shared) */
    if (total_nodes < 16384) n_thr = 1; /* (waking the team costs more than such a job) */
    if (total_nodes / 4096 + 1 < n_thr) n_thr = (int)(total_nodes / 4096 + 1);
    if (n_thr > n_trees) n_thr = n_trees;
    if (n_thr <= 1 || pthread_mutex_trylock(&g_pool.busy) != 0) {
        tm.chunk = n_trees > 0 ? n_trees : 1;
        tree_team_worker(&tm);
        return tm.rc;
    }
    /* about eight visits to the counter per thread, at most 64 trees each */
    tm.chunk = n_trees / (n_thr * 8);
    if (tm.chunk < 1) tm.chunk = 1;
    if (tm.chunk > 64) tm.chunk = 64;
    pthread_mutex_lock(&g_pool.m);
    while (g_pool.started < n_thr - 1 && g_pool.started < 63) {
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        const int rc = pthread_create(&g_pool.th[g_pool.started], &at, pool_worker, 0);
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        g_pool.started += 1;
    }
    const int helpers = n_thr - 1 < g_pool.started ? n_thr - 1 : g_pool.started;
    g_pool.job = &tm;
    g_pool.want = helpers;
    g_pool.joined = 0;
    g_pool.active = helpers;
    g_pool.generation += 1;
    if (helpers > 0) pthread_cond_broadcast(&g_pool.cv_work);
    pthread_mutex_unlock(&g_pool.m);
    tree_team_worker(&tm); /* the caller is a member of the team */
    pthread_mutex_lock(&g_pool.m);
    while (g_pool.active > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.m);
    g_pool.job = 0;
    g_pool.want = 0;
    pthread_mutex_unlock(&g_pool.m);
    pthread_mutex_unlock(&g_pool.busy);
    return tm.rc;
}

/*
 * Pass 1 of a restriction: sizes of the result.
 *   keep[taxon] != 0 marks the taxa to keep.
 *   out_tree_keep[t]  1 if tree t survives (>= 2 kept leaves)
 *   out_nodes[t]      nodes of the restricted tree (0 if dropped)
 * Returns 0 or a negative error.
 */
static int restrict_tree_count(const int32_t *parent, const int32_t *taxon, int32_t k,
                               const uint8_t *keep, int32_t *cnt, int32_t *nkc) {
    /* cnt[i]: kept leaves below i; nkc[i]: children with cnt > 0 */
    for (int32_t i = 0; i < k; ++i) {
        cnt[i] = (taxon[i] >= 0 && keep[taxon[i]]) ? 1 : 0;
        nkc[i] = 0;
    }
    for (int32_t i = k - 1; i > 0; --i) {
        const int32_t p = parent[i];
        if (p < 0 || p >= i) return SCS_HOST_EINVAL; /* not preorder */
        if (cnt[i] > 0) {
            cnt[p] += cnt[i];
            nkc[p] += 1;
        }
    }
    return SCS_HOST_OK;
}

typedef struct {
    const int64_t *node_off;
    const int32_t *parent, *taxon;
    const uint8_t *keep;
    uint8_t *out_tree_keep;
    int32_t *out_nodes;
    int32_t max_k;
} sizes_ctx;

static int restrict_sizes_tree(int32_t t, void *scratch, void *vctx) {
    const sizes_ctx *c = (const sizes_ctx *)vctx;
    int32_t *cnt = (int32_t *)scratch, *nkc = cnt + c->max_k;
    const int64_t off = c->node_off[t];
    const int32_t k = (int32_t)(c->node_off[t + 1] - off);
    const int rc = restrict_tree_count(c->parent + off, c->taxon + off, k, c->keep, cnt, nkc);
    if (rc != SCS_HOST_OK) return rc;
    if (cnt[0] < 2) {
        c->out_tree_keep[t] = 0;
        c->out_nodes[t] = 0;
        return SCS_HOST_OK;
    }
    int32_t kept = 0;
    for (int32_t i = 0; i < k; ++i)
        if ((c->taxon[off + i] >= 0 && cnt[i] == 1) || (c->taxon[off + i] < 0 && nkc[i] >= 2)) ++kept;
    c->out_tree_keep[t] = 1;
    c->out_nodes[t] = kept;
    return SCS_HOST_OK;
}

int scs_host_restrict_sizes(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                            const int32_t *taxon, const uint8_t *keep, uint8_t *out_tree_keep,
                            int32_t *out_nodes) {
    int32_t max_k = 1;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k < 1 || k > INT32_MAX) return SCS_HOST_EINVAL;
        if (k > max_k) max_k = (int32_t)k;
    }
    sizes_ctx c = {node_off, parent, taxon, keep, out_tree_keep, out_nodes, max_k};
    return for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0,
                         sizeof(int32_t) * 2 * (size_t)max_k, restrict_sizes_tree, &c);
}

/*
 * Pass 2: fill the restricted forest.  new_node_off (for the surviving trees, in order)
 * is the exclusive scan of out_nodes over surviving trees, computed by the caller.
 */
typedef struct {
    const int64_t *node_off;
    const int32_t *parent, *taxon;
    const double *length, *support;
    const uint8_t *keep, *tree_keep;
    const int64_t *new_node_off;
    const int32_t *out_index; /* tree -> its index among the surviving trees */
    int32_t *new_parent, *new_taxon;
    double *new_length, *new_support;
    int32_t max_k;
} fill_ctx;

static int restrict_fill_tree(int32_t t, void *scratch, void *vctx) {
    const fill_ctx *c = (const fill_ctx *)vctx;
    if (!c->tree_keep[t]) return SCS_HOST_OK;
    int32_t *cnt = (int32_t *)scratch, *nkc = cnt + c->max_k;
    int32_t *newidx = nkc + c->max_k; /* kept node -> new index */
    int32_t *anc = newidx + c->max_k; /* nearest kept ancestor-or-self (old index), -1 above the new root */
    const int64_t off = c->node_off[t];
    const int32_t k = (int32_t)(c->node_off[t + 1] - off);
    const int32_t *par = c->parent + off, *tax = c->taxon + off;
    const double *len = c->length + off, *sup = c->support + off;
    const int rc = restrict_tree_count(par, tax, k, c->keep, cnt, nkc);
    if (rc != SCS_HOST_OK) return rc;
    const int64_t noff = c->new_node_off[c->out_index[t]];
    int32_t next = 0;
    /* preorder: a parent is numbered before its children */
    for (int32_t i = 0; i < k; ++i) {
        const int kept = (tax[i] >= 0 && cnt[i] == 1) || (tax[i] < 0 && nkc[i] >= 2);
        const int32_t up = i == 0 ? -1 : anc[par[i]];
        if (!kept) {
            anc[i] = up; /* unary, empty, or above the new root: look through */
            newidx[i] = -1;
            continue;
        }
        anc[i] = i;
        newidx[i] = next;
        c->new_parent[noff + next] = up < 0 ? -1 : newidx[up];
        c->new_taxon[noff + next] = tax[i];
        c->new_support[noff + next] = sup[i];
        /* merged length: fold the spliced chain bottom-up, parent's length in front */
        double acc = len[i];
        for (int32_t u = i == 0 ? -1 : par[i]; u >= 0 && u != up; u = par[u]) {
            /* u lies strictly between the node and its kept ancestor: it had one child left */
            if (!isnan(len[u]) && !isnan(acc)) acc = len[u] + acc;
        }
        /* (a node that becomes the root absorbs the chain up to the old root the same way;
         * a root's length is never used) */
        c->new_length[noff + next] = acc;
        ++next;
    }
    return SCS_HOST_OK;
}

int scs_host_restrict_fill(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                           const int32_t *taxon, const double *length, const double *support,
                           const uint8_t *keep, const uint8_t *tree_keep,
                           const int64_t *new_node_off, int32_t *new_parent, int32_t *new_taxon,
                           double *new_length, double *new_support) {
    int32_t max_k = 1;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k > max_k) max_k = (int32_t)k;
    }
    int32_t *out_index = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_trees > 0 ? n_trees : 1));
    if (!out_index) return SCS_HOST_ENOMEM;
    int32_t out_t = 0;
    for (int32_t t = 0; t < n_trees; ++t) out_index[t] = tree_keep[t] ? out_t++ : -1;
    fill_ctx c = {node_off, parent, taxon, length, support, keep, tree_keep, new_node_off, out_index,
                  new_parent, new_taxon, new_length, new_support, max_k};
    const int rc = for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0,
                                 sizeof(int32_t) * 4 * (size_t)max_k, restrict_fill_tree, &c);
    free(out_index);
    return rc;
}

/* mark[x] = 1 for every taxon that is a leaf of some tree (mark zeroed by the caller; the
 * threads may store the same 1 twice) */
typedef struct {
    const int64_t *node_off;
    const int32_t *taxon;
    uint8_t *mark;
} mark_ctx;

static int present_tree(int32_t t, void *scratch, void *vctx) {
    (void)scratch;
    const mark_ctx *c = (const mark_ctx *)vctx;
    /* (test before the store: once a taxon is marked its cache line stays shared among the
     * threads instead of bouncing between them) */
    for (int64_t i = c->node_off[t]; i < c->node_off[t + 1]; ++i) {
        const int32_t x = c->taxon[i];
        if (x >= 0 && !c->mark[x]) c->mark[x] = 1;
    }
    return SCS_HOST_OK;
}

int scs_host_present(int32_t n_trees, const int64_t *node_off, const int32_t *taxon, uint8_t *mark) {
    mark_ctx c = {node_off, taxon, mark};
    return for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0, 0, present_tree,
                         &c);
}

/* leaves of every tree (out_counts[t]) */
typedef struct {
    const int64_t *node_off;
    const int32_t *taxon;
    int64_t *out;
} count_ctx;

static int leaf_count_tree(int32_t t, void *scratch, void *vctx) {
    (void)scratch;
    const count_ctx *c = (const count_ctx *)vctx;
    int64_t n = 0;
    for (int64_t i = c->node_off[t]; i < c->node_off[t + 1]; ++i) n += c->taxon[i] >= 0;
    c->out[t] = n;
    return SCS_HOST_OK;
}

int scs_host_leaf_counts(int32_t n_trees, const int64_t *node_off, const int32_t *taxon,
                         int64_t *out_counts) {
    count_ctx c = {node_off, taxon, out_counts};
    return for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0, 0,
                         leaf_count_tree, &c);
}

/*
 * Flatten a forest into the device tables.
 *   strategy: 0 one, 1 depth, 2 branch, 3 bootstrap
 *   leaf_off[t]: first leaf slot of tree t (exclusive scan of leaf counts, n_trees + 1)
 *   monotone_out: set to 0 if a negative internal length is met under `branch`
 * Outputs sized leaf_off[n_trees].
 */
typedef struct {
    const int64_t *node_off;
    const int32_t *parent, *taxon;
    const double *length, *support;
    int32_t strategy;
    const int64_t *leaf_off;
    int32_t *leaf_taxon, *adj_depth;
    double *adj_val;
    int32_t *monotone_out;
    int32_t max_k;
    const int32_t *renumber; /* taxon id -> id written to leaf_taxon, or null */
} flatten_ctx;

static int flatten_tree(int32_t t, void *scratch, void *vctx) {
    const flatten_ctx *c = (const flatten_ctx *)vctx;
    double *val = (double *)scratch;
    int32_t *depth = (int32_t *)(val + c->max_k), *nch = depth + c->max_k;
    const int32_t strategy = c->strategy;
    int32_t *leaf_taxon = c->leaf_taxon, *adj_depth = c->adj_depth;
    double *adj_val = c->adj_val;
    const int64_t off = c->node_off[t];
    const int32_t k = (int32_t)(c->node_off[t + 1] - off);
    const int32_t *par = c->parent + off, *tax = c->taxon + off;
    const double *len = c->length + off, *sup = c->support + off;
    int64_t slot = c->leaf_off[t];
    const int64_t slot_end = c->leaf_off[t + 1];
    int rc = SCS_HOST_OK;
    for (int32_t i = 0; i < k; ++i) nch[i] = 0;
    for (int32_t i = 1; i < k; ++i) nch[par[i]] += 1;
    int first_leaf = 1;
    int32_t pend_depth = 0;
    double pend_val = 0.0;
    depth[0] = 0;
    val[0] = 0.0;
    if (tax[0] >= 0) { /* a single-leaf tree */
        if (slot >= slot_end) return SCS_HOST_EINVAL;
        leaf_taxon[slot] = c->renumber ? c->renumber[tax[0]] : tax[0];
        adj_depth[slot] = 0;
        adj_val[slot] = 0.0;
        return SCS_HOST_OK;
    }
    for (int32_t i = 1; i < k && rc == SCS_HOST_OK; ++i) {
        const int32_t u = par[i];
        if (i != u + 1) { /* not the first child: the next leaf's LCA with the previous one is u */
            pend_depth = depth[u];
            pend_val = val[u];
        }
        if (tax[i] >= 0) {
            if (slot >= slot_end) return SCS_HOST_EINVAL;
            if (!first_leaf) {
                adj_depth[slot - 1] = pend_depth;
                adj_val[slot - 1] = pend_val;
            }
            first_leaf = 0;
            leaf_taxon[slot++] = c->renumber ? c->renumber[tax[i]] : tax[i];
            continue;
        }
        depth[i] = depth[u] + 1;
        double v;
        switch (strategy) {
            case 0:
                v = 1.0;
                break;
            case 1:
                v = val[u] + 1.0;
                break;
            case 2:
                v = val[u] + (isnan(len[i]) ? 1.0 : len[i]);
                if (!isnan(len[i]) && len[i] < 0.0) __atomic_store_n(c->monotone_out, 0, __ATOMIC_RELAXED);
                break;
            default:
                v = sup[i];
                if (isnan(v)) {
                    if (nch[i] >= 2) rc = SCS_HOST_ENOSUPPORT;
                    v = 0.0;
                }
                break;
        }
        val[i] = v;
    }
    if (rc != SCS_HOST_OK) return rc;
    if (slot != slot_end) return SCS_HOST_EINVAL;
    /* padding slot so adj_* share the offsets of leaf_taxon */
    adj_depth[slot_end - 1] = 0;
    adj_val[slot_end - 1] = 0.0;
    return SCS_HOST_OK;
}

int scs_host_flatten(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                     const int32_t *taxon, const double *length, const double *support,
                     int32_t strategy, const int64_t *leaf_off, int32_t *leaf_taxon,
                     int32_t *adj_depth, double *adj_val, int32_t *monotone_out,
                     const int32_t *renumber) {
    if (strategy < 0 || strategy > 3) return SCS_HOST_EINVAL;
    int32_t max_k = 1;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k < 1 || k > INT32_MAX) return SCS_HOST_EINVAL;
        if (k > max_k) max_k = (int32_t)k;
    }
    flatten_ctx c = {node_off, parent, taxon, length, support, strategy, leaf_off, leaf_taxon,
                     adj_depth, adj_val, monotone_out, max_k, renumber};
    return for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0,
                         (sizeof(double) + 2 * sizeof(int32_t)) * (size_t)max_k, flatten_tree, &c);
}

/* ---------------------------------------------------------------------------
 * Newick text -> tree arrays (reference: load.py:7-23, one
 * tree per line, every line handed to the parser).  The grammar accepted is the one of
 * spectralclustersupertree_amd/tree.py:make_tree, with the same meaning: a label on a
 * node with children that parses as a number is its support; ':' introduces a branch
 * length (empty = None); '[...]' comments are skipped; labels may be quoted with ''
 * as the escaped quote; everything after ';' on a line is ignored.
 *
 * Two passes over the same text:
 *   scs_host_newick_scan   counts trees (lines), nodes and the bytes of the leaf names
 *   scs_host_newick_parse  fills parent / length / support, and for every leaf the offset
 *                          of its NUL-terminated name in name_pool (-1 for internal nodes)
 * then scs_host_names_rank maps the leaf names to the ranks of the sorted distinct names
 * (the taxon ids the rest of the package uses) with an open-addressing hash table.
 * Errors: a negative code; *err_line receives the 0-based line of the offending tree.
 * ------------------------------------------------------------------------- */
#define SCS_HOST_EPARSE (-4)

static int is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

/* parse one line [s, e); mode 0 counts, mode 1 fills.  Node indices are relative to the tree. */
static int newick_line(const char *s, const char *e, int mode, int64_t *n_nodes, int64_t *name_bytes,
                       int32_t *parent, double *length, double *support, int64_t *name_off,
                       char *pool, int64_t *pool_at, int32_t *nchild, int32_t *stack, int64_t stack_cap) {
    int64_t count = 1; /* the root */
    int64_t depth = 0; /* stack[depth] = current node */
    int seen_any = 0;
    int32_t cur = 0;
    if (mode) {
        parent[0] = -1;
        length[0] = NAN;
        support[0] = NAN;
        name_off[0] = -1;
        nchild[0] = 0;
        stack[0] = 0;
    }
    const char *p = s;
    while (p < e && is_space(*p)) ++p;
    if (p >= e) return SCS_HOST_EPARSE; /* empty line */
    while (p < e) {
        const char c = *p;
        if (is_space(c)) {
            ++p;
        } else if (c == '[') {
            const char *q = p;
            while (q < e && *q != ']') ++q;
            if (q >= e) return SCS_HOST_EPARSE;
            p = q + 1;
        } else if (c == '(' || c == ',') {
            if (c == ',') {
                if (depth == 0) return SCS_HOST_EPARSE; /* ',' at top level */
                --depth;                              /* back to the parent */
            }
            if (depth + 1 >= stack_cap) return SCS_HOST_EPARSE;
            if (mode) {
                const int32_t par = stack[depth];
                cur = (int32_t)count;
                parent[cur] = par;
                length[cur] = NAN;
                support[cur] = NAN;
                name_off[cur] = -1;
                nchild[cur] = 0;
                nchild[par] += 1;
                stack[depth + 1] = cur;
            }
            ++depth;
            ++count;
            seen_any = seen_any || c == '(';
            ++p;
        } else if (c == ')') {
            if (depth == 0) return SCS_HOST_EPARSE;
            --depth;
            if (mode) cur = stack[depth];
            ++p;
        } else if (c == ';') {
            break;
        } else if (c == ':') {
            const char *q = p + 1;
            while (q < e && *q != ',' && *q != '(' && *q != ')' && *q != ';' && *q != '[') ++q;
            if (mode) {
                const char *a = p + 1, *b = q;
                while (a < b && is_space(*a)) ++a;
                while (b > a && is_space(b[-1])) --b;
                if (a == b) {
                    length[cur] = NAN;
                } else {
                    char buf[64];
                    const size_t len = (size_t)(b - a);
                    if (len >= sizeof(buf)) return SCS_HOST_EPARSE;
                    memcpy(buf, a, len);
                    buf[len] = 0;
                    char *end = NULL;
                    const double v = strtod(buf, &end);
                    if (end == buf || *end != 0) return SCS_HOST_EPARSE;
                    length[cur] = v;
                }
            }
            p = q;
        } else {
            /* a label: quoted or bare */
            const char *a, *b; /* [a, b) raw label text */
            int quoted = 0;
            if (c == '\'') {
                quoted = 1;
                const char *q = p + 1;
                for (;;) {
                    if (q >= e) return SCS_HOST_EPARSE;
                    if (*q == '\'') {
                        if (q + 1 < e && q[1] == '\'') {
                            q += 2;
                            continue;
                        }
                        break;
                    }
                    ++q;
                }
                a = p + 1;
                b = q;
                p = q + 1;
            } else {
                const char *q = p;
                while (q < e && *q != ',' && *q != '(' && *q != ')' && *q != ':' && *q != ';' && *q != '[') ++q;
                a = p;
                b = q;
                while (b > a && is_space(b[-1])) --b;
                p = q;
            }
            seen_any = 1;
            const int64_t raw = b - a;
            if (!mode) {
                *name_bytes += raw + 1; /* upper bound: every label could be a leaf's */
            } else {
                /* unescape into the pool (tentatively) */
                const int64_t at = *pool_at;
                int64_t w = at;
                for (const char *q = a; q < b; ++q) {
                    pool[w++] = *q;
                    if (quoted && *q == '\'' && q + 1 < b && q[1] == '\'') ++q;
                }
                pool[w] = 0;
                if (nchild[cur] > 0) {
                    /* internal node: a numeric label is the support; the name is not kept */
                    char *end = NULL;
                    const double v = strtod(pool + at, &end);
                    if (end != pool + at && *end == 0 && w > at) support[cur] = v;
                } else {
                    name_off[cur] = at;
                    *pool_at = w + 1;
                }
            }
        }
    }
    if (depth != 0) return SCS_HOST_EPARSE; /* missing ')' */
    if (!seen_any) return SCS_HOST_EPARSE;
    *n_nodes = count;
    return SCS_HOST_OK;
}

static const char *line_end(const char *p, const char *end) {
    while (p < end && *p != '\n') ++p;
    return p;
}

int scs_host_newick_scan(const char *text, int64_t len, int64_t *n_trees, int64_t *n_nodes,
                         int64_t *name_bytes, int64_t *max_depth_nodes, int64_t *err_line) {
    const char *p = text, *end = text + len;
    int64_t trees = 0, nodes = 0, bytes = 0, max_nodes = 0;
    while (p < end) {
        const char *e = line_end(p, end);
        int64_t k = 0;
        const int rc = newick_line(p, e, 0, &k, &bytes, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                                   INT64_MAX);
        if (rc != SCS_HOST_OK) {
            *err_line = trees;
            return rc;
        }
        nodes += k;
        if (k > max_nodes) max_nodes = k;
        ++trees;
        p = e < end ? e + 1 : e;
    }
    *n_trees = trees;
    *n_nodes = nodes;
    *name_bytes = bytes + 1;
    *max_depth_nodes = max_nodes;
    return SCS_HOST_OK;
}

int scs_host_newick_parse(const char *text, int64_t len, int64_t n_trees, int64_t max_nodes,
                          int64_t *node_off, int32_t *parent, double *length, double *support,
                          int64_t *name_off, char *name_pool, int64_t *pool_used, int64_t *err_line) {
    const char *p = text, *end = text + len;
    int32_t *nchild = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_nodes + 1));
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_nodes + 2));
    if (!nchild || !stack) {
        free(nchild);
        free(stack);
        return SCS_HOST_ENOMEM;
    }
    int64_t at = 0, pool_at = 0, t = 0;
    int rc = SCS_HOST_OK;
    node_off[0] = 0;
    while (p < end && t < n_trees) {
        const char *e = line_end(p, end);
        int64_t k = 0, unused = 0;
        rc = newick_line(p, e, 1, &k, &unused, parent + at, length + at, support + at, name_off + at,
                         name_pool, &pool_at, nchild, stack, max_nodes + 2);
        if (rc != SCS_HOST_OK) {
            *err_line = t;
            break;
        }
        at += k;
        node_off[++t] = at;
        p = e < end ? e + 1 : e;
    }
    *pool_used = pool_at;
    free(nchild);
    free(stack);
    return rc;
}

static uint64_t fnv1a(const char *s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
    return h;
}

static const char *g_sort_pool;
static int cmp_name(const void *a, const void *b) {
    return strcmp(g_sort_pool + *(const int64_t *)a, g_sort_pool + *(const int64_t *)b);
}

/*
 * taxon[i] = rank of node i's name among the sorted distinct leaf names (-1 for internal
 * nodes).  uniq_off (capacity n_nodes) receives the pool offsets of the distinct names in
 * sorted order; *n_taxa their count.  (Not re-entrant: uses one static for qsort.)
 */
int scs_host_names_rank(const char *name_pool, const int64_t *name_off, int64_t n_nodes,
                        int32_t *taxon, int64_t *uniq_off, int64_t *n_taxa) {
    int64_t leaves = 0;
    for (int64_t i = 0; i < n_nodes; ++i) leaves += name_off[i] >= 0;
    uint64_t cap = 16;
    while (cap < (uint64_t)leaves * 2 + 1) cap <<= 1;
    int64_t *slot_off = (int64_t *)malloc(sizeof(int64_t) * cap);
    int32_t *slot_id = (int32_t *)malloc(sizeof(int32_t) * cap);
    if (!slot_off || !slot_id) {
        free(slot_off);
        free(slot_id);
        return SCS_HOST_ENOMEM;
    }
    for (uint64_t i = 0; i < cap; ++i) slot_off[i] = -1;
    int64_t n_uniq = 0;
    /* first pass: distinct names in order of first appearance; taxon = provisional id */
    for (int64_t i = 0; i < n_nodes; ++i) {
        if (name_off[i] < 0) {
            taxon[i] = -1;
            continue;
        }
        const char *nm = name_pool + name_off[i];
        uint64_t h = fnv1a(nm) & (cap - 1);
        while (slot_off[h] >= 0 && strcmp(name_pool + slot_off[h], nm) != 0) h = (h + 1) & (cap - 1);
        if (slot_off[h] < 0) {
            slot_off[h] = name_off[i];
            slot_id[h] = (int32_t)n_uniq;
            uniq_off[n_uniq++] = name_off[i];
        }
        taxon[i] = slot_id[h];
    }
    /* rank of every provisional id in sorted-name order */
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_uniq > 0 ? n_uniq : 1));
    int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_uniq > 0 ? n_uniq : 1));
    if (!order || !rank) {
        free(order);
        free(rank);
        free(slot_off);
        free(slot_id);
        return SCS_HOST_ENOMEM;
    }
    memcpy(order, uniq_off, sizeof(int64_t) * (size_t)n_uniq);
    g_sort_pool = name_pool;
    qsort(order, (size_t)n_uniq, sizeof(int64_t), cmp_name);
    /* provisional id of a sorted name: look it up again */
    for (int64_t r = 0; r < n_uniq; ++r) {
        const char *nm = name_pool + order[r];
        uint64_t h = fnv1a(nm) & (cap - 1);
        while (strcmp(name_pool + slot_off[h], nm) != 0) h = (h + 1) & (cap - 1);
        rank[slot_id[h]] = (int32_t)r;
    }
    for (int64_t i = 0; i < n_nodes; ++i)
        if (taxon[i] >= 0) taxon[i] = rank[taxon[i]];
    memcpy(uniq_off, order, sizeof(int64_t) * (size_t)n_uniq);
    *n_taxa = n_uniq;
    free(order);
    free(rank);
    free(slot_off);
    free(slot_id);
    return SCS_HOST_OK;
}

/* ---------------------------------------------------------------------------
 * Contraction groups from the flattened tables (reference: scs.py:298-334; the equivalence
 * with "identical (tree, root side) signatures" is argued in flatten.contraction_groups,
 * whose result this reproduces): partition refinement, one tree at a time, hashing the
 * pair (current class, root side in this tree; 0 = absent).  groups[x] is the class of taxon
 * x numbered by smallest member.  O(n_taxa) per tree, stops as soon as every taxon is alone.
 * ------------------------------------------------------------------------- */
int scs_host_contraction_groups(int32_t n_taxa, int32_t n_trees, const int64_t *tree_off,
                                const int32_t *leaf_taxon, const int32_t *adj_depth,
                                int32_t *groups) {
    if (n_taxa <= 0) return SCS_HOST_OK;
    int32_t *cls = (int32_t *)calloc((size_t)n_taxa, sizeof(int32_t));
    int32_t *side = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_taxa);
    uint64_t cap = 16;
    int cap_bits = 4;
    while (cap < (uint64_t)n_taxa * 2 + 1) {
        cap <<= 1;
        ++cap_bits;
    }
    uint64_t *hkey = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    int32_t *hval = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_taxa);
    if (!cls || !side || !hkey || !hval || !first) {
        free(cls);
        free(side);
        free(hkey);
        free(hval);
        free(first);
        return SCS_HOST_ENOMEM;
    }
    int32_t n_cls = 1;
    for (int32_t t = 0; t < n_trees && n_cls < n_taxa; ++t) {
        memset(side, 0, sizeof(int32_t) * (size_t)n_taxa);
        int32_t s = 1;
        for (int64_t p = tree_off[t]; p < tree_off[t + 1]; ++p) {
            side[leaf_taxon[p]] = s;
            if (p + 1 < tree_off[t + 1] && adj_depth[p] == 0) ++s; /* a root gap: next side */
        }
        for (uint64_t i = 0; i < cap; ++i) hkey[i] = UINT64_MAX;
        int32_t next = 0;
        for (int32_t x = 0; x < n_taxa; ++x) {
            const uint64_t key = ((uint64_t)(uint32_t)cls[x] << 32) | (uint32_t)side[x];
            /* the TOP bits of the product: the class sits in the key's upper half and only
             * reaches the upper half of the product (bits 17.. of it ignored the class and made
             * every probe sequence collide: quadratic) */
            uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64 - cap_bits);
            while (hkey[h] != UINT64_MAX && hkey[h] != key) h = (h + 1) & (cap - 1);
            if (hkey[h] == UINT64_MAX) {
                hkey[h] = key;
                hval[h] = next++;
            }
            cls[x] = hval[h];
        }
        n_cls = next;
    }
    /* number the classes by smallest member: classes were numbered in order of first
     * appearance over x = 0, 1, ..., which is exactly that order */
    for (int32_t c = 0; c < n_cls; ++c) first[c] = -1;
    int32_t next = 0;
    for (int32_t x = 0; x < n_taxa; ++x) {
        if (first[cls[x]] < 0) first[cls[x]] = next++;
        groups[x] = first[cls[x]];
    }
    free(cls);
    free(side);
    free(hkey);
    free(hval);
    free(first);
    return SCS_HOST_OK;
}

/* ---------------------------------------------------------------------------
 * Connected components of the proper cluster graph from the flattened tables
 * (reference: scs.py:458-492, 651-652): two taxa are adjacent iff they share a root side
 * in some tree, so the components are those of "union the leaves of every root side".
 * Union-find (smaller index becomes the root, path halving); labels[x] = component of x,
 * components numbered by smallest member.
 * ------------------------------------------------------------------------- */
static int32_t uf_find(int32_t *parent, int32_t x) {
    while (parent[x] != x) {
        parent[x] = parent[parent[x]];
        x = parent[x];
    }
    return x;
}

int scs_host_components(int32_t n_taxa, int32_t n_trees, const int64_t *tree_off,
                        const int32_t *leaf_taxon, const int32_t *adj_depth, int32_t *labels) {
    if (n_taxa <= 0) return SCS_HOST_OK;
    int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_taxa);
    if (!parent) return SCS_HOST_ENOMEM;
    for (int32_t x = 0; x < n_taxa; ++x) parent[x] = x;
    int32_t n_sets = n_taxa; /* one set left: every further union is a no-op, stop reading */
    for (int32_t t = 0; t < n_trees && n_sets > 1; ++t) {
        int32_t rep = -1; /* first leaf of the current root side */
        for (int64_t p = tree_off[t]; p < tree_off[t + 1] && n_sets > 1; ++p) {
            const int32_t x = leaf_taxon[p];
            if (rep < 0) {
                rep = x;
            } else {
                int32_t a = uf_find(parent, rep), b = uf_find(parent, x);
                if (a != b) {
                    if (a < b) parent[b] = a;
                    else parent[a] = b;
                    --n_sets;
                }
            }
            if (adj_depth[p] == 0) rep = -1; /* a root gap (or the tree's padding slot) ends the side */
        }
    }
    int32_t next = 0;
    for (int32_t x = 0; x < n_taxa; ++x) {
        const int32_t r = uf_find(parent, x);
        /* roots are smallest members, and they are met in increasing order */
        if (r == x) labels[x] = next++;
        else labels[x] = labels[r];
    }
    free(parent);
    return SCS_HOST_OK;
}

/* ---------------------------------------------------------------------------
 * One-pass multi-way restriction (reference: scs.py:139-155 with :411-455 -- the recursion
 * restricts the trees to EVERY part of a split).  scs_host_restrict_* walk the whole parent
 * forest once per child; here one sweep per tree serves all parts at once and the work per
 * part is proportional to the part, not to the parent:
 *   - a preorder sweep keeps the current root path on a stack; for a leaf y of part c whose
 *     previous leaf of the same part was x, LCA(x, y) is the deepest entry of the stack with
 *     preorder index <= x (binary search: the stack's indices increase with depth);
 *   - the restricted tree of part c consists of c's leaves and those LCAs (exactly the
 *     internal nodes left with two or more non-empty children); in preorder they are simply
 *     sorted by their old index, and a node's new parent is its nearest ancestor in that
 *     list (interval test with the end of the old subtree);
 *   - merged branch lengths fold the spliced chain bottom-up with the parent's length in
 *     front, as scs_host_restrict_fill and tree.py:get_sub_tree do: same bits.
 * Trees left with fewer than two leaves of a part are dropped for that part.  Taxa are
 * renumbered through new_id[] (the recursion numbers a child's taxa 0..k-1), so nothing a
 * child does later is proportional to the number of taxa of the whole input.
 * The plan holds every part's trees (thread-local arenas); scs_host_split_fill copies one
 * part out in tree order.
 * ------------------------------------------------------------------------- */
typedef struct {
    int32_t tree, part, n_nodes, n_leaves;
    int64_t off; /* into the plan's node region */
} split_entry;

typedef struct {
    split_entry *entries;
    int64_t n_entries, cap_entries;
} split_arena;

/* The restricted trees of ALL parts are written into one region, tree t owning the slice
 * [slice_off[t], slice_off[t + 1]) of 2 * leaves(t) nodes (the parts of a tree hold its leaves
 * once and at most leaves - 1 LCAs between them): no growing buffers, so the threads never
 * meet in the allocator (a realloc that moves a mapping takes the address space's write lock
 * and stalls every other thread's page faults -- the first version did not scale past one
 * thread).  The region is kept per host thread and reused by the next split (fresh mappings
 * cost a page fault per 4 KiB: seconds at 10^9 nodes). */
typedef struct {
    int32_t *parent, *taxon;
    double *length, *support;
    int64_t cap;
} split_region;

static __thread split_region t_region = {0, 0, 0, 0, 0};

static int region_reserve(int64_t nodes) {
    split_region *r = &t_region;
    if (nodes <= r->cap) return SCS_HOST_OK;
    free(r->parent);
    free(r->taxon);
    free(r->length);
    free(r->support);
    r->cap = 0;
    const int64_t cap = nodes + nodes / 8 + 1024;
    r->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    r->taxon = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    r->length = (double *)malloc(sizeof(double) * (size_t)cap);
    r->support = (double *)malloc(sizeof(double) * (size_t)cap);
    if (!r->parent || !r->taxon || !r->length || !r->support) {
        free(r->parent);
        free(r->taxon);
        free(r->length);
        free(r->support);
        r->parent = r->taxon = 0;
        r->length = r->support = 0;
        return SCS_HOST_ENOMEM;
    }
    r->cap = cap;
    return SCS_HOST_OK;
}

typedef struct scs_split_plan {
    int32_t n_trees, n_parts, n_arenas, arenas_used;
    split_arena *arenas;
    split_region region; /* (borrowed from the calling thread's t_region) */
    int64_t *slice_off;  /* [n_trees + 1] */
    split_entry *sorted;
    int64_t n_sorted;
    int64_t *part_first; /* [n_parts + 1] */
} scs_split_plan;

typedef struct {
    const int64_t *node_off;
    const int32_t *parent, *taxon;
    const double *length, *support;
    const int32_t *part_of, *new_id;
    int32_t n_parts, max_k;
    scs_split_plan *plan;
} split_ctx;

static int arena_reserve(split_arena *a) {
    if (a->n_entries + 1 > a->cap_entries) {
        int64_t cap = a->cap_entries ? a->cap_entries * 2 : 256;
        split_entry *e = (split_entry *)realloc(a->entries, sizeof(split_entry) * (size_t)cap);
        if (!e) return SCS_HOST_ENOMEM;
        a->entries = e;
        a->cap_entries = cap;
    }
    return SCS_HOST_OK;
}

static int cmp_i32(const void *a, const void *b) {
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* scratch of a worker: [int32 arena_plus1][pad] then the per-tree arrays (see split_tree) */
static int split_tree(int32_t t, void *scratch, void *vctx) {
    const split_ctx *c = (const split_ctx *)vctx;
    scs_split_plan *plan = c->plan;
    int32_t *hdr = (int32_t *)scratch;
    if (hdr[0] == 0) hdr[0] = 1 + __atomic_fetch_add(&plan->arenas_used, 1, __ATOMIC_RELAXED);
    if (hdr[0] > plan->n_arenas) return SCS_HOST_EINVAL;
    split_arena *ar = &plan->arenas[hdr[0] - 1];
    const split_region *rg = &plan->region;
    const int32_t mk = c->max_k, np = c->n_parts;
    int32_t *sub_end = hdr + 4;
    int32_t *stack = sub_end + mk;
    int32_t *ev_part = stack + mk;          /* up to 2 mk events */
    int32_t *ev_node = ev_part + 2 * (size_t)mk;
    int32_t *sorted_node = ev_node + 2 * (size_t)mk;
    int32_t *touched = sorted_node + 2 * (size_t)mk; /* mk */
    int32_t *vstack = touched + mk;                  /* mk: positions in the part's node list */
    int32_t *last = vstack + mk;                     /* np: 1 + index of the part's latest leaf in this tree, 0 = none (kept clean) */
    int32_t *cnt = last + np;                        /* np leaves of the part in this tree */
    int32_t *fill = cnt + np;                        /* np write cursor of the bucket sort */

    const int64_t off = c->node_off[t];
    const int32_t k = (int32_t)(c->node_off[t + 1] - off);
    const int32_t *par = c->parent + off, *tax = c->taxon + off;
    const double *len = c->length + off, *sup = c->support + off;

    for (int32_t i = 0; i < k; ++i) sub_end[i] = i;
    for (int32_t i = k - 1; i > 0; --i) {
        const int32_t p = par[i];
        if (p < 0 || p >= i) return SCS_HOST_EINVAL; /* not preorder */
        if (sub_end[i] > sub_end[p]) sub_end[p] = sub_end[i];
    }
    /* Up to eight parts (a spectral split has two): the nodes a part keeps -- its leaves and the LCAs
     * of consecutive ones -- are MARKED, one bit per part, and read off in index order by one
     * branch-free pass; no event lists, no sorting, no duplicates.  (The sorting path below spent
     * most of its time in mispredicted compares of the insertion sort: thousands of trees of a few
     * leaves each.)  The marks live in the event array's space and are cleared by the pass. */
    const int by_marks = np <= 8;
    uint8_t *mark = (uint8_t *)ev_part;
    int32_t sp = 0, n_ev = 0, n_touched = 0;
    for (int32_t i = 0; i < k; ++i) {
        while (sp > 0 && i > sub_end[stack[sp - 1]]) --sp;
        if (tax[i] < 0) {
            stack[sp++] = i;
            continue;
        }
        const int32_t pc = c->part_of[tax[i]];
        if (pc < 0) continue;
        const int32_t x = last[pc] - 1;
        if (x < 0) {
            touched[n_touched++] = pc;
            cnt[pc] = 0;
        } else {
            /* deepest ancestor of i whose index is <= x: the LCA of x and i */
            int32_t lo = 0, hi = sp - 1; /* stack[0] = root <= x */
            while (lo < hi) {
                const int32_t mid = (lo + hi + 1) >> 1;
                if (stack[mid] <= x) lo = mid;
                else hi = mid - 1;
            }
            if (by_marks) {
                mark[stack[lo]] |= (uint8_t)(1u << pc);
            } else {
                ev_part[n_ev] = pc;
                ev_node[n_ev++] = stack[lo];
            }
        }
        if (by_marks) {
            mark[i] |= (uint8_t)(1u << pc);
        } else {
            ev_part[n_ev] = pc;
            ev_node[n_ev++] = i;
        }
        last[pc] = i + 1;
        cnt[pc] += 1;
    }
    /* the node lists of the parts, one after the other (parts with >= 2 leaves only) */
    int32_t total = 0;
    for (int32_t q = 0; q < n_touched; ++q) {
        const int32_t pc = touched[q];
        fill[pc] = total;
        if (cnt[pc] >= 2) total += 2 * cnt[pc] - 1;
    }
    int32_t list_start[8], list_end[8];
    if (by_marks) {
        /* one write cursor per bit.  A cursor stores every node and advances only on its own bit, so
         * it keeps writing one slot past its list: the lists are laid out with a spare slot between
         * them, and the bits of absent or dropped parts write to a dummy */
        int32_t dummy = 0, *cur[8];
        unsigned live = 0;
        int32_t at2 = 0;
        for (int b = 0; b < 8; ++b) cur[b] = &dummy;
        for (int32_t q = 0; q < n_touched; ++q) {
            const int32_t pc = touched[q];
            if (cnt[pc] >= 2) {
                list_start[pc] = at2;
                cur[pc] = sorted_node + at2;
                at2 += 2 * cnt[pc]; /* 2 cnt - 1 nodes at most, and the spare slot */
                live |= 1u << pc;
            }
        }
        for (int32_t i = 0; i < k; ++i) {
            const unsigned m = mark[i] & live;
            mark[i] = 0;
            for (int b = 0; b < np; ++b) {
                *cur[b] = i;
                cur[b] += (m >> b) & 1u;
            }
        }
        for (int32_t q = 0; q < n_touched; ++q) {
            const int32_t pc = touched[q];
            if (cnt[pc] >= 2) list_end[pc] = (int32_t)(cur[pc] - sorted_node);
        }
    } else {
        for (int32_t e = 0; e < n_ev; ++e) {
            const int32_t pc = ev_part[e];
            if (cnt[pc] >= 2) sorted_node[fill[pc]++] = ev_node[e];
        }
    }
    int rc = SCS_HOST_OK;
    int32_t at = 0;
    int64_t cursor = plan->slice_off[t];
    const int64_t slice_end = plan->slice_off[t + 1];
    for (int32_t q = 0; q < n_touched && rc == SCS_HOST_OK; ++q) {
        const int32_t pc = touched[q];
        const int32_t leaves = cnt[pc];
        last[pc] = 0; /* leave the per-part state clean for the next tree */
        if (leaves < 2) continue;
        const int32_t ne = 2 * leaves - 1;
        int32_t *nodes = sorted_node + at;
        at += ne;
        int32_t nv = 0;
        if (by_marks) {
            nodes = sorted_node + list_start[pc]; /* sorted and distinct already */
            nv = list_end[pc] - list_start[pc];
        } else
        /* the leaves come in order already; the LCAs (every second event) do not.  Deep levels
         * of the recursion are thousands of trees of a handful of leaves: a call into qsort per
         * (tree, part) cost more than everything else there */
        if (ne <= 48) {
            for (int32_t j = 1; j < ne; ++j) {
                const int32_t x = nodes[j];
                int32_t i = j - 1;
                while (i >= 0 && nodes[i] > x) {
                    nodes[i + 1] = nodes[i];
                    --i;
                }
                nodes[i + 1] = x;
            }
        } else {
            qsort(nodes, (size_t)ne, sizeof(int32_t), cmp_i32);
        }
        if (!by_marks)
            for (int32_t j = 0; j < ne; ++j)
                if (nv == 0 || nodes[j] != nodes[nv - 1]) nodes[nv++] = nodes[j];
        rc = arena_reserve(ar);
        if (rc != SCS_HOST_OK) break;
        if (cursor + nv > slice_end) {
            rc = SCS_HOST_EINVAL; /* (cannot happen: a tree's parts hold <= 2 * leaves nodes) */
            break;
        }
        const int64_t base = cursor;
        int32_t vs = 0;
        for (int32_t j = 0; j < nv; ++j) {
            const int32_t v = nodes[j];
            while (vs > 0 && v > sub_end[nodes[vstack[vs - 1]]]) --vs;
            const int32_t upj = vs > 0 ? vstack[vs - 1] : -1;
            const int32_t up = upj < 0 ? -1 : nodes[upj];
            rg->parent[base + j] = upj;
            rg->taxon[base + j] = tax[v] >= 0 ? c->new_id[tax[v]] : -1;
            rg->support[base + j] = sup[v];
            double acc = len[v];
            for (int32_t u = v == 0 ? -1 : par[v]; u >= 0 && u != up; u = par[u])
                if (!isnan(len[u]) && !isnan(acc)) acc = len[u] + acc;
            rg->length[base + j] = acc;
            if (tax[v] < 0) vstack[vs++] = j;
        }
        split_entry *en = &ar->entries[ar->n_entries++];
        en->tree = t;
        en->part = pc;
        en->n_nodes = nv;
        en->n_leaves = leaves;
        en->off = base;
        cursor += nv;
    }
    /* (on an error the remaining touched parts still have to be cleaned) */
    for (int32_t q = 0; q < n_touched; ++q) last[touched[q]] = 0;
    return rc;
}

void scs_host_split_end(scs_split_plan *plan) {
    if (!plan) return;
    for (int32_t i = 0; i < plan->n_arenas; ++i) free(plan->arenas[i].entries);
    free(plan->arenas);
    free(plan->slice_off);
    free(plan->sorted);
    free(plan->part_first);
    free(plan);
}

/*
 *   part_of[x]   part of taxon x (0 .. n_parts-1) or -1: the taxon is dropped
 *   new_id[x]    id of taxon x inside its part
 *   part_trees[c], part_nodes[c]  (out) surviving trees / nodes of part c
 */
int scs_host_split_begin(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                         const int32_t *taxon, const double *length, const double *support,
                         const int64_t *leaf_counts, const int32_t *part_of, const int32_t *new_id,
                         int32_t n_parts, scs_split_plan **out_plan, int64_t *part_trees,
                         int64_t *part_nodes) {
    if (n_parts < 1 || !out_plan) return SCS_HOST_EINVAL;
    int32_t max_k = 1;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k < 1 || k > INT32_MAX / 4) return SCS_HOST_EINVAL;
        if (k > max_k) max_k = (int32_t)k;
    }
    scs_split_plan *plan = (scs_split_plan *)calloc(1, sizeof(scs_split_plan));
    if (!plan) return SCS_HOST_ENOMEM;
    plan->n_trees = n_trees;
    plan->n_parts = n_parts;
    plan->n_arenas = 64; /* for_each_tree runs at most 64 threads */
    plan->arenas = (split_arena *)calloc((size_t)plan->n_arenas, sizeof(split_arena));
    plan->part_first = (int64_t *)calloc((size_t)n_parts + 1, sizeof(int64_t));
    plan->slice_off = (int64_t *)calloc((size_t)n_trees + 1, sizeof(int64_t));
    if (!plan->arenas || !plan->part_first || !plan->slice_off) {
        scs_host_split_end(plan);
        return SCS_HOST_ENOMEM;
    }
    for (int32_t t = 0; t < n_trees; ++t) plan->slice_off[t + 1] = plan->slice_off[t] + 2 * leaf_counts[t];
    if (region_reserve(plan->slice_off[n_trees]) != SCS_HOST_OK) {
        scs_host_split_end(plan);
        return SCS_HOST_ENOMEM;
    }
    plan->region = t_region;
    split_ctx c = {node_off, parent, taxon, length, support, part_of, new_id, n_parts, max_k, plan};
    /* scratch: header + 10 max_k + 3 n_parts ints, zeroed by the worker (header: no arena
     * yet; last[]: no leaf seen) */
    const size_t ints = 4 + (size_t)10 * max_k + (size_t)3 * n_parts;
    int rc = for_each_tree(n_trees, n_trees > 0 ? node_off[n_trees] - node_off[0] : 0,
                           sizeof(int32_t) * ints, split_tree, &c);
    if (rc != SCS_HOST_OK) {
        scs_host_split_end(plan);
        return rc;
    }
    int64_t n = 0;
    for (int32_t i = 0; i < plan->n_arenas; ++i) n += plan->arenas[i].n_entries;
    plan->sorted = (split_entry *)malloc(sizeof(split_entry) * (size_t)(n ? n : 1));
    if (!plan->sorted) {
        scs_host_split_end(plan);
        return SCS_HOST_ENOMEM;
    }
    /* entries by (part, tree): two stable counting passes (least significant key first: the
     * tree, then the part) -- linear, whatever order the threads produced them in */
    for (int32_t p = 0; p < n_parts; ++p) {
        part_trees[p] = 0;
        part_nodes[p] = 0;
    }
    plan->n_sorted = n;
    {
        split_entry *tmp = (split_entry *)malloc(sizeof(split_entry) * (size_t)(n ? n : 1));
        int64_t *cnt_tree = (int64_t *)calloc((size_t)n_trees + 1, sizeof(int64_t));
        if (!tmp || !cnt_tree) {
            free(tmp);
            free(cnt_tree);
            scs_host_split_end(plan);
            return SCS_HOST_ENOMEM;
        }
        for (int32_t i = 0; i < plan->n_arenas; ++i)
            for (int64_t e = 0; e < plan->arenas[i].n_entries; ++e) {
                const split_entry *en = &plan->arenas[i].entries[e];
                cnt_tree[en->tree + 1] += 1;
                part_trees[en->part] += 1;
                part_nodes[en->part] += en->n_nodes;
            }
        for (int32_t t = 0; t < n_trees; ++t) cnt_tree[t + 1] += cnt_tree[t];
        for (int32_t i = 0; i < plan->n_arenas; ++i)
            for (int64_t e = 0; e < plan->arenas[i].n_entries; ++e) {
                const split_entry *en = &plan->arenas[i].entries[e];
                tmp[cnt_tree[en->tree]++] = *en;
            }
        plan->part_first[0] = 0;
        for (int32_t p = 0; p < n_parts; ++p) plan->part_first[p + 1] = plan->part_first[p] + part_trees[p];
        int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_parts);
        if (!cursor) {
            free(tmp);
            free(cnt_tree);
            scs_host_split_end(plan);
            return SCS_HOST_ENOMEM;
        }
        for (int32_t p = 0; p < n_parts; ++p) cursor[p] = plan->part_first[p];
        for (int64_t e = 0; e < n; ++e) plan->sorted[cursor[tmp[e].part]++] = tmp[e];
        free(cursor);
        free(cnt_tree);
        free(tmp);
    }
    *out_plan = plan;
    return SCS_HOST_OK;
}

/* Part `part` in tree order: node_off_out [trees + 1], tree_index_out [trees] (index of the
 * tree in the parent forest), leaf_count_out [trees], the node arrays, and present_out[id] = 1
 * for every (new) taxon id that occurs (zeroed by the caller).  Call from the thread that
 * called scs_host_split_begin, before its next split (the node region belongs to it). */
typedef struct {
    const scs_split_plan *plan;
    int64_t first;
    const int64_t *node_off_out;
    int32_t *parent_out, *taxon_out;
    double *length_out, *support_out;
    uint8_t *present_out;
} fill2_ctx;

static int split_fill_entry(int32_t j, void *scratch, void *vctx) {
    (void)scratch;
    const fill2_ctx *c = (const fill2_ctx *)vctx;
    const split_entry *e = &c->plan->sorted[c->first + j];
    const split_region *a = &c->plan->region;
    const int64_t at = c->node_off_out[j];
    memcpy(c->parent_out + at, a->parent + e->off, sizeof(int32_t) * (size_t)e->n_nodes);
    memcpy(c->taxon_out + at, a->taxon + e->off, sizeof(int32_t) * (size_t)e->n_nodes);
    memcpy(c->length_out + at, a->length + e->off, sizeof(double) * (size_t)e->n_nodes);
    memcpy(c->support_out + at, a->support + e->off, sizeof(double) * (size_t)e->n_nodes);
    if (c->present_out)
        for (int32_t q = 0; q < e->n_nodes; ++q) {
            const int32_t x = a->taxon[e->off + q];
            if (x >= 0 && !c->present_out[x]) c->present_out[x] = 1;
        }
    return SCS_HOST_OK;
}

int scs_host_split_fill(const scs_split_plan *plan, int32_t part, int64_t *node_off_out,
                        int32_t *tree_index_out, int64_t *leaf_count_out, int32_t *parent_out,
                        int32_t *taxon_out, double *length_out, double *support_out,
                        uint8_t *present_out) {
    if (!plan || part < 0 || part >= plan->n_parts) return SCS_HOST_EINVAL;
    int64_t at = 0, j = 0;
    node_off_out[0] = 0;
    for (int64_t i = plan->part_first[part]; i < plan->part_first[part + 1]; ++i, ++j) {
        const split_entry *e = &plan->sorted[i];
        tree_index_out[j] = e->tree;
        leaf_count_out[j] = e->n_leaves;
        at += e->n_nodes;
        node_off_out[j + 1] = at;
    }
    if (j > INT32_MAX) return SCS_HOST_EINVAL;
    fill2_ctx c = {plan, plan->part_first[part], node_off_out, parent_out, taxon_out, length_out,
                   support_out, present_out};
    return for_each_tree((int32_t)j, at, 0, split_fill_entry, &c);
}


/* ---------------------------------------------------------------------------
 * The recursion allocates and drops numpy arrays of a few hundred kilobytes thousands of times a
 * second (every split hands each child fresh node arrays).  glibc serves such sizes by mmap and gives
 * them back by munmap: every array is born as untouched pages and costs a page fault per 4 KiB --
 * 14 % of a whole recursion of 5 000-tree forests on the measured host.  While a recursion runs the
 * threshold is raised so that these arrays come from the heap and are reused warm (on: 1), and put
 * back to glibc's static defaults afterwards (on: 0; the dynamic adjustment of the threshold cannot be
 * re-armed).  Process-wide, which is why scs.py does it only for the duration of a recursion and
 * SCS_MALLOC_TUNE=0 leaves the allocator alone.
 * ------------------------------------------------------------------------- */
int scs_host_malloc_tune(int on) {
    int ok = 1;
    if (on) {
        ok &= mallopt(M_MMAP_THRESHOLD, 32 << 20);
        ok &= mallopt(M_TRIM_THRESHOLD, 1 << 30);
        ok &= mallopt(M_TOP_PAD, 64 << 20);
    } else {
        ok &= mallopt(M_MMAP_THRESHOLD, 128 * 1024);
        ok &= mallopt(M_TRIM_THRESHOLD, 128 * 1024);
        ok &= mallopt(M_TOP_PAD, 128 * 1024);
        malloc_trim(0);
    }
    return ok ? SCS_HOST_OK : SCS_HOST_EINVAL;
}
