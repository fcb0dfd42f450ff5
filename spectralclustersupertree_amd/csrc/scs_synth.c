/*
 * Deterministic synthetic source trees for benchmarks and parity tests
 * (host only, no GPU code).  SURVEY.md section 8d: iid random-join (Yule-like)
 * rooted binary trees, Exp(mean 0.1) branch lengths, supports in 50..100.
 *
 * Tree t of a set is a pure function of (seed, t, n_taxa, n_leaves): the same
 * call always yields the same topology, lengths and supports, so the device
 * path and the CPU oracle can be fed identical inputs at any size.  The tables
 * are emitted directly in the flattened layout of include/scs_hip.h
 * (leaf_taxon / adj_depth / adj_val), the way
 * spectralclustersupertree_amd/flatten.py produces them from tree objects.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t s;
} rng_t;

static uint64_t rng_next(rng_t *r) { /* splitmix64 */
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static uint64_t rng_below(rng_t *r, uint64_t k) { /* floor(u * k / 2^64) */
    return (uint64_t)(((unsigned __int128)rng_next(r) * (unsigned __int128)k) >> 64);
}

static double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static void rng_seed(rng_t *r, uint64_t seed, uint64_t tree) {
    r->s = seed * 0x9E3779B97F4A7C15ull + tree * 0xD1B54A32D192ED03ull + 0x2545F4914F6CDD1Dull;
    (void)rng_next(r);
}

/*
 * One tree.  Leaves are node ids 0..k-1, internal nodes k..2k-2 (root = 2k-2).
 *   strategy: 0 one, 1 depth, 2 branch, 3 bootstrap
 *   leaf_taxon/adj_depth/adj_val: k slots each (last adj slot is padding = 0)
 * Optional structure outputs (may be NULL), sized 2k-1: left, right (children,
 * -1 for leaves), length, support; taxon_of_leaf sized k.
 * Returns 0, or -1 on allocation failure / bad arguments.
 */
static int synth_tree(uint64_t seed, int64_t tree, int32_t n_taxa, int32_t k, int32_t strategy,
                      int32_t n_spr, int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val,
                      int32_t *left, int32_t *right, double *length, double *support,
                      int32_t *taxon_of_leaf);

int scs_synth_tree(uint64_t seed, int64_t tree, int32_t n_taxa, int32_t k, int32_t strategy,
                   int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val, int32_t *left,
                   int32_t *right, double *length, double *support, int32_t *taxon_of_leaf) {
    return synth_tree(seed, tree, n_taxa, k, strategy, -1, leaf_taxon, adj_depth, adj_val, left,
                      right, length, support, taxon_of_leaf);
}

/* random joins over leaves 0..k-1: internal nodes k..2k-2, root = 2k-2 */
static void random_joins(rng_t *rng, int32_t k, int32_t *lf, int32_t *rt, int32_t *roots) {
    for (int32_t i = 0; i < k; ++i) roots[i] = i;
    int32_t cnt = k;
    for (int32_t step = 0; step < k - 1; ++step) {
        int32_t i = (int32_t)rng_below(rng, (uint64_t)cnt);
        int32_t j = (int32_t)rng_below(rng, (uint64_t)(cnt - 1));
        if (j >= i) ++j;
        const int32_t v = k + step;
        lf[v] = roots[i];
        rt[v] = roots[j];
        roots[i] = v;
        roots[j] = roots[cnt - 1];
        --cnt;
    }
}

/* one subtree-prune-and-regraft move that keeps the root node and its two sides' node ids:
 * prune x (parent p, p != root), close the gap, re-insert p on the edge above y */
static void spr_move(rng_t *rng, int32_t nn, int32_t *lf, int32_t *rt, int32_t *par) {
    const int32_t root = nn - 1;
    for (int attempt = 0; attempt < 64; ++attempt) {
        const int32_t x = (int32_t)rng_below(rng, (uint64_t)(nn - 1));
        const int32_t p = par[x];
        if (p == root) continue;
        const int32_t y = (int32_t)rng_below(rng, (uint64_t)(nn - 1));
        if (y == x || y == p) continue;
        int inside = 0; /* y in the subtree of x (or of p: p moves along)? */
        for (int32_t a = y; a != root; a = par[a])
            if (a == x || a == p) {
                inside = 1;
                break;
            }
        const int32_t sib = lf[p] == x ? rt[p] : lf[p];
        if (inside && y != sib) {
            int under_x = 0;
            for (int32_t a = y; a != root; a = par[a])
                if (a == x) {
                    under_x = 1;
                    break;
                }
            if (under_x) continue;
        }
        if (y == sib) continue; /* re-inserting above the sibling changes nothing */
        /* close the gap: the sibling takes p's place under g */
        const int32_t g = par[p];
        if (lf[g] == p) lf[g] = sib; else rt[g] = sib;
        par[sib] = g;
        /* insert p above y */
        const int32_t q = par[y];
        if (lf[q] == y) lf[q] = p; else rt[q] = p;
        par[p] = q;
        lf[p] = x;
        rt[p] = y;
        par[y] = p;
        return;
    }
}

/* n_spr < 0: an independent random-join tree (the iid sets).  n_spr >= 0: the PLANTED sets of
 * SURVEY.md 8d -- the model tree of the set (random joins drawn from (seed, tree = -1)) with
 * n_spr random SPR moves, lengths and supports drawn per tree. */
static int synth_tree(uint64_t seed, int64_t tree, int32_t n_taxa, int32_t k, int32_t strategy,
                      int32_t n_spr, int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val,
                      int32_t *left, int32_t *right, double *length, double *support,
                      int32_t *taxon_of_leaf) {
    if (k < 1 || k > n_taxa || strategy < 0 || strategy > 3) return -1;
    rng_t rng;
    rng_seed(&rng, seed, (uint64_t)tree);
    const int32_t nn = 2 * k - 1;
    int32_t *lf = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    int32_t *rt = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    double *len = (double *)malloc(sizeof(double) * (size_t)nn);
    double *sup = (double *)malloc(sizeof(double) * (size_t)nn);
    int32_t *tax = (int32_t *)malloc(sizeof(int32_t) * (size_t)(k > n_taxa ? k : n_taxa));
    int32_t *roots = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    int32_t *sdepth = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
    double *sval = (double *)malloc(sizeof(double) * (size_t)nn);
    int8_t *state = (int8_t *)malloc((size_t)nn);
    if (!lf || !rt || !len || !sup || !tax || !roots || !stack || !sdepth || !sval || !state) {
        free(lf); free(rt); free(len); free(sup); free(tax); free(roots); free(stack);
        free(sdepth); free(sval); free(state);
        return -1;
    }
    /* taxa of this tree: first k entries of a partial Fisher-Yates shuffle */
    for (int32_t i = 0; i < n_taxa; ++i) tax[i] = i;
    if (k < n_taxa) {
        for (int32_t i = 0; i < k; ++i) {
            int32_t j = i + (int32_t)rng_below(&rng, (uint64_t)(n_taxa - i));
            int32_t tmp = tax[i];
            tax[i] = tax[j];
            tax[j] = tmp;
        }
    }
    /* lengths and supports for every node, in node-id order */
    for (int32_t v = 0; v < nn; ++v) {
        len[v] = -0.1 * log(1.0 - rng_unit(&rng));
        sup[v] = 50.0 + (double)rng_below(&rng, 51);
        lf[v] = rt[v] = -1;
    }
    if (n_spr < 0) {
        random_joins(&rng, k, lf, rt, roots);
    } else {
        rng_t model;
        rng_seed(&model, seed, 0xFFFFFFFFFFFFFFFFull);
        random_joins(&model, k, lf, rt, roots);
        int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (size_t)nn);
        if (!par) {
            free(lf); free(rt); free(len); free(sup); free(tax); free(roots); free(stack);
            free(sdepth); free(sval); free(state);
            return -1;
        }
        for (int32_t v = 0; v < nn; ++v) par[v] = -1;
        for (int32_t v = k; v < nn; ++v) {
            par[lf[v]] = v;
            par[rt[v]] = v;
        }
        if (nn >= 7)
            for (int32_t i = 0; i < n_spr; ++i) spr_move(&rng, nn, lf, rt, par);
        free(par);
    }
    const int32_t root = nn - 1;
    /* depth-first flatten; value carried below a node per weighting strategy
     * (reference: src/sc_supertree/scs.py:555-564, started at 0 below the root, :577) */
    int32_t sp = 0, nleaf = 0;
    int32_t pend_depth = 0;
    double pend_val = 0.0;
    stack[0] = root;
    sdepth[0] = 0;
    sval[0] = 0.0;
    state[0] = 0;
    if (k == 1) {
        leaf_taxon[0] = tax[0];
        adj_depth[0] = 0;
        adj_val[0] = 0.0;
    } else {
        while (sp >= 0) {
            const int32_t v = stack[sp];
            if (lf[v] < 0) { /* leaf */
                if (nleaf > 0) {
                    adj_depth[nleaf - 1] = pend_depth;
                    adj_val[nleaf - 1] = pend_val;
                }
                leaf_taxon[nleaf++] = tax[v];
                --sp;
                continue;
            }
            if (state[sp] == 2) {
                --sp;
                continue;
            }
            const int32_t child = state[sp] == 0 ? lf[v] : rt[v];
            if (state[sp] == 1) { /* entering the second child: v is the next LCA */
                pend_depth = sdepth[sp];
                pend_val = sval[sp];
            }
            ++state[sp];
            const int32_t d = sdepth[sp];
            const double val = sval[sp];
            ++sp;
            stack[sp] = child;
            state[sp] = 0;
            sdepth[sp] = d + 1;
            if (lf[child] < 0) {
                sval[sp] = 0.0;
            } else if (strategy == 0) {
                sval[sp] = 1.0;
            } else if (strategy == 1) {
                sval[sp] = (d == 0 ? 0.0 : val) + 1.0;
            } else if (strategy == 2) {
                sval[sp] = (d == 0 ? 0.0 : val) + len[child];
            } else {
                sval[sp] = sup[child];
            }
        }
        adj_depth[k - 1] = 0;
        adj_val[k - 1] = 0.0;
    }
    if (left) memcpy(left, lf, sizeof(int32_t) * (size_t)nn);
    if (right) memcpy(right, rt, sizeof(int32_t) * (size_t)nn);
    if (length) memcpy(length, len, sizeof(double) * (size_t)nn);
    if (support) memcpy(support, sup, sizeof(double) * (size_t)nn);
    if (taxon_of_leaf) memcpy(taxon_of_leaf, tax, sizeof(int32_t) * (size_t)k);
    free(lf); free(rt); free(len); free(sup); free(tax); free(roots); free(stack);
    free(sdepth); free(sval); free(state);
    return 0;
}

/* A whole set: trees [0, n_trees), tree t written at slot offset t*k.
 * weight_mode 0: all weights 1.0; 1: Uniform(0.5, 2.0) drawn per tree. */
static int synth_tables(uint64_t seed, int32_t n_taxa, int32_t n_trees, int32_t k,
                        int32_t strategy, int32_t weight_mode, int32_t n_spr, int64_t *tree_off,
                        int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val, double *tree_w);

int scs_synth_tables(uint64_t seed, int32_t n_taxa, int32_t n_trees, int32_t k, int32_t strategy,
                     int32_t weight_mode, int64_t *tree_off, int32_t *leaf_taxon,
                     int32_t *adj_depth, double *adj_val, double *tree_w) {
    return synth_tables(seed, n_taxa, n_trees, k, strategy, weight_mode, -1, tree_off, leaf_taxon,
                        adj_depth, adj_val, tree_w);
}

/* The planted sets: every tree = the set's model tree over all n_taxa taxa
 * + n_spr random SPR moves. */
int scs_synth_tables_planted(uint64_t seed, int32_t n_taxa, int32_t n_trees, int32_t strategy,
                             int32_t weight_mode, int32_t n_spr, int64_t *tree_off,
                             int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val,
                             double *tree_w) {
    if (n_spr < 0) return -1;
    return synth_tables(seed, n_taxa, n_trees, n_taxa, strategy, weight_mode, n_spr, tree_off,
                        leaf_taxon, adj_depth, adj_val, tree_w);
}

static int synth_tables(uint64_t seed, int32_t n_taxa, int32_t n_trees, int32_t k,
                        int32_t strategy, int32_t weight_mode, int32_t n_spr, int64_t *tree_off,
                        int32_t *leaf_taxon, int32_t *adj_depth, double *adj_val, double *tree_w) {
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t off = (int64_t)t * k;
        tree_off[t] = off;
        if (synth_tree(seed, t, n_taxa, k, strategy, n_spr, leaf_taxon + off, adj_depth + off,
                       adj_val + off, 0, 0, 0, 0, 0) != 0)
            return -1;
        if (weight_mode == 1) {
            rng_t rng;
            rng_seed(&rng, seed ^ 0xA5A5A5A5A5A5A5A5ull, (uint64_t)t);
            tree_w[t] = 0.5 + 1.5 * rng_unit(&rng);
        } else {
            tree_w[t] = 1.0;
        }
    }
    tree_off[n_trees] = (int64_t)n_trees * k;
    return 0;
}

/* Tree t of a set as preorder node arrays, the layout of treearrays.TreeArrays (what
 * TreeArrays.from_trees makes of synthetic.tree_objects' tree t): 2k-1 nodes; parent = index
 * within the tree (-1 at the root); taxon = the leaf's taxon id, -1 for internal nodes; length
 * NaN at the root; support NaN at the leaves.  Children keep their order (left first). */
int scs_synth_tree_nodes(uint64_t seed, int64_t tree, int32_t n_taxa, int32_t k, int32_t *parent,
                         int32_t *taxon, double *length, double *support) {
    if (k < 1) return -1;
    const int32_t nn = 2 * k - 1;
    int32_t *lt = malloc(sizeof(int32_t) * (size_t)k), *ad = malloc(sizeof(int32_t) * (size_t)k);
    double *av = malloc(sizeof(double) * (size_t)k);
    int32_t *lf = malloc(sizeof(int32_t) * (size_t)nn), *rt = malloc(sizeof(int32_t) * (size_t)nn);
    double *len = malloc(sizeof(double) * (size_t)nn), *sup = malloc(sizeof(double) * (size_t)nn);
    int32_t *tax = malloc(sizeof(int32_t) * (size_t)k);
    int32_t *stack = malloc(sizeof(int32_t) * 2 * (size_t)nn);
    int rc = -1;
    if (lt && ad && av && lf && rt && len && sup && tax && stack &&
        synth_tree(seed, tree, n_taxa, k, 0, -1, lt, ad, av, lf, rt, len, sup, tax) == 0) {
        int32_t top = 0, out = 0;
        stack[0] = nn - 1; /* root */
        stack[1] = -1;
        top = 1;
        while (top > 0) {
            --top;
            const int32_t v = stack[2 * top], par = stack[2 * top + 1];
            const int32_t me = out++;
            parent[me] = par;
            if (v < k) {
                taxon[me] = tax[v];
                length[me] = par < 0 ? NAN : len[v];
                support[me] = NAN;
            } else {
                taxon[me] = -1;
                length[me] = par < 0 ? NAN : len[v];
                support[me] = sup[v];
                stack[2 * top] = rt[v]; /* popped second */
                stack[2 * top + 1] = me;
                ++top;
                stack[2 * top] = lf[v];
                stack[2 * top + 1] = me;
                ++top;
            }
        }
        rc = out == nn ? 0 : -1;
    }
    free(lt); free(ad); free(av); free(lf); free(rt); free(len); free(sup); free(tax); free(stack);
    return rc;
}
