/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/scs_oracle.py for the rules).
 *
 * Plain-C restatement of the reference's proper-cluster-graph accumulation
 * (reference: src/sc_supertree/scs.py:569-581 driver loop, :586-663
 * _dfs_pcg_weights) over the flattened tables of include/scs_hip.h, producing
 * the dense fp64 matrix the reference fills at scs.py:246-250.
 *
 * For every tree, in tree order, every leaf pair (a, b) whose lowest common
 * ancestor is not the root receives `value(LCA) * tree_weight` (one rounded
 * multiply, scs.py:656) added to its running sum (`edge_weights.get(edge, 0) +
 * ...`, a rounded add).  The LCA of leaves a < b (DFS positions) is the entry
 * of minimum depth in adj_depth[a..b-1]; a pair is proper iff that depth > 0.
 * Each pair of a tree is visited exactly once, as in the reference's
 * cross-children loops (scs.py:644-658), so per-cell addends and their order
 * (tree order) are the reference's.  Compiled with -ffp-contract=off.
 *
 * Validated against the dict-based restatement oracle/scs_oracle.py on the
 * reference's own fixtures (tests/test_oracle_tables.py).
 */
#include <stdint.h>
#include <string.h>

/* w: n_taxa x n_taxa row-major, zero-initialised by the caller (diagonal stays 0).
 * Trees [t_begin, t_end) are accumulated, so a caller can time a bounded sample.
 * Returns the number of (pair, tree) updates performed. */
int64_t scs_oracle_pcg_dense(int32_t n_taxa, int32_t t_begin, int32_t t_end,
                             const int64_t *tree_off, const int32_t *leaf_taxon,
                             const int32_t *adj_depth, const double *adj_val,
                             const double *tree_w, double *w) {
    int64_t updates = 0;
    for (int32_t t = t_begin; t < t_end; ++t) {
        const int64_t off = tree_off[t];
        const int32_t n = (int32_t)(tree_off[t + 1] - off);
        const int32_t *tax = leaf_taxon + off;
        const int32_t *dep = adj_depth + off;
        const double *val = adj_val + off;
        const double wt = tree_w[t];
        for (int32_t a = 0; a + 1 < n; ++a) {
            int32_t md = dep[a];
            double mv = val[a];
            const int64_t ra = (int64_t)tax[a] * n_taxa;
            for (int32_t b = a + 1; b < n; ++b) {
                if (b > a + 1 && dep[b - 1] < md) {
                    md = dep[b - 1];
                    mv = val[b - 1];
                }
                if (md == 0) break; /* the root separates a from every later leaf */
                const double add = mv * wt;
                const int64_t rb = (int64_t)tax[b] * n_taxa;
                const double s = w[ra + tax[b]] + add;
                w[ra + tax[b]] = s;
                w[rb + tax[a]] = s;
                ++updates;
            }
        }
    }
    return updates;
}

/* Contraction of consecutive index ranges: out[g][h] = max over member pairs,
 * diagonal 0 (reference: scs.py:336-387; see include/scs_hip.h scs_graph_contract). */
void scs_oracle_contract(int32_t n, const double *w, int32_t n_groups,
                         const int32_t *group_start, double *out) {
    for (int32_t g = 0; g < n_groups; ++g)
        for (int32_t h = 0; h < n_groups; ++h) {
            double best = 0.0;
            if (g != h) {
                best = w[(int64_t)group_start[g] * n + group_start[h]];
                for (int32_t r = group_start[g]; r < group_start[g + 1]; ++r)
                    for (int32_t c = group_start[h]; c < group_start[h + 1]; ++c)
                        if (w[(int64_t)r * n + c] > best) best = w[(int64_t)r * n + c];
            }
            out[(int64_t)g * n_groups + h] = best;
        }
}

/* Selected rows of W only (full-size spot checks): out is n_rows x n_taxa,
 * zero-initialised by the caller; same addends, same tree order as above. */
void scs_oracle_pcg_rows(int32_t n_taxa, int32_t n_trees, const int64_t *tree_off,
                         const int32_t *leaf_taxon, const int32_t *adj_depth,
                         const double *adj_val, const double *tree_w, int32_t n_rows,
                         const int32_t *rows, const int32_t *pos_scratch /* unused */,
                         double *out) {
    (void)pos_scratch;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t off = tree_off[t];
        const int32_t n = (int32_t)(tree_off[t + 1] - off);
        const int32_t *tax = leaf_taxon + off;
        const int32_t *dep = adj_depth + off;
        const double *val = adj_val + off;
        const double wt = tree_w[t];
        for (int32_t r = 0; r < n_rows; ++r) {
            int32_t a = -1;
            for (int32_t p = 0; p < n; ++p)
                if (tax[p] == rows[r]) {
                    a = p;
                    break;
                }
            if (a < 0) continue;
            double *o = out + (int64_t)r * n_taxa;
            /* leaves to the right of a */
            if (a + 1 < n) {
                int32_t md = dep[a];
                double mv = val[a];
                for (int32_t b = a + 1; b < n; ++b) {
                    if (b > a + 1 && dep[b - 1] < md) {
                        md = dep[b - 1];
                        mv = val[b - 1];
                    }
                    if (md == 0) break;
                    o[tax[b]] = o[tax[b]] + mv * wt;
                }
            }
            /* leaves to the left of a */
            if (a > 0) {
                int32_t md = dep[a - 1];
                double mv = val[a - 1];
                for (int32_t b = a - 1; b >= 0; --b) {
                    if (b < a - 1 && dep[b] < md) {
                        md = dep[b];
                        mv = val[b];
                    }
                    if (md == 0) break;
                    o[tax[b]] = o[tax[b]] + mv * wt;
                }
            }
        }
    }
}

/* ---- all-core variant (bench.py's cpu_baseline, SURVEY.md 8d "all host cores") -------------
 * Thread k owns the rows of a contiguous range of taxa and sweeps, for every tree in tree
 * order, left and right from each of its leaves (as scs_oracle_pcg_rows does): every cell
 * still receives the reference's addends in the reference's order, from exactly one thread. */
#include <pthread.h>
#include <stdlib.h>

typedef struct {
    int32_t n_taxa, t_begin, t_end, r0, r1;
    const int64_t *tree_off;
    const int32_t *leaf_taxon, *adj_depth;
    const double *adj_val, *tree_w;
    double *w;
} mt_job;

static void *mt_worker(void *arg) {
    const mt_job *j = (const mt_job *)arg;
    for (int32_t t = j->t_begin; t < j->t_end; ++t) {
        const int64_t off = j->tree_off[t];
        const int32_t n = (int32_t)(j->tree_off[t + 1] - off);
        const int32_t *tax = j->leaf_taxon + off;
        const int32_t *dep = j->adj_depth + off;
        const double *val = j->adj_val + off;
        const double wt = j->tree_w[t];
        for (int32_t a = 0; a < n; ++a) {
            if (tax[a] < j->r0 || tax[a] >= j->r1) continue;
            double *o = j->w + (int64_t)tax[a] * j->n_taxa;
            if (a + 1 < n) {
                int32_t md = dep[a];
                double mv = val[a];
                for (int32_t b = a + 1; b < n; ++b) {
                    if (b > a + 1 && dep[b - 1] < md) {
                        md = dep[b - 1];
                        mv = val[b - 1];
                    }
                    if (md == 0) break;
                    o[tax[b]] = o[tax[b]] + mv * wt;
                }
            }
            if (a > 0) {
                int32_t md = dep[a - 1];
                double mv = val[a - 1];
                for (int32_t b = a - 1; b >= 0; --b) {
                    if (b < a - 1 && dep[b] < md) {
                        md = dep[b];
                        mv = val[b];
                    }
                    if (md == 0) break;
                    o[tax[b]] = o[tax[b]] + mv * wt;
                }
            }
        }
    }
    return 0;
}

/* w zero-initialised by the caller; returns 0, or -1 when threads cannot be started */
int scs_oracle_pcg_dense_mt(int32_t n_taxa, int32_t t_begin, int32_t t_end,
                            const int64_t *tree_off, const int32_t *leaf_taxon,
                            const int32_t *adj_depth, const double *adj_val,
                            const double *tree_w, double *w, int32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_taxa) n_threads = n_taxa;
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)n_threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!jobs || !th) {
        free(jobs);
        free(th);
        return -1;
    }
    int started = 0, rc = 0;
    for (int32_t k = 0; k < n_threads; ++k) {
        mt_job j = {n_taxa, t_begin, t_end, (int32_t)((int64_t)n_taxa * k / n_threads),
                    (int32_t)((int64_t)n_taxa * (k + 1) / n_threads), tree_off, leaf_taxon,
                    adj_depth, adj_val, tree_w, w};
        jobs[k] = j;
        if (pthread_create(&th[k], 0, mt_worker, &jobs[k]) != 0) {
            rc = -1;
            break;
        }
        ++started;
    }
    for (int k = 0; k < started; ++k) pthread_join(th[k], 0);
    if (rc != 0) /* finish the rows nobody took on this thread */
        for (int32_t k = started; k < n_threads; ++k) {
            mt_job j = {n_taxa, t_begin, t_end, (int32_t)((int64_t)n_taxa * k / n_threads),
                        (int32_t)((int64_t)n_taxa * (k + 1) / n_threads), tree_off, leaf_taxon,
                        adj_depth, adj_val, tree_w, w};
            mt_worker(&j);
        }
    free(jobs);
    free(th);
    return 0;
}
