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
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SCS_HOST_OK 0
#define SCS_HOST_ENOMEM (-1)
#define SCS_HOST_EINVAL (-2)
#define SCS_HOST_ENOSUPPORT (-3) /* bootstrap strategy met an internal node without support */

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

int scs_host_restrict_sizes(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                            const int32_t *taxon, const uint8_t *keep, uint8_t *out_tree_keep,
                            int32_t *out_nodes) {
    int32_t max_k = 0;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k < 1 || k > INT32_MAX) return SCS_HOST_EINVAL;
        if (k > max_k) max_k = (int32_t)k;
    }
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_k > 0 ? max_k : 1));
    int32_t *nkc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(max_k > 0 ? max_k : 1));
    if (!cnt || !nkc) {
        free(cnt);
        free(nkc);
        return SCS_HOST_ENOMEM;
    }
    int rc = SCS_HOST_OK;
    for (int32_t t = 0; t < n_trees && rc == SCS_HOST_OK; ++t) {
        const int64_t off = node_off[t];
        const int32_t k = (int32_t)(node_off[t + 1] - off);
        rc = restrict_tree_count(parent + off, taxon + off, k, keep, cnt, nkc);
        if (rc != SCS_HOST_OK) break;
        if (cnt[0] < 2) {
            out_tree_keep[t] = 0;
            out_nodes[t] = 0;
            continue;
        }
        int32_t kept = 0;
        for (int32_t i = 0; i < k; ++i)
            if ((taxon[off + i] >= 0 && cnt[i] == 1) || (taxon[off + i] < 0 && nkc[i] >= 2)) ++kept;
        out_tree_keep[t] = 1;
        out_nodes[t] = kept;
    }
    free(cnt);
    free(nkc);
    return rc;
}

/*
 * Pass 2: fill the restricted forest.  new_node_off (for the surviving trees, in order)
 * is the exclusive scan of out_nodes over surviving trees, computed by the caller.
 */
int scs_host_restrict_fill(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                           const int32_t *taxon, const double *length, const double *support,
                           const uint8_t *keep, const uint8_t *tree_keep,
                           const int64_t *new_node_off, int32_t *new_parent, int32_t *new_taxon,
                           double *new_length, double *new_support) {
    int32_t max_k = 0;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k > max_k) max_k = (int32_t)k;
    }
    const size_t cap = (size_t)(max_k > 0 ? max_k : 1);
    int32_t *cnt = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *nkc = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *newidx = (int32_t *)malloc(sizeof(int32_t) * cap); /* kept node -> new index */
    int32_t *anc = (int32_t *)malloc(sizeof(int32_t) * cap);    /* nearest kept ancestor-or-self (old index), -1 above the new root */
    if (!cnt || !nkc || !newidx || !anc) {
        free(cnt);
        free(nkc);
        free(newidx);
        free(anc);
        return SCS_HOST_ENOMEM;
    }
    int rc = SCS_HOST_OK;
    int32_t out_t = 0;
    for (int32_t t = 0; t < n_trees && rc == SCS_HOST_OK; ++t) {
        if (!tree_keep[t]) continue;
        const int64_t off = node_off[t];
        const int32_t k = (int32_t)(node_off[t + 1] - off);
        const int32_t *par = parent + off, *tax = taxon + off;
        const double *len = length + off, *sup = support + off;
        rc = restrict_tree_count(par, tax, k, keep, cnt, nkc);
        if (rc != SCS_HOST_OK) break;
        const int64_t noff = new_node_off[out_t];
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
            new_parent[noff + next] = up < 0 ? -1 : newidx[up];
            new_taxon[noff + next] = tax[i];
            new_support[noff + next] = sup[i];
            /* merged length: fold the spliced chain bottom-up, parent's length in front */
            double acc = len[i];
            for (int32_t u = i == 0 ? -1 : par[i]; u >= 0 && u != up; u = par[u]) {
                /* u lies strictly between the node and its kept ancestor: it had one child left */
                if (!isnan(len[u]) && !isnan(acc)) acc = len[u] + acc;
            }
            /* (a node that becomes the root absorbs the chain up to the old root the same way;
             * a root's length is never used) */
            new_length[noff + next] = acc;
            ++next;
        }
        ++out_t;
    }
    free(cnt);
    free(nkc);
    free(newidx);
    free(anc);
    return rc;
}

/*
 * Flatten a forest into the device tables.
 *   strategy: 0 one, 1 depth, 2 branch, 3 bootstrap
 *   leaf_off[t]: first leaf slot of tree t (exclusive scan of leaf counts, n_trees + 1)
 *   monotone_out: set to 0 if a negative internal length is met under `branch`
 * Outputs sized leaf_off[n_trees].
 */
int scs_host_flatten(int32_t n_trees, const int64_t *node_off, const int32_t *parent,
                     const int32_t *taxon, const double *length, const double *support,
                     int32_t strategy, const int64_t *leaf_off, int32_t *leaf_taxon,
                     int32_t *adj_depth, double *adj_val, int32_t *monotone_out) {
    if (strategy < 0 || strategy > 3) return SCS_HOST_EINVAL;
    int32_t max_k = 0;
    for (int32_t t = 0; t < n_trees; ++t) {
        const int64_t k = node_off[t + 1] - node_off[t];
        if (k < 1 || k > INT32_MAX) return SCS_HOST_EINVAL;
        if (k > max_k) max_k = (int32_t)k;
    }
    const size_t cap = (size_t)(max_k > 0 ? max_k : 1);
    int32_t *depth = (int32_t *)malloc(sizeof(int32_t) * cap);
    double *val = (double *)malloc(sizeof(double) * cap);
    int32_t *nch = (int32_t *)malloc(sizeof(int32_t) * cap);
    if (!depth || !val || !nch) {
        free(depth);
        free(val);
        free(nch);
        return SCS_HOST_ENOMEM;
    }
    int rc = SCS_HOST_OK;
    for (int32_t t = 0; t < n_trees && rc == SCS_HOST_OK; ++t) {
        const int64_t off = node_off[t];
        const int32_t k = (int32_t)(node_off[t + 1] - off);
        const int32_t *par = parent + off, *tax = taxon + off;
        const double *len = length + off, *sup = support + off;
        int64_t slot = leaf_off[t];
        const int64_t slot_end = leaf_off[t + 1];
        for (int32_t i = 0; i < k; ++i) nch[i] = 0;
        for (int32_t i = 1; i < k; ++i) nch[par[i]] += 1;
        int first_leaf = 1;
        int32_t pend_depth = 0;
        double pend_val = 0.0;
        depth[0] = 0;
        val[0] = 0.0;
        if (tax[0] >= 0) { /* a single-leaf tree */
            if (slot >= slot_end) {
                rc = SCS_HOST_EINVAL;
                break;
            }
            leaf_taxon[slot] = tax[0];
            adj_depth[slot] = 0;
            adj_val[slot] = 0.0;
            continue;
        }
        for (int32_t i = 1; i < k && rc == SCS_HOST_OK; ++i) {
            const int32_t u = par[i];
            if (i != u + 1) { /* not the first child: the next leaf's LCA with the previous one is u */
                pend_depth = depth[u];
                pend_val = val[u];
            }
            if (tax[i] >= 0) {
                if (slot >= slot_end) {
                    rc = SCS_HOST_EINVAL;
                    break;
                }
                if (!first_leaf) {
                    adj_depth[slot - 1] = pend_depth;
                    adj_val[slot - 1] = pend_val;
                }
                first_leaf = 0;
                leaf_taxon[slot++] = tax[i];
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
                    if (!isnan(len[i]) && len[i] < 0.0) *monotone_out = 0;
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
        if (rc != SCS_HOST_OK) break;
        if (slot != slot_end) {
            rc = SCS_HOST_EINVAL;
            break;
        }
        /* padding slot so adj_* share the offsets of leaf_taxon */
        adj_depth[slot_end - 1] = 0;
        adj_val[slot_end - 1] = 0.0;
    }
    free(depth);
    free(val);
    free(nch);
    return rc;
}
