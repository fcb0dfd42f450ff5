"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.

CPU restatement of the reference's spectral-cluster-supertree algorithm
(rmcar17/SpectralClusterSupertree, ``src/sc_supertree/scs.py``) used as the
checker for the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``spectralclustersupertree_amd``) never does.

Why a restatement and not the reference itself: the reference cannot be
imported in the build container -- ``cogent3`` and ``citeable`` are not
installed (``ModuleNotFoundError``) and ``scs.py:12-15`` uses PEP 695 ``type``
statements, a ``SyntaxError`` under the container's Python 3.10.  Those are
ordinary errors, nothing was denied.

How the restatement is pinned (see tests/test_oracle_reference_cases.py):

* every known-answer case the reference's own tests hold for this path -- the
  inline cases of ``tests/test_spectral_cluster_supertree.py:30-274`` and the
  three file fixtures under ``tests/test_data`` (committed as data under
  ``tests/golden/reference_data``) -- is reproduced topologically;
* the eigen-solve and label assignment are not restated at all: the oracle
  calls scikit-learn's ``SpectralClustering`` exactly as the reference does
  (``scs.py:235-241,252``), i.e. the real third-party numerics
  (scikit-learn 1.7.2 / scipy 1.15.3 in the container).

The dictionaries below are keyed the way the reference keys them (vertices are
sorted tuples of taxon names, edges are ordered pairs of vertices) so that each
function can be read against the cited lines.  The one deliberate difference:
the vertex order fed to the spectral step is ``sorted(vertices)`` instead of
set-iteration order (``scs.py:244``), because the reference's order depends on
the process hash seed and parity needs a fixed one.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent.parent
if str(_ROOT) not in sys.path:  # tests import the host tree model from the package
    sys.path.insert(0, str(_ROOT))

from spectralclustersupertree_amd.tree import (  # noqa: E402
    TreeNode,
    connect_trees,
    is_not_completed,
    make_tree,
    tip_names_to_tree,
)

STRATEGIES = ("one", "branch", "depth", "bootstrap")


def _pair(u, v):
    """Canonical unordered pair (reference: scs.py:666-686)."""
    return (u, v) if u < v else (v, u)


def _step_value(strategy, carried, node):
    """Value carried below ``node`` (reference: scs.py:555-564)."""
    if strategy == "one":
        return 1
    if strategy == "depth":
        return carried + 1
    if strategy == "branch":
        return carried + (1 if node.length is None else node.length)
    return node.support  # bootstrap


def build_pcg(vertices, trees, weights, strategy):
    """Proper cluster graph of weighted trees (reference: scs.py:495-583).

    Returns (adjacency, weight, occurrences, co_occurrences).
    """
    if strategy not in STRATEGIES:
        msg = f"Invalid weighting strategy selected: '{strategy}'"
        raise ValueError(msg)
    adjacency = {v: set() for v in vertices}
    occurrences = dict.fromkeys(vertices, 0)
    weight: dict = {}
    together: dict = {}

    def descend(node, tree_weight, carried):
        # reference: scs.py:586-663
        if node.is_tip():
            return [(node.name,)]
        carried = _step_value(strategy, carried, node)
        below = [descend(child, tree_weight, carried) for child in node]
        for hi in range(1, len(below)):
            for lo in range(hi):
                for a in below[hi]:
                    for b in below[lo]:
                        adjacency[a].add(b)
                        adjacency[b].add(a)
                        key = _pair(a, b)
                        weight[key] = weight.get(key, 0) + carried * tree_weight
                        together[key] = together.get(key, 0) + 1
        merged = below[0]
        for extra in below[1:]:
            merged.extend(extra)
        return merged

    for tree, tree_weight in zip(trees, weights):
        for side in tree:
            for vertex in descend(side, tree_weight, 0):
                occurrences[vertex] += 1
    return adjacency, weight, occurrences, together


def graph_components(vertices, adjacency):
    """Connected components by graph search (reference: scs.py:458-492)."""
    todo = set(vertices)
    out = []
    while todo:
        seed = todo.pop()
        comp = {seed}
        frontier = [seed]
        while frontier:
            here = frontier.pop()
            for nxt in adjacency[here]:
                if nxt not in comp:
                    comp.add(nxt)
                    frontier.append(nxt)
        out.append(comp)
        todo -= comp
    return out


def contract_pcg(vertices, adjacency, weight, occurrences, together):
    """In-place contraction of always-together taxa (reference: scs.py:261-387)."""
    full_adj: dict = {}
    for (u, v), count in together.items():
        if count == max(occurrences[u], occurrences[v]):  # scs.py:302-305
            full_adj.setdefault(u, set()).add(v)
            full_adj.setdefault(v, set()).add(u)
    groups = graph_components(set(full_adj), full_adj)  # scs.py:316
    merged_name = []
    for group in groups:
        names = []
        for vertex in group:
            names.extend(vertex)
        merged_name.append(tuple(sorted(names)))  # scs.py:324, 689-705
    renamed = {}
    for group, new in zip(groups, merged_name):
        for vertex in group:
            renamed[vertex] = new

    collected: dict = {}
    for group, new in zip(groups, merged_name):
        vertices.difference_update(group)
        inside = set(new)
        for vertex in group:
            for nb in adjacency[vertex]:
                old = _pair(vertex, nb)
                if inside.issuperset(nb):  # scs.py:352-354
                    weight.pop(old, None)
                    continue
                tgt = _pair(new, renamed.get(nb, nb))
                collected.setdefault(tgt, []).append(weight[old])  # scs.py:364-366
                adjacency[nb].remove(vertex)
                del weight[old]
            del adjacency[vertex]
    for new in merged_name:
        vertices.add(new)
        adjacency.setdefault(new, set())
    for (u, v), values in collected.items():
        adjacency[u].add(v)
        adjacency[v].add(u)
        weight[(u, v)] = max(values)  # scs.py:387


def dense_matrix(vertex_list, weight):
    """V x V float64 fill, zero where no edge (reference: scs.py:246-250)."""
    n = len(vertex_list)
    a = np.zeros((n, n))
    for i, u in enumerate(vertex_list):
        for j, v in enumerate(vertex_list):
            a[i, j] = weight.get(_pair(u, v), 0)
    return a


def spectral_labels(matrix, random_state):
    """Labels exactly as the reference obtains them (reference: scs.py:235-252)."""
    from sklearn.cluster import SpectralClustering

    sc = SpectralClustering(
        2,
        affinity="precomputed",
        assign_labels="kmeans",
        n_jobs=1,
        random_state=random_state,
    )
    return sc.fit_predict(matrix)


def spectral_maps(matrix, random_state):
    """The V x 2 embedding ``fit_predict`` clusters.

    Public twin of the private call at sklearn/cluster/_spectral.py:748-755.
    Consumes ``random_state`` exactly like the first half of ``fit_predict``
    (the ARPACK start vector, sklearn/utils/_arpack.py:31-33).
    """
    from sklearn.manifold import spectral_embedding

    return spectral_embedding(
        matrix,
        n_components=2,
        eigen_solver=None,
        random_state=random_state,
        eigen_tol="auto",
        drop_first=False,
    )


def spectral_bipartition(vertices, weight, random_state, vertex_list=None, trace_entry=None, steer=None):
    """Two vertex sets, the set of label 0 first (reference: scs.py:210-258), fixed vertex
    order.  ``trace_entry`` (a dict) receives the label vector and the generator state in
    front of the call; ``steer(trace_entry, labels) -> labels`` may replace the labels the
    walk continues with (tests: keep two walks aligned across an exact tie)."""
    if vertex_list is None:
        vertex_list = sorted(vertices)
    if trace_entry is not None:
        trace_entry["rng_state"] = random_state.get_state()
    labels = spectral_labels(dense_matrix(vertex_list, weight), random_state)
    if trace_entry is not None:
        trace_entry["labels"] = np.asarray(labels).copy()
    if steer is not None:
        labels = steer(trace_entry, labels)
    parts = [set(), set()]
    for vertex, lab in zip(vertex_list, labels):
        parts[lab].add(vertex)
    return parts


def _all_tips(trees):
    names = set()
    for tree in trees:
        names.update(tree.get_tip_names())
    return names


def induce(names, trees, weights):
    """Restrict trees to ``names`` (reference: scs.py:411-455)."""
    out_trees, out_weights = [], []
    for tree, w in zip(trees, weights):
        if len(names.intersection(tree.get_tip_names())) < 2:
            continue
        sub = tree.get_sub_tree(names, ignore_missing=True, as_rooted=True)
        sub.name = "root"
        out_trees.append(sub)
        out_weights.append(w)
    return out_trees, out_weights


def construct_supertree_oracle(
    trees,
    weights=None,
    pcg_weighting="one",
    *,
    contract_edges=True,
    random_state=None,
    trace=None,
    trace_matrices=True,
    steer=None,
):
    """Whole algorithm on the CPU (reference: scs.py:18-174).

    ``trace`` (a list) receives one dict per spectral call -- vertex order,
    dense matrix (dropped again when ``trace_matrices`` is False), label vector -- so
    tests can harvest golden vectors and compare the recursion node by node.

    Child order: the two parts of a spectral split are visited label 0 first, as the
    reference's ``partition`` list is (scs.py:254-258, 139) -- the order decides where in the
    shared RandomState stream each child's draws fall.  The parts of a component split come
    in set order in the reference (hash-seed dependent, scs.py:458-492); here: by smallest
    taxon name.  ``steer``: see ``spectral_bipartition`` (needs ``trace``).
    """
    if random_state is None:
        random_state = np.random.RandomState()
    if len(trees) == 0:
        raise ValueError("There must be at least one tree to make a supertree.")
    if pcg_weighting not in STRATEGIES:
        raise ValueError(f"Invalid weighting strategy selected: '{pcg_weighting}'")
    if weights is None:
        weights = [1.0] * len(trees)
    if len(trees) != len(weights):
        msg = (
            f"The number of trees ({len(trees)}) "
            f"and tree weights ({len(weights)}) must match."
        )
        raise ValueError(msg)
    kept = [(t, w) for t, w in zip(trees, weights) if not is_not_completed(t)]
    if not kept:
        raise ValueError("There must be at least one tree to make a supertree.")
    trees = [t for t, _ in kept]
    weights = [w for _, w in kept]

    if len(trees) == 1:  # scs.py:96-98
        only = trees[0].copy()
        for node in only.iter_nontips(include_self=True):
            node.name = ""
        return make_tree(only.get_newick())

    names = _all_tips(trees)
    if len(names) <= 2:
        return tip_names_to_tree(sorted(names))

    vertices = {(n,) for n in names}
    adjacency, weight, occ, together = build_pcg(vertices, trees, weights, pcg_weighting)
    parts = graph_components(vertices, adjacency)
    if len(parts) == 1:
        if contract_edges:
            contract_pcg(vertices, adjacency, weight, occ, together)
        order = sorted(vertices)
        entry = None
        if trace is not None:
            entry = {"vertices": order}
            if trace_matrices:
                entry["matrix"] = dense_matrix(order, weight)
            trace.append(entry)
        parts = spectral_bipartition(vertices, weight, random_state, order, entry, steer)
    else:
        parts = sorted(parts, key=lambda p: min(p))

    children = []
    for part in parts:
        taxa = set()
        for vertex in part:
            taxa.update(vertex)
        if len(taxa) <= 2:
            children.append(tip_names_to_tree(sorted(taxa)))
            continue
        sub_trees, sub_weights = induce(taxa, trees, weights)
        children.append(
            construct_supertree_oracle(
                sub_trees,
                sub_weights,
                pcg_weighting,
                contract_edges=contract_edges,
                random_state=random_state,
                trace=trace,
                trace_matrices=trace_matrices,
                steer=steer,
            )
        )
        lost = taxa.difference(_all_tips(sub_trees))
        children.extend(TreeNode(name) for name in sorted(lost))
    return connect_trees(children)
