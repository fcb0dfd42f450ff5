"""cogent3 app plug-ins (reference: src/sc_supertree/_app.py:34-96, entry points
pyproject.toml:54-57): ``load_trees``, ``sc_supertree`` and ``outgroup_root`` over this package's
``construct_supertree``.  With cogent3 installed they are ``define_app`` composables as in the
reference; without it (this build's image) they are plain callables with the same arguments and
error behaviour, so pipelines written against the reference's apps read the same."""

from __future__ import annotations

import os
from collections.abc import Sequence

import numpy as np

from spectralclustersupertree_amd.load import load_trees as _load_trees
from spectralclustersupertree_amd.scs import construct_supertree as _construct

try:  # pragma: no cover - cogent3 is not installable in the build image
    from cogent3.app.composable import define_app as _define_app  # type: ignore[import-not-found]
except ImportError:
    def _define_app(fn=None, **_):
        return fn if fn is not None else (lambda f: f)


@_define_app
def load_trees(source_tree_file):
    """Line-separated Newick file -> list of trees; ``TypeError`` for anything but a path."""
    if not isinstance(source_tree_file, (str, os.PathLike)):
        msg = f"Invalid Path Type: '{type(source_tree_file)}'."
        raise TypeError(msg)
    return _load_trees(source_tree_file)


@_define_app
def sc_supertree(
    trees,
    weights: Sequence[float] | None = None,
    pcg_weighting: str = "one",
    *,
    contract_edges: bool = True,
    random_state: np.random.RandomState | None = None,
):
    return _construct(trees, weights, pcg_weighting, contract_edges=contract_edges,
                      random_state=random_state)


@_define_app
def outgroup_root(tree, *, priority_outgroups: Sequence[str]):
    """Root the tree at the first of ``priority_outgroups`` it contains."""
    tip_names = set(tree.get_tip_names())
    for name in priority_outgroups:
        if name in tip_names:
            return tree.rooted(name)
    msg = f"Tree does not contain any tip names in: {priority_outgroups}"
    raise ValueError(msg)
