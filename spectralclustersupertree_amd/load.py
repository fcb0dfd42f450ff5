"""``load_trees``: one Newick tree per line (reference: src/sc_supertree/load.py:7-23)."""

from __future__ import annotations

import os
from pathlib import Path


def _make_tree(newick: str):
    """cogent3's parser when it is installed, else the built-in one."""
    try:
        from cogent3 import make_tree  # type: ignore[import-not-found]
    except ImportError:
        from spectralclustersupertree_amd.tree import make_tree
    return make_tree(newick)


def load_trees(source_tree_file: str | os.PathLike) -> list:
    """Load a line-separated file of Newick-formatted trees.

    Same contract as the reference: every line is handed to the tree parser
    (blank lines are not skipped, reference: load.py:21-22).
    """
    with Path(source_tree_file).open() as f:
        return [_make_tree(line.strip()) for line in f]


def load_tree_arrays(source_tree_file: str | os.PathLike):
    """The same file as flat tree arrays, parsed in C without building tree objects
    (``treearrays.TreeArrays.from_newick_file``); ``construct_supertree`` accepts the result
    in place of the list of trees.  For inputs too large for Python objects."""
    from spectralclustersupertree_amd.treearrays import TreeArrays

    return TreeArrays.from_newick_file(source_tree_file)
