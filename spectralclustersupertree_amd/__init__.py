"""MI355X-native spectral-clustering core for Spectral Cluster Supertree.

Public API mirrors the reference package (reference: src/sc_supertree/__init__.py:6-9):
``construct_supertree`` and ``load_trees``.
"""

from spectralclustersupertree_amd.load import load_trees
from spectralclustersupertree_amd.scs import construct_supertree

__all__ = ["construct_supertree", "load_trees"]
__version__ = "0.1.0"
