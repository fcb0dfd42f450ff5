"""The matrix-free operator (``scs_graph_matrix_free``, csrc/scs_matfree.h: a measured comparison, not the
product path) against the dense path on the same tables: degrees (W applied to the vector of ones) against
the row sums of the built matrix, and the Fiedler pair of both solves.  Reference semantics:
src/sc_supertree/scs.py:569-663 (the pair weights), :246-257 (the eigen-solve)."""

from __future__ import annotations

import numpy as np
import pytest

from spectralclustersupertree_amd import _native as nv
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    d = Device(0)
    yield d
    d.close()


@pytest.mark.parametrize(("n", "m", "strategy", "coverage", "weights", "block"), [
    (300, 12, "branch", 1.0, True, 4), (700, 40, "depth", 0.7, False, 4), (1500, 9, "one", 1.0, True, 8),
    (900, 25, "bootstrap", 1.0, True, 4), (2500, 6, "branch", 0.5, True, 8), (5000, 30, "branch", 1.0, False, 4)])
def test_matrix_free_operator_and_solve_agree_with_the_dense_path(dev, n, m, strategy, coverage, weights, block):
    lpt = None if coverage == 1.0 else int(coverage * n)
    tables = synthetic.make_tables(n + m, n, m, strategy, leaves_per_tree=lpt, random_weights=weights)
    dtab = dev.upload(tables)
    g = dtab.build()
    gm = dtab.matrix_free_graph(max_block=8)
    try:
        deg, deg_mf = g.degrees(), gm.degrees()
        assert np.allclose(deg_mf, deg, rtol=1e-12, atol=1e-12 * float(np.max(deg)))
        maps, st = g.fiedler(None, block=block)
        maps_mf, st_mf = gm.fiedler(None, block=block)
        with pytest.raises(nv.ScsError):
            gm.download()
    finally:
        gm.free()
        g.free()
        dtab.free()
    assert st["converged"] == 1 and st_mf["converged"] == 1
    assert st_mf["block"] == block and st_mf["n_apply32"] == 0
    assert abs(st["lambda"][1] - st_mf["lambda"][1]) <= 1e-12
    gap = abs(st["lambda"][1] - st["lambda_next"])
    if gap > 1e-6:  # a repeated eigenvalue has no unique vector
        scale = float(np.max(np.abs(maps[:, 1])))
        assert float(np.max(np.abs(maps_mf[:, 1] - maps[:, 1]))) <= 1e-9 * scale / min(1.0, gap * 1e3)
