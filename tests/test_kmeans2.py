"""kmeans2.labels against scikit-learn's public k_means: labels AND the position of the
RandomState stream afterwards, bit for bit (the recursion's next draw depends on it)."""

import warnings

import numpy as np
import pytest

from spectralclustersupertree_amd import kmeans2


def _same(x, seed):
    ra, rb = np.random.RandomState(seed), np.random.RandomState(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        la, lb = kmeans2.labels(x, ra), kmeans2._public(x, rb)
    sa, sb = ra.get_state(), rb.get_state()
    return np.array_equal(la, lb) and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


def test_fast_path_is_active_with_the_pinned_scikit_learn():
    import sklearn

    if sklearn.__version__ not in kmeans2._KNOWN:
        pytest.skip("restatement not written against this scikit-learn: the public function is used")
    assert kmeans2.fast_path_active()


def test_labels_and_stream_match_k_means():
    rs = np.random.RandomState(3)
    for _ in range(400):
        n = int(rs.choice([2, 3, 4, 5, 6, 8, 11, 17, 40, 64, 129, 256, 257, 700]))
        kind = rs.randint(4)
        if kind == 0:
            x = rs.standard_normal((n, 2))
        elif kind == 1:  # the shape of a spectral embedding: a constant column and a small one
            x = np.column_stack([np.full(n, rs.rand()),
                                 np.sign(rs.standard_normal(n)) * rs.rand(n) * 10.0 ** rs.randint(-6, 3)])
        elif kind == 2:  # many duplicates
            x = np.round(rs.standard_normal((n, 2)), 1)
        else:  # two tight groups
            x = np.column_stack([rs.rand(n) * 1e-3 + 0.5, np.r_[rs.rand(n // 2) - 3, rs.rand(n - n // 2) + 3]])
        assert _same(x, int(rs.randint(1 << 30))), (n, kind)


def test_degenerate_inputs_match_k_means():
    assert _same(np.zeros((5, 2)), 1)  # one distinct point: ConvergenceWarning on both sides
    assert _same(np.array([[0.0, 1.0], [0.0, -1.0]]), 2)
    assert _same(np.array([[1.0, 2.0], [1.0, 2.0], [1.0, 2.0], [5.0, 2.0]]), 3)


def test_forced_public_function(monkeypatch):
    monkeypatch.setenv("SCS_KMEANS", "sklearn")
    assert not kmeans2.fast_path_active()
    assert _same(np.random.RandomState(0).standard_normal((9, 2)), 4)


def test_inputs_outside_the_fast_range_use_the_public_function():
    x = np.random.RandomState(0).standard_normal((kmeans2._MAX_SAMPLES + 1, 2))
    assert _same(x, 5)
    # a seed instead of a generator is the public function's business
    lab = kmeans2.labels(x[:30], 11)
    assert set(np.unique(lab)) <= {0, 1}


def test_single_thread_lloyd_equals_the_public_call_above_one_chunk():
    # the fast path runs sklearn's Lloyd kernel on ONE thread also above 256 samples (several
    # chunks): same labels and the same generator state as the public, team-parallel call
    import warnings

    from spectralclustersupertree_amd import kmeans2

    if not kmeans2.fast_path_active():
        pytest.skip("fast path not active with this scikit-learn")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for seed in range(24):
            rs = np.random.RandomState(seed)
            n = int(rs.choice([257, 300, 511, 512, 513, 1000, 2049, 4096]))
            x = rs.standard_normal((n, 2)) * [1.0, 10.0 ** rs.randint(-4, 1)]
            if seed % 3 == 0:
                x[:, 1] = np.where(rs.rand(n) < 0.4, -1, 1) * 0.1 + x[:, 1] * 1e-3
            ra, rb = np.random.RandomState(7 + seed), np.random.RandomState(7 + seed)
            assert np.array_equal(kmeans2._fast(x, ra), kmeans2._public(x, rb)), (seed, n)
            assert np.array_equal(ra.get_state()[1], rb.get_state()[1]) and ra.get_state()[2:] == rb.get_state()[2:]
