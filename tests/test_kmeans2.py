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


def _starts(x, tol, native):
    from sklearn.cluster import _k_means_common as kc
    from sklearn.cluster import _k_means_lloyd as kl

    ref = kmeans2._lloyd_cython(x, np.ones(len(x)), tol, kl, kc)
    return ref, kmeans2._lloyd_native(x, tol, ref, *native)


def test_native_lloyd_iteration_equals_the_cython_kernel_start_by_start():
    # csrc/scs_kmeans.c against lloyd_iter_chunked_dense / _inertia_dense: labels, inertia bits and
    # the number of iterations, from seeded and from arbitrary centres
    if not kmeans2.native_lloyd_active():
        pytest.skip("the C iteration is not active here")
    native = kmeans2._native()
    rs = np.random.RandomState(99)
    checked = 0
    for case in range(300):
        n = int(rs.choice([2, 3, 4, 5, 7, 16, 33, 64, 100, 255, 256, 257, 511, 513, 1500, 4096]))
        x = rs.standard_normal((n, 2)) * [1.0, 10.0 ** rs.randint(-8, 2)]
        if case % 4 == 1:
            x = np.round(x, 1)
        if case % 4 == 2:
            x[:, 0] = rs.rand()
        x = np.ascontiguousarray(x - x.mean(axis=0))
        tol = float(np.mean(np.var(x, axis=0)) * 1e-4)
        ref, nat = _starts(x, tol, native)
        for start in range(3):
            c = x[rs.choice(n, 2, replace=False)].copy() if start < 2 else rs.standard_normal((2, 2))
            got, want = nat(c.copy(), True), ref(c.copy(), True)
            assert np.array_equal(got[0], want[0]), (case, n, start)
            assert got[1] == want[1] and got[2] == want[2], (case, n, start, got[1:], want[1:])
            checked += 1
    assert checked == 900


def test_native_lloyd_hands_an_empty_cluster_back():
    # every point nearer to the first centre: scikit-learn relocates the empty cluster's centre with
    # numpy operations -- the C side returns 1 and the start runs on the Cython kernel
    if not kmeans2.native_lloyd_active():
        pytest.skip("the C iteration is not active here")
    import ctypes as C

    lib, ptr = kmeans2._native()
    rs = np.random.RandomState(5)
    x = np.ascontiguousarray(rs.standard_normal((40, 2)))
    x -= x.mean(axis=0)
    far = np.array([[0.0, 0.0], [50.0, 50.0]])
    lab = np.empty(40, dtype=np.int32)
    inertia, iters = C.c_double(), C.c_int32()
    rc = lib.scs_host_lloyd2(ptr, 40, x.ctypes.data, far.ctypes.data, 1e-4, 300, lab.ctypes.data,
                             C.byref(inertia), C.byref(iters))
    assert rc == 1
    ref, nat = _starts(x, 1e-4, (lib, ptr))
    got, want = nat(far.copy(), True), ref(far.copy(), True)
    assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
    assert lib.scs_host_lloyd2(None, 40, x.ctypes.data, far.ctypes.data, 1e-4, 300, lab.ctypes.data,
                               C.byref(inertia), C.byref(iters)) == -1


def test_forced_cython_kernel(monkeypatch):
    monkeypatch.setenv("SCS_KMEANS", "cython")
    saved = dict(kmeans2._state)
    try:
        kmeans2._state.update(checked=False, ok=False, native=None)
        assert kmeans2._native() is None
        if kmeans2.fast_path_active():
            assert not kmeans2.native_lloyd_active()
        assert _same(np.random.RandomState(1).standard_normal((21, 2)), 8)
    finally:
        kmeans2._state.update(saved)


def test_native_seeding_equals_the_numpy_seeding_bit_for_bit():
    # scs_host_kmeans2 (seedings through numpy's own OpenBLAS entry points + the C iteration) against
    # the numpy / Cython path on the same draws: seeds, potentials, labels (kmeans2._whole_agrees)
    if not kmeans2.native_seeding_active():
        pytest.skip("the C seeding is not active here")
    assert kmeans2._whole_agrees()
    rs = np.random.RandomState(17)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for case in range(300):
            n = int(rs.randint(2, 257)) if case % 6 else int(rs.choice([257, 600, 1000, 2048, 4096]))
            x = rs.standard_normal((n, 2)) * [1.0, 10.0 ** rs.randint(-7, 2)]
            if case % 5 == 0:
                x = np.round(x, 1)
            ra, rb = np.random.RandomState(case), np.random.RandomState(case)
            la, lb = kmeans2._fast(x, ra), kmeans2._fast(x, rb, use_native=False)
            assert np.array_equal(la, lb), (case, n)
            assert np.array_equal(ra.get_state()[1], rb.get_state()[1]) and ra.get_state()[2:] == rb.get_state()[2:]


def test_native_seeding_hands_an_emptied_cluster_back():
    # one distinct point: both centres coincide, the second cluster stays empty, the C call returns 1
    # and the numpy / Cython path finishes the call on the same draws
    if not kmeans2.native_seeding_active():
        pytest.skip("the C seeding is not active here")
    x = np.zeros((6, 2))
    x_sq = np.zeros(6)
    cdf = np.arange(1, 7) / 6.0
    draws = np.random.RandomState(0).random_sample(30)
    assert kmeans2._whole_call(kmeans2._whole(), x, x_sq, cdf, draws, 0.0) is None
    assert _same(x, 9)
    assert _same(np.array([[1.0, 2.0]] * 3 + [[1.0, 2.0 + 1e-300]]), 10)


def test_first_use_from_two_threads_at_once():
    """Two ranks of an in-process team (or the walk and its look-ahead worker) can reach their
    first label assignment together: the self-test must run once, on one thread -- run
    concurrently, threadpoolctl's library scan (loader lock) and a scikit-learn import
    (interpreter lock) waited for each other (profiles/r04_team_first_use_deadlock.txt).  A
    fresh interpreter, so that nothing is imported or self-tested yet."""
    import subprocess
    import sys
    from pathlib import Path

    code = (
        "import threading, numpy as np\n"
        "from spectralclustersupertree_amd import kmeans2\n"
        "x = np.random.RandomState(3).standard_normal((40, 2))\n"
        "out = [None, None]\n"
        "gate = threading.Barrier(2)\n"
        "def work(i):\n"
        "    gate.wait()\n"
        "    out[i] = kmeans2.labels(x, np.random.RandomState(5))\n"
        "ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "assert np.array_equal(out[0], out[1])\n"
        "print('ok', kmeans2.fast_path_active())\n")
    root = Path(__file__).resolve().parent.parent
    res = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.split()[0] == "ok"


def test_centre_and_tolerance_is_numpys_var_and_mean_bit_for_bit():
    # kmeans2._centre_and_tolerance replaces np.mean(np.var(x, axis=0)) * 1e-4 and x -= x.mean(axis=0)
    # (sklearn/cluster/_kmeans.py: _tolerance and the centring in fit) by the same operations without the calls'
    # bookkeeping: the centred array and the tolerance must be the calls', bit for bit, at every size the fast path
    # takes -- including constant columns, tiny and huge scales, ties and duplicated columns
    from spectralclustersupertree_amd.kmeans2 import _centre_and_tolerance

    rs = np.random.RandomState(11)
    sizes = list(range(2, 140)) + [255, 256, 257, 511, 512, 513, 1000, 2047, 2048, 4095, 4096]
    for n in sizes:
        for kind in range(6):
            x = rs.standard_normal((n, 2))
            if kind == 1:
                x[:, 0] = 1.0 / np.sqrt(n)  # (the embedding's first column: constant)
            elif kind == 2:
                x *= 1e-8
            elif kind == 3:
                x = np.round(x, 1)
            elif kind == 4:
                x[:, 1] = x[:, 0]
            elif kind == 5:
                x[:, 0] *= 1e150
            a, b = np.array(x, copy=True), np.array(x, copy=True)
            want_tol = np.mean(np.var(a, axis=0)) * 1e-4
            a -= a.mean(axis=0)
            got_tol = _centre_and_tolerance(b)
            assert np.array_equal(a, b), (n, kind)
            assert np.float64(want_tol).tobytes() == np.float64(got_tol).tobytes(), (n, kind)
