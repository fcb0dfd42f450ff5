"""The device-memory arena on the device (csrc/scs_arena.h; its logic alone: tests/test_arena_cpu.py): blocks of
every context of the process are carved out of slabs that stay with the process, a block one context released
serves another context, ``trim`` hands whole free slabs back, ``reserve`` makes room ahead of a call -- and none
of that changes a bit of the results."""
import numpy as np
import pytest

from oracle import tables_oracle as to
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

pytestmark = pytest.mark.gpu


def _build_and_solve(dev, tables, v0):
    dtab = dev.upload(tables)
    g = dtab.build()
    w = g.download()
    maps, _ = g.fiedler(v0)
    g.free()
    dtab.free()
    return w, maps


def test_a_second_context_is_served_from_what_the_first_released():
    n, m = 3000, 40
    tables = synthetic.make_tables(11, n, m, "branch")
    w_ref, _ = to.pcg_dense(tables)
    v0 = np.random.RandomState(1).uniform(-1, 1, n)
    with Device(0) as a:
        a.trim()  # (whatever earlier tests left free goes back first: the counts below are this test's)
        w_a, maps_a = _build_and_solve(a, tables, v0)
        after_a = a.arena_stats()
        assert after_a["slab_bytes"] >= n * n * 8 and after_a["driver_allocations"] >= 1
        with Device(0) as b:
            w_b, maps_b = _build_and_solve(b, tables, v0)
            after_b = b.arena_stats()
        # the second context's W, tables and scratch came out of the slabs the first had filled: no driver call
        assert after_b["driver_allocations"] == after_a["driver_allocations"]
        assert after_b["slab_bytes"] == after_a["slab_bytes"]
        assert after_b["requests"] > after_a["requests"]
    assert np.array_equal(w_a, w_ref) and np.array_equal(w_b, w_ref)
    assert np.array_equal(maps_a, maps_b)


def test_trim_hands_free_slabs_back_and_reserve_takes_them_ahead_of_a_call():
    with Device(0) as dev:
        dev.trim()
        base = dev.arena_stats()
        dev.reserve(3 << 30)
        held = dev.arena_stats()
        assert held["slab_bytes"] >= base["slab_bytes"] + (3 << 30)
        assert held["used_bytes"] == base["used_bytes"]  # (reserved, not in use)
        # a graph whose W fits the reservation: the driver is not asked again for it
        n = 15000
        tables = synthetic.make_tables(5, n, 8, "one")
        dtab = dev.upload(tables)
        before = dev.arena_stats()["driver_allocations"]
        g = dtab.build()
        w_bytes = n * ((n + 511) // 512 * 512) * 8
        assert w_bytes < (3 << 30)
        during = dev.arena_stats()
        assert during["used_bytes"] >= base["used_bytes"] + w_bytes
        g.free()
        dtab.free()
        assert dev.arena_stats()["driver_allocations"] - before <= 2  # (small scratch slabs at most; not W)
        dev.synchronize()
        dev.trim()
        after = dev.arena_stats()
        assert after["slab_bytes"] < held["slab_bytes"] - (2 << 30)
        assert after["driver_releases"] > held["driver_releases"]
        dev.trim(keep_bytes=1 << 40)  # (keeps everything: nothing to do)
        assert dev.arena_stats()["driver_releases"] == after["driver_releases"]


def test_results_do_not_depend_on_where_the_arena_puts_a_block():
    """The same build + solve with the arena empty, and with it fragmented by blocks of other sizes that are
    alive or were released in between: bit-identical W and embedding (recycled memory is never assumed clean)."""
    n, m = 2200, 30
    tables = synthetic.make_tables(21, n, m, "depth", leaves_per_tree=1500)
    v0 = np.random.RandomState(3).uniform(-1, 1, n)
    w_ref, _ = to.pcg_dense(tables)
    with Device(0) as dev:
        dev.trim()
        w0, maps0 = _build_and_solve(dev, tables, v0)
        # fragment: graphs of other sizes, some kept alive across the second run
        keep = []
        for i, k in enumerate((700, 1900, 1200, 2600)):
            t = synthetic.make_tables(30 + i, k, 6, "branch")
            dt = dev.upload(t)
            g = dt.build()
            if i % 2 == 0:
                keep.append((g, dt))
            else:
                g.free()
                dt.free()
        w1, maps1 = _build_and_solve(dev, tables, v0)
        for g, dt in keep:
            g.free()
            dt.free()
    assert np.array_equal(w0, w_ref) and np.array_equal(w1, w_ref)
    assert np.array_equal(maps0, maps1)


def test_an_object_freed_after_its_context_releases_its_own_memory_only():
    """A graph that outlives its context (the documented order is the other way round; interpreter shutdown does
    not always keep it): its blocks are orphaned when the context goes, NOT released -- so that its late release
    cannot hit the memory another context has been given in between."""
    n = 1800
    tables = synthetic.make_tables(41, n, 12, "branch")
    w_ref, _ = to.pcg_dense(tables)
    a = Device(0)
    dtab = a.upload(tables)
    g = dtab.build()
    dtab.free()
    used_before = a.arena_stats()["used_bytes"]
    a.close()  # (g still holds W)
    with Device(0) as b:
        assert b.arena_stats()["used_bytes"] >= n * n * 8  # W is still accounted for
        # the other context allocates and computes while the orphan is alive ...
        w_b, maps_b = _build_and_solve(b, tables, None)
        g.free()  # ... and after it is gone
        w_c, maps_c = _build_and_solve(b, tables, None)
        assert b.arena_stats()["used_bytes"] < used_before
    assert np.array_equal(w_b, w_ref) and np.array_equal(w_c, w_ref)
    assert np.array_equal(maps_b, maps_c)


def test_trim_releases_the_idle_small_solve_slots_too():
    """The staging and scratch blocks of a batched small solve stay with their slot for the next batch -- and go back
    with ``trim()`` (they used to stay until the context went away: ADVICE r05); the next batch makes them anew and
    returns the same bits."""
    nodes = [(synthetic.make_tables(90 + i, n, 40, "branch"), None) for i, n in enumerate((30, 64, 100, 128))]
    with Device(0) as dev:
        dev.trim()
        base = dev.arena_stats()["used_bytes"]
        want = dev.small_solve(nodes, want_w=True)
        held = dev.arena_stats()["used_bytes"]
        assert held > base  # (the slot keeps its blocks)
        dev.trim()
        assert dev.arena_stats()["used_bytes"] <= held - (2 << 20)  # (staging and scratch: a MiB each at least)
        got = dev.small_solve(nodes, want_w=True)
        for g, w in zip(got, want):
            assert all(np.array_equal(a, b) for a, b in zip(g, w))
        ticket = dev.small_solve_begin(nodes)  # a slot a ticket holds is left alone
        dev.trim()
        out = ticket.result()
        for g, w in zip(out, want):
            assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1])
