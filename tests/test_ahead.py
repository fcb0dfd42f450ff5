"""ahead.Ahead: the queue that lets the device work on recursion nodes before the walk visits
them (pure host logic; the jobs here are plain Python callables)."""

import threading
import time

import pytest

from spectralclustersupertree_amd.ahead import Ahead


def _until(cond, seconds=5.0):
    end = time.monotonic() + seconds
    while not cond() and time.monotonic() < end:
        time.sleep(0.002)
    assert cond()


class FakeDevice:
    made = 0
    closed = 0

    def __init__(self):
        FakeDevice.made += 1
        self.owner = threading.current_thread().name

    def close(self):
        FakeDevice.closed += 1


def test_results_do_not_depend_on_who_ran_the_job():
    mine = FakeDevice()
    with Ahead(FakeDevice) as q:
        jobs = [q.submit(lambda dev, i=i: i * i) for i in range(200)]
        assert [q.result(j, mine) for j in jobs] == [i * i for i in range(200)]
        assert q.stats["submitted"] == 200
        assert q.stats["by_worker"] + q.stats["by_walk"] == 200


def test_the_worker_has_a_device_of_its_own_and_a_needed_job_runs_on_the_callers():
    gate = threading.Event()
    made, closed = FakeDevice.made, FakeDevice.closed
    mine = FakeDevice()

    def work(dev, tag, wait=False):
        if wait:
            gate.wait(5)
        return tag, dev.owner

    with Ahead(FakeDevice) as q:
        first = q.submit(lambda dev: work(dev, "first", wait=True))  # the worker picks it up and blocks
        _until(lambda: first.state == 1)
        later = [q.submit(lambda dev, i=i: work(dev, i)) for i in range(5)]
        # a queued job: the asker runs it itself, at once, on its own device
        assert q.result(later[3], mine) == (3, mine.owner)
        gate.set()
        assert q.result(first, mine) == ("first", "scs-ahead-0")
        assert [q.result(j, mine)[0] for j in later] == list(range(5))
        assert q.stats["by_walk"] >= 1
    assert FakeDevice.made == made + 2  # mine + the worker's, made on the worker thread
    assert FakeDevice.closed == closed + 1  # the worker closes its own


def test_a_failure_travels_with_the_result_and_close_drops_what_is_queued():
    def boom(dev):
        msg = "no good"
        raise ValueError(msg)

    mine = FakeDevice()
    q = Ahead(FakeDevice)
    job = q.submit(boom)
    with pytest.raises(ValueError, match="no good"):
        q.result(job, mine)
    block = threading.Event()
    blocker = q.submit(lambda dev: block.wait(5))
    _until(lambda: blocker.state == 1)
    never = [q.submit(lambda dev: 1 / 0) for _ in range(3)]
    block.set()
    q.close()
    for j in never:  # dropped, not run
        if j.state == 0:
            with pytest.raises(RuntimeError, match="closed"):
                q.result(j, mine)
    with pytest.raises(RuntimeError):
        q.submit(lambda dev: 1)
    q.close()  # idempotent


def test_a_worker_that_cannot_make_its_device_fails_the_job_not_the_process():
    def no_device():
        msg = "no GPU"
        raise OSError(msg)

    with Ahead(no_device) as q:
        job = q.submit(lambda dev: 1)
        _until(lambda: job.state != 0)
        with pytest.raises(OSError, match="no GPU"):
            q.result(job, None)


def test_several_workers_take_jobs_side_by_side_each_on_a_device_of_its_own():
    """Round 6: the level-synchronous recursion hands over all larger nodes of a level at once."""
    gate = threading.Event()
    running = []
    lock = threading.Lock()

    def work(dev, tag):
        with lock:
            running.append(dev.owner)
        gate.wait(5)
        return tag, dev.owner

    mine = FakeDevice()
    made = FakeDevice.made
    with Ahead(FakeDevice, workers=3) as q:
        jobs = [q.submit(lambda dev, i=i: work(dev, i)) for i in range(6)]
        _until(lambda: len(running) == 3)  # three jobs under way at once, none waiting for another
        assert len(set(running)) == 3
        gate.set()
        got = [q.result(j, mine) for j in jobs]
        assert [g[0] for g in got] == list(range(6))
        assert q.stats["by_worker"] + q.stats["by_walk"] == 6
    assert FakeDevice.made == made + 3


# ---------------------------------------------------------------------------------------------------------
# levels._Lazy: the nodes above the cap of one level, one after the other in the order of the visit
# ---------------------------------------------------------------------------------------------------------
class _FakeEngine:
    """What levels._Lazy asks of an engine: the queue and a job per node."""

    def __init__(self, queue, work):
        self.ahead, self.work = queue, work

    def _large_job(self, lev, k, relabel, gs_patch):
        return lambda dev: self.work(k, dev)


def test_lazy_nodes_are_embedded_one_after_the_other_in_the_order_of_the_visit():
    from spectralclustersupertree_amd.levels import _Lazy

    running, most, order = [0], [0], []
    lock = threading.Lock()

    def work(k, dev):
        with lock:
            running[0] += 1
            most[0] = max(most[0], running[0])
            order.append(k)
        time.sleep(0.02)
        with lock:
            running[0] -= 1
        return ("maps", k, dev.owner)

    mine = FakeDevice()
    with Ahead(FakeDevice, workers=3) as q:
        lazy = _Lazy(_FakeEngine(q, work), None, [4, 7, 9], None, None)
        assert lazy.pending(4) and lazy.pending(9) and not lazy.pending(5)
        _until(lambda: lazy.job.state == 1)  # (a worker has the sequence: the walk only waits)
        got = [lazy.fetch(k, mine) for k in (4, 7, 9)]
    assert [g[1] for g in got] == [4, 7, 9]
    assert order == [4, 7, 9] and most[0] == 1  # never two of them side by side
    assert not lazy.pending(4) and not lazy.value and not lazy.fns  # handed over once, nothing kept


def test_a_lazy_sequence_nobody_picked_up_is_computed_by_the_walk_node_by_node():
    from spectralclustersupertree_amd.levels import _Lazy

    gate = threading.Event()
    mine = FakeDevice()
    with Ahead(FakeDevice, workers=1) as q:
        blocker = q.submit(lambda dev: gate.wait(5))  # the only worker is busy
        _until(lambda: blocker.state == 1)
        lazy = _Lazy(_FakeEngine(q, lambda k, dev: (k, dev.owner)), None, [1, 2], None, None)
        assert lazy.fetch(1, mine) == (1, mine.owner)  # here, now -- and only this node
        assert 2 not in lazy.claimed
        gate.set()
        q.result(blocker, mine)
        _until(lambda: lazy.done[2].is_set())  # the worker, free again, takes the rest and skips node 1
        assert lazy.fetch(2, mine) == (2, "scs-ahead-0")


def test_a_lazy_node_that_failed_on_the_workers_context_is_solved_again_on_the_walks_own():
    from spectralclustersupertree_amd.levels import _Lazy

    mine = FakeDevice()

    def work(k, dev):
        if k == 2 and dev is not mine:
            raise RuntimeError("out of device memory beside another node")
        if k == 3:
            raise ValueError("the node itself")
        return (k, dev.owner)

    with Ahead(FakeDevice, workers=2) as q:
        lazy = _Lazy(_FakeEngine(q, work), None, [1, 2, 3], None, None)
        _until(lambda: lazy.done[3].is_set())
        assert lazy.fetch(1, mine)[0] == 1
        assert lazy.fetch(2, mine) == (2, mine.owner)
        with pytest.raises(ValueError, match="the node itself"):
            lazy.fetch(3, mine)
