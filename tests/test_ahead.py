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
