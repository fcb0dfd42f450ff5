"""Device work queued ahead of the recursion's visit.

The embedding of a recursion node depends on the node's forest only -- no random draw enters
it (the reference's ARPACK start vector is drawn from the stream when the node is VISITED, to
keep the stream where the reference has it, but the LOBPCG solve does not use it) -- so it
can be computed as soon as the node's parent has been split, long before the depth-first
walk of ``construct_supertree`` arrives there (reference order of the walk: scs.py:139-171).

The walk itself is a chain -- a node's labels need its embedding, its children need its
labels -- and everything on that chain stays on the walk's own thread and context.  What is
NOT on it are the right siblings: while the walk is busy with a left child's whole subtree,
``Ahead`` builds and solves the right child on a worker thread with a context (stream, scratch)
of its own on the same GPU.  A node of a few hundred or thousand taxa is a latency-bound solve
that leaves most of the chip idle, so the two streams really run side by side; the library
calls release the GIL.

Rules that keep it simple and deterministic:
  * jobs are taken in submission order by the worker; a job the walk needs and that is still
    queued is run by the walk itself, on the walk's context;
  * a job's result does not depend on when, where or on which context it ran, so neither
    does the supertree.
"""

from __future__ import annotations

import collections
import threading

_QUEUED, _RUNNING, _DONE = 0, 1, 2


class Job:
    __slots__ = ("fn", "state", "value", "error", "event")

    def __init__(self, fn) -> None:
        self.fn = fn
        self.state = _QUEUED
        self.value = None
        self.error = None
        self.event = None  # created by the first waiter

    def _run(self, dev) -> None:
        try:
            self.value = self.fn(dev)
        except BaseException as exc:  # noqa: BLE001 -- handed to whoever asks for the result
            self.error = exc
        self.fn = None


class Ahead:
    """``make_device()`` is called once, on the worker thread, for the worker's own context;
    a job is a callable taking the device it runs on."""

    def __init__(self, make_device, workers: int = 1) -> None:
        self._make_device = make_device
        self._lock = threading.Lock()
        self._cv = threading.Condition(self._lock)
        self._queue: collections.deque[Job] = collections.deque()
        # (round 6) several workers, each with a context of its own: the level-synchronous recursion hands over
        # ALL larger nodes of a level at once -- latency-bound solves of a few hundred to a few thousand
        # vertices that run side by side on the chip.  A thread is started when a job finds none idle.
        self._max_workers = max(1, int(workers))
        self._threads: list[threading.Thread] = []
        self._idle = 0
        self._closed = False
        self.stats = {"submitted": 0, "by_worker": 0, "by_walk": 0}

    # ---- the walk's side --------------------------------------------------------------
    def submit(self, fn) -> Job:
        """Queue ``fn(device)``; returns the job to ask ``result`` of."""
        job = Job(fn)
        with self._cv:
            if self._closed:
                msg = "submit on a closed queue"
                raise RuntimeError(msg)
            self._queue.append(job)
            self.stats["submitted"] += 1
            if self._idle < len(self._queue) and len(self._threads) < self._max_workers:
                thread = threading.Thread(target=self._worker, name=f"scs-ahead-{len(self._threads)}", daemon=True)
                self._threads.append(thread)
                thread.start()
            self._cv.notify()
        return job

    def result(self, job: Job, own_device):
        """The job's value; a job nobody has started yet runs here, on ``own_device``."""
        steal = False
        with self._lock:
            if job.state == _QUEUED and self._closed:
                msg = "the queue was closed before this job ran"
                raise RuntimeError(msg)
            if job.state == _QUEUED:
                self._queue.remove(job)
                job.state = _RUNNING
                steal = True
            elif job.state == _RUNNING and job.event is None:
                job.event = threading.Event()
        if steal:
            job._run(own_device)
            job.state = _DONE
            self.stats["by_walk"] += 1
        elif job.state != _DONE:
            job.event.wait()
        if job.error is not None:
            raise job.error
        return job.value

    def close(self) -> None:
        with self._cv:
            self._closed = True
            self._queue.clear()
            self._cv.notify_all()
        for thread in self._threads:
            thread.join()
        self._threads = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- the worker ------------------------------------------------------------------
    def _worker(self) -> None:
        dev = None
        try:
            while True:
                with self._cv:
                    self._idle += 1
                    while not self._queue and not self._closed:
                        self._cv.wait()
                    self._idle -= 1
                    if self._closed:
                        return
                    job = self._queue.popleft()
                    job.state = _RUNNING
                if dev is None:
                    try:
                        dev = self._make_device()
                    except BaseException as exc:  # noqa: BLE001 -- the job's asker hears about it
                        job.error = exc
                if job.error is None:
                    job._run(dev)
                with self._lock:
                    job.state = _DONE
                    event = job.event
                    self.stats["by_worker"] += 1
                if event is not None:
                    event.set()
        finally:
            if dev is not None:
                dev.close()
