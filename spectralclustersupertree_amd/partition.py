"""Row partition of the proper-cluster-graph matrix and the ranks that share it.

North star: the Laplacian is row-partitioned over the GPUs of one node, one
process per GPU, RCCL all-gather of the Krylov block -- but only at the top
recursion levels, where V justifies it; deeper sub-problems run on single
devices.  This module holds the host side of that: the split computation
(aligned to the build's 64-row tiles and, under contraction, to the groups --
SURVEY.md section 8f-1), the rendezvous (``hoststore``: a small TCP star, no torch), and the ``Team`` a recursion walks with.

The reference has no counterpart (single process, ``n_jobs=1``,
src/sc_supertree/scs.py:239).
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

TILE_ROWS = 64  # scs_pcg_build's tile height

# Below this many vertices a node is solved on one device: the per-iteration collective
# (tens of microseconds) and the tile exchange outweigh the streamed bytes it saves
# ("below ~N = 8-16k a single device is faster"; one MI355X takes 25 ms for
# the whole 10 000-taxa step).  SCS_SHARD_MIN_VERTICES overrides.
DEFAULT_SHARD_MIN_VERTICES = 16384


def shard_min_vertices() -> int:
    v = os.environ.get("SCS_SHARD_MIN_VERTICES")
    return int(v) if v else DEFAULT_SHARD_MIN_VERTICES


def row_splits(n: int, world: int, group_start=None) -> list[int]:
    """``world + 1`` increasing row indices, ``[0, ..., n]``: rank r owns rows
    ``[splits[r], splits[r+1])``.

    Without ``group_start`` the boundaries are the multiples of 64 nearest to an even split
    (whole tiles per rank).  With ``group_start`` (contraction: vertex g of the contracted
    graph is rows ``[group_start[g], group_start[g+1])``) every boundary is a group boundary
    -- ``scs_graph_contract`` needs whole groups per rank -- chosen nearest to the even split,
    a multiple of 64 when one is within reach.  Raises ``ValueError`` when there are fewer
    blocks (or groups) than ranks.
    """
    if world < 1 or n < world:
        msg = f"cannot split {n} rows over {world} ranks"
        raise ValueError(msg)
    if group_start is None:
        blocks = (n + TILE_ROWS - 1) // TILE_ROWS
        if blocks < world:
            msg = f"{n} rows are {blocks} tiles of {TILE_ROWS}: fewer than {world} ranks"
            raise ValueError(msg)
        out = [0]
        for r in range(1, world):
            out.append(min(n, (blocks * r // world) * TILE_ROWS))
        out.append(n)
        return out
    gs = np.asarray(group_start, dtype=np.int64)
    if gs[0] != 0 or gs[-1] != n or np.any(np.diff(gs) <= 0):
        msg = "group_start must increase from 0 to n"
        raise ValueError(msg)
    n_groups = len(gs) - 1
    if n_groups < world:
        msg = f"{n_groups} groups cannot be split over {world} ranks"
        raise ValueError(msg)
    out = [0]
    for r in range(1, world):
        ideal = n * r / world
        # group boundaries that leave at least one group to every remaining rank
        lo = int(np.searchsorted(gs, out[-1], side="right"))
        hi = n_groups - (world - r)
        cand = gs[lo:hi + 1]
        k = int(np.argmin(np.abs(cand - ideal)))
        best = int(cand[k])
        # prefer a boundary on a tile edge if one lies within a tile of the ideal split
        near = cand[(np.abs(cand - ideal) <= TILE_ROWS) & (cand % TILE_ROWS == 0)]
        if len(near):
            best = int(near[int(np.argmin(np.abs(near - ideal)))])
        out.append(best)
    out.append(n)
    return out


UPPER_ALIGN = 256  # SCS_BUILD_UPPER: a rank's first row is a multiple of the build's tile width


def row_splits_upper(n: int, world: int) -> list[int]:
    """Row splits for ``SCS_BUILD_UPPER`` jobs: rank r keeps the part of the upper triangle that
    lies in its rows, a trapezoid of (V - row) cells per row, so equal ROWS would give rank 0
    almost twice the average work and the last rank next to none.  The splits equalise the
    trapezoids' areas -- row s_r = V (1 - sqrt(1 - r / world)) -- rounded to multiples of 256,
    every rank at least one 256-row block.  Raises ``ValueError`` when V has fewer such blocks
    than ranks."""
    blocks = (n + UPPER_ALIGN - 1) // UPPER_ALIGN
    if world < 1 or blocks < world:
        msg = f"{n} rows are {blocks} blocks of {UPPER_ALIGN}: fewer than {world} ranks"
        raise ValueError(msg)
    out = [0]
    for r in range(1, world):
        ideal = n * (1.0 - (1.0 - r / world) ** 0.5)
        s = int(round(ideal / UPPER_ALIGN)) * UPPER_ALIGN
        s = max(s, out[-1] + UPPER_ALIGN)  # at least one block for the previous rank
        s = min(s, (blocks - (world - r)) * UPPER_ALIGN)  # and for every later one
        out.append(s)
    out.append(n)
    return out


def group_splits(splits: list[int], group_start) -> list[int]:
    """The same partition in group indices: rank r owns groups
    ``[gsplits[r], gsplits[r+1])`` of the contracted graph."""
    gs = np.asarray(group_start, dtype=np.int64)
    idx = np.searchsorted(gs, np.asarray(splits, dtype=np.int64))
    if not np.array_equal(gs[idx], np.asarray(splits, dtype=np.int64)):
        msg = "a split is not a group boundary"
        raise ValueError(msg)
    return [int(i) for i in idx]


@dataclass
class Team:
    """The ranks that walk one recursion together (SPMD: every rank makes the same calls with
    the same inputs and the same RandomState stream, so every rank takes the same decisions).

    ``device``   this rank's context in the job-wide communicator (collective build / solve),
    ``solo``     a single-rank context on the same GPU for the nodes below the sharding
                 threshold,
    ``allgather`` host-side exchange of small Python objects (child subtrees in "forked" mode).
    """

    rank: int = 0
    world: int = 1
    device: object = None
    solo: object = None
    allgather: object = None
    shard_min: int = field(default_factory=shard_min_vertices)
    # "shared": children are solved by every rank on its own device with the one shared
    # RandomState stream (the reference's stream, scs.py:164) -- identical results; the
    # device work of the larger nodes below the threshold is dealt over the ranks by the level
    # engine (``level_engine`` below), everything else is done by every rank.  "forked": sibling sub-problems below the threshold are
    # dealt to the ranks round-robin, each with a RandomState forked from the parent's
    # stream, and the subtrees are exchanged -- one sub-problem per device at a time.
    child_rng: str = "shared"
    # "shared" teams: subtrees go through levels.Engine on every rank and the larger nodes of a level are dealt
    # over the ranks (nodes of shard_min vertices and more: collectively), their embeddings exchanged through
    # ``allgather`` -- the single-device labels, the device work of the mid-size nodes spread.  False: every rank
    # walks every node by itself on the node-by-node path (rounds 2-5; kept for comparison).
    level_engine: bool = True

    def close(self) -> None:
        for d in (self.device, self.solo):
            if d is not None:
                d.close()
        self.device = self.solo = None


def rendezvous_host(rank: int, world: int, make_unique_id):
    """Rendezvous of the HOST side: returns ``(group, unique_id)`` -- a ``hoststore.HostGroup``
    (allgather / broadcast / barrier / max over a small TCP star; no torch: the launcher only
    provides RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and rank 0's 128-byte RCCL id on every
    rank.  ``make_unique_id`` is called on rank 0 only.  No GPU call is made here
    (tests/test_multirank_cpu.py drives exactly this with world size 2)."""
    if world <= 1:
        return None, None
    from spectralclustersupertree_amd.hoststore import HostGroup

    group = HostGroup(rank, world)
    uid = None
    if rank == 0:
        uid = bytes(make_unique_id())
        if len(uid) != 128:
            msg = "unique id must be 128 bytes"
            raise ValueError(msg)
    return group, group.broadcast(uid)


def rendezvous(rank: int, world: int, dev_index: int):
    """``(group, Device)``: the host-side group (None for a single rank) and the job-wide
    context of this rank (RCCL communicator when world > 1)."""
    from spectralclustersupertree_amd.backend import Device

    group, uid = rendezvous_host(rank, world, Device.unique_id)
    return group, Device(dev_index, rank, world, uid)


def team_from_env() -> Team | None:
    """A team from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK as set by
    ``torch.distributed.run``), or None for a plain single-process run.  ``SCS_DEVICE``
    pins the GPU index (default: LOCAL_RANK)."""
    from spectralclustersupertree_amd.backend import Device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    rank = int(os.environ.get("RANK", "0"))
    dev_index = int(os.environ.get("SCS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    group, dev = rendezvous(rank, world, dev_index)
    return Team(rank=rank, world=world, device=dev, solo=Device(dev_index), allgather=group.allgather,
                child_rng=os.environ.get("SCS_CHILD_RNG", "shared"))


class LocalTeams:
    """``world`` teams inside ONE process, one host thread each, on one GPU: the in-process
    communicator of libscs_hip (thread barrier + device copies instead of RCCL).  Exercises the
    sharded code path of ``construct_supertree`` where only one device is at hand (tests,
    rehearsals); a real job uses one process per GPU and ``team_from_env``."""

    def __init__(self, world: int, device: int = 0, shard_min: int | None = None,
                 child_rng: str = "shared") -> None:
        import ctypes as C
        import threading

        from spectralclustersupertree_amd import _native as nv
        from spectralclustersupertree_amd.backend import Device

        self.world = world
        self._lib = nv.load_library()
        self._group = C.c_void_p()
        nv.check(self._lib.scs_local_group_create(world, C.byref(self._group)))
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self.teams = []
        for r in range(world):
            self.teams.append(Team(
                rank=r, world=world, device=None, solo=Device(device),
                allgather=self._make_allgather(r), child_rng=child_rng,
                **({"shard_min": shard_min} if shard_min is not None else {})))
        self._device_index = device

    def _make_allgather(self, rank):
        def allgather(obj):
            self._slots[rank] = obj
            self._barrier.wait()
            out = list(self._slots)
            self._barrier.wait()
            return out

        return allgather

    def run(self, fn):
        """``fn(team)`` on one thread per rank (the job-wide contexts are created on their own
        threads, as the library requires); returns the list of results, re-raises failures."""
        import threading

        from spectralclustersupertree_amd.backend import Device

        out, err = [None] * self.world, [None] * self.world

        def worker(r):
            team = self.teams[r]
            try:
                team.device = Device(self._device_index, r, self.world, _local_group=self._group)
                out[r] = fn(team)
            except BaseException as e:  # noqa: BLE001
                err[r] = e
                self._barrier.abort()
            finally:
                if team.device is not None:
                    team.device.close()
                    team.device = None

        threads = [threading.Thread(target=worker, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out

    def close(self) -> None:
        for t in self.teams:
            if t.solo is not None:
                t.solo.close()
                t.solo = None
        if self._group:
            self._lib.scs_local_group_destroy(self._group)
            self._group = None
