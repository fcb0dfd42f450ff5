"""Several ranks walking ONE level engine (levels.Engine under a team's shared stream): the deal of a
level's larger nodes over the ranks and the exchange of their embeddings -- host logic, no GPU.

The deal must be a pure function of what every rank holds (the ranks never talk about it), the exchange must
hand every rank every embedding bit for bit, and a failure on one rank must raise on all of them instead of
leaving the others waiting.  (The device side -- a recursion walked by 2 and 3 in-process ranks against the
single-device result -- is tests/test_gpu_team.py::test_level_engine_under_a_team_*.)
"""

import socket
import threading

import numpy as np
import pytest

from spectralclustersupertree_amd import levels
from spectralclustersupertree_amd.partition import Team, row_splits


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _level(sizes, trees=100):
    k = len(sizes)
    n_pres = np.asarray(sizes, dtype=np.int64)
    return np.arange(k), n_pres, n_pres.astype(np.int32), np.full(k, trees, dtype=np.int64)


def test_deal_is_a_pure_function_and_balances_the_tile_work():
    large, n_pres, n_groups, m = _level([900, 130, 700, 700, 300, 5000, 129, 2500])
    a = levels.deal(large, n_pres, n_groups, m, {}, 3, 1 << 30)
    b = levels.deal(large[::-1], n_pres, n_groups, m, {}, 3, 1 << 30)
    assert a == b  # whatever order the caller lists the nodes in
    collective, owner = a
    assert collective == [] and sorted(owner) == list(range(8))
    # longest job first onto the least loaded rank: the largest node alone, the rest shared out behind it
    assert owner[5] == 0 and owner[7] == 1 and owner[0] == 2
    load = [sum(float(n_groups[k]) ** 2 * 100 for k in owner if owner[k] == r) for r in range(3)]
    assert max(load) == load[0] == 5000.0 ** 2 * 100  # nothing was put on top of the longest job
    assert owner == {5: 0, 7: 1, 0: 2, 2: 2, 3: 2, 4: 2, 1: 2, 6: 2}  # (everything else fits beside 2 500 squared)
    # equal loads: ties go to the lower rank, equal sizes to the lower node -- no dependence on dict or sort order
    large, n_pres, n_groups, m = _level([400, 400, 400, 400])
    assert levels.deal(large, n_pres, n_groups, m, {}, 2, 1 << 30)[1] == {0: 0, 1: 1, 2: 0, 3: 1}


def test_deal_sends_nodes_above_the_threshold_to_all_ranks_in_one_order():
    large, n_pres, n_groups, m = _level([3000, 200, 2600, 1500, 2600])
    collective, owner = levels.deal(large, n_pres, n_groups, m, {}, 2, 2000)
    assert [k for k, _ in collective] == [0, 2, 4]  # largest first, ties by node
    for k, splits in collective:
        assert splits == row_splits(int(n_pres[k]), 2, None)
        assert splits[0] == 0 and splits[-1] == n_pres[k] and len(splits) == 3
    assert sorted(owner) == [1, 3]
    # a contracted node: the threshold looks at its VERTICES, the splits are group-aligned rows of its taxa
    group_start = np.arange(0, 3001, 2, dtype=np.int32)  # 1 500 groups of two taxa
    n_groups2 = n_groups.copy()
    n_groups2[0] = 1500
    collective, owner = levels.deal(large, n_pres, n_groups2, m, {0: group_start}, 2, 2000)
    assert [k for k, _ in collective] == [2, 4] and 0 in owner
    n_groups2[0] = 2400
    gs = np.concatenate((np.arange(0, 1200, 2), np.arange(1200, 3001))).astype(np.int32)
    collective, _ = levels.deal(large, n_pres, n_groups2, m, {0: gs}, 2, 2000)
    splits = dict(collective)[0]
    assert all(s in set(gs.tolist()) for s in splits)  # every split is a group boundary


def test_deal_falls_back_to_one_owner_when_a_node_has_fewer_row_blocks_than_ranks():
    large, n_pres, n_groups, m = _level([300, 260])
    # 300 rows are two 256-row blocks: eight ranks cannot share them -- the node is dealt like a smaller one
    with pytest.raises(ValueError):
        row_splits(300, 8, None)
    collective, owner = levels.deal(large, n_pres, n_groups, m, {}, 8, 250)
    assert collective == [] and owner == {0: 0, 1: 1}


class _Engine(levels.Engine):
    def __init__(self, team) -> None:  # (only what _spread / _exchange touch)
        self.team = team


def _run_ranks(world, fn):
    """``fn(team)`` on one thread per rank over a real hoststore.HostGroup (sockets, pickling)."""
    from spectralclustersupertree_amd.hoststore import HostGroup

    port = _free_port()
    out, err = [None] * world, [None] * world

    def worker(r):
        g = None
        try:
            g = HostGroup(r, world, addr="127.0.0.1", port=port, timeout=30.0, tag="levels")
            out[r] = fn(Team(rank=r, world=world, allgather=g.allgather))
        except BaseException as e:  # noqa: BLE001
            err[r] = e
        finally:
            if g is not None:
                g.close()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    return out, err


def test_exchange_hands_every_rank_every_embedding_bit_for_bit():
    rs = np.random.RandomState(5)
    maps = {k: rs.standard_normal((n, 2)) for k, n in ((3, 140), (4, 999), (7, 131), (9, 2048))}
    owner = {3: 0, 4: 1, 7: 2, 9: 1}
    coll = rs.standard_normal((5000, 2))  # a collective node: every rank holds it already

    def rank(team):
        mine = [k for k in sorted(owner) if owner[k] == team.rank]
        got = {k: maps[k].copy() for k in mine}
        got[1] = coll
        e = _Engine(team)
        assert e._spread()
        return e._exchange(got, None, [(1, [0, 2500, 5000])], mine)

    levels.reset_stats()
    out, err = _run_ranks(3, rank)
    assert err == [None] * 3, err
    for got in out:
        assert sorted(got) == [1, 3, 4, 7, 9]
        assert got[1] is coll
        for k in maps:
            assert got[k].dtype == np.float64 and np.array_equal(got[k], maps[k])
    assert levels.stats["team_dealt"] == 4 and levels.stats["team_received"] == 8  # (summed over the three ranks)


def test_a_failure_on_one_rank_raises_on_every_rank():
    def rank(team):
        e = _Engine(team)
        if team.rank == 1:
            return e._exchange({}, "rank 1: ScsError: out of device memory", [], [5])
        return e._exchange({2 + team.rank: np.zeros((130, 2))}, None, [], [2 + team.rank])

    out, err = _run_ranks(3, rank)
    assert out == [None] * 3
    for e in err:
        assert isinstance(e, RuntimeError) and "rank 1: ScsError: out of device memory" in str(e)


def test_ranks_out_of_step_are_told_apart_from_an_exchange():
    def rank(team):
        if team.rank == 0:
            return team.allgather(({0: [-1, 0, 0]}, None))  # (a "forked" team's subtree exchange, say)
        return _Engine(team)._exchange({1: np.zeros((130, 2))}, None, [], [1])

    out, err = _run_ranks(2, rank)
    assert isinstance(err[1], RuntimeError) and "out of step" in str(err[1])


def test_a_single_rank_or_no_team_never_exchanges():
    assert not _Engine(None)._spread()
    assert not _Engine(Team(rank=0, world=1, allgather=lambda x: [x]))._spread()
