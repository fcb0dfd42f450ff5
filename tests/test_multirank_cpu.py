"""World-size-2 tests (CPU, two processes) of the N > 1 host logic: the row-partition plan
(tile- and group-aligned), the product's own rendezvous (partition.rendezvous_host over
hoststore.HostGroup, as bench.py and construct_supertree use it) up to the GPU call, the
recursion walked by two ranks with sibling sub-problems dealt one per rank, and -- over a
gloo process group, the CPU stand-in for the RCCL all-gather -- the slice-gather
order the device path relies on (rank r contributes rows [split_r, split_{r+1})
of S @ X; concatenating the gathered slices in rank order must equal the full
product).  The arithmetic here is the oracle's numpy restatement -- the device
kernels themselves are covered by the -m gpu tests."""

import json
import os
import socket
import sys
import time
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    import bench
    from oracle import tables_oracle as to
    from spectralclustersupertree_amd import partition, synthetic

    # 1. the product's own rendezvous (partition.rendezvous_host: process group + broadcast of
    # rank 0's 128-byte id), everything up to -- not including -- the GPU context
    made = []

    def make_id():
        made.append(rank)
        return bytes(range(128))

    group, uid = partition.rendezvous_host(rank, world, make_id)
    assert uid == bytes(range(128)) and made == ([0] if rank == 0 else [])
    # the host group's own collectives (what bench.py and the teams use)
    assert group.allgather({"rank": rank, "x": np.arange(3) + rank})[1 - rank]["rank"] == 1 - rank
    assert group.max(1.0 + rank) == float(world)
    assert group.broadcast("from zero" if rank == 0 else None) == "from zero"
    group.barrier()
    # the gather-order arithmetic below runs over gloo (what RCCL does on the device)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 2. row-partitioned operator apply, slices gathered in rank order
    n, b = 333, 8
    tables = synthetic.make_tables(3, n, 12, "branch", leaves_per_tree=300)
    w, _ = to.pcg_dense(tables)
    s, _ = to.normalized_operator(w)
    x = np.random.RandomState(0).standard_normal((n, b))
    splits = bench.even_splits(n, world)
    lo, hi = splits[rank], splits[rank + 1]
    chunk = max(splits[r + 1] - splits[r] for r in range(world))
    send = torch.zeros(chunk * b, dtype=torch.float64)
    send[: (hi - lo) * b] = torch.from_numpy((s[lo:hi] @ x).ravel())
    recv = [torch.zeros(chunk * b, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(recv, send)
    y = np.concatenate([recv[r].numpy()[: (splits[r + 1] - splits[r]) * b] for r in range(world)]).reshape(n, b)
    assert np.array_equal(y, np.concatenate([s[splits[r] : splits[r + 1]] @ x for r in range(world)]))
    assert np.allclose(y, s @ x, rtol=0, atol=1e-13)
    # 3. max-over-ranks timing reduction
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    # 4. the recursion walked by two ranks, sibling sub-problems dealt one per rank ("forked"
    # streams) and the subtrees exchanged through the process group; the bipartition itself
    # is the CPU oracle's here (the device one is covered by the -m gpu tests)
    from reference_cases import DATA_DIR, FILE_CASES
    from test_treearrays import cpu_bipartition

    from spectralclustersupertree_amd import scs
    from spectralclustersupertree_amd.load import load_tree_arrays
    from spectralclustersupertree_amd.tree import load_tree

    allgather = group.allgather
    team = partition.Team(rank=rank, world=world, device=None, solo=None, allgather=allgather,
                          shard_min=10**9, child_rng="forked")
    name, src, exp, weighting = next(c for c in FILE_CASES if "supertriplets" in c[0])
    arrays = load_tree_arrays(DATA_DIR / src)
    got = scs._construct(arrays, weighting, True, np.random.RandomState(0), cpu_bipartition, team)
    assert got.sorted().same_shape(load_tree(DATA_DIR / exp).sorted())
    newicks = allgather(got.sorted().get_newick())
    assert newicks[0] == newicks[1]
    # 5. no seed given: rank 0 draws it for every rank (one stream, not one per rank)
    seen = {}

    def capture(arr, weighting_, contract, rs, *a, **kw):
        seen["state"] = rs.get_state()[1].tobytes()
        return scs.tip_names_to_tree(["a", "b"])

    real_construct = scs._construct
    scs._construct = capture
    try:
        scs.construct_supertree(arrays, pcg_weighting=weighting, team=team)
    finally:
        scs._construct = real_construct
    states = allgather(seen["state"])
    assert states[0] == states[1]
    # 6. a sub-problem that fails on ONE rank makes EVERY rank raise (nobody is left waiting in
    # the exchange)
    calls = {"n": 0}

    def failing(tables, random_state, *, contract_edges):
        calls["n"] += 1
        if rank == 1 and calls["n"] == 2:  # the first node of rank 1's own dealt sub-problem
            raise ArithmeticError("planted failure")
        return cpu_bipartition(tables, random_state, contract_edges=contract_edges)

    forest = synthetic.tree_arrays(3, 120, 8, 90)  # a top-level split with work for both ranks
    with pytest.raises(RuntimeError, match="planted failure"):
        scs._construct(forest, "branch", True, np.random.RandomState(0), failing, team)
    assert calls["n"] >= 2
    dist.barrier()
    dist.destroy_process_group()
    group.close()
    Path(out_dir, f"ok{rank}").write_text("ok")


def test_even_splits_tile_aligned_and_cover():
    sys.path.insert(0, str(ROOT))
    import bench

    for n in (64, 65, 333, 1000, 10000, 50000):
        for world in (1, 2, 4, 8):
            if (n + 63) // 64 < world:
                with pytest.raises(ValueError):
                    bench.even_splits(n, world)
                continue
            sp = bench.even_splits(n, world)
            assert sp[0] == 0 and sp[-1] == n and len(sp) == world + 1
            assert all(a < b for a, b in zip(sp, sp[1:]))
            assert all(x % 64 == 0 for x in sp[1:-1])


def test_group_aligned_splits():
    # contraction under row partitioning: every split is a group boundary,
    # every rank gets at least one group, splits stay near the even ones
    from spectralclustersupertree_amd.partition import group_splits, row_splits

    rs = np.random.RandomState(0)
    for n, world in ((700, 2), (700, 3), (5000, 8), (64, 4), (9, 3)):
        sizes = []
        while sum(sizes) < n:
            sizes.append(min(n - sum(sizes), int(rs.choice([1, 1, 1, 2, 3, 40]))))
        gs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        sp = row_splits(n, world, gs)
        assert sp[0] == 0 and sp[-1] == n and all(a < b for a, b in zip(sp, sp[1:]))
        assert set(sp) <= set(gs.tolist())
        gsp = group_splits(sp, gs)
        assert gsp[0] == 0 and gsp[-1] == len(gs) - 1 and all(a < b for a, b in zip(gsp, gsp[1:]))
        assert max(abs(sp[r] - n * r / world) for r in range(1, world)) <= 64 + 40
    with pytest.raises(ValueError):
        row_splits(10, 4, [0, 5, 10])
    with pytest.raises(ValueError):
        group_splits([0, 3, 10], [0, 5, 10])


def test_two_rank_gloo_rendezvous_and_gather(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _bench_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """bench.py's N > 1 control flow end to end (rendezvous, timed loop, max over ranks, parity
    gate, the one-GPU reference pass, the JSON line) with the device replaced by a stand-in that
    answers from the oracle: no GPU here, and every key the real library reports is present."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import contextlib
    import io
    import json

    from oracle import scs_oracle as so
    from oracle import tables_oracle as to
    from spectralclustersupertree_amd import backend

    class FakeGraph:
        def __init__(self, dev, tables, rb, re_, shared):
            self.w, _ = to.pcg_dense(tables)
            self.rb, self.re = rb, re_
            n, m = tables.n_taxa, tables.n_trees
            self.build_stats = {
                "n_taxa": n, "n_trees": m, "row_begin": rb, "row_end": re_, "symmetric": 2 if shared else 0,
                "n_tiles": 3, "n_batches": 1, "cell_trees": 1.0 * n * n * m, "prep_ms": 0.1,
                "accumulate_ms": 1.0, "exchange_ms": 0.2 if shared else 0.0, "total_ms": 1.4,
                "bytes_w": 8.0 * (re_ - rb) * n, "bytes_tables": 16.0 * tables.n_taxa * m,
                "exchange_bytes": 1e6 if shared else 0.0}

        def fiedler(self, v0, tol=1e-13, max_iter=2000, block=0):
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                maps = to.sign_flip_columns(so.spectral_maps(self.w, np.random.RandomState(0)))
            n = self.w.shape[0]
            return maps, {"n_vertices": n, "block": 4, "iterations": 7, "n_apply": 8, "converged": 1,
                          "used_constraint": 1, "lambda": [1.0, 0.5], "resid": [0.0, 1e-14], "lambda_next": 0.4,
                          "apply_ms_total": 0.8, "apply_ms_min": 0.1, "solve_ms": 2.0,
                          "apply_bytes": 8.0 * (self.re - self.rb) * n, "allgather_ms_total": 0.07,
                          "allgather_bytes": 8.0 * n * 4, "n_allgather": 7}

        def download_rows(self, first, count):
            return self.w[first:first + count].copy()

        def free(self):
            pass

    class FakeTables:
        def __init__(self, dev, tables):
            self.dev, self.tables = dev, tables

        def build(self, rb=0, re_=None, shared=None, upper=False):
            re_ = self.tables.n_taxa if re_ is None else re_
            assert not (shared and upper)
            assert not upper or rb % 256 == 0
            return FakeGraph(self.dev, self.tables, rb, re_, bool(shared))

        def free(self):
            pass

    class FakeDevice:
        SMALL_MAX_TAXA = 64

        def __init__(self, device=0, rank=0, world=1, unique_id=None, **kw):
            assert world == 1 or (unique_id is not None and len(unique_id) == 128)
            self.rank, self.world = rank, world

        @staticmethod
        def unique_id():
            return bytes(range(128))

        def upload(self, tables):
            return FakeTables(self, tables)

        def synchronize(self):
            pass

        def comm_info(self):
            return {"kind": "rccl", "world": self.world, "rank": self.rank, "reported_world": self.world,
                    "reported_rank": self.rank}

        def close(self):
            pass

    backend.Device = FakeDevice
    import bench

    # the upper-triangle job (the job keeps the upper triangle only) ...
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--workload", "custom",
                "--taxa", "600", "--trees", "4", "--strategy", "depth", "--no-extra", "--multi-rank-mode", "upper"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main()
    assert rc == 0
    if rank == 0:
        line = json.loads(buf.getvalue().strip().splitlines()[-1])
        assert "upper triangle" in line["config"]["parallelism"] and line["parity"]["w_cells_mismatched"] == 0
        assert line["stages"]["build_exchange_ms"] == 0
    # ... and the DEFAULT, the row-partitioned one with the tile exchange -- which also times the other
    # layout (other_modes.upper) and holds both embeddings against the one-GPU run of the same input
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--workload", "custom",
                "--taxa", "600", "--trees", "5", "--strategy", "depth"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.main()
    assert rc == 0
    if rank == 0:
        line = json.loads(buf.getvalue().strip().splitlines()[-1])
        assert line["n_gpus"] == world and line["steps"] == 2 and line["unit"] == "s"
        assert line["higher_is_better"] is False and line["scaling"] == "strong"
        for key in ("metric", "value", "ms_per_step", "config", "roofline", "roofline_other", "roofline_path",
                    "stages", "parity", "same_workload_on_one_gpu"):
            assert key in line, key
        assert line["parity"]["w_cells_mismatched"] == 0
        assert line["stages"]["build_exchange_ms"] > 0 and line["stages"]["build_exchange_bytes_received"] > 0
        assert "workload" in line["config"] and "parallelism" in line["config"]
        assert line["stages"]["multi_rank_mode"] == "shared" and "row-partitioned" in line["config"]["parallelism"]
        assert line["communicator"]["reported_world"] == world == line["communicator"]["launcher_world_size"]
        assert line["stages"]["allgather_bytes_received_per_rank"] > 0 and line["stages"]["allgathers_per_step"] == 7
        alt = line["other_modes"]["upper"]
        assert "upper triangle" in alt["config"]["parallelism"] and alt["parity"]["w_cells_mismatched"] == 0
        assert alt["stages"]["build_exchange_ms"] == 0
        for rep in (line, alt):  # (the fake solver returns one embedding whatever the layout)
            assert rep["parity"]["embedding_vs_one_gpu_max_abs"] == 0.0
    else:
        assert buf.getvalue().strip() == ""
    Path(out_dir, f"bench_ok{rank}").write_text("ok")


def test_bench_two_rank_control_flow(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "bench_ok0").exists() and (tmp_path / "bench_ok1").exists()


def test_bench_launches_its_own_ranks(tmp_path, capsys, monkeypatch):
    """`python bench.py --gpus N` without a launcher: bench.launch_ranks starts N fresh processes
    with the launcher's environment, relays rank 0's line and fails if any rank fails (the
    ranks here are a stand-in script: no GPU in this container; what a rank does with that
    environment is test_bench_two_rank_control_flow)."""
    import bench

    script = tmp_path / "rank.py"
    script.write_text(
        "import json, os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert 1024 < int(os.environ['MASTER_PORT']) < 65536 - 16\n"
        "open(os.path.join(sys.argv[1], f'seen{r}'), 'w').write(os.environ['MASTER_PORT'])\n"
        "if sys.argv[2] == 'fail' and r == 1: sys.exit(7)\n"
        "if sys.argv[2] == 'fail': time.sleep(60)\n"
        "print(json.dumps({'rank': r, 'n_gpus': w}))\n")
    rc = bench.launch_ranks(3, [sys.executable, str(script), str(tmp_path), "ok"])
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 0 and out == [json.dumps({"rank": 0, "n_gpus": 3})]
    ports = {(tmp_path / f"seen{r}").read_text() for r in range(3)}
    assert len(ports) == 1
    # a failing rank: its code comes back, the ranks still waiting are stopped
    t0 = time.perf_counter()
    rc = bench.launch_ranks(2, [sys.executable, str(script), str(tmp_path), "fail"])
    assert rc == 7 and time.perf_counter() - t0 < 30
    # main() takes that road when no launcher set WORLD_SIZE
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    called = {}
    monkeypatch.setattr(bench, "launch_ranks", lambda n, cmd=None: called.setdefault("n", n) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    assert bench.main() == 0 and called["n"] == 4


def test_subtrees_travel_in_flat_form_at_the_default_recursion_limit():
    # "forked" teams exchange subtrees; a linked tree pickles by recursion (one level per tree
    # level), the flat form does not: a 5 000-level caterpillar at a recursion limit of 1 000
    import pickle

    from spectralclustersupertree_amd.tree import TreeNode

    node = TreeNode("t0", None, 0.5)
    for i in range(1, 5000):
        node = TreeNode(None, [node, TreeNode(f"t{i}", None, 0.25 * i, None)], 1.0 / i, float(i))
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(1000)
    try:
        back = pickle.loads(pickle.dumps(node))
        again = TreeNode.from_flat(node.to_flat())
    finally:
        sys.setrecursionlimit(old)
    for other in (back, again):
        assert other.to_flat() == node.to_flat()
        assert other.get_newick(with_distances=True) == node.get_newick(with_distances=True)


def test_upper_triangle_splits_balance_the_trapezoids():
    from spectralclustersupertree_amd.partition import row_splits_upper

    for n, world in ((50000, 2), (50000, 4), (50000, 8), (100000, 8), (1300, 2), (2100, 4), (1024, 4)):
        sp = row_splits_upper(n, world)
        assert sp[0] == 0 and sp[-1] == n and len(sp) == world + 1
        assert all(a < b for a, b in zip(sp, sp[1:])) and all(x % 256 == 0 for x in sp[:-1])
        area = [(b - a) * (n - (a + b) / 2) for a, b in zip(sp, sp[1:])]
        if n >= 50000:  # fine-grained enough to balance: every rank within 5 % of the mean
            assert max(area) <= 1.05 * sum(area) / world
    with pytest.raises(ValueError):
        row_splits_upper(700, 4)


def test_host_group_three_ranks_skips_a_stranger_on_the_first_port():
    # hoststore.HostGroup: rank 0 binds the first free port of the probe list, the others find
    # it by handshake -- a foreign listener on the first port is passed over
    import threading

    from spectralclustersupertree_amd.hoststore import HostGroup

    port = _free_port()
    stranger = socket.socket()
    stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        stranger.bind(("127.0.0.1", port))
        stranger.listen(4)
    except OSError:
        pytest.skip("port taken meanwhile")
    out, err = [None] * 3, [None] * 3

    def worker(r):
        try:
            g = HostGroup(r, 3, addr="127.0.0.1", port=port, timeout=30.0, tag="t")
            got = g.allgather(("rank", r))
            top = g.max(10.0 * r)
            word = g.broadcast("hello" if r == 0 else None)
            g.barrier()
            g.close()
            out[r] = (got, top, word)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    stranger.close()
    assert err == [None, None, None], err
    for got, top, word in out:
        assert got == [("rank", 0), ("rank", 1), ("rank", 2)] and top == 20.0 and word == "hello"


def test_rendezvous_unpickles_plain_data_only():
    # the sockets are unauthenticated: a message that names a global outside the whitelist
    # (here os.system through __reduce__) is refused, not executed; what the ranks really send
    # -- None, numbers, bytes, flat trees, numpy values -- round-trips
    import os
    import pickle

    from spectralclustersupertree_amd import hoststore as hs

    for obj in (None, 1.5, 3, "a", b"\x00" * 128, ({4: ([-1, 0, 0], ["r", "a", "b"], [None, 1.0, 2.5], [None, None, None])}, None),
                np.float64(2.0)):
        assert hs._loads(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)) == obj
    arr = np.arange(7, dtype=np.int32)
    assert np.array_equal(hs._loads(pickle.dumps(arr, protocol=pickle.HIGHEST_PROTOCOL)), arr)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    with pytest.raises(pickle.UnpicklingError, match="refused to unpickle"):
        hs._loads(pickle.dumps(Evil()))
