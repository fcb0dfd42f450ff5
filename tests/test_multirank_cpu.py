"""World-size-2 gloo tests (CPU) of the N > 1 host logic: the row-partition plan,
the unique-id/barrier/max rendezvous bench.py performs, and the slice-gather
order the device path relies on (rank r contributes rows [split_r, split_{r+1})
of S @ X; concatenating the gathered slices in rank order must equal the full
product).  The arithmetic here is the oracle's numpy restatement -- the device
kernels themselves are covered by the -m gpu tests."""

import os
import socket
import sys
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    import bench
    from oracle import tables_oracle as to
    from spectralclustersupertree_amd import synthetic

    dist.init_process_group("gloo", rank=rank, world_size=world)
    # 1. unique-id style broadcast (128 bytes from rank 0)
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.arange(128, dtype=torch.uint8)
    dist.broadcast(buf, 0)
    assert buf.tolist() == list(range(128))
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
    dist.barrier()
    dist.destroy_process_group()
    Path(out_dir, f"ok{rank}").write_text("ok")


def test_even_splits_tile_aligned_and_cover():
    sys.path.insert(0, str(ROOT))
    import bench

    for n in (64, 65, 333, 1000, 10000, 50000):
        for world in (1, 2, 4, 8):
            sp = bench.even_splits(n, world)
            assert sp[0] == 0 and sp[-1] == n and len(sp) == world + 1
            assert all(a <= b for a, b in zip(sp, sp[1:]))
            assert all(x % 64 == 0 for x in sp[1:-1])
            if n >= 64 * world:
                assert all(a < b for a, b in zip(sp, sp[1:]))


def test_two_rank_gloo_rendezvous_and_gather(tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
