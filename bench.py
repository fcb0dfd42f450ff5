#!/usr/bin/env python3
"""Benchmark of the hot path: top-level PCG build + Fiedler solve (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A *step* is one pass of the hot path over one synthetic input whose flattened
tree tables are already resident in HBM: ``scs_pcg_build`` (W rows of this rank)
followed by ``scs_fiedler`` (degrees + LOBPCG).  N = 1 runs BASELINE.json
configs[2] (10 000 taxa / 500 trees / branch), the largest configuration
BASELINE.json assigns to a single MI355X; N > 1 runs configs[3] (50 000 taxa /
2 000 trees, row-partitioned W, RCCL all-gather of the Krylov block), one
process per GPU as launched by ``torch.distributed.run``.  torch is used only
for the host-side rendezvous (gloo: unique-id broadcast, barrier, max over
ranks); every number is produced by libscs_hip.so through its C-ABI.

Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate

WORKLOADS = {
    # name: (n_taxa, n_trees, strategy, random tree weights, BASELINE.json config index)
    "cfg1": (1000, 100, "depth", False, 1),
    "cfg2": (10000, 500, "branch", False, 2),
    "cfg3": (50000, 2000, "branch", False, 3),
    "cfg4": (100000, 5000, "branch", True, 4),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["custom"])
    ap.add_argument("--taxa", type=int, default=2000)
    ap.add_argument("--trees", type=int, default=50)
    ap.add_argument("--strategy", default="branch")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--tol", type=float, default=1e-13)
    ap.add_argument("--max-iter", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shared", action="store_true",
                    help="multi-rank runs: every rank evaluates all cells of its own rows "
                         "instead of sharing the upper-triangle tiles")
    ap.add_argument("--no-parity", action="store_true",
                    help="profiling runs: skip the oracle gates (never for a reported number)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the short reference passes over the other single-GPU configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU time of each leg of the bounded cpu_baseline sample")
    return ap.parse_args()


def even_splits(n: int, world: int) -> list[int]:
    """Contiguous row blocks, boundaries on multiples of 64 (the build's tile height)."""
    blocks = (n + 63) // 64
    out = [0]
    for r in range(1, world):
        out.append(min(n, (blocks * r // world) * 64))
    out.append(n)
    return out


def cpu_baseline(tables, graph, args, n, m):
    """Oracle timed on a bounded sample of the same workload (rank 0, N = 1 only)."""
    import threadpoolctl

    from oracle import scs_oracle as so
    from oracle import tables_oracle as to

    # leg 1: the C restatement of the reference's accumulation, one thread, a
    # prefix of the trees sized to ~cpu_seconds, scaled linearly to all trees
    t0 = time.perf_counter()
    w = np.zeros((n, n))
    _, upd = to.pcg_dense(tables, 0, 1, out=w)
    per_tree = max(time.perf_counter() - t0, 1e-4)
    sample_trees = int(max(1, min(m, args.cpu_seconds / per_tree)))
    w[:] = 0
    t0 = time.perf_counter()
    _, upd = to.pcg_dense(tables, 0, sample_trees, out=w)
    t_build_sample = time.perf_counter() - t0
    t_build = t_build_sample * m / sample_trees
    del w

    # leg 2: scikit-learn's spectral_embedding (the reference's eigen-solve) on a
    # leading principal block of the device-built W, scaled by (V / Vs)^3 (dense LU)
    vs = n
    if n > 3000:
        vs = 3000
    blockw = graph.download_rows(0, vs)[:, :vs].copy()
    t0 = time.perf_counter()
    so.spectral_maps(blockw, np.random.RandomState(0))
    t_eig_sample = time.perf_counter() - t0
    if vs < n and t_eig_sample < args.cpu_seconds / 4 and n >= 6000:
        vs = 6000
        blockw = graph.download_rows(0, vs)[:, :vs].copy()
        t0 = time.perf_counter()
        so.spectral_maps(blockw, np.random.RandomState(0))
        t_eig_sample = time.perf_counter() - t0
    t_eig = t_eig_sample * (n / vs) ** 3
    blas_threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    return {
        "value": round(t_build + t_eig, 3),
        "unit": "s",
        "cores": int(blas_threads),
        "kind": "port",
        "sample": (
            f"build: oracle/pcg_oracle.c on trees [0,{sample_trees}) of {m}, 1 thread, "
            f"{t_build_sample:.2f} s, scaled x{m / sample_trees:.1f} -> {t_build:.1f} s; "
            f"eig: sklearn.manifold.spectral_embedding (reference's ARPACK shift-invert path) on the "
            f"leading {vs}x{vs} block of W, {blas_threads} BLAS threads, {t_eig_sample:.2f} s, "
            f"scaled (V/Vs)^3 = x{(n / vs) ** 3:.1f} -> {t_eig:.1f} s; host has {os.cpu_count()} cores"
        ),
        "build_s": round(t_build, 3),
        "eig_s": round(t_eig, 3),
    }


def run_workload(name, args, dev, dist, rank, world, steps, warmup, full):
    """Time `steps` passes of build + solve on one named workload; returns the report."""
    from spectralclustersupertree_amd import synthetic

    if name == "custom":
        n, m, strategy, rw, cfg_idx = args.taxa, args.trees, args.strategy, False, -1
    else:
        n, m, strategy, rw, cfg_idx = WORKLOADS[name]

    t_gen0 = time.perf_counter()
    tables = synthetic.make_tables(args.seed, n, m, strategy, random_weights=rw)
    t_gen = time.perf_counter() - t_gen0
    splits = even_splits(n, world)
    rb, re_ = splits[rank], splits[rank + 1]

    t_up0 = time.perf_counter()
    dtab = dev.upload(tables)
    dev.synchronize()
    t_upload = time.perf_counter() - t_up0
    v0 = np.random.RandomState(args.seed).uniform(-1, 1, n)

    def barrier():
        dev.synchronize()
        if dist is not None:
            dist.barrier()

    def one_step(keep=False):
        graph = dtab.build(rb, re_, shared=(world > 1 and not args.no_shared))
        maps, stats = graph.fiedler(v0, tol=args.tol, max_iter=args.max_iter, block=args.block)
        bstats = graph.build_stats
        if keep:
            return graph, maps, stats, bstats
        graph.free()
        return None, maps, stats, bstats

    for _ in range(warmup):
        one_step()

    acc = {"apply_ms": 0.0, "n_apply": 0, "iters": 0, "build_ms": 0.0, "acc_ms": 0.0, "prep_ms": 0.0,
           "solve_ms": 0.0, "exch_ms": 0.0}
    barrier()
    t0 = time.perf_counter()
    last = None
    kept = None
    for i in range(steps):
        # multi-rank runs keep the last graph for the parity gate (no extra collective step)
        kept, maps, stats, bstats = one_step(keep=(world > 1 and i == steps - 1))
        acc["apply_ms"] += stats["apply_ms_total"]
        acc["n_apply"] += stats["n_apply"]
        acc["iters"] += stats["iterations"]
        acc["build_ms"] += bstats["total_ms"]
        acc["acc_ms"] += bstats["accumulate_ms"]
        acc["prep_ms"] += bstats["prep_ms"]
        acc["exch_ms"] += bstats["exchange_ms"]
        acc["solve_ms"] += stats["solve_ms"]
        last = (maps, stats, bstats)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    maps, stats, bstats = last
    steps = max(steps, 1)
    sec_per_step = elapsed / steps
    apply_avg_ms = acc["apply_ms"] / max(acc["n_apply"], 1)
    achieved = stats["apply_bytes"] / (apply_avg_ms * 1e-3) / 1e9 if apply_avg_ms > 0 else 0.0
    acc_ms = acc["acc_ms"] / steps
    cell_rate = bstats["cell_trees"] / (acc_ms * 1e-3) if acc_ms > 0 else 0.0
    build_bytes = bstats["bytes_w"] + bstats["bytes_tables"]
    build_gbs = build_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0

    report = {
        "value": round(sec_per_step, 6),
        "ms_per_step": round(sec_per_step * 1e3, 3),
        "steps": steps,
        "config": {
            "workload": (
                f"BASELINE.json configs[{cfg_idx}]: synthetic {n} taxa / {m} random-join rooted trees, "
                f"pcg_weighting='{strategy}'" + (", per-tree weights" if rw else "")
                if cfg_idx >= 0 else f"custom: {n} taxa / {m} trees / {strategy}"
            ),
            "n_taxa": n,
            "n_trees": m,
            "pcg_weighting": strategy,
            "seed": args.seed,
            "parallelism": "single device, symmetric tile schedule" if world == 1
            else f"W row-partitioned over {world} ranks ("
                 + ("upper-triangle tiles split round-robin, one RCCL all-gather of the packed tiles"
                    if bstats["symmetric"] == 2 else "every rank evaluates all cells of its rows")
                 + "), RCCL all-gather of the Krylov block per iteration",
            "lobpcg_block": stats["block"],
            "tol": args.tol,
        },
        "roofline": {
            "kernel": f"k_symm<{stats['block']}> (S*X: the kernel that streams the N x N matrix, "
                      "HBM-bound stage of the path)",
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "bytes_per_launch": stats["apply_bytes"],
            "avg_launch_ms": round(apply_avg_ms, 5),
            "launches_per_step": acc["n_apply"] / steps,
        },
        # the PCG accumulation is LDS/issue-bound by construction (0.5*V^2*M cell-tree
        # evaluations against 8*V^2 bytes written once): its HBM fraction is reported as is
        "roofline_build": {
            "kernel": "k_accumulate_mono / k_accumulate (PCG weights; largest single kernel by time)",
            "bound": "hbm",
            "achieved": round(build_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(build_gbs / HBM_PEAK_GBS, 5),
            "bytes_per_launch": build_bytes,
            "avg_launch_ms": round(acc_ms, 3),
            "cell_trees_per_launch": bstats["cell_trees"],
            "cell_trees_per_s": round(cell_rate, 0),
            "note": "not HBM-bound: limited by LDS gathers, VALU issue and L2 gather latency",
        },
        "stages": {
            "build_ms": round(acc["build_ms"] / steps, 3),
            "build_prep_ms": round(acc["prep_ms"] / steps, 3),
            "build_accumulate_ms": round(acc_ms, 3),
            "build_exchange_ms": round(acc["exch_ms"] / steps, 3),
            "fiedler_ms": round(acc["solve_ms"] / steps, 3),
            "fiedler_symm_ms": round(acc["apply_ms"] / steps, 3),
            "lobpcg_iterations": acc["iters"] / steps,
            "converged": stats["converged"],
            "lambda2": stats["lambda"][1],
            "lambda3": stats["lambda_next"],
            "residual": stats["resid"][1],
            "tables_upload_s": round(t_upload, 4),
            "tables_generate_s": round(t_gen, 3),
        },
    }
    # HBM traffic of k_symm from the committed PMC pass of this workload, if any
    try:
        pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get(name)
        if pmc and world == 1 and pmc["kernel"].startswith(f"k_symm<{stats['block']},"):
            report["roofline"]["traffic"] = pmc["traffic"]
            report["roofline"]["traffic_source"] = pmc["source"]
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0 and world == 1 and not args.no_parity:
        # parity gate at full size: rows of W against the oracle, bit for bit
        from oracle import tables_oracle as to

        graph, maps2, stats2, _ = one_step(keep=True)
        rows = np.unique(np.random.RandomState(1).randint(0, n, size=8)).astype(np.int32)
        want = to.pcg_rows(tables, rows)
        mismatch = 0
        for i, r in enumerate(rows):
            got = graph.download_rows(int(r), 1)[0]
            mismatch += int(np.count_nonzero(got != want[i]))
        report["parity"] = {
            "w_rows_checked": int(len(rows)),
            "w_cells_mismatched": mismatch,
            "fiedler_residual": stats2["resid"][1],
            "maps_repeatable": bool(np.array_equal(maps, maps2)),
        }
        if full and not args.no_cpu_baseline:
            report["cpu_baseline"] = cpu_baseline(tables, graph, args, n, m)
        graph.free()
    if world > 1:
        if rank == 0:
            from oracle import tables_oracle as to

            rows = (rb + np.unique(np.random.RandomState(1).randint(0, re_ - rb, size=4))).astype(np.int32)
            want = to.pcg_rows(tables, rows)
            mismatch = 0
            for i, r in enumerate(rows):
                mismatch += int(np.count_nonzero(kept.download_rows(int(r), 1)[0] != want[i]))
            report["parity"] = {
                "w_rows_checked": int(len(rows)),
                "w_cells_mismatched": mismatch,
                "fiedler_residual": stats["resid"][1],
            }
        kept.free()
    dtab.free()
    return report


def main() -> int:
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.gpus > 1 and world == 1:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run "
              f"--nproc-per-node {args.gpus}", file=sys.stderr)
        return 2

    dist = None
    if world > 1:
        import torch.distributed as dist  # host-side rendezvous only (gloo)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from spectralclustersupertree_amd.backend import Device

    uid = None
    if world > 1:
        import torch

        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = torch.frombuffer(bytearray(Device.unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(buf, 0)
        uid = bytes(buf.numpy().tobytes())
    # SCS_BENCH_DEVICE pins every rank to one device index (single-GPU rehearsal of the
    # multi-rank path); the driver's runs leave it unset: one GPU per local rank
    dev_index = int(os.environ.get("SCS_BENCH_DEVICE", local_rank))
    dev = Device(dev_index, rank, world, uid)

    name = args.workload or ("cfg2" if world == 1 else "cfg3")
    main_rep = run_workload(name, args, dev, dist, rank, world, args.steps, args.warmup, full=True)

    result = {
        "metric": "top-level PCG build + Fiedler solve wall-time (s) at N taxa, 1/2/4/8 MI355X",
        "value": main_rep["value"],
        "unit": "s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": main_rep["ms_per_step"],
        "higher_is_better": False,
        "scaling": "strong",
        "scaling_note": "N > 1 runs BASELINE.json configs[3]; its one-GPU time, the strong-scaling "
                        "baseline, is other_workloads.cfg3.value of the N = 1 line and "
                        "same_workload_on_one_gpu.value of every N > 1 line",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
    }
    for key in ("config", "roofline", "roofline_build", "stages", "parity", "cpu_baseline"):
        if key in main_rep:
            result[key] = main_rep[key]

    # the other single-GPU configurations of BASELINE.json, one short pass each, for
    # reference (configs[3] is the workload the N > 1 runs use: its N = 1 time is the
    # strong-scaling baseline)
    if world == 1 and args.workload is None and not args.no_extra:
        others = {}
        for extra, st in (("cfg1", 3), ("cfg3", 1)):
            try:
                rep = run_workload(extra, args, dev, dist, rank, world, st, 1,
                                   full=False)
                others[extra] = {k: rep[k] for k in ("value", "steps", "config", "roofline",
                                                     "roofline_build", "stages", "parity") if k in rep}
            except Exception as exc:  # noqa: BLE001 - report, never hide the main line
                others[extra] = {"error": str(exc)}
        result["other_workloads"] = others

    dev.close()
    # N > 1: rank 0 also times the SAME workload alone on its GPU (one warm-up + one pass,
    # single-rank context, the other ranks wait), so that every multi-GPU line carries the
    # one-device time it is to be compared with
    if world > 1 and rank == 0 and not args.no_extra:
        try:
            solo = Device(dev_index, 0, 1)
            try:
                rep = run_workload(name, args, solo, None, 0, 1, 1, 1, full=False)
                result["same_workload_on_one_gpu"] = {k: rep[k] for k in ("value", "stages", "roofline", "parity")
                                                      if k in rep}
            finally:
                solo.close()
        except Exception as exc:  # noqa: BLE001 - report, never hide the main line
            result["same_workload_on_one_gpu"] = {"error": str(exc)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
