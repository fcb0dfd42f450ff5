#!/usr/bin/env python3
"""Benchmark of the hot path: top-level PCG build + Fiedler solve (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A *step* is one pass of the hot path over one synthetic input by the protocol of
SURVEY.md 8d / BASELINE.md 3.3: the flattened tree tables go host -> HBM
(``scs_tables_upload`` from page-locked memory), ``scs_pcg_build`` (W rows of this
rank), ``scs_fiedler`` (degrees + LOBPCG) and the V x 2 embedding comes back to the
host.  The same pass on tables already resident in HBM is reported beside it as
``value_tables_resident``.  N = 1 runs BASELINE.json
configs[2] (10 000 taxa / 500 trees / branch), the largest configuration
BASELINE.json assigns to a single MI355X; N > 1 runs configs[3] (50 000 taxa /
2 000 trees, row-partitioned W, RCCL all-gather of the Krylov block), one
process per GPU -- launched by ``torch.distributed.run`` or, when ``--gpus N`` is given
to a plain ``python bench.py``, by this script itself (``launch_ranks``: N fresh child
processes before anything touches the GPU) -- launchers only: the
host-side rendezvous (unique-id broadcast, barrier, max over ranks) is the
package's own TCP star (hoststore.py), no torch import anywhere; every number
is produced by libscs_hip.so through its C-ABI.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline            the kernel with the most device time per step (HIP-event-timed on the
                      library's stream inside the timed region), `roofline_other` the other one
                      of the two that matter (k_accumulate_mono, k_symm), `roofline_path` the
                      path-level figure (B_A + B_C) / t of SURVEY.md 8d;
  value_tables_resident  the same step without the tables' upload (`value` includes it);
  parity              the gates of SURVEY.md 8d at FULL size: rows of W vs the C oracle, the
                      embedding vs scikit-learn on both scales, labels;
  cpu_baseline        the CPU path timed on this box: all-core C restatement of the build +
                      scikit-learn's eigen-solve on the full matrix, and a bounded sample of
                      the reference's literal dict-based build (oracle/scs_oracle.build_pcg);
  seeds / planted     seeds 0, 1, 2 (median) and the planted input, with lambda2 / lambda3.
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

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy rate); fp64 VALU non-FMA
# issue 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz; LDS ~150 TB/s aggregate for ds_read_b64
HBM_PEAK_GBS = 8000.0
F64_VALU_TOPS = 39.3
LDS_PEAK_TBS = 150.0

WORKLOADS = {
    # name: (n_taxa, n_trees, strategy, random tree weights, BASELINE.json config index)
    "cfg1": (1000, 100, "depth", False, 1),
    "cfg2": (10000, 500, "branch", False, 2),
    "cfg3": (50000, 2000, "branch", False, 3),
    "cfg4": (100000, 5000, "branch", True, 4),
    # partial-coverage forests (not BASELINE.json configs; the reference's own fixtures are of this kind --
    # scs.py:644-658 does one update per pair of leaves OF A TREE): a sixth field = leaves per tree
    "sparse10": (10000, 500, "branch", False, -2, 1000),   # every tree holds 10 % of the taxa: the dense walk
    "sparse05": (20000, 2000, "branch", False, -3, 100),   # 0.5 %: every tile walks its own list of trees
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
    ap.add_argument("--planted", action="store_true", help="planted input (model tree + SPR moves)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--tol", type=float, default=1e-13)
    ap.add_argument("--max-iter", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi-rank-mode", default="shared", choices=["upper", "shared", "rows"],
                    help="N > 1, the headline layout: 'shared' (default: north_star's row-partitioned W, upper "
                         "tiles computed once and exchanged, RCCL all-gather of the Krylov block per iteration); "
                         "'upper' = the job keeps only the upper triangle of W (no exchange, the solve adds the "
                         "ranks' partial products); 'rows' = every rank evaluates all cells of its rows.  The "
                         "other one of shared / upper is timed too and printed under other_modes")
    ap.add_argument("--no-parity", action="store_true",
                    help="profiling runs: skip the oracle gates (never for a reported number)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip seeds 1/2, the planted input and the other single-GPU configs")
    ap.add_argument("--cpu-seconds", type=float, default=8.0,
                    help="target CPU time of the one-thread leg of cpu_baseline")
    ap.add_argument("--quick-reference-leg", action="store_true",
                    help="cpu_baseline.reference_style_build on 300 taxa / 30 trees (seconds) instead of the "
                         "full configs[1] (a minute or two)")
    return ap.parse_args()


def even_splits(n: int, world: int, upper: bool = False) -> list[int]:
    """Contiguous row blocks: boundaries on multiples of 64 (the build's tile height), or --
    upper-triangle jobs -- on multiples of 256 with equal trapezoid areas."""
    from spectralclustersupertree_amd.partition import row_splits, row_splits_upper

    return row_splits_upper(n, world) if upper else row_splits(n, world)


def make_input(name, args, seed, planted):
    """Synthetic tables in page-locked host memory (the upload inside a step is then a DMA)."""
    from spectralclustersupertree_amd import synthetic

    lpt = None
    if name == "custom":
        n, m, strategy, rw, cfg_idx = args.taxa, args.trees, args.strategy, False, -1
    else:
        n, m, strategy, rw, cfg_idx, *rest = WORKLOADS[name]
        lpt = rest[0] if rest else None
    t0 = time.perf_counter()
    spr = int(np.ceil(0.02 * n)) if planted else None
    try:
        tables = synthetic.make_tables(seed, n, m, strategy, leaves_per_tree=lpt, random_weights=rw, planted_spr=spr,
                                       pinned=True)
        tables.pinned = True
    except (RuntimeError, OSError):
        # no HIP device to pin memory on (the CPU-only control-flow tests): pageable arrays
        tables = synthetic.make_tables(seed, n, m, strategy, leaves_per_tree=lpt, random_weights=rw, planted_spr=spr)
        tables.pinned = False
    tables.leaves_per_tree = lpt
    return tables, (n, m, strategy, rw, cfg_idx), time.perf_counter() - t0


def sklearn_gates(w, maps, labels_dev):
    """scikit-learn on the device-built W exactly as SpectralClustering.fit goes about it: the
    embedding (draws the ARPACK start vector first), then k_means on the same stream.
    Returns (gates, seconds of the eigen-solve, BLAS threads)."""
    import threadpoolctl
    from sklearn.cluster import k_means

    from oracle import scs_oracle as so
    from oracle import tables_oracle as to

    rs = np.random.RandomState(0)
    t0 = time.perf_counter()
    ref = so.spectral_maps(w, rs)
    t_eig = time.perf_counter() - t0
    ref = to.sign_flip_columns(ref)
    _, labels_ref, _ = k_means(ref, 2, random_state=rs, n_init=10, verbose=False)
    _, dd = to.normalized_operator(w)
    n = len(labels_ref)
    mism = int(np.count_nonzero(labels_dev != labels_ref))
    blas = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    return {
        "maps_vs_sklearn_max_abs": float(np.max(np.abs(maps[:, 1] - ref[:, 1]))),
        "unit_vec_max_abs": float(np.max(np.abs((maps[:, 1] - ref[:, 1]) * dd))),
        "col0_max_abs": float(np.max(np.abs(maps[:, 0] - ref[:, 0]))),
        "labels_mismatched": min(mism, n - mism),  # label names are arbitrary
    }, t_eig, int(blas)


def reference_style_leg(n, m, strategy, sample_taxa=1000, sample_trees=100, sample_strategy="depth"):
    """The reference's LITERAL loop structure -- dicts keyed by tuples of names, one Python
    update per leaf pair (scs.py:569-658, restated in oracle/scs_oracle.build_pcg) -- TIMED IN FULL
    at BASELINE.json configs[1] (1 000 taxa / 100 trees / depth: SURVEY.md 8d "timed only where
    feasible (config 1 ...)", a minute or two of one CPython thread), with its cost per pair update
    and what that extrapolates to at the workload's 0.336 N^2 M pair updates."""
    from oracle import scs_oracle as so
    from spectralclustersupertree_amd import synthetic

    trees = synthetic.tree_objects(0, sample_taxa, sample_trees)
    names = sorted(so._all_tips(trees))
    t0 = time.perf_counter()
    _, weight, _, together = so.build_pcg({(x,) for x in names}, trees, [1.0] * sample_trees, sample_strategy)
    secs = time.perf_counter() - t0
    updates = int(sum(together.values()))
    per = secs / max(updates, 1)
    return {"kind": "reference-style (oracle/scs_oracle.build_pcg: the reference's dict-of-tuples loops)",
            "sample": (f"BASELINE.json configs[1] in full: {sample_taxa} taxa / {sample_trees} trees / "
                       f"{sample_strategy}, one thread (CPython), measured -- not a sample of it"),
            "seconds": round(secs, 3), "pair_updates": updates, "us_per_pair_update": round(per * 1e6, 3),
            "extrapolated_s_at_workload": round(per * 0.336 * n * n * m, 0),
            "extrapolation": "us_per_pair_update x 0.336 N^2 M pair updates of THIS workload "
                             f"({n} taxa / {m} trees / {strategy})"}


def cpu_build_legs(tables, args, n, m):
    """The C restatement of the reference's accumulation (oracle/pcg_oracle.c) on this box:
    every tree on all host cores, and a bounded prefix of the trees on one thread."""
    from oracle import tables_oracle as to

    cores = os.cpu_count() or 1
    w = np.zeros((n, n))
    t0 = time.perf_counter()
    to.pcg_dense_mt(tables, cores, out=w)
    t_all = time.perf_counter() - t0
    w[:] = 0
    t0 = time.perf_counter()
    to.pcg_dense(tables, 0, 1, out=w)
    per_tree = max(time.perf_counter() - t0, 1e-4)
    sample = int(max(1, min(m, args.cpu_seconds / per_tree)))
    w[:] = 0
    t0 = time.perf_counter()
    to.pcg_dense(tables, 0, sample, out=w)
    t_one_sample = time.perf_counter() - t0
    return {"cores": cores, "all_core_s": t_all, "one_thread_sample_s": t_one_sample,
            "one_thread_sample_trees": sample, "one_thread_scaled_s": t_one_sample * m / sample}


def run_workload(name, args, dev, dist, rank, world, steps, warmup, full, seed=None, planted=False, mode=None):
    """Time `steps` passes of build + solve on one named workload; returns the report (`_maps`: the
    last step's embedding, for the callers that compare layouts with each other)."""
    seed = args.seed if seed is None else seed
    tables, (n, m, strategy, rw, cfg_idx), t_gen = make_input(name, args, seed, planted)
    mode = (mode or args.multi_rank_mode) if world > 1 else "single"
    splits = even_splits(n, world, upper=(mode == "upper"))
    rb, re_ = splits[rank], splits[rank + 1]

    t_up0 = time.perf_counter()
    dtab = dev.upload(tables)
    dev.synchronize()
    t_upload_first = time.perf_counter() - t_up0
    v0 = np.random.RandomState(seed).uniform(-1, 1, n)

    def barrier():
        dev.synchronize()
        if dist is not None:
            dist.barrier()

    def one_step(keep=False, tab=None):
        """build + solve on resident tables (`tab`, default the set uploaded above)"""
        graph = (tab or dtab).build(rb, re_, shared=(mode == "shared"), upper=(mode == "upper"))
        maps, stats = graph.fiedler(v0, tol=args.tol, max_iter=args.max_iter, block=args.block)
        bstats = graph.build_stats
        if keep:
            return graph, maps, stats, bstats
        graph.free()
        return None, maps, stats, bstats

    def protocol_step(keep=False):
        """tables host -> HBM, build, solve, embedding to the host"""
        tab = dev.upload(tables)
        try:
            return one_step(keep=keep, tab=tab)
        finally:
            tab.free()

    for _ in range(warmup):
        protocol_step()

    acc = {"apply_ms": 0.0, "n_apply": 0, "iters": 0, "build_ms": 0.0, "acc_ms": 0.0, "prep_ms": 0.0,
           "solve_ms": 0.0, "exch_ms": 0.0, "spec_ms": 0.0, "ag_ms": 0.0, "n_ag": 0, "apply32_ms": 0.0, "n_apply32": 0}
    barrier()
    t0 = time.perf_counter()
    last = None
    kept = None
    for i in range(steps):
        # multi-rank runs keep the last graph for the parity gate (no extra collective step)
        kept, maps, stats, bstats = protocol_step(keep=(world > 1 and i == steps - 1))
        acc["apply_ms"] += stats["apply_ms_total"]
        acc["n_apply"] += stats["n_apply"]
        acc["apply32_ms"] += stats.get("apply32_ms_total", 0.0)
        acc["n_apply32"] += stats.get("n_apply32", 0)
        acc["iters"] += stats["iterations"]
        acc["build_ms"] += bstats["total_ms"]
        acc["acc_ms"] += bstats["accumulate_ms"]
        acc["prep_ms"] += bstats["prep_ms"]
        acc["exch_ms"] += bstats["exchange_ms"]
        acc["solve_ms"] += stats["solve_ms"]
        acc["spec_ms"] += bstats.get("spec_ms", 0.0)
        acc["ag_ms"] += stats.get("allgather_ms_total", 0.0)
        acc["n_ag"] += stats.get("n_allgather", 0)
        last = (maps, stats, bstats)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = dist.max(elapsed)

    maps, stats, bstats = last
    steps = max(steps, 1)
    sec_per_step = elapsed / steps
    # (mixed-precision loop: the launches that streamed the single-precision image of W are timed and
    # priced on their own -- half the bytes; apply_ms_total covers the double-precision launches)
    n_apply64 = acc["n_apply"] - acc["n_apply32"]
    apply_avg_ms = acc["apply_ms"] / max(n_apply64, 1)
    apply32_avg_ms = acc["apply32_ms"] / max(acc["n_apply32"], 1)
    apply32_bytes = stats.get("apply32_bytes", 0.0)
    symm_gbs = stats["apply_bytes"] / (apply_avg_ms * 1e-3) / 1e9 if apply_avg_ms > 0 else 0.0
    acc_ms = acc["acc_ms"] / steps
    n_batches = max(int(bstats["n_batches"]), 1)
    cell_rate = bstats["cell_trees"] / (acc_ms * 1e-3) if acc_ms > 0 else 0.0
    build_bytes = bstats["bytes_w"] + bstats["bytes_tables"]
    build_gbs = build_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    symm_ms_step = (acc["apply_ms"] + acc["apply32_ms"]) / steps
    n_apply_step = n_apply64 / steps
    n_apply32_step = acc["n_apply32"] / steps
    # B_C = n_apply (W bytes of one apply + 16 V b) + n_iter 72 V b with the W
    # bytes the SELECTED kernel must stream -- 4 V^2 (upper tiles) when the symmetric SYMM ran,
    # 8 rows V otherwise: stats["apply_bytes"] is exactly that; B_A = bytes_w + tables
    symm_tri = mode == "upper" or stats["apply_bytes"] < 6.0 * (re_ - rb) * n
    b_c = (n_apply_step * stats["apply_bytes"] + n_apply32_step * apply32_bytes +
           (acc["iters"] / steps) * 72.0 * n * stats["block"])
    path_gbs = (build_bytes + b_c) / sec_per_step / 1e9

    roof_symm = {
        "kernel": (f"k_symm_tri<{stats['block']}> (S*X from the upper-triangle tiles of the symmetric "
                   "N x N matrix: 4 V^2 bytes per LOBPCG iteration)" if symm_tri else
                   f"k_symm<{stats['block']}> (S*X: streams this rank's rows of the N x N matrix once per "
                   "LOBPCG iteration, 8 rows V bytes)"),
        "bound": "hbm",
        "achieved": round(symm_gbs, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(symm_gbs / HBM_PEAK_GBS, 4),
        "traffic": None,
        "bytes_per_launch": stats["apply_bytes"],
        "avg_launch_ms": round(apply_avg_ms, 5),
        "launches_per_step": n_apply_step,
        "device_ms_per_step": round(acc["apply_ms"] / steps, 3),
        # (what the event pair itself adds on this stream, measured by an empty pair at the end of the solve:
        # a profiler's kernel time is about that much below avg_launch_ms; `achieved` uses the raw figure)
        "event_pair_ms": round(stats.get("event_pair_ms", 0.0), 5),
    }
    if acc["n_apply32"]:
        # the loop's launches stream the single-precision image; whichever kind takes more device time per step
        # is the entry, the other one rides inside it
        g32 = apply32_bytes / (apply32_avg_ms * 1e-3) / 1e9 if apply32_avg_ms > 0 else 0.0
        roof_img = {
            "kernel": (f"k_symm_tri_tf<{stats['block']}, float> (S*R from the single-precision image of the upper-triangle "
                       "tiles: 2 V^2 bytes per LOBPCG iteration; products and sums in double precision)"),
            "bound": "hbm", "achieved": round(g32, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(g32 / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": apply32_bytes, "avg_launch_ms": round(apply32_avg_ms, 5),
            "launches_per_step": n_apply32_step, "device_ms_per_step": round(acc["apply32_ms"] / steps, 3),
            "event_pair_ms": round(stats.get("event_pair_ms", 0.0), 5),
            "renewals_of_SX_SP_through_W_per_solve": stats.get("lowp_renewals", 0),
        }
        if acc["apply32_ms"] >= acc["apply_ms"]:
            roof_img["double_precision_launches"] = roof_symm
            roof_symm = roof_img
        else:
            roof_symm["image_launches"] = roof_img
    # the PCG accumulation does 0.5 V^2 M cell-tree evaluations (one ds_read_b64, one v_min_f64,
    # one v_add_f64 each) against 8 V^2 bytes written once: its HBM fraction is tiny by
    # construction and is reported as is, next to the fractions of the units that do bound it
    spec_b = int(bstats.get("spec_batches", 0))
    # the launches of the producer / consumer kernel and those of the 4-wave kernels, apart: each
    # with the average duration of ITS launches (what a rocprofv3 kernel-stats row shows)
    spec_ms = acc["spec_ms"] / steps
    spec_trees = int(bstats.get("spec_trees", 0))
    cells_per_tree = bstats["cell_trees"] / max(int(bstats["n_trees"]), 1)
    launches = []
    if spec_b:
        launches.append({"kernel": "k_accumulate_spec", "launches_per_step": spec_b, "trees": spec_trees,
                         "avg_launch_ms": round(spec_ms / spec_b, 4),
                         "cell_trees_per_s": round(cells_per_tree * spec_trees / (spec_ms * 1e-3), 0) if spec_ms > 0 else 0.0})
    if n_batches > spec_b:
        rest_ms, rest_trees = acc_ms - spec_ms, int(bstats["n_trees"]) - spec_trees
        launches.append({"kernel": "k_accumulate_mono" if tables.monotone else "k_accumulate_gen",
                         "launches_per_step": n_batches - spec_b, "trees": rest_trees,
                         "avg_launch_ms": round(rest_ms / (n_batches - spec_b), 4),
                         "cell_trees_per_s": round(cells_per_tree * rest_trees / (rest_ms * 1e-3), 0) if rest_ms > 0 else 0.0})
    main_launch = max(launches, key=lambda e: e["avg_launch_ms"] * e["launches_per_step"]) if launches else None
    if main_launch and main_launch["cell_trees_per_s"]:
        # the roofline of the DOMINANT kernel is that of its own launches (the other kind is listed beside it)
        cell_rate_k = main_launch["cell_trees_per_s"]
    else:
        cell_rate_k = cell_rate
    lds_gbs = 8.0 * cell_rate_k / 1e9
    if not tables.monotone:
        acc_kernel = "k_accumulate_gen"
    elif spec_b == 0:
        acc_kernel = "k_accumulate_mono"
    elif spec_b == n_batches:
        acc_kernel = "k_accumulate_spec"
    else:
        acc_kernel = (f"k_accumulate_spec ({spec_b} of the {n_batches} tree batches; the short ones: "
                      "k_accumulate_mono)")
    roof_acc = {
        "kernel": acc_kernel + " (PCG weights: tree-ordered sums of LCA values into 64 x 256 tiles)",
        # the binding unit: every cell-tree evaluation is one 8-byte ds_read_b64 of the tile's
        # row-row table (plus one v_min_f64 and one v_add_f64); HBM sees W once
        "bound": "lds",
        "achieved": round(lds_gbs, 1),
        "peak": LDS_PEAK_TBS * 1e3,
        "unit": "GB/s",
        "frac": round(lds_gbs / (LDS_PEAK_TBS * 1e3), 4),
        "traffic": None,
        "hbm": {"achieved": round(build_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(build_gbs / HBM_PEAK_GBS, 5),
                "bytes_per_launch": build_bytes / n_batches},
        "bytes_per_launch": build_bytes / n_batches,
        "avg_launch_ms": main_launch["avg_launch_ms"] if main_launch else round(acc_ms / n_batches, 4),
        "launches_per_step": main_launch["launches_per_step"] if main_launch else n_batches,
        "launches": launches,
        "device_ms_per_step": round(acc_ms, 3),
        "tile_list_batches": int(bstats.get("listed_batches", 0)),
        "cell_trees_per_step": bstats["cell_trees"],
        "cell_trees_per_s": round(cell_rate_k, 0),
        "cell_trees_per_s_all_launches": round(cell_rate, 0),
        "frac_f64_valu": round(2.0 * cell_rate_k / (F64_VALU_TOPS * 1e12), 4),
        "frac_lds": round(8.0 * cell_rate_k / (LDS_PEAK_TBS * 1e12), 4),
        "note": "not HBM-bound by construction (SURVEY.md 8d): 0.5 V^2 M cell-tree evaluations against "
                "8 V^2 bytes of W written once.  frac = LDS bytes of the cell loop (8 per cell-tree, "
                "ds_read_b64) over ~150 TB/s; the reads are 2-way bank-conflicted by construction (64 "
                "table rows over 32 eight-byte bank pairs), so 0.5 is the ceiling of this fraction; "
                "frac_f64_valu prices the 2 fp64 VALU ops per cell-tree against 39.3 Tops/s; hbm.* is the "
                "algorithmic HBM figure (W + tables once per build).  achieved / avg_launch_ms are those of the "
                "kernel named first in `launches` by device time (its own launches: a rocprofv3 kernel-stats "
                "row), the other kind of launch is listed beside it.  Round 5 measured the inner loops alone "
                "(profiles/r05_cells_probe_rank_halved.txt): a conflict-free table layout runs no faster, the "
                "loop is bound by how fast 2-3 waves per SIMD issue its dependent ds_read / min / add stream",
    }
    dominant, other = (roof_acc, roof_symm) if acc_ms >= symm_ms_step else (roof_symm, roof_acc)
    report = {
        "value": round(sec_per_step, 6),
        "ms_per_step": round(sec_per_step * 1e3, 3),
        "steps": steps,
        "config": {
            "workload": (
                f"BASELINE.json configs[{cfg_idx}]: synthetic {n} taxa / {m} "
                + ("planted trees (model tree + ceil(0.02 N) SPR moves)" if planted
                   else "random-join rooted trees")
                + f", pcg_weighting='{strategy}'" + (", per-tree weights" if rw else "")
                if cfg_idx >= 0 else
                (f"partial coverage: {n} taxa / {m} trees of {tables.leaves_per_tree} leaves each / {strategy}"
                 if getattr(tables, "leaves_per_tree", None) else f"custom: {n} taxa / {m} trees / {strategy}")
            ),
            "n_taxa": n,
            "n_trees": m,
            "pcg_weighting": strategy,
            "seed": seed,
            "parallelism": "single device, symmetric tile schedule" if world == 1
            else (f"the upper triangle of W over {world} ranks by rows (equal trapezoids, no exchange); per "
                  "iteration every rank applies its tiles directly and transposed and one RCCL all-gather "
                  "collects the V x b partial products, added in rank order" if mode == "upper" else
                  f"W row-partitioned over {world} ranks ("
                  + ("upper-triangle tiles split round-robin, packed tiles exchanged once"
                     if bstats["symmetric"] == 2 else "every rank evaluates all cells of its rows")
                  + "), RCCL all-gather of the Krylov block per iteration"),
            "lobpcg_block": stats["block"],
            "tol": args.tol,
            # the arithmetic is fp64 throughout; what the loop's operator STREAMS is disclosed here
            "precision": ("fp64 arithmetic; the LOBPCG loop applies the operator from a single-precision image of W "
                          "(search directions only), S X / S P renewed and the result confirmed through the fp64 W"
                          if acc["n_apply32"] > 0 else "fp64 throughout (no single-precision image of W)"),
        },
        "roofline": dominant,
        "roofline_other": other,
        "roofline_path": {
            "what": "(B_A + B_C) / t of SURVEY.md 8d: algorithmic bytes of build + solve over the step time; "
                    + ("B_C counts 4 V^2 bytes per apply (symmetric SYMM, upper tiles only)" if symm_tri
                       else "B_C counts 8 rows V bytes per apply"),
            "bound": "hbm",
            "achieved": round(path_gbs, 1),
            "peak": HBM_PEAK_GBS * world,
            "unit": "GB/s",
            "frac": round(path_gbs / (HBM_PEAK_GBS * world), 4),
            "bytes_build": build_bytes,
            "bytes_solve": b_c,
        },
        "stages": {
            "build_ms": round(acc["build_ms"] / steps, 3),
            "build_prep_ms": round(acc["prep_ms"] / steps, 3),
            "build_accumulate_ms": round(acc_ms, 3),
            "build_exchange_ms": round(acc["exch_ms"] / steps, 3),
            "fiedler_ms": round(acc["solve_ms"] / steps, 3),
            "fiedler_symm_ms": round(symm_ms_step, 3),
            "lobpcg_iterations": acc["iters"] / steps,
            "converged": stats["converged"],
            "lambda2": stats["lambda"][1],
            "lambda3": stats["lambda_next"],
            "residual": stats["resid"][1],
            "tables_upload_first_s": round(t_upload_first, 4),
            "tables_generate_s": round(t_gen, 3),
        },
    }
    report["_maps"] = maps
    if world > 1:
        report["stages"]["build_exchange_bytes_received"] = bstats.get("exchange_bytes", 0.0)
        # the one collective of an iteration: device time (HIP events around every fourth, scaled)
        # and the bytes it delivers to a rank -- shared: world x rows x b x 8 (the Krylov block's
        # slices), upper: world x V x b x 8 (the ranks' partial products)
        n_ag = max(acc["n_ag"], 1)
        report["stages"]["allgather_ms_per_iteration"] = round(acc["ag_ms"] / n_ag, 5)
        report["stages"]["allgather_ms_per_step"] = round(acc["ag_ms"] / steps, 4)
        report["stages"]["allgathers_per_step"] = acc["n_ag"] / steps
        report["stages"]["allgather_bytes_received_per_rank"] = stats.get("allgather_bytes", 0.0)
        report["stages"]["multi_rank_mode"] = mode
        # what THIS rank (rank 0 prints its own) streams per operator application: the double-precision launches
        # (8 bytes a cell of its rows, or 4 V^2 / world in the upper-triangle job) and -- round 6, row-partitioned
        # layout -- the launches on the single-precision image of its rows (4 bytes a cell)
        report["stages"]["rank_rows"] = int(re_ - rb)
        report["stages"]["rank_apply_bytes_per_launch"] = stats["apply_bytes"]
        report["stages"]["rank_apply32_bytes_per_launch"] = apply32_bytes
        report["stages"]["rank_launches_per_step"] = {"double": n_apply_step, "image": n_apply32_step}
    # HBM traffic from the committed PMC passes of this workload, if any
    try:
        pmc_key = name if name != "custom" else f"custom_{n}_{m}_{strategy}"
        pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text()).get(pmc_key, [])
        nested = [roof_symm.get(k) for k in ("double_precision_launches", "image_launches") if roof_symm.get(k)]
        for roof in (roof_symm, roof_acc, *nested):
            short = roof["kernel"].split("<")[0].split(" ")[0]
            for entry in pmc if isinstance(pmc, list) else [pmc]:
                if entry and world == 1 and entry.get("kernel", "").split("<")[0] == short:
                    roof["traffic"] = entry["traffic"]
                    roof["traffic_source"] = entry["source"]
    except (OSError, ValueError, KeyError, AttributeError):
        pass

    if full:
        # the same step on tables that are already resident in HBM, and the upload alone
        k = min(steps, 5)
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            one_step()
        barrier()
        report["value_tables_resident"] = round((time.perf_counter() - t0) / k, 6)
        t0 = time.perf_counter()
        for _ in range(k):
            dev.upload(tables).free()
        dev.synchronize()
        report["stages"]["tables_upload_ms"] = round((time.perf_counter() - t0) / k * 1e3, 3)
        # (the whole copy, drained; inside a step only the first tree batch's worth is waited for --
        # the rest arrives on the copy stream behind the first batch's kernels, include/scs_hip.h)
        report["stages"]["tables_upload_exposed_ms"] = round(
            (sec_per_step - report["value_tables_resident"]) * 1e3, 3)
        report["stages"]["tables_bytes"] = int(16 * tables.n_leaves + 16 * m + 8)
        report["stages"]["tables_in_pinned_host_memory"] = bool(getattr(tables, "pinned", False))

    if rank == 0 and world == 1 and not args.no_parity:
        from oracle import tables_oracle as to

        graph, maps2, stats2, _ = one_step(keep=True)
        rows = np.unique(np.random.RandomState(1).randint(0, n, size=16 if full else 8)).astype(np.int32)
        want = to.pcg_rows(tables, rows)
        mismatch = 0
        for i, r in enumerate(rows):
            got = graph.download_rows(int(r), 1)[0]
            mismatch += int(np.count_nonzero(got != want[i]))
        report["parity"] = {
            "w_rows_checked": int(len(rows)),
            "w_cells_mismatched": mismatch,
            "w_max_rel_diff": 0.0 if mismatch == 0 else None,
            "fiedler_residual": stats2["resid"][1],
            "maps_repeatable": bool(np.array_equal(maps, maps2)),
        }
        if full and n <= 20000 and not args.no_cpu_baseline:
            # the SURVEY.md 8d gates at full size; the same scikit-learn run is the eigen-solve
            # leg of cpu_baseline
            from sklearn.cluster import k_means

            w = graph.download()
            rs = np.random.RandomState(0)
            v0g = rs.uniform(-1, 1, n)
            maps_g, _ = graph.fiedler(v0g, tol=args.tol, max_iter=args.max_iter, block=args.block)
            _, labels_dev, _ = k_means(maps_g, 2, random_state=rs, n_init=10, verbose=False)
            gates, t_eig, blas = sklearn_gates(w, maps_g, labels_dev)
            report["parity"].update(gates)
            report["parity"]["w_symmetric"] = bool(np.array_equal(w, w.T))
            del w
            legs = cpu_build_legs(tables, args, n, m)
            ref_leg = (reference_style_leg(n, m, strategy, 300, 30, strategy) if args.quick_reference_leg
                       else reference_style_leg(n, m, strategy))
            report["cpu_baseline"] = {
                "value": round(legs["all_core_s"] + t_eig, 3),
                "unit": "s",
                "cores": legs["cores"],
                "kind": "port",
                "sample": (
                    f"the full workload, nothing scaled: build = oracle/pcg_oracle.c (C restatement of "
                    f"scs.py:495-663) on all {m} trees over {legs['cores']} threads, {legs['all_core_s']:.2f} s; "
                    f"eig = sklearn.manifold.spectral_embedding (the reference's ARPACK shift-invert "
                    f"path) on the full {n} x {n} matrix, {blas} BLAS threads, {t_eig:.2f} s.  One thread: "
                    f"trees [0,{legs['one_thread_sample_trees']}) in {legs['one_thread_sample_s']:.2f} s, "
                    f"x{m / legs['one_thread_sample_trees']:.1f} -> {legs['one_thread_scaled_s']:.1f} s (scaled).  "
                    f"The reference's literal dict-of-tuples build (reference_style_build): {ref_leg['sample']}, "
                    f"{ref_leg['seconds']:.2f} s for {ref_leg['pair_updates']} pair updates; EXTRAPOLATED to "
                    f"{ref_leg['extrapolated_s_at_workload']:.0f} s for this workload -- not part of `value`"
                ),
                "build_s": round(legs["all_core_s"], 3),
                "build_one_thread_scaled_s": round(legs["one_thread_scaled_s"], 2),
                "eig_s": round(t_eig, 3),
                "blas_threads": blas,
                "reference_style_build": ref_leg,
            }
        graph.free()
    if world > 1:
        if rank == 0 and not args.no_parity:
            from oracle import tables_oracle as to

            rows = (rb + np.unique(np.random.RandomState(1).randint(0, re_ - rb, size=4))).astype(np.int32)
            want = to.pcg_rows(tables, rows)
            mismatch = 0
            for i, r in enumerate(rows):
                # (an upper-triangle job stores a row from its 256-column diagonal tile on)
                c0 = (int(r) // 256 * 256) if mode == "upper" else 0
                mismatch += int(np.count_nonzero(kept.download_rows(int(r), 1)[0][c0:] != want[i][c0:]))
            report["parity"] = {
                "w_rows_checked": int(len(rows)),
                "w_cells_mismatched": mismatch,
                "fiedler_residual": stats["resid"][1],
            }
        kept.free()
    dtab.free()
    return report


def launch_ranks(n: int, cmd: list[str] | None = None) -> int:
    """Start `n` copies of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT as torch.distributed.run would set them), wait for all of them,
    print what rank 0 printed, and return non-zero if any rank failed.  The parent never touches
    the GPU; a rank that fails takes the others down with it (their exact PIDs).  `cmd` replaces
    the command line of a rank (tests)."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free port; the host store binds MASTER_PORT + 1 ... + 16
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    port = min(port, 65535 - 32)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd or [sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    timeout_s = float(os.environ.get("SCS_BENCH_LAUNCH_TIMEOUT", "3000"))
    deadline = time.monotonic() + timeout_s
    rc = 0
    live = set(range(n))
    out0 = ""
    import threading

    def drain():  # rank 0's stdout is read while it runs (a full pipe would block it)
        nonlocal out0
        out0 = procs[0].stdout.read()

    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        if live and (rc != 0 or time.monotonic() > deadline):
            if rc == 0:
                rc = 124
                print(f"bench.py: ranks {sorted(live)} still running after {timeout_s:.0f} s", file=sys.stderr)
            for r in live:
                procs[r].terminate()
            for r in live:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
        elif live:
            time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc


def main() -> int:
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher (it has made no
        # HIP call and makes none) -- one fresh child per GPU, rank 0's JSON line relayed
        return launch_ranks(args.gpus)

    from spectralclustersupertree_amd.partition import rendezvous

    # SCS_BENCH_DEVICE pins every rank to one device index (single-GPU rehearsal of the
    # multi-rank path); the driver's runs leave it unset: one GPU per local rank
    dev_index = int(os.environ.get("SCS_BENCH_DEVICE", local_rank))
    dist, dev = rendezvous(rank, world, dev_index)

    name = args.workload or ("cfg2" if world == 1 else "cfg3")
    # the box's own copy rate beside the nominal 8 TB/s (SURVEY.md 8d "Peak to quote")
    try:
        copy_gbs = dev.copy_bandwidth() if rank == 0 else None
    except Exception as exc:  # noqa: BLE001 - a report field, never fatal
        copy_gbs = None
        print(f"bench.py: copy bandwidth not measured: {exc}", file=sys.stderr)
    main_rep = run_workload(name, args, dev, dist, rank, world, args.steps, args.warmup, full=True,
                            planted=args.planted)
    maps_by_mode = {main_rep["stages"].get("multi_rank_mode", "single"): main_rep.pop("_maps", None)}
    other_modes = {}
    if world > 1 and not args.no_extra and args.multi_rank_mode in ("shared", "upper"):
        # the other layout in the same invocation (a first real multi-GPU run then decides between
        # them): timed, rows of W against the oracle, embedding against the one-GPU run below
        alt = "upper" if args.multi_rank_mode == "shared" else "shared"
        try:
            rep = run_workload(name, args, dev, dist, rank, world, min(args.steps, 3), 1, full=False, mode=alt)
            maps_by_mode[alt] = rep.pop("_maps", None)
            other_modes[alt] = {k: rep[k] for k in ("value", "ms_per_step", "steps", "config", "roofline",
                                                    "roofline_other", "stages", "parity") if k in rep}
        except Exception as exc:  # noqa: BLE001 - report, never hide the main line
            other_modes[alt] = {"error": f"{type(exc).__name__}: {exc}"}

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
        "scaling_note": "N > 1 runs BASELINE.json configs[3] in the row-partitioned layout north_star names "
                        "(--multi-rank-mode shared; the upper-triangle job is other_modes.upper of the same "
                        "line); its one-GPU time, the strong-scaling baseline, is other_workloads.cfg3.value "
                        "of the N = 1 line and same_workload_on_one_gpu.value of every N > 1 line",
        "vs_baseline": None,
        "dtype": ("f64 (loop operator: fp32 image of W; fp64 renewal + confirmation)"
                  if main_rep.get("config", {}).get("precision", "").startswith("fp64 arithmetic;") else "f64"),
        "data": "synthetic",
        "value_definition": "SURVEY.md 8d / BASELINE.md 3.3 protocol: tables host -> HBM (page-locked source; "
                            "the chunks behind the first tree batch overlap the build's first kernels, all of "
                            "them inside the timed region), scs_pcg_build, scs_fiedler, V x 2 embedding to the "
                            "host; the same step on HBM-resident tables is value_tables_resident",
    }
    try:
        from spectralclustersupertree_amd import kmeans2

        result["kmeans2_fast_path_active"] = bool(kmeans2.fast_path_active())
    except Exception as exc:  # noqa: BLE001 - a report field, never fatal
        result["kmeans2_fast_path_active"] = f"error: {exc}"
    for key in ("config", "roofline", "roofline_other", "roofline_path", "value_tables_resident", "stages",
                "parity", "cpu_baseline"):
        if key in main_rep:
            result[key] = main_rep[key]
    if other_modes:
        result["other_modes"] = other_modes
    if world > 1:
        try:
            result["communicator"] = dev.comm_info()  # RCCL's own rank count beside WORLD_SIZE
            result["communicator"]["launcher_world_size"] = world
        except Exception as exc:  # noqa: BLE001 - a report field, never fatal
            result["communicator"] = {"error": str(exc)}
    if copy_gbs:
        # every HBM roofline of the line also against what this box's copy engine-free DtoD copy reaches
        result["hbm_copy_measured_gbs"] = round(copy_gbs, 1)
        for key in ("roofline", "roofline_other", "roofline_path"):
            roof = result.get(key)
            for part in (roof, (roof or {}).get("hbm")):
                if isinstance(part, dict) and part.get("bound", "hbm") == "hbm" and part.get("unit") == "GB/s":
                    scale = world if key == "roofline_path" else 1
                    part["peak_measured"] = round(copy_gbs * scale, 1)
                    part["frac_of_measured"] = round(part["achieved"] / (copy_gbs * scale), 4)

    t_bench0 = time.perf_counter()
    budget_s = float(os.environ.get("SCS_BENCH_BUDGET_S", "400"))  # wall clock the extra legs below may use in all
    if world == 1 and not args.no_extra and "fp32 image" in result.get("dtype", ""):
        # the same workload with the loop in double precision throughout (SCS_LOWP=0): the price of NOT using the
        # image, in the same line as the headline
        old_lowp = os.environ.get("SCS_LOWP")
        os.environ["SCS_LOWP"] = "0"
        try:
            rep = run_workload(name, args, dev, dist, rank, world, min(args.steps, 5), 1, full=False, planted=args.planted)
            rep.pop("_maps", None)
            result["value_all_double"] = rep["value"]
            result["all_double"] = {"value": rep["value"], "ms_per_step": rep["ms_per_step"],
                                    "fiedler_ms": rep["stages"]["fiedler_ms"],
                                    "lobpcg_iterations": rep["stages"]["lobpcg_iterations"],
                                    "precision": rep["config"]["precision"]}
        except Exception as exc:  # noqa: BLE001 - report, never hide the main line
            result["all_double"] = {"error": f"{type(exc).__name__}: {exc}"}
        finally:
            if old_lowp is None:
                del os.environ["SCS_LOWP"]
            else:
                os.environ["SCS_LOWP"] = old_lowp
    if world == 1 and args.workload is None and not args.no_extra:
        # seeds 0, 1, 2 (median) and one planted input, lambda2 / lambda3 printed
        seeds = {str(args.seed): {"value": main_rep["value"], "lambda2": main_rep["stages"]["lambda2"],
                                  "lambda3": main_rep["stages"]["lambda3"],
                                  "iterations": main_rep["stages"]["lobpcg_iterations"]}}
        for sd in (1, 2):
            try:
                rep = run_workload(name, args, dev, dist, rank, world, 3, 1, full=False, seed=sd)
                rep.pop("_maps", None)
                seeds[str(sd)] = {"value": rep["value"], "lambda2": rep["stages"]["lambda2"],
                                  "lambda3": rep["stages"]["lambda3"],
                                  "iterations": rep["stages"]["lobpcg_iterations"],
                                  "w_cells_mismatched": rep.get("parity", {}).get("w_cells_mismatched")}
            except Exception as exc:  # noqa: BLE001 - report, never hide the main line
                seeds[str(sd)] = {"error": str(exc)}
        vals = sorted(v["value"] for v in seeds.values() if "value" in v)
        seeds["median"] = vals[len(vals) // 2] if vals else None
        result["seeds"] = seeds
        try:
            rep = run_workload(name, args, dev, dist, rank, world, 3, 1, full=False, planted=True)
            rep.pop("_maps", None)
            result["planted"] = {k: rep[k] for k in ("value", "config", "stages", "parity") if k in rep}
        except Exception as exc:  # noqa: BLE001
            result["planted"] = {"error": str(exc)}
        # the other single-GPU configurations of BASELINE.json, one short pass each (configs[3]
        # is the workload the N > 1 runs use: its N = 1 time is the strong-scaling baseline)
        others = {}
        for extra, st in (("cfg1", 3), ("cfg3", 1)):
            try:
                rep = run_workload(extra, args, dev, dist, rank, world, st, 1, full=False)
                rep.pop("_maps", None)
                others[extra] = {k: rep[k] for k in ("value", "steps", "config", "roofline", "roofline_other",
                                                     "roofline_path", "stages", "parity") if k in rep}
            except Exception as exc:  # noqa: BLE001 - report, never hide the main line
                others[extra] = {"error": str(exc)}
        # configs[4] (100 000 taxa / 5 000 weighted trees, W = 80 GB): its top level, one step; and its "full
        # recursion" leg -- construct_supertree's whole walk with the property checks of
        # tools/full_recursion_check.py -- so that the driver's own run carries both.  Each is skipped (and says
        # so) when the time the extras may use is nearly spent.
        spent = time.perf_counter() - t_bench0
        if os.environ.get("SCS_BENCH_CFG4", "1") != "0":
            if spent + 90 <= budget_s:
                try:
                    t0 = time.perf_counter()
                    # no warm-up step here (6.6 s each); what a warm-up would leave in the device's arena -- the
                    # 80 GB of W in one piece -- is reserved instead, untimed: on this pool the driver CLEARS memory
                    # another process has used (~25 ms per GB, 1.9 s for W; tools/probes/alloc_big_probe.hip), a
                    # cost of the box's history, not of the step
                    n4 = WORKLOADS["cfg4"][0]
                    t_res = time.perf_counter()
                    dev.reserve(n4 * ((n4 + 511) // 512 * 512) * 8)
                    t_res = time.perf_counter() - t_res
                    rep = run_workload("cfg4", args, dev, dist, rank, world, 1, 0, full=False)
                    rep.pop("_maps", None)
                    others["cfg4"] = {k: rep[k] for k in ("value", "steps", "config", "roofline", "roofline_other",
                                                           "roofline_path", "stages", "parity") if k in rep}
                    others["cfg4"]["arena_reserve_untimed_s"] = round(t_res, 3)
                    others["cfg4"]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
                except Exception as exc:  # noqa: BLE001 - report, never hide the main line
                    others["cfg4"] = {"error": f"{type(exc).__name__}: {exc}"}
            else:
                others["cfg4"] = {"skipped": f"{spent:.0f} s of the {budget_s:.0f} s for extra legs already used"}
        result["other_workloads"] = others
        spent = time.perf_counter() - t_bench0
        if os.environ.get("SCS_BENCH_RECURSION", "1") != "0":
            if spent + 150 <= budget_s:
                try:
                    # (the recursion runs on the process-wide context; the device's arena -- csrc/scs_arena.h --
                    # is shared by all contexts of the process, so the memory the legs above left free serves it)
                    sys.path.insert(0, str(ROOT / "tools"))
                    import full_recursion_check

                    t0 = time.perf_counter()
                    rec = full_recursion_check.run(100000, 5000, True, log=lambda s_: None)
                    rec["leg_wall_s"] = round(time.perf_counter() - t0, 1)
                    rec["workload"] = "configs[4] full recursion: construct_supertree on 100 000 taxa / 5 000 weighted trees, branch"
                    result["recursion"] = rec
                except Exception as exc:  # noqa: BLE001 - report, never hide the main line
                    result["recursion"] = {"error": f"{type(exc).__name__}: {exc}"}
            else:
                result["recursion"] = {"skipped": f"{spent:.0f} s of the {budget_s:.0f} s for extra legs already used"}

    if rank == 0:
        try:
            st = dev.arena_stats()
            result["device_memory_arena"] = {
                "what": "csrc/scs_arena.h (DESIGN.md 13): slabs the process keeps from the driver, every block of every "
                        "context carved out of them; counts at the end of this run",
                "slab_GB": round(st["slab_bytes"] / 2**30, 2), "slabs": st["slabs"],
                "driver_allocations": st["driver_allocations"], "driver_releases": st["driver_releases"],
                "requests_served": st["requests"]}
        except Exception as exc:  # noqa: BLE001 - a report field, never fatal
            result["device_memory_arena"] = {"error": str(exc)}
    dev.close()
    # N > 1: rank 0 also times the SAME workload alone on its GPU (one warm-up + one pass,
    # single-rank context, the other ranks wait), so that every multi-GPU line carries the
    # one-device time it is to be compared with
    if world > 1 and rank == 0 and not args.no_extra:
        from spectralclustersupertree_amd.backend import Device

        try:
            solo = Device(dev_index, 0, 1)
            try:
                rep = run_workload(name, args, solo, None, 0, 1, 1, 1, full=False)
                maps_one = rep.pop("_maps", None)
                result["same_workload_on_one_gpu"] = {k: rep[k] for k in ("value", "stages", "roofline", "parity")
                                                      if k in rep}
                # the embedding of every multi-rank layout against the one-GPU run of the same input
                # (Fiedler column; the 1e-10 bar of north_star is on the unit-norm eigenvector, of
                # which the embedding is a row-scaled copy: both are printed where the degrees allow)
                for md, mp in maps_by_mode.items():
                    where = result if md == args.multi_rank_mode else result.get("other_modes", {}).get(md)
                    if mp is None or maps_one is None or not isinstance(where, dict):
                        continue
                    d = float(np.max(np.abs(mp[:, 1] - maps_one[:, 1])))
                    scale = float(np.max(np.abs(maps_one[:, 1])))
                    where.setdefault("parity", {})["embedding_vs_one_gpu_max_abs"] = d
                    where["parity"]["embedding_vs_one_gpu_rel_to_largest_entry"] = d / scale if scale > 0 else None
            finally:
                solo.close()
        except Exception as exc:  # noqa: BLE001 - report, never hide the main line
            result["same_workload_on_one_gpu"] = {"error": str(exc)}
    if dist is not None:
        dist.barrier()
        dist.close()
    if rank == 0:
        print(json.dumps(result))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
