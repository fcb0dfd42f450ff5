"""Randomised parity sweep on the device (run on the GPU box): W bit-exact against the C
oracle and the Fiedler column against scikit-learn on random sizes, strategies, coverage
and weights.  Prints a summary line; exits non-zero on any failure.

    python tests/fuzz_parity.py [--seconds 120] [--seed 0]
"""
import os as _os

_os.environ.setdefault("SCS_DEBUG", "1")  # (the sweeps force probe paths: csrc/scs_internal.h scs_dbg)
import argparse, sys, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from oracle import scs_oracle as so
from oracle import tables_oracle as to
from spectralclustersupertree_amd import synthetic
from spectralclustersupertree_amd.backend import Device

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
warnings.simplefilter("ignore")
rng = np.random.RandomState(args.seed)
dev = Device(0)
t_end = time.time() + args.seconds
n_cases = n_fiedler = n_multi = n_small = n_upper = n_treepar = n_spec = 0
worst = 0.0
fails = []
while time.time() < t_end:
    n = int(rng.choice([rng.randint(3, 65), rng.randint(65, 400), rng.randint(400, 2500)]))
    if rng.randint(0, 12) == 0:
        n = int(rng.randint(4096, 9001))  # (round 6: the sizes where the mixed-precision loop is the default)
    m = int(rng.randint(1, 30))
    if n > 128 and n <= 900 and rng.randint(0, 4) == 0:
        # (forests of many trees on a few tiles: the tree-parallel build; above ~800 taxa the
        # producer / consumer kernel on a few dozen tiles)
        m = int(rng.randint(128, 420))
    strategy = str(rng.choice(["one", "depth", "branch", "bootstrap"]))
    k = int(rng.randint(max(2, n // 2), n + 1))
    rw = bool(rng.randint(0, 2))
    seed = int(rng.randint(0, 1 << 30))
    tag = f"seed={seed} n={n} m={m} {strategy} k={k} rw={rw}"
    tables = synthetic.make_tables(seed, n, m, strategy, leaves_per_tree=k, random_weights=rw)
    w_ref, _ = to.pcg_dense(tables)
    dtab = dev.upload(tables)
    g = dtab.build()
    w = g.download()
    if not np.array_equal(w, w_ref):
        fails.append(f"W mismatch: {tag}: {int(np.sum(w != w_ref))} cells")
    n_cases += 1
    if g.build_stats["tree_parallel_batches"]:
        n_treepar += 1
    if g.build_stats["spec_batches"]:
        n_spec += 1
    if n <= dev.SMALL_MAX_TAXA:
        # the fused small-node kernel on the same input: same W, same embedding as the general path
        (maps_s, lam_s, w_s), = dev.small_solve([(tables, None)], want_w=True)
        n_small += 1
        if not np.array_equal(w_s, w_ref):
            fails.append(f"small-path W mismatch: {tag}: {int(np.sum(w_s != w_ref))} cells")
        ev = np.sort(np.linalg.eigvalsh(to.normalized_operator(w_ref)[0]))[::-1]
        if np.max(np.abs(lam_s[: min(3, n)] - ev[:3])) > 1e-11:
            fails.append(f"small-path eigenvalues: {tag}: {lam_s} vs {ev[:3]}")
    if n_cases % 200 == 0:
        print(f"... {n_cases} cases, {len(fails)} failures", flush=True)
    deg = w_ref.sum(axis=1)
    if n >= 3 and np.all(deg > 0):
        s_op = to.normalized_operator(w_ref)[0]
        lam = np.sort(np.linalg.eigvalsh(s_op))[::-1]
        # compare only where the Fiedler vector is well defined (scikit-learn itself is only
        # accurate to ~1e-14 / gap)
        if lam[1] - lam[2] > 1e-3 and lam[0] - lam[1] > 1e-3:
            v0 = np.random.RandomState(seed % 1000).uniform(-1, 1, n)
            ref = to.sign_flip_columns(so.spectral_maps(w_ref, np.random.RandomState(seed % 1000)))
            maps, stats = g.fiedler(v0)
            # the sign convention (largest |entry| positive) is ill-defined when two entries tie
            # in magnitude: accept either sign per column there
            err = 0.0
            dd = np.sqrt(deg)  # errors are taken on the unit-norm eigenvector scale as well
            for c in range(2):
                d_minus, d_plus = maps[:, c] - ref[:, c], maps[:, c] + ref[:, c]
                col_err = max(float(np.max(np.abs(d_minus))), float(np.max(np.abs(d_minus * dd))))
                mags = np.sort(np.abs(ref[:, c]))[::-1]
                if len(mags) > 1 and mags[0] - mags[1] <= 1e-9 * mags[0]:
                    col_err = min(col_err, max(float(np.max(np.abs(d_plus))),
                                               float(np.max(np.abs(d_plus * dd)))))
                err = max(err, col_err)
            worst = max(worst, err)
            n_fiedler += 1
            if err > 1e-10:
                fails.append(f"Fiedler {err:.2e}: {tag} gap {lam[1]-lam[2]:.2e} {stats}")
    g.free()
    dtab.free()
    # every few cases: the same input row-partitioned over 2-4 in-process ranks, shared build
    if n >= 130 and n_cases % 4 == 0:
        import threading
        from spectralclustersupertree_amd import _native as nv
        world = int(rng.randint(2, 5))
        # alternately: row-partitioned with shared tiles (arbitrary splits), or an upper-triangle
        # job (SCS_BUILD_UPPER: splits on multiples of 256, equal trapezoids)
        upper_job = (n_cases // 4) % 2 == 1 and n >= 256 * world
        if upper_job:
            from spectralclustersupertree_amd.partition import row_splits_upper
            splits = row_splits_upper(n, world)
        else:
            cuts = sorted(set(int(x) for x in rng.choice(np.arange(1, n), world - 1, replace=False)))
            splits = [0] + cuts + [n]
        world = len(splits) - 1
        lib = nv.load_library()
        group = nv.C.c_void_p()
        nv.check(lib.scs_local_group_create(world, nv.C.byref(group)))
        outs, errs = [None] * world, [None] * world
        v0m = np.random.RandomState(seed % 1000).uniform(-1, 1, n)
        def worker(r):
            try:
                d = Device(0, r, world, _local_group=group)
                dt = d.upload(tables)
                gg = dt.build(splits[r], splits[r + 1], shared=not upper_job, upper=upper_job)
                ww = gg.download()
                mm, _ = gg.fiedler(v0m)
                outs[r] = (ww, mm)
                gg.free(); dt.free(); d.close()
            except BaseException as e:  # noqa: BLE001
                errs[r] = e
        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(timeout=300) for t in th]
        lib.scs_local_group_destroy(group)
        n_multi += 1
        if any(errs):
            fails.append(f"multi-rank error: {tag} splits={splits}: {errs}")
        else:
            for r in range(world):
                want_rows = w_ref[splits[r]:splits[r + 1]].copy()
                if upper_job:  # a row is stored from its 256-column diagonal tile on
                    for i in range(splits[r], splits[r + 1]):
                        want_rows[i - splits[r], : i // 256 * 256] = 0.0
                    n_upper += 1
                if not np.array_equal(outs[r][0], want_rows):
                    fails.append(f"multi-rank W mismatch: {tag} splits={splits} upper={upper_job} rank {r}")
                if not np.array_equal(outs[r][1], outs[0][1]):
                    fails.append(f"multi-rank maps differ between ranks: {tag} splits={splits} rank {r}")
dev.close()
print(f"fuzz: {n_cases} builds bit-exact checked, {n_fiedler} Fiedler comparisons, worst |diff| {worst:.2e}, "
      f"{n_multi} multi-rank builds + solves ({n_upper} rank-blocks of upper-triangle jobs), {n_small} fused small-node "
      f"solves, {n_treepar} tree-parallel builds, {n_spec} builds by the producer / consumer kernel, {len(fails)} failures")
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
