"""Randomised parity sweep on the device (run on the GPU box): W bit-exact against the C
oracle and the Fiedler column against scikit-learn on random sizes, strategies, coverage
and weights.  Prints a summary line; exits non-zero on any failure.

    python tools/fuzz_parity.py [--seconds 120] [--seed 0]
"""
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
n_cases = n_fiedler = 0
worst = 0.0
fails = []
while time.time() < t_end:
    n = int(rng.choice([rng.randint(3, 65), rng.randint(65, 400), rng.randint(400, 2500)]))
    m = int(rng.randint(1, 30))
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
            err = float(np.max(np.abs(maps - ref)))
            worst = max(worst, err)
            n_fiedler += 1
            if err > 1e-10:
                fails.append(f"Fiedler {err:.2e}: {tag} gap {lam[1]-lam[2]:.2e} {stats}")
    g.free()
    dtab.free()
dev.close()
print(f"fuzz: {n_cases} builds bit-exact checked, {n_fiedler} Fiedler comparisons, worst |diff| {worst:.2e}, {len(fails)} failures")
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
