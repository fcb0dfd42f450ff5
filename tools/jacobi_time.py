"""Time the one-workgroup Jacobi kernel by size (run under rocprofv3 --kernel-trace)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd.backend import Device

dev = Device(0)
rng = np.random.RandomState(0)
def run(a, tag):
    t0 = time.perf_counter()
    w, v = dev.debug_jacobi(a)
    dt = time.perf_counter() - t0
    print(tag, a.shape[0], f"{dt*1e6:.0f} us host", np.max(np.abs(v @ np.diag(w) @ v.T - a)))
for n in (4, 4, 8, 12, 12, 24, 24):
    x = rng.standard_normal((n + 3, n))
    a = x.T @ x
    d = 1 / np.sqrt(np.diag(a))
    run(a * d[:, None] * d[None, :], "dense")
for eps in (1e-2, 1e-5, 1e-8):
    x = rng.standard_normal((12, 1)) + eps * rng.standard_normal((12, 4))
    a = x.T @ x
    d = 1 / np.sqrt(np.diag(a))
    run(a * d[:, None] * d[None, :], f"rank1+{eps}")
a = np.diag([3.0, 2.0, 1.0, 0.5]) + 1e-3 * rng.standard_normal((4, 4)); a = 0.5 * (a + a.T)
run(a, "neardiag")
dev.close()
