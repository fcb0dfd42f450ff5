import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from spectralclustersupertree_amd.backend import Device
dev = Device(0)
mats = [np.array([[0, 2/np.sqrt(6), 0], [2/np.sqrt(6), 0, 1/np.sqrt(3)], [0, 1/np.sqrt(3), 0]]),
        np.array([[0, 1.0], [1.0, 0]]),
        np.array([[0, 1.0, 1.0], [1.0, 0, 1.0], [1.0, 1.0, 0]]) / 2]
rng = np.random.RandomState(3)
for n in (3, 5, 6, 7, 9, 11, 13, 17, 23, 24):
    x = rng.standard_normal((n, n)); mats.append(x + x.T)
for a in mats:
    w, v = dev.debug_jacobi(np.ascontiguousarray(a))
    wr = np.linalg.eigvalsh(a)[::-1]
    print(a.shape[0], "eig err", np.abs(w - wr).max(), "recon", np.abs(v @ np.diag(w) @ v.T - a).max(), "orth", np.abs(v.T @ v - np.eye(len(w))).max())
    if a.shape[0] <= 3: print(w, "\n", v)
dev.close()
