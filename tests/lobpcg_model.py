"""Measurement aid, not product code (lives with the tests because it calls the oracle): the
eigen-solver's iteration counts on the REAL spectrum of a benchmark matrix.

    python tests/lobpcg_model.py [n_taxa n_trees]

Builds W of the synthetic workload with the oracle's C restatement, takes all eigenvalues of S with
numpy (cached under /tmp), and runs LOBPCG and a Chebyshev-filtered subspace iteration IN THE
EIGENBASIS (a diagonal operator with those eigenvalues, trivial pair deflated): iteration and
operator-application counts to a residual of 1e-13.  profiles/r05_lobpcg_notes.txt.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def spectrum(n, m):
    path = f"/tmp/scs_spectrum_{n}_{m}.npy"
    if os.path.exists(path):
        return np.load(path)
    from oracle import tables_oracle as to
    from spectralclustersupertree_amd import synthetic

    t = time.time()
    tab = synthetic.make_tables(0, n, m, "branch")
    w = to.pcg_dense_mt(tab, os.cpu_count() or 1)
    s, _ = to.normalized_operator(w)
    del w
    ev = np.linalg.eigvalsh(s)
    print(f"spectrum of {n} x {n} in {time.time() - t:.0f} s", flush=True)
    np.save(path, ev)
    return ev


N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 500)
ev = spectrum(N, M)
lam = ev[:-1].copy()          # deflate trivial eigenvalue 1
n = len(lam)
rng = np.random.RandomState(1)
def resid(X):
    # X orthonormal columns (n x b) in eigenbasis; returns Ritz values (desc) & residual norms, rotated X
    AX = lam[:,None]*X
    H = X.T@AX
    th, C = np.linalg.eigh(H)
    idx = np.argsort(-th); th=th[idx]; C=C[:,idx]
    X = X@C; AX = lam[:,None]*X
    R = AX - X*th
    return th, np.linalg.norm(R,axis=0), X, R
def orth(X):
    Q,_ = np.linalg.qr(X); return Q
def lobpcg(b, tol=1e-13, maxit=300):
    X = orth(rng.randn(n,b)); P=None
    for it in range(maxit):
        th, rn, X, R = resid(X)
        if rn[0] <= tol: return it
        R = R - X@(X.T@R)
        basis = [X, R] + ([P] if P is not None else [])
        Q = orth(np.hstack(basis))
        H = Q.T@(lam[:,None]*Q)
        w, C = np.linalg.eigh(H); idx=np.argsort(-w)[:b]
        Xn = Q@C[:,idx]
        # P = component of Xn outside X
        P = Xn - X@(X.T@Xn)
        X = orth(Xn)
    return maxit
def chfsi(b, d, tol=1e-13, maxouter=100, lmin=None, adapt=True):
    X = orth(rng.randn(n,b))
    applies=0; outer=0
    a = lam.min() if lmin is None else lmin
    while outer < maxouter:
        th, rn, X, R = resid(X); applies+=1
        if rn[0] <= tol: return outer, applies
        cut = th[-1]   # lowest Ritz value in block: damp [a, cut]
        c = (cut + a)/2; e = (cut - a)/2
        # Chebyshev recurrence on (S - c)/e
        Y0 = X; Y1 = (lam[:,None]*X - c*X)/e; applies+=1
        for k in range(2, d+1):
            Y2 = 2*(lam[:,None]*Y1 - c*Y1)/e - Y0; applies+=1
            Y0, Y1 = Y1, Y2
            # scale to avoid overflow
            s = np.abs(Y1).max(); Y0/=s; Y1/=s
        X = orth(Y1)
        outer+=1
    return outer, applies
if __name__=='__main__':
    print("spectrum: top", lam[-10:][::-1], "min", lam[:3])
    for b in (4,8):
        print("lobpcg b",b, lobpcg(b))
    for b in (4,8):
        for d in (3,4,6,8,10):
            print("chfsi b",b,"d",d, chfsi(b,d))
