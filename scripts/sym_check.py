"""Symbolic phase alone (no numeric kernel runs): row pointer of C against the structural product
computed by scipy on the patterns.  For bisecting changes to smm_symbolic without risking the
downstream kernels on a wrong list."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from sparse_matrix_mult_amd.engine import Context  # noqa: E402



def wide_random(m, n, per_row, seed):
    """~per_row distinct random columns per row; scipy.sparse.random permutes m*n positions and cannot do 1e10."""
    rng = np.random.default_rng(seed)
    cols = np.sort(rng.integers(0, n, size=(m, per_row)), axis=1)
    keep = np.ones_like(cols, dtype=bool); keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    indptr = np.zeros(m + 1, np.int64); indptr[1:] = np.cumsum(keep.sum(axis=1))
    idx = cols[keep].astype(np.int32)
    return sp.csr_matrix((rng.random(idx.size) + 0.5, idx, indptr.astype(np.int32)), shape=(m, n))


ctx = Context(0)
cases = [(1, 1, 1, 1.0), (7, 5, 9, 0.5), (60, 50, 70, 0.2), (300, 300, 300, 0.05), (200, 100, 5000, 0.05),
         (2000, 2000, 2000, 0.01), (3000, 3000, 3000, 0.05), (500, 400, 300000, 0.002),
         # hash-marker classes (<= 256 / <= 2048 products) and their boundaries with the bitmap kernel
         (20000, 20000, 1000000, None, 10, 10), (5000, 5000, 200000, None, 16, 16), (5000, 5000, 200000, None, 45, 45),
         (3000, 3000, 70000, None, 12, 20)]
bad = 0
for case in cases:
    m, k, n, d = case[:4]
    da, db = (d, d) if d is not None else (case[4] / k, case[5] / n)       # None: nnz per row given instead
    for sym in (False, True):
        if d is None:
            A, B = wide_random(m, k, case[4], 1), wide_random(k, n, case[5], 2)
        else:
            A = sp.random(m, k, density=da, format="csr", random_state=1, dtype=np.float64)
            B = sp.random(k, n, density=db, format="csr", random_state=2, dtype=np.float64)
        Ap, Bp = A.copy(), B.copy(); Ap.data[:] = 1.0; Bp.data[:] = 1.0
        P = (Ap @ Bp).tocsr()
        if sym:
            P = sp.triu(P).tocsr()
        want = np.diff(P.indptr)
        a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
        plan = ctx.spgemm_plan(a, b, symmetric=sym)
        got = np.diff(plan.indptr_host())
        nbad = int((got != want).sum())
        bad += nbad
        print(f"{m}x{k}x{n} d={da:.2g}/{db:.2g} sym={sym}: nnz got {plan.nnz} want {int(want.sum())} rows differing {nbad}"
              + (f" first {np.flatnonzero(got != want)[:5]} got {got[got != want][:5]} want {want[got != want][:5]}" if nbad else ""), flush=True)
        plan.close(); a.close(); b.close()
print("BAD" if bad else "OK")
