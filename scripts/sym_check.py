"""Symbolic phase alone (no numeric kernel runs): row pointer of C against the structural product
computed by scipy on the patterns.  For bisecting changes to smm_symbolic without risking the
downstream kernels on a wrong list."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from sparse_matrix_mult_amd.engine import Context  # noqa: E402

ctx = Context(0)
cases = [(1, 1, 1, 1.0), (7, 5, 9, 0.5), (60, 50, 70, 0.2), (300, 300, 300, 0.05), (200, 100, 5000, 0.05),
         (2000, 2000, 2000, 0.01), (3000, 3000, 3000, 0.05), (500, 400, 300000, 0.002)]
bad = 0
for (m, k, n, d) in cases:
    for sym in (False, True):
        A = sp.random(m, k, density=d, format="csr", random_state=1, dtype=np.float64)
        B = sp.random(k, n, density=d, format="csr", random_state=2, dtype=np.float64)
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
        print(f"{m}x{k}x{n} d={d} sym={sym}: nnz got {plan.nnz} want {int(want.sum())} rows differing {nbad}"
              + (f" first {np.flatnonzero(got != want)[:5]} got {got[got != want][:5]} want {want[got != want][:5]}" if nbad else ""), flush=True)
        plan.close(); a.close(); b.close()
print("BAD" if bad else "OK")
