"""Diagnostic: symbolic phase only of the tall-skinny product (1e6 x 2000 d=0.005 times 2000 x 2000 d=0.05)."""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, ".")
from sparse_matrix_mult_amd import engine
rng = np.random.default_rng(7)
A = sp.random(1_000_000, 2000, 0.005, "csr", random_state=rng); B = sp.random(2000, 2000, 0.05, "csr", random_state=rng)
A.sort_indices(); B.sort_indices()
ctx = engine.Context(0); ctx.timing(True)
a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
for rep in range(3):
    ctx.timing_reset()
    plan = ctx.spgemm_plan(a, b)
    ctx.synchronize()
    print({k: round(ctx.kernel_time(k)[0], 3) for k in ("smm_symbolic", "smm_row_work", "smm_bin_rows", "smm_scan")}, plan.nnz, flush=True)
    plan.close()
