import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import rand_csr, arrays
from oracle import oracle
from sparse_matrix_mult_amd.engine import Context
ctx = Context(0)
m, k, n, da, db = 300, 2000, 20000, 0.01, 0.004
A, B = rand_csr(m, k, da, 1), rand_csr(k, n, db, 2)
want = oracle.sparse(arrays(A), arrays(B), n)
a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
for hcfg in ((256, 2048), (0, 0)):
    ctx.tune_hash(*hcfg)
    for lds, w in ((17000, 16), (17000, 8), (17000, 1), (5000, 16)):
        ctx.tune(lds, w)
        p, i, v = ctx.spgemm_host(a, b, exact=True)
        bad = np.nonzero(v != want[2])[0]
        rows = np.unique(np.searchsorted(want[0], bad, side="right") - 1)
        cnt = np.diff(want[0])
        print("hash", hcfg, "lds", lds, "waves", w, "idx ok", np.array_equal(i, want[1]), "bad values", len(bad), "in rows", len(rows),
              "row nnz of bad rows", cnt[rows][:8], "max cnt", cnt.max())
        if len(bad):
            r = rows[0]; s0 = want[0][r]
            bb = bad[bad >= s0][:5]
            print("   row", r, "slots", bb - s0, "cols", want[1][bb], "got", v[bb], "want", want[2][bb], "A row len", A.indptr[r+1]-A.indptr[r])
