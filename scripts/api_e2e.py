"""End-to-end timings through the drop-in Python API (scipy in, scipy/numpy out, PCIe included)
on the orientation configs of BASELINE.md section 2."""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from sparse_matrix_mult_amd import sparse_matrix_multiply  # noqa: E402

for n, d in ((1000, 0.05), (5000, 0.01), (10000, 0.005)):
    A = sp.random(n, n, density=d, format="csr", random_state=np.random.default_rng(1))
    B = sp.random(n, n, density=d, format="csr", random_state=np.random.default_rng(2))
    res = {}
    for name, kw in (("sparse", dict(output_format="sparse")), ("sparse_sym", dict(output_format="sparse", symmetric=True)),
                     ("dense", dict(output_format="dense"))):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); C = sparse_matrix_multiply(A, B, **kw); best = min(best, time.perf_counter() - t0)
        res[name] = round(best * 1e3, 2)
    t0 = time.perf_counter(); S = A @ B; res["scipy"] = round((time.perf_counter() - t0) * 1e3, 2)
    print(json.dumps({"n": n, "d": d, "nnzC": int(S.nnz), "ms": res}))
