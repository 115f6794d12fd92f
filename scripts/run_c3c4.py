"""One-off measurements of BASELINE configs[2] (dense output) and configs[3] (triple product)
on the GPU box; prints one JSON line per config.  Not part of bench.py's contract."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from sparse_matrix_mult_amd.synthetic import gen_csr_device  # noqa: E402
from sparse_matrix_mult_amd.engine import Context  # noqa: E402

dev = torch.device("cuda", 0)
ctx = Context(0, torch.cuda.current_stream().cuda_stream)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
exact = "--exact" in sys.argv
if which == "c3":
    m = n = 50000
    A = ctx.csr_from_torch(m, n, *gen_csr_device(torch, m, n, 0.01, 1, dev))
    B = ctx.csr_from_torch(n, n, *gen_csr_device(torch, n, n, 0.01, 2, dev))
    out = torch.empty((m, n), dtype=torch.float64, device=dev)
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.dense_into(A, B, out.data_ptr(), exact=exact)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    alg = 12 * (A.nnz + B.nnz) + 8 * m * n
    print(json.dumps({"config": "c3 50k x 50k d=0.01 -> dense", "exact": exact, "ms": dt * 1e3, "algorithmic_GB": alg / 1e9,
                      "achieved_GBs": alg / dt / 1e9, "checksum": float(out.sum())}))
else:
    scale = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 1.0
    n, k = int(20000 * scale), int(80000 * scale)
    H = ctx.csr_from_torch(n, k, *gen_csr_device(torch, n, k, 0.02, 3, dev))
    # Q = S + S^T (symmetric, d ~ 0.005) assembled on the host with scipy, uploaded once
    import numpy as np
    import scipy.sparse as sp
    ip, ix, dv = (t.cpu().numpy() for t in gen_csr_device(torch, k, k, 0.0025, 4, dev))
    S = sp.csr_matrix((dv, ix, ip), shape=(k, k))
    Qs = (S + S.T).tocsr(); Qs.sort_indices()
    Q = ctx.csr_from_scipy(Qs)
    out = torch.empty((n, n), dtype=torch.float64, device=dev)
    for it in range(2):
        if it == 1:
            ctx.timing(True); ctx.timing_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.triple_into(H, Q, out.data_ptr(), exact=exact)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1 = ctx.kernel_time("smm_numeric_dense")[0]; s2 = ctx.kernel_time("smm_triple_stage2")[0]
    fma = H.nnz * (Q.nnz / k) + n * (n + 1) / 2 * (H.nnz / n)
    print(json.dumps({"config": f"c4 triple H {n}x{k} d=0.02, Q sym d~0.005", "exact": exact, "ms": dt * 1e3,
                      "stage1_ms": s1, "stage2_ms": s2,
                      "gather_fma": fma, "GFMA_s": fma / dt / 1e9, "checksum": float(out.sum())}))
