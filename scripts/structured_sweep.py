"""Round 4: operands with STRUCTURE through the same engine -- BASELINE's configs are uniform random, the reference's
users (H Q H^T with covariance matrices) are not.  For every family: device-resident CSR x CSR -> CSR (plan + numeric,
best of 3), the kernels that ran, nnz/s, algorithmic GB/s (SURVEY 8d bytes), and a check: C x = A (B x) for a random x
(every family) and nnz(C) = scipy's structural product where that finishes in seconds.

    python scripts/structured_sweep.py [family ...]      # on the GPU box; one JSON line per family
"""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import torch  # noqa: E402
from sparse_matrix_mult_amd import engine  # noqa: E402

KERNELS = ("smm_validate", "smm_segptr", "smm_idx16", "smm_pack_fill", "smm_ccs_fill", "smm_row_work", "smm_scan", "smm_bin_rows",
           "smm_symbolic", "smm_symbolic_hash", "smm_symbolic_tiny", "smm_numeric_tiny", "smm_copy_lists", "smm_runs", "smm_numeric", "smm_numeric_hash", "smm_numeric_general",
           "smm_dense_slab", "smm_emit", "smm_plan_check")


def banded(n, half, rng):
    offs = np.arange(-half, half + 1)
    diags = [rng.standard_normal(n - abs(o)) for o in offs]
    return sp.diags(diags, offs, shape=(n, n), format="csr")


def powerlaw(n, avg, rng, alpha=1.6, cap=None):
    w = rng.pareto(alpha, n) + 1.0
    lens = np.minimum(np.maximum((w * avg / w.mean()).astype(np.int64), 1), cap or n)
    indptr = np.concatenate(([0], np.cumsum(lens)))
    cols = np.concatenate([np.sort(rng.choice(n, int(k), replace=False)) for k in lens])
    return sp.csr_matrix((rng.standard_normal(indptr[-1]), cols.astype(np.int32), indptr), shape=(n, n))


def blockdiag(nb, bs, d, rng):
    return sp.block_diag([sp.random(bs, bs, density=d, format="csr", random_state=rng) for _ in range(nb)], format="csr")


def arrow(n, rng):
    A = sp.lil_matrix((n, n))
    A.setdiag(rng.standard_normal(n))
    A[0, :] = rng.standard_normal(n)
    A[:, 0] = rng.standard_normal((n, 1))
    return A.tocsr()


def two_per_row(n, rng):
    c1 = rng.integers(0, n, size=n, dtype=np.int64)
    c2 = (c1 + 1 + rng.integers(0, n - 1, size=n, dtype=np.int64)) % n          # != c1
    cols = np.stack([np.minimum(c1, c2), np.maximum(c1, c2)], axis=1).reshape(-1).astype(np.int32)
    return sp.csr_matrix((rng.standard_normal(2 * n), cols, np.arange(0, 2 * n + 1, 2, dtype=np.int64)), shape=(n, n))


def families(rng):
    yield "uniform 20k d=0.01", lambda: (sp.random(20000, 20000, 0.01, "csr", random_state=rng),) * 2
    yield "banded n=2M half-width 8", lambda: (banded(2_000_000, 8, rng),) * 2
    yield "banded n=100k half-width 200", lambda: (banded(100_000, 200, rng),) * 2
    yield "power-law rows n=100k avg 20 (max 20k)", lambda: (powerlaw(100_000, 20, rng, cap=20000),) * 2
    yield "block diagonal 500 x (400 x 400, d=0.5)", lambda: (blockdiag(500, 400, 0.5, rng),) * 2
    yield "hypersparse n=4M, 4 per row", lambda: (sp.random(4_000_000, 4_000_000, 1e-6, "csr", random_state=rng),) * 2
    yield "tall-skinny 1M x 2000 d=0.005 times 2000 x 2000 d=0.05", lambda: (
        sp.random(1_000_000, 2000, 0.005, "csr", random_state=rng), sp.random(2000, 2000, 0.05, "csr", random_state=rng))
    yield "arrow n=30k (dense first row and column)", lambda: (arrow(30000, rng),) * 2
    yield "30M rows, 2 per row", lambda: (two_per_row(30_000_000, rng),) * 2


def main():
    want = sys.argv[1:]
    rng = np.random.default_rng(7)
    ctx = engine.Context(0)
    ctx.set_check(True)
    ctx.timing(True)
    for name, make in families(rng):
        if want and not any(w in name for w in want):
            continue
        A, B = make()
        A.sort_indices(); B.sort_indices()
        a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
        products = int(ctx.row_products(a, b).sum())
        best, kern = 1e30, {}
        for rep in range(3):
            ctx.timing_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            indptr, indices, data = ctx.spgemm_torch(a, b)
            ctx.synchronize(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dt < best:
                best = dt
                kern = {k: round(ctx.kernel_time(k)[0], 3) for k in KERNELS if ctx.kernel_time(k)[1]}
            if rep < 2:
                del indptr, indices, data
        nnz = int(indptr[-1].item())
        # check: C x = A (B x)
        x = rng.standard_normal(B.shape[1])
        want_y = A @ (B @ x)
        xd = torch.from_numpy(x).to(data.device)
        rows = torch.repeat_interleave(torch.arange(A.shape[0], device=data.device), indptr[1:] - indptr[:-1])
        got_y = torch.zeros(A.shape[0], dtype=torch.float64, device=data.device).index_add_(0, rows, data * xd[indices.long()]).cpu().numpy()
        scale = np.abs(A) @ (np.abs(B) @ np.abs(x)) + 1e-300
        err = float(np.max(np.abs(got_y - want_y) / scale))
        rec = {"family": name, "rows": A.shape[0], "nnzA": int(A.nnz), "nnzB": int(B.nnz), "products": products, "nnzC": nnz,
               "ms": round(best * 1e3, 3), "kernels_ms": kern, "nnz_per_s": round(nnz / best, 1),
               "algorithmic_GBps": round((12 * (A.nnz + B.nnz + nnz) + 16 * A.shape[0]) / best / 1e9, 1),
               "checksum_rel_err": err}
        if products <= 400_000_000:
            t0 = time.perf_counter()
            S = A @ B
            rec["scipy_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            # scipy drops nothing structurally here (no explicit zeros are eliminated by A @ B)
            rec["nnz_equals_scipy"] = bool(S.nnz == nnz)
        assert err < 1e-10, rec
        print(json.dumps(rec), flush=True)
        del indptr, indices, data, rows, xd
        a.close(); b.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
