"""Every BASELINE.json config at its FULL size on the MI355X, through the v2 C ABI.

The CPU oracle cannot finish these sizes in seconds, so each case checks
  (1) a contiguous sample of rows against the oracle run on the very same arrays
      (indptr / indices bit-exact in first-touch order; values bit-exact with SMM_EXACT, within
      the north star's 1e-10 relative in the default mode), and
  (2) size-independent properties over the WHOLE result: nnz bookkeeping (indptr[-1] == symbolic
      nnz, > 2^31 at configs[1]), column range, no duplicate column inside sampled rows, the
      lower triangle exactly 0.0 for the upper-triangle outputs, and LINEARITY as a checksum of
      every value: C.1 = A.(B.1) (row sums), resp. (U + U^T - diag U).x = H.(Q.(H^T.x)).
Operands are generated on the device (sparse_matrix_mult_amd/synthetic.py: the distribution of
scipy.sparse.random) and copied to the host once for the oracle.
Reference workload definitions: /root/reference/tests/test_computation_speed.py:37-87 (same
products, smaller), src/sparsework.cpp:56-129, src/sparse_sparse_dense.cpp:108-129,185-220.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-10          # north star: float64 values within 1e-10 relative
LIN_RTOL = 1e-9       # linearity checksums re-associate sums of up to 1e6 positive terms
MODES = [pytest.param(False, id="default"), pytest.param(True, id="exact")]


@pytest.fixture(scope="module")
def tdev():
    import torch
    return torch, torch.device("cuda", 0)


def _host(t3):
    return tuple(t.cpu().numpy() for t in t3)


def _scipy(h3, shape):
    return sp.csr_matrix((h3[2], h3[1], h3[0]), shape=shape)


def _same(got, want, exact):
    if exact:
        assert np.array_equal(np.asarray(got).view(np.int64), np.asarray(want).view(np.int64)), \
            f"values differ bitwise (max rel {rel_err(got, want):.3e})"
    else:
        assert rel_err(got, want) <= RTOL, f"max rel {rel_err(got, want):.3e}"


def _check_sparse_sample(oracle, a_h, b_h, n, res, r0, r1, exact):
    """rows [r0, r1) of the device result (indptr int64, indices, data tensors) against the oracle."""
    indptr, indices, data = res
    ptr = indptr[r0:r1 + 1].cpu().numpy()
    lo, hi = int(ptr[0]), int(ptr[-1])
    idx, val = indices[lo:hi].cpu().numpy(), data[lo:hi].cpu().numpy()
    cnt, oidx, oval = oracle.sparse_rows(a_h, b_h, n, r0, r1)
    assert np.array_equal(np.diff(ptr), cnt), "per-row counts differ"
    assert np.array_equal(idx, oidx), "indices differ (first-touch order)"
    _same(val, oval, exact)
    for i in range(0, r1 - r0, max(1, (r1 - r0) // 16)):          # no duplicate column inside a row
        row = idx[ptr[i] - lo:ptr[i + 1] - lo]
        assert len(np.unique(row)) == len(row)


def _row_sums(torch, indptr, data, rows_per_chunk=2000):
    """Row sums of a device CSR (indptr int64): chunked cumsum + differences at the row ends."""
    m = indptr.numel() - 1
    out = torch.empty(m, dtype=torch.float64, device=data.device)
    for r0 in range(0, m, rows_per_chunk):
        r1 = min(m, r0 + rows_per_chunk)
        ptr = indptr[r0:r1 + 1]
        lo, hi = int(ptr[0]), int(ptr[-1])
        cs = torch.zeros(hi - lo + 1, dtype=torch.float64, device=data.device)
        torch.cumsum(data[lo:hi], 0, out=cs[1:])
        out[r0:r1] = cs[ptr[1:] - lo] - cs[ptr[:-1] - lo]
    return out


# ---------------------------------------------------------------------------------------------
# configs[1] and [2]: 50 000 x 50 000 times 50 000 x 50 000, d = 0.01
@pytest.fixture(scope="module")
def c1(ctx, tdev):
    torch, dev = tdev
    m = n = 50000
    a_t = gen(torch, m, n, 0.01, 1, dev)
    b_t = gen(torch, n, n, 0.01, 2, dev)
    a_h, b_h = _host(a_t), _host(b_t)
    A, B = ctx.csr_from_torch(m, n, *a_t), ctx.csr_from_torch(n, n, *b_t)
    As, Bs = _scipy(a_h, (m, n)), _scipy(b_h, (n, n))
    lin = As @ (Bs @ np.ones(n))                  # C.1 = A.(B.1)
    yield dict(m=m, n=n, A=A, B=B, a_h=a_h, b_h=b_h, lin=lin)
    A.close(); B.close()


def gen(torch, *a):
    from sparse_matrix_mult_amd.synthetic import gen_csr_device
    return gen_csr_device(torch, *a)


@pytest.mark.parametrize("exact", MODES)
def test_config1_50k_sparse(ctx, oracle, tdev, c1, exact):
    torch, dev = tdev
    m, n = c1["m"], c1["n"]
    plan = ctx.spgemm_plan(c1["A"], c1["B"], exact=exact)
    try:
        assert plan.nnz > 2 ** 31                                  # 2.48e9
        indptr = torch.empty(m + 1, dtype=torch.int64, device=dev)
        indices = torch.empty(plan.nnz, dtype=torch.int32, device=dev)
        data = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
        plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        ctx.synchronize()
        nnz = plan.nnz
    finally:
        plan.close()
    assert int(indptr[0]) == 0 and int(indptr[-1]) == nnz
    cnt = indptr[1:] - indptr[:-1]
    assert int(cnt.min()) >= 0 and int(cnt.max()) <= n
    assert int(indices.min()) >= 0 and int(indices.max()) < n
    # expected fill 1 - (1 - d^2)^n = 0.99326
    assert abs(nnz / (m * n) - 0.99326) < 2e-4
    for r0 in (0, 31337):                                          # two contiguous samples, 500 rows in all
        _check_sparse_sample(oracle, c1["a_h"], c1["b_h"], n, (indptr, indices, data), r0, r0 + 250, exact)
    rs = _row_sums(torch, indptr, data).cpu().numpy()
    assert rel_err(rs, c1["lin"]) <= LIN_RTOL


@pytest.mark.parametrize("exact", MODES)
@pytest.mark.parametrize("symmetric", [False, True])
def test_config2_50k_dense(ctx, oracle, tdev, c1, exact, symmetric):
    torch, dev = tdev
    m, n = c1["m"], c1["n"]
    out = torch.empty((m, n), dtype=torch.float64, device=dev)
    ctx.dense_into(c1["A"], c1["B"], out.data_ptr(), symmetric=symmetric, exact=exact)
    ctx.synchronize()
    for r0 in (0, 40000):
        want = oracle.dense(c1["a_h"], c1["b_h"], n, symmetric=symmetric, row_begin=r0, row_end=r0 + 150)
        _same(out[r0:r0 + 150].cpu().numpy(), want, exact)
    if symmetric:
        for r0 in range(0, m, 5000):                               # lower triangle exactly 0.0 everywhere
            blk = out[r0:r0 + 5000]
            assert not bool(blk[:, :r0].any())
            assert not bool(torch.tril(blk[:, r0:r0 + 5000], -1).any())
    else:
        assert rel_err(out.sum(dim=1).cpu().numpy(), c1["lin"]) <= LIN_RTOL


# ---------------------------------------------------------------------------------------------
# configs[3]: H Q H^T, H 20 000 x 80 000 d = 0.02, Q 80 000 x 80 000 symmetric d = 0.005
@pytest.mark.parametrize("exact", MODES)
def test_config3_triple_product(ctx, oracle, tdev, exact):
    from sparse_matrix_mult_amd.synthetic import gen_symmetric_csr_device
    torch, dev = tdev
    n, k = 20000, 80000
    h_t = gen(torch, n, k, 0.02, 3, dev)
    q_t = gen_symmetric_csr_device(torch, k, 0.005, 4, dev)
    h_h, q_h = _host(h_t), _host(q_t)
    Qs = _scipy(q_h, (k, k))
    assert abs(Qs - Qs.T).nnz == 0 and abs(Qs.nnz / (k * k) - 0.005) < 2e-4
    H, Q = ctx.csr_from_torch(n, k, *h_t), ctx.csr_from_torch(k, k, *q_t)
    try:
        out = torch.empty((n, n), dtype=torch.float64, device=dev)
        ctx.triple_into(H, Q, out.data_ptr(), exact=exact)
        ctx.synchronize()
    finally:
        H.close(); Q.close()
    for r0, nr in ((6000, 12), (19900, 100)):                      # stage 2 of an early row is 22 M gather-FMAs on the CPU
        want = oracle.triple(h_h, q_h, k, 0, r0, r0 + nr)[r0:r0 + nr]
        got = out[r0:r0 + nr].cpu().numpy()
        # stage 2 sums in H's stored order in both modes; stage 1 (T = H Q) follows the mode
        _same(got, want, exact)
    assert not bool(torch.tril(out, -1).any())                     # compute_full_matrix=0: lower triangle 0.0
    # linearity over the whole result: (U + U^T - diag U) x = H (Q (H^T x))
    Hs = _scipy(h_h, (n, k))
    x = np.random.default_rng(7).random(n)
    want = Hs @ (Qs @ (Hs.T @ x))
    xt = torch.from_numpy(x).to(dev)
    got = (out @ xt + out.T @ xt - torch.diagonal(out) * xt).cpu().numpy()
    assert rel_err(got, want) <= LIN_RTOL


# ---------------------------------------------------------------------------------------------
# configs[4]: 200 000 x 200 000, d = 0.005, row-sharded x8: ONE rank's share (25 000 rows) on one GPU
@pytest.mark.parametrize("exact", MODES)
def test_config4_one_rank_share_of_200k(ctx, oracle, tdev, exact):
    torch, dev = tdev
    n, world, rank = 200000, 8, 3
    m = n // world
    a_t = gen(torch, m, n, 0.005, 1 + 1000 * rank, dev)            # rows [rank*m, (rank+1)*m) of the global A
    b_t = gen(torch, n, n, 0.005, 2, dev)
    a_h, b_h = _host(a_t), _host(b_t)
    A, B = ctx.csr_from_torch(m, n, *a_t), ctx.csr_from_torch(n, n, *b_t)
    try:
        indptr, indices, data = ctx.spgemm_torch(A, B, row_offset=rank * m, exact=exact)
        ctx.synchronize()
    finally:
        A.close(); B.close()
    nnz = int(indptr[-1])
    assert nnz == indices.numel() == data.numel() and nnz > 2 ** 32          # ~4.97e9 per rank
    assert abs(nnz / (m * n) - 0.99326) < 2e-4
    assert int(indices.min()) >= 0 and int(indices.max()) < n
    _check_sparse_sample(oracle, a_h, b_h, n, (indptr, indices, data), 12345, 12345 + 120, exact)
    lin = _scipy(a_h, (m, n)) @ (_scipy(b_h, (n, n)) @ np.ones(n))
    assert rel_err(_row_sums(torch, indptr, data).cpu().numpy(), lin) <= LIN_RTOL


def test_legacy_abi_refuses_a_result_beyond_int32_loudly(c1, capfd):
    """SURVEY 8b "Limits": the reference's structs hold nnz / rowPtr as int.  configs[1] has 2.48e9 output
    nonzeros: through the legacy symbol the call must fail LOUDLY -- a message on stderr and an empty output
    struct (nzmax 0, NULL arrays), never a truncated matrix (reference error convention)."""
    import ctypes

    class SparseMat(ctypes.Structure):
        _fields_ = [("nzmax", ctypes.c_int), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                    ("rowPtr", ctypes.POINTER(ctypes.c_int)), ("colInd", ctypes.POINTER(ctypes.c_int)),
                    ("values", ctypes.POINTER(ctypes.c_double))]
    from sparse_matrix_mult_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    sp_ = ctypes.POINTER(SparseMat)
    lib.create_sparsemat.argtypes = [ctypes.c_int] * 3; lib.create_sparsemat.restype = sp_
    lib.sparse_nosym.argtypes = [sp_, sp_, sp_, ctypes.c_int]; lib.sparse_nosym.restype = None
    lib.destroy_sparsemat.argtypes = [sp_]
    ops = []
    for h in (c1["a_h"], c1["b_h"]):
        s = lib.create_sparsemat(c1["m"], c1["n"], int(h[0][-1])).contents
        for dst, src in ((s.rowPtr, h[0]), (s.colInd, h[1]), (s.values, h[2])):
            ctypes.memmove(dst, src.ctypes.data, src.nbytes)
        ops.append(s)
    out = SparseMat()
    lib.sparse_nosym(ctypes.byref(ops[0]), ctypes.byref(ops[1]), ctypes.byref(out), 5)
    err = capfd.readouterr().err
    assert out.nzmax == 0 and not out.colInd and not out.values
    assert "2^31" in err or "int32" in err or "INT_MAX" in err or "overflow" in err.lower(), err
    for s in ops:
        lib.destroy_sparsemat(ctypes.byref(s))


# ---------------------------------------------------------------------------------------------
# configs[1] on the north star's LITERAL operands: scipy.sparse.random(50000, 50000, 0.01, random_state=
# default_rng(1 | 2)) (SURVEY 8d's seeds: A = 1, B = 2), uploaded from the host like a caller's matrices.
# Generation costs ~10 s per matrix on the host (sampling 2.5e7 of 2.5e9 cells without replacement) -- the
# reason the other full-size cases draw their operands on the device (same distribution, other stream).
def test_config1_on_literal_scipy_sparse_random_operands(ctx, oracle, tdev):
    import time
    from helpers import arrays
    torch, dev = tdev
    m = n = 50000
    t0 = time.perf_counter()
    A = sp.random(m, n, density=0.01, format="csr", random_state=np.random.default_rng(1), dtype=np.float64)
    B = sp.random(n, n, density=0.01, format="csr", random_state=np.random.default_rng(2), dtype=np.float64)
    print(f"scipy.sparse.random x2: {time.perf_counter() - t0:.1f} s")
    assert A.nnz == B.nnz == 25000000 and A.has_sorted_indices and B.has_sorted_indices
    a_h, b_h = arrays(A), arrays(B)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        indptr, indices, data = ctx.spgemm_torch(a, b)
        ctx.synchronize()
    finally:
        a.close(); b.close()
    nnz = int(indptr[-1])
    assert nnz > 2 ** 31 and abs(nnz / (m * n) - 0.99326) < 2e-4
    for r0 in (7, 44444):
        _check_sparse_sample(oracle, a_h, b_h, n, (indptr, indices, data), r0, r0 + 120, False)
    lin = A @ (B @ np.ones(n))
    assert rel_err(_row_sums(torch, indptr, data).cpu().numpy(), lin) <= LIN_RTOL
