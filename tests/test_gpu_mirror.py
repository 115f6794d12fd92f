"""SURVEY 8f-2 for CSR results: callers of symmetric=True get only i <= col (src/sparsework.cpp:217); the opt-in
mirror epilogue returns the full symmetric CSR, built on the device.  Parity: for a symmetric product the
mirrored matrix has exactly the pattern and the values of the reference's symmetric=False product (bit for bit
with SMM_EXACT: C[i,j] and C[j,i] sum the same products in the same order), in the documented order: row i =
mirrored entries (columns < i) ascending, then the reference's upper-triangle row in first-touch order."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, rand_csr, wide_csr

pytestmark = pytest.mark.gpu


def _sorted_rows(ptr, idx, val):
    out_i, out_v = idx.copy(), val.copy()
    for r in range(len(ptr) - 1):
        s, e = int(ptr[r]), int(ptr[r + 1])
        o = np.argsort(idx[s:e], kind="stable")
        out_i[s:e], out_v[s:e] = idx[s:e][o], val[s:e][o]
    return out_i, out_v


CASES = [
    pytest.param(lambda: wide_csr(3000, 40000, 10, 1), id="wide-hash-small-rows"),
    pytest.param(lambda: rand_csr(900, 5000, 0.004, 2), id="hash-medium-rows"),
    pytest.param(lambda: rand_csr(600, 400, 0.05, 3), id="segments-over-64-lds-sort"),
    pytest.param(lambda: rand_csr(64, 30, 0.3, 4), id="tiny"),
]


@pytest.mark.parametrize("make", CASES)
@pytest.mark.parametrize("exact", [False, True])
def test_mirrored_csr_equals_the_full_product(oracle, make, exact):
    import sparse_matrix_mult_amd as pkg
    A = make()
    At = A.T.tocsr()
    At.sort_indices()
    n = A.shape[0]
    old_e, old_f = pkg.set_exact(exact), pkg.set_full_symmetric(True)
    try:
        full = pkg.sparse_matrix_multiply(A, At, symmetric=True)
        pkg.set_full_symmetric(False)
        upper = pkg.sparse_matrix_multiply(A, At, symmetric=True)
    finally:
        pkg.set_exact(old_e); pkg.set_full_symmetric(old_f)
    want = oracle.sparse(arrays(A), arrays(At), n, symmetric=False)
    assert np.array_equal(full.indptr, want[0])                      # same row lengths as the full product
    gi, gv = _sorted_rows(full.indptr, full.indices, full.data)
    wi, wv = _sorted_rows(*want)
    assert np.array_equal(gi, wi)
    if exact:
        assert np.array_equal(gv.view(np.int64), wv.view(np.int64))
    else:
        assert np.allclose(gv, wv, rtol=1e-10, atol=0)
    # the documented order: mirrored part ascending, then the upper row as it stands
    for r in list(range(0, n, max(1, n // 50))) + [n - 1]:
        row = full.indices[full.indptr[r]:full.indptr[r + 1]]
        own = upper.indices[upper.indptr[r]:upper.indptr[r + 1]]
        k = len(row) - len(own)
        assert np.array_equal(row[k:], own) and (row[:k] < r).all() and (np.diff(row[:k]) > 0).all()
        fo, uo = full.data[full.indptr[r]:full.indptr[r + 1]][k:], upper.data[upper.indptr[r]:upper.indptr[r + 1]]
        assert np.array_equal(fo, uo) if exact else np.allclose(fo, uo, rtol=1e-10, atol=0)      # (two runs: default-mode sums differ in rounding)
    assert abs(full - full.T).nnz == 0                               # symmetric, value for value


def test_mirror_refuses_what_it_cannot_sort_and_what_is_not_upper(ctx):
    from sparse_matrix_mult_amd.engine import SmmError
    A = rand_csr(9000, 40, 0.6, 5)                                   # A A^T is full: the last rows get ~9000 mirrored entries
    At = A.T.tocsr()
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(At)
    try:
        with pytest.raises(SmmError, match="mirrored entries"):
            ctx.spgemm_host_mirrored(a, b)
    finally:
        a.close(); b.close()
    # a CSR with an entry left of the diagonal is not an upper-triangle result
    M = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [3.0, 4.0, 0.0], [0.0, 0.0, 5.0]]))
    ptr, idx = M.indptr.astype(np.int64), M.indices.astype(np.int32)
    lib = ctx.lib
    bufs = []
    try:
        d = []
        for arr in (ptr, idx):
            p = ctypes.c_void_p()
            assert lib.smm_device_malloc(ctx.handle, arr.nbytes, ctypes.byref(p)) == 0
            assert lib.smm_memcpy_h2d(ctx.handle, p, ctypes.c_void_p(arr.ctypes.data), arr.nbytes) == 0
            d.append(p); bufs.append(p)
        fp = ctypes.c_void_p()
        assert lib.smm_device_malloc(ctx.handle, 8 * 4, ctypes.byref(fp)) == 0
        bufs.append(fp)
        nnz = ctypes.c_int64()
        rc = lib.smm_csr_mirror_symbolic(ctx.handle, 3, d[0], d[1], fp, ctypes.byref(nnz))
        assert rc == -2 and b"left of the diagonal" in lib.smm_last_error()
    finally:
        for p in bufs:
            lib.smm_device_free(ctx.handle, p)


def test_mirror_off_by_default(oracle):
    import sparse_matrix_mult_amd as pkg
    A = rand_csr(200, 150, 0.05, 6)
    At = A.T.tocsr()
    C = pkg.sparse_matrix_multiply(A, At, symmetric=True)            # the reference's behaviour: upper triangle only
    want = oracle.sparse(arrays(A), arrays(At), 200, symmetric=True)
    assert np.array_equal(C.indptr, want[0]) and np.array_equal(C.indices, want[1])
