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


def _oracle_rows_parallel(oracle, a, b, n, threads=16):
    """The whole symmetric=False product from the oracle, row ranges on host threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    m = len(a[0]) - 1
    bounds = [m * t // threads for t in range(threads + 1)]
    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(lambda t: oracle.sparse_rows(a, b, n, bounds[t], bounds[t + 1]), range(threads)))
    cnt = np.concatenate([p[0] for p in parts])
    return (np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]))


def _check_full_against_oracle(oracle, A, exact, sample_rows=None):
    """sparse_matrix_multiply(A, A^T, symmetric=True) with the mirror epilogue == the reference's symmetric=False product:
    row pointer, pattern and values (bit for bit with SMM_EXACT), in the documented order.  sample_rows: compare that
    many rows with the oracle instead of all of them, plus whole-result checksums (results of 1e8+ entries)."""
    import sparse_matrix_mult_amd as pkg
    At = A.T.tocsr()
    At.sort_indices()
    n = A.shape[0]
    old_e, old_f = pkg.set_exact(exact), pkg.set_full_symmetric(True)
    try:
        full = pkg.sparse_matrix_multiply(A, At, symmetric=True)
    finally:
        pkg.set_exact(old_e); pkg.set_full_symmetric(old_f)
        pkg.clear_cache()
    ptr = full.indptr.astype(np.int64)
    # the documented order over the WHOLE result: inside a row first the mirrored entries (columns < row) in strictly
    # ascending column order, then the row's own entries (columns >= row)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(ptr))
    low = full.indices < rows
    same = rows[1:] == rows[:-1]
    assert not (low[1:] & ~low[:-1] & same).any()                           # no mirrored entry behind an own one
    assert (full.indices[1:] > full.indices[:-1])[low[1:] & low[:-1] & same].all()
    del rows, low, same
    a_, b_ = arrays(A), arrays(At)
    if sample_rows is None:
        wp, wi, wv = _oracle_rows_parallel(oracle, a_, b_, n)
        assert np.array_equal(ptr, wp)
        want = sp.csr_matrix((wv, wi, wp), shape=(n, n)); want.sort_indices()
        got = sp.csr_matrix((full.data.copy(), full.indices.copy(), full.indptr.copy()), shape=(n, n)); got.sort_indices()
        assert np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data.view(np.int64), want.data.view(np.int64)) if exact else np.allclose(got.data, want.data, rtol=1e-10, atol=0)
    else:
        r = np.random.default_rng(5)
        for row in sorted(set(r.integers(0, n, size=sample_rows).tolist()) | {0, n - 1}):
            cnt, wi, wv = oracle.sparse_rows(a_, b_, n, row, row + 1)
            gi, gv = full.indices[ptr[row]:ptr[row + 1]], full.data[ptr[row]:ptr[row + 1]]
            assert len(gi) == int(cnt[0])
            og, ow = np.argsort(gi, kind="stable"), np.argsort(wi, kind="stable")
            assert np.array_equal(gi[og], wi[ow])
            assert np.array_equal(gv[og].view(np.int64), wv[ow].view(np.int64)) if exact else np.allclose(gv[og], wv[ow], rtol=1e-10, atol=0)
        # whole result: C x = A (A^T x), and u^T C w = w^T C u (symmetric)
        x, u = r.uniform(0.5, 1.5, n), r.uniform(0.5, 1.5, n)
        assert np.allclose(full @ x, A @ (At @ x), rtol=1e-9, atol=0)
        assert np.isclose(u @ (full @ x), x @ (full @ u), rtol=1e-11, atol=0)
    return int(np.diff(ptr).max())


@pytest.mark.parametrize("exact", [False, True])
def test_mirror_of_full_rows_ranks_the_long_segments(oracle, exact):
    """Round 4: no limit on a row.  A A^T of a 9000 x 40 matrix at d = 0.6 is full: rows beyond 8192 receive more
    mirrored entries than the LDS sort holds (round 3 refused them) and are placed by rank (smm_mirror_rank); the
    rows before them still take the LDS sorts.  The whole result is compared with the oracle's symmetric=False product."""
    assert _check_full_against_oracle(oracle, rand_csr(9000, 40, 0.6, 5), exact) == 9000


def test_mirror_at_baseline_fill_20000(oracle):
    """The verdict's case: 20 000 x 20 000, d = 0.02 (n d^2 = 8: rows 99.97 % full, up to 19 999 mirrored entries per
    row, 4e8 nonzeros): pattern and values of oracle.sparse(symmetric=False), bit for bit under SMM_EXACT, on 300
    sampled rows; order, C x = A (A^T x) and symmetry on the whole result."""
    assert _check_full_against_oracle(oracle, rand_csr(20000, 20000, 0.02, 7), True, sample_rows=300) > 19900


def test_mirror_rank_walks_rows_wider_than_one_bitmap_range(oracle):
    """600 000 rows: 8300 of them share column 0, so the last of those receives 8299 mirrored entries whose columns
    span more than the 524 288 columns one pass of the rank kernel's bitmap covers."""
    n, k = 600000, 8300
    r = np.random.default_rng(11)
    shared = np.unique(np.concatenate([r.choice(n - 1, size=k - 1, replace=False), [n - 1]]))
    own = sp.csr_matrix((r.uniform(0.5, 1.5, n), np.arange(1, n + 1, dtype=np.int32), np.arange(n + 1, dtype=np.int32)),
                        shape=(n, n + 1))                                  # a private column for every row ...
    extra = sp.csr_matrix((r.uniform(0.5, 1.5, len(shared)), (shared, np.zeros(len(shared), dtype=np.int64))), shape=(n, n + 1))
    A = (own + extra).tocsr(); A.sort_indices()                           # ... and column 0 for the shared ones
    assert _check_full_against_oracle(oracle, A, True) == len(shared)


def test_mirror_refuses_what_is_not_upper(ctx):
    M = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [3.0, 4.0, 0.0], [0.0, 0.0, 5.0]]))
    # a CSR with an entry left of the diagonal is not an upper-triangle result
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
