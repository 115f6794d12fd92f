"""The reference's own test-suite, restated against this package's drop-in API and run on the
MI355X (reference tests/test_matrix_multiply.py, test_edge_case.py, test_computation_speed.py,
test_with_dense.py, test_basic.py).  Same matrices, same assertions (np.allclose against
numpy/scipy), plus what the reference never checks: return types, dtypes, index order and the
legacy C symbols driven exactly as the reference's matrix_ops.py drives them."""
import ctypes
import os

import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse import csr_matrix

from helpers import arrays, rand_csr

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))


def _dense(case, tag):
    shape = tuple(GOLD[f"{case}/{tag}_shape"])
    return csr_matrix((GOLD[f"{case}/{tag}_data"], GOLD[f"{case}/{tag}_indices"], GOLD[f"{case}/{tag}_indptr"]),
                      shape=shape).toarray()


@pytest.fixture(scope="module")
def smm():
    from sparse_matrix_mult_amd import sparse_matrix_multiply
    return sparse_matrix_multiply


# ---- reference tests/test_matrix_multiply.py:89-112 (dense ndarray inputs, ints included)
def test_sparse_nonsym(smm):
    C, D = _dense("ref_CxD", "a"), _dense("ref_CxD", "b")
    result = smm(C.astype(np.int64), D, output_format='sparse', symmetric=False)
    expected = np.matmul(C, D)
    assert sp.isspmatrix_csr(result) and result.dtype == np.float64 and result.indices.dtype == np.int32
    assert result.shape == expected.shape
    assert np.allclose(result.toarray(), expected)


def test_dense_nonsym(smm):
    C, D = _dense("ref_CxD", "a"), _dense("ref_CxD", "b")
    result = smm(C, D, output_format='dense', symmetric=False)
    assert isinstance(result, np.ndarray) and result.flags.c_contiguous and result.dtype == np.float64
    assert np.allclose(result, np.matmul(C, D))


def test_dense_sym(smm):
    C, F = _dense("ref_CxF_square", "a"), _dense("ref_CxF_square", "b")
    result = smm(C, F, output_format='dense', symmetric=True)
    expected = np.matmul(C, F)
    assert result.shape == expected.shape
    assert np.allclose(np.triu(result), np.triu(expected))
    assert np.all(np.tril(result, -1) == 0.0)          # the reference leaves calloc's zeros there


def test_sparse_sym(smm):
    C, F = _dense("ref_CxF_square", "a"), _dense("ref_CxF_square", "b")
    result = smm(C, F, output_format='sparse', symmetric=True)
    expected = np.matmul(C, F)
    assert np.allclose(np.triu(result.toarray()), np.triu(expected))
    assert np.all(np.tril(result.toarray(), -1) == 0.0)


# ---- reference tests/test_edge_case.py:42-71
def test_one_by_one_multiplication(smm):
    assert np.allclose(smm(np.array([[5]]), np.array([[2]]), output_format='dense', symmetric=True), [[10]])


def test_matrix_with_zero_rows(smm):
    Z = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [0, 0, 0], [0, 0, 0], [0, 0, 0]])
    B = np.random.default_rng(0).random((3, 4))
    assert np.allclose(smm(Z, B, output_format='dense'), Z @ B)
    r = smm(csr_matrix(Z), csr_matrix(B), output_format='sparse')
    assert r.shape == (6, 4) and np.allclose(r.toarray(), Z @ B)
    assert np.array_equal(np.diff(r.indptr), [4, 4, 4, 0, 0, 0])


# ---- reference tests/test_computation_speed.py:37-87 (500 x 500, density 0.3, uniform values)
@pytest.fixture(scope="module")
def speed_matrices():
    return rand_csr(500, 500, 0.3, 42), rand_csr(500, 500, 0.3, 43)


@pytest.mark.parametrize("output_format", ["sparse", "dense"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_speed_suite(smm, speed_matrices, output_format, symmetric):
    A, B = speed_matrices
    got = smm(A, B, output_format=output_format, symmetric=symmetric)
    got = got.toarray() if output_format == "sparse" else got
    want = A.dot(B).toarray()
    if symmetric:
        assert np.allclose(np.triu(want), np.triu(got))
    else:
        assert np.allclose(want, got)


def test_triple_product(smm, speed_matrices):
    A, B = speed_matrices
    got = smm(A, B, use_triple_product=True, compute_full_matrix=0)
    want = A.dot(B).dot(A.transpose()).toarray()
    assert got.ndim == 2 and np.allclose(np.triu(want), np.triu(got))
    # output_format / symmetric are ignored on this route (reference matrix_ops.py:325)
    again = smm(A, B, output_format='sparse', use_triple_product=True)
    assert isinstance(again, np.ndarray) and np.allclose(np.triu(again), np.triu(got), rtol=1e-12)


def test_triple_product_full_matrix_reproduces_reference(smm, oracle):
    """compute_full_matrix=1: off-diagonals hold S[i,k]+S[k,i], exactly as the reference."""
    H = rand_csr(40, 60, 0.2, 7); S = rand_csr(60, 60, 0.05, 8); Q = (S + S.T).tocsr()
    got = smm(H, Q, use_triple_product=True, compute_full_matrix=1)
    want = oracle.triple(arrays(H), arrays(Q), 60, full=1)
    assert np.allclose(got, want, rtol=1e-10, atol=0)


# ---- reference tests/test_with_dense.py:30-109 (its `.A` is `.toarray()` on current SciPy)
@pytest.mark.parametrize("shape_a,shape_b,density", [((5, 5), (5, 5), 0.3), ((6, 6), (6, 6), 0.1),
                                                     ((500, 400), (400, 500), 0.1), ((1000, 1000), (1000, 1000), 0.01)])
def test_with_dense_suite(smm, shape_a, shape_b, density):
    A = sp.random(*shape_a, density=density, format='csr', random_state=np.random.default_rng(1))
    B = sp.random(*shape_b, density=density, format='csr', random_state=np.random.default_rng(2))
    got = smm(A, B, output_format='sparse', symmetric=False)
    assert np.allclose(got.toarray(), A.dot(B).toarray())


def test_identity(smm):
    A = rand_csr(500, 500, 0.1, 3)
    got = smm(A, sp.identity(500, format='csr'))
    assert np.allclose(got.toarray(), A.toarray())
    assert np.array_equal(got.indices, A.indices) and np.array_equal(got.indptr, A.indptr)


def test_result_is_in_first_touch_order_and_matches_oracle(smm, oracle):
    from sparse_matrix_mult_amd import set_exact
    A, B = rand_csr(300, 250, 0.05, 5), rand_csr(250, 300, 0.05, 6)
    want = oracle.sparse(arrays(A), arrays(B), 300)
    old = set_exact(True)
    try:
        got = smm(A, B)
    finally:
        set_exact(old)
    assert not got.has_sorted_indices
    assert np.array_equal(got.indptr, want[0]) and np.array_equal(got.indices, want[1])
    assert np.array_equal(got.data, want[2])


def test_other_input_formats_are_coerced(smm):
    A, B = rand_csr(60, 50, 0.2, 9), rand_csr(50, 70, 0.2, 10)
    want = (A @ B).toarray()
    for a in (A.tocoo(), A.tocsc(), A.toarray()):
        assert np.allclose(smm(a, B.tolil(), output_format='dense'), want)


# ---- the legacy C ABI, driven exactly as the reference's matrix_ops.py:187-240,338-365 drives it
class SparseMat(ctypes.Structure):
    _fields_ = [("nzmax", ctypes.c_int), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("rowPtr", ctypes.POINTER(ctypes.c_int)), ("colInd", ctypes.POINTER(ctypes.c_int)),
                ("values", ctypes.POINTER(ctypes.c_double))]


class DArray(ctypes.Structure):
    _fields_ = [("array", ctypes.POINTER(ctypes.c_double)), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


@pytest.fixture(scope="module")
def legacy():
    from sparse_matrix_mult_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    sp_, dp_ = ctypes.POINTER(SparseMat), ctypes.POINTER(DArray)
    lib.create_sparsemat.argtypes = [ctypes.c_int] * 3; lib.create_sparsemat.restype = sp_
    for f in ("sparse_nosym", "sparse_sym"):
        getattr(lib, f).argtypes = [sp_, sp_, sp_, ctypes.c_int]; getattr(lib, f).restype = None
    for f in ("sparsework_nosym", "sparsework_sym"):
        getattr(lib, f).argtypes = [sp_, sp_, sp_, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    for f in ("dense_nosym", "dense_sym"):
        getattr(lib, f).argtypes = [sp_, sp_, dp_]; getattr(lib, f).restype = None
    lib.triple_product.argtypes = [sp_, sp_, dp_, ctypes.c_int]
    lib.destroy_sparsemat.argtypes = [sp_]; lib.destroy_darray.argtypes = [dp_]
    return lib


def _to_sparsemat(lib, m):          # reference csr_to_sparsemat, matrix_ops.py:187-202
    s = lib.create_sparsemat(m.shape[0], m.shape[1], m.nnz).contents
    ip, ix, dv = arrays(m)
    ctypes.memmove(s.rowPtr, ip.ctypes.data, ip.nbytes)
    ctypes.memmove(s.colInd, ix.ctypes.data, ix.nbytes)
    ctypes.memmove(s.values, dv.ctypes.data, dv.nbytes)
    return s


def _from_sparsemat(s):             # reference sparsemat_to_csr, matrix_ops.py:205-228
    if s.nzmax == 0:
        return np.zeros(s.rows + 1, np.int32), np.zeros(0, np.int32), np.zeros(0)
    return (np.ctypeslib.as_array(s.rowPtr, shape=(s.rows + 1,)).copy(),
            np.ctypeslib.as_array(s.colInd, shape=(s.nzmax,)).copy(),
            np.ctypeslib.as_array(s.values, shape=(s.nzmax,)).copy())


@pytest.mark.parametrize("symmetric", [False, True])
def test_legacy_sparse_symbols(legacy, oracle, symmetric):
    A, B = rand_csr(120, 90, 0.1, 11), rand_csr(90, 120, 0.1, 12)
    a, b, out = _to_sparsemat(legacy, A), _to_sparsemat(legacy, B), SparseMat()
    (legacy.sparse_sym if symmetric else legacy.sparse_nosym)(ctypes.byref(a), ctypes.byref(b), ctypes.byref(out), 5)
    ptr, idx, val = _from_sparsemat(out)
    want = oracle.sparse(arrays(A), arrays(B), 120, symmetric=symmetric)
    assert (out.rows, out.cols, out.nzmax) == (120, 120, want[0][-1])
    assert np.array_equal(ptr, want[0]) and np.array_equal(idx, want[1]) and np.allclose(val, want[2], rtol=1e-10, atol=0)
    legacy.destroy_sparsemat(ctypes.byref(out))
    assert out.nzmax == 0 and not out.colInd
    # sparsework_*: inclusive row range, per-row COUNTS in rowPtr (sparsework.cpp:116)
    part = SparseMat()
    (legacy.sparsework_sym if symmetric else legacy.sparsework_nosym)(ctypes.byref(a), ctypes.byref(b),
                                                                      ctypes.byref(part), 30, 79, 0)
    counts = np.ctypeslib.as_array(part.rowPtr, shape=(50,)).copy()
    assert part.rows == 50 and np.array_equal(counts, np.diff(want[0])[30:80])
    got_idx = np.ctypeslib.as_array(part.colInd, shape=(part.nzmax,)).copy()
    assert np.array_equal(got_idx, want[1][want[0][30]:want[0][80]])
    legacy.destroy_sparsemat(ctypes.byref(part))
    for s in (a, b):
        legacy.destroy_sparsemat(ctypes.byref(s))


def test_legacy_dense_and_triple_symbols(legacy, oracle):
    A, B = rand_csr(80, 60, 0.15, 13), rand_csr(60, 80, 0.15, 14)
    a, b = _to_sparsemat(legacy, A), _to_sparsemat(legacy, B)
    for symmetric, fn in ((False, legacy.dense_nosym), (True, legacy.dense_sym)):
        d = DArray()
        fn(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d))
        got = np.ctypeslib.as_array(d.array, shape=(d.rows, d.cols)).copy()       # darray_to_numpy
        assert np.allclose(got, oracle.dense(arrays(A), arrays(B), 80, symmetric=symmetric), rtol=1e-10, atol=0)
        legacy.destroy_darray(ctypes.byref(d))
    S = rand_csr(60, 60, 0.05, 15); Q = (S + S.T).tocsr(); q = _to_sparsemat(legacy, Q)
    for full in (0, 1):
        d = DArray()
        legacy.triple_product(ctypes.byref(a), ctypes.byref(q), ctypes.byref(d), full)
        got = np.ctypeslib.as_array(d.array, shape=(d.rows, d.cols)).copy()
        assert np.allclose(got, oracle.triple(arrays(A), arrays(Q), 60, full), rtol=1e-10, atol=0)
        legacy.destroy_darray(ctypes.byref(d))


def test_legacy_zero_operand_and_mismatch(legacy):
    Z = csr_matrix((3, 3)); B = rand_csr(3, 4, 0.9, 16)
    z, b, out = _to_sparsemat(legacy, Z), _to_sparsemat(legacy, B), SparseMat()
    legacy.sparse_nosym(ctypes.byref(z), ctypes.byref(b), ctypes.byref(out), 5)     # sparse_sparse_sparse.cpp:181-185
    assert (out.rows, out.cols, out.nzmax) == (3, 4, 0) and [out.rowPtr[i] for i in range(4)] == [0, 0, 0, 0]
    legacy.destroy_sparsemat(ctypes.byref(out))
    d = DArray()
    legacy.dense_nosym(ctypes.byref(b), ctypes.byref(b), ctypes.byref(d))           # 3x4 times 3x4: incompatible
    assert not d.array


def test_legacy_symbols_honour_smm_exact(legacy, oracle, monkeypatch):
    """SMM_EXACT=1 in the environment: the legacy symbols return the CPU loop's values bit for bit."""
    monkeypatch.setenv("SMM_EXACT", "1")
    A, B = rand_csr(300, 250, 0.2, 17), rand_csr(250, 300, 0.2, 18)
    a, b, out, d = _to_sparsemat(legacy, A), _to_sparsemat(legacy, B), SparseMat(), DArray()
    legacy.sparse_nosym(ctypes.byref(a), ctypes.byref(b), ctypes.byref(out), 5)
    ptr, idx, val = _from_sparsemat(out)
    want = oracle.sparse(arrays(A), arrays(B), 300)
    assert np.array_equal(ptr, want[0]) and np.array_equal(idx, want[1]) and np.array_equal(val, want[2])
    legacy.dense_sym(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d))
    got = np.ctypeslib.as_array(d.array, shape=(d.rows, d.cols)).copy()
    assert np.array_equal(got, oracle.dense(arrays(A), arrays(B), 300, symmetric=True))
    legacy.destroy_darray(ctypes.byref(d)); legacy.destroy_sparsemat(ctypes.byref(out))
    for s in (a, b):
        legacy.destroy_sparsemat(ctypes.byref(s))


# ---- operand cache: a repeated operand is neither uploaded nor re-indexed
def test_operand_cache_skips_upload_and_tile_index(smm, oracle):
    import sparse_matrix_mult_amd as pkg
    from sparse_matrix_mult_amd.engine import default_context
    ctx = default_context()
    pkg.clear_cache()
    old = pkg.set_operand_cache(4)
    ctx.tune_hash(0, 0)                                   # every row through the tile kernels (they need the tile index)
    try:
        B = rand_csr(600, 700, 0.05, 2)
        mats = [rand_csr(300, 600, 0.05, 10 + i) for i in range(3)]
        ctx.timing(True); ctx.timing_reset()
        first = smm(mats[0], B)
        n_val, n_seg, n_loc = (ctx.kernel_time(k)[1] for k in ("smm_validate", "smm_segptr", "smm_pack_fill"))
        assert n_val == 2 and n_seg >= 1 and n_loc >= 1   # both operands validated, B indexed and packed
        for A in mats[1:]:                                # many A against one B (reference README.md:5,13)
            C = smm(A, B)
            want = oracle.sparse(arrays(A), arrays(B), 700)
            assert np.array_equal(C.indptr, want[0]) and np.array_equal(C.indices, want[1])
            assert np.allclose(C.data, want[2], rtol=1e-10, atol=0)
        again = smm(mats[0], B)                           # both operands cached now
        assert np.array_equal(again.indices, first.indices) and np.allclose(again.data, first.data, rtol=1e-10, atol=0)
        v2, s2, l2 = (ctx.kernel_time(k)[1] for k in ("smm_validate", "smm_segptr", "smm_pack_fill"))
        assert (v2, s2, l2) == (n_val + 2, n_seg, n_loc)  # only the two new A's were validated; B untouched
        # an operand edited in place where the sample looks is seen as a new one
        B2 = B.copy(); B2.data[:] *= 2.0
        assert np.allclose(smm(mats[0], B2).data, 2.0 * first.data, rtol=1e-12)
        # switching the cache off goes back to upload-per-call
        pkg.set_operand_cache(0)
        before = ctx.kernel_time("smm_validate")[1]
        smm(mats[0], B)
        assert ctx.kernel_time("smm_validate")[1] == before + 2
    finally:
        ctx.timing(False)
        ctx.tune_hash(256, 2048)
        pkg.set_operand_cache(old)
        pkg.clear_cache()


def test_wide_index_result_is_a_usable_scipy_matrix(ctx, oracle):
    """nnz >= 2^31 cannot be int32 (BASELINE configs[1] has 2.48e9): the host result then carries
    int64 indptr AND indices.  The widening path is forced here on a small product and the matrix is used
    the way a caller would (matvec, toarray, slicing); tests/test_gpu_baseline_configs.py covers the size."""
    from sparse_matrix_mult_amd.matrix_ops import _result_csr
    A, B = rand_csr(400, 300, 0.05, 1), rand_csr(300, 500, 0.05, 2)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        narrow = ctx.spgemm_host(a, b, exact=True)
        wide = ctx.spgemm_host(a, b, exact=True, index_dtype=np.int64)
    finally:
        a.close(); b.close()
    assert narrow[1].dtype == np.int32 and wide[1].dtype == np.int64 and wide[0].dtype == np.int64
    assert np.array_equal(narrow[1], wide[1]) and np.array_equal(narrow[2], wide[2])
    C = _result_csr(*wide, (400, 500))
    assert C.indptr.dtype == np.int64 and C.indices.dtype == np.int64
    want = oracle.dense(arrays(A), arrays(B), 500)
    x = np.random.default_rng(3).random(500)
    assert np.allclose(C @ x, want @ x) and np.array_equal(C.toarray(), want) and np.array_equal(C[10:20].toarray(), want[10:20])


def test_large_result_takes_the_pipelined_download(ctx, oracle):
    """Results >= 256 MB come back through the ring of pinned buffers + host copy threads (smm_api.hip: download()):
    a 6200 x 6200 dense result (307 MB) and a CSR result whose value array alone is > 256 MB, against the same
    products left on the device and fetched with plain copies."""
    import torch
    n = 6200
    A, B = rand_csr(n, 500, 0.02, 51), rand_csr(500, n, 0.02, 52)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        got = ctx.dense_host(a, b, exact=True)
        dev = torch.empty((n, n), dtype=torch.float64, device=torch.device("cuda", ctx.device))
        ctx.dense_into(a, b, dev.data_ptr(), exact=True)
        ctx.synchronize()
        assert got.nbytes >= 256 << 20 and np.array_equal(got, dev.cpu().numpy())
        want = oracle.dense(arrays(A), arrays(B), n, row_begin=3000, row_end=3040)
        assert np.array_equal(got[3000:3040], want)
    finally:
        a.close(); b.close()
    A2, B2 = rand_csr(n, 2000, 0.04, 53), rand_csr(2000, n, 0.04, 54)       # 3.2 products per cell: C is 96 % full
    a, b = ctx.csr_from_scipy(A2), ctx.csr_from_scipy(B2)
    try:
        ptr, idx, val = ctx.spgemm_host(a, b, exact=True)
        dp, di, dv = ctx.spgemm_torch(a, b, exact=True)
        assert val.nbytes >= 256 << 20
        assert np.array_equal(ptr, dp.cpu().numpy()) and np.array_equal(idx, di.cpu().numpy()) and np.array_equal(val, dv.cpu().numpy())
    finally:
        a.close(); b.close()
