"""The HIP path against the REFERENCE'S OWN OUTPUTS, directly (no oracle in between).

tests/golden/ref_vectors.npz holds, for the matrices the reference's tests carry as data and a
few seeded random ones, what the reference's own sources produced when compiled in the build
container (tests/golden/make_golden.py): dense_nosym / dense_sym / triple_product results, and
the per-row counts, first-touch colInd order and values of its row kernel src/sparsework.cpp.
Here the GPU results are compared with those arrays bit for bit (SMM_EXACT) and within the
north star's 1e-10 (default mode), through the v2 C ABI and through sparse_matrix_multiply().
Operands with unsorted rows of B take the general numeric path (global f64 atomics): indices
bit-exact, values to 1e-10 in both modes.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_err

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if "/a_indptr" in k})
TRIPLES = sorted({k.split("/")[0] for k in GOLD.files if "/h_indptr" in k})
MODES = [pytest.param(False, id="default"), pytest.param(True, id="exact")]
RTOL = 1e-10


def _upload(ctx, case, tag, shape):
    ip, ix, dv = GOLD[f"{case}/{tag}_indptr"], GOLD[f"{case}/{tag}_indices"], GOLD[f"{case}/{tag}_data"]
    return ctx.csr_from_arrays(int(shape[0]), int(shape[1]), ip, ix, dv)


def _b_unsorted(case):
    ip, ix = GOLD[f"{case}/b_indptr"], GOLD[f"{case}/b_indices"]
    return any(np.any(np.diff(ix[ip[i]:ip[i + 1]]) < 0) for i in range(len(ip) - 1))


def _same(got, want, bits):
    got, want = np.ascontiguousarray(got, np.float64), np.ascontiguousarray(want, np.float64)
    if bits:
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), f"bitwise mismatch, max rel {rel_err(got, want):.3e}"
    else:
        assert rel_err(got, want) <= RTOL


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("exact", MODES)
def test_dense_equals_reference_build(ctx, case, exact):
    sa, sb = GOLD[f"{case}/a_shape"], GOLD[f"{case}/b_shape"]
    a, b = _upload(ctx, case, "a", sa), _upload(ctx, case, "b", sb)
    bits = exact and not _b_unsorted(case)
    try:
        _same(ctx.dense_host(a, b, exact=exact), GOLD[f"{case}/ref_dense"], bits)
        if f"{case}/ref_dense_sym" in GOLD.files:
            got = ctx.dense_host(a, b, symmetric=True, exact=exact)
            _same(got, GOLD[f"{case}/ref_dense_sym"], bits)
            assert np.all(np.tril(got, -1) == 0.0)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("exact", MODES)
def test_sparse_equals_reference_row_kernel(ctx, case, exact):
    """per-row counts, first-touch colInd order and values of src/sparsework.cpp (refloop_*)."""
    sa, sb = GOLD[f"{case}/a_shape"], GOLD[f"{case}/b_shape"]
    a, b = _upload(ctx, case, "a", sa), _upload(ctx, case, "b", sb)
    bits = exact and not _b_unsorted(case)
    try:
        ptr, idx, val = ctx.spgemm_host(a, b, exact=exact)
        assert np.array_equal(np.diff(ptr), GOLD[f"{case}/refloop_counts"])
        assert np.array_equal(idx, GOLD[f"{case}/refloop_indices"])
        _same(val, GOLD[f"{case}/refloop_values"], bits)
        if f"{case}/refloop_sym_counts" in GOLD.files:
            ptr, idx, val = ctx.spgemm_host(a, b, symmetric=True, exact=exact)
            assert np.array_equal(np.diff(ptr), GOLD[f"{case}/refloop_sym_counts"])
            assert np.array_equal(idx, GOLD[f"{case}/refloop_sym_indices"])
            # values of the symmetric product: the reference-built dense_sym result holds them
            C = sp.csr_matrix((val, idx, ptr), shape=(int(sa[0]), int(sb[1])))
            _same(C.toarray(), GOLD[f"{case}/ref_dense_sym"], bits)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("case", TRIPLES)
@pytest.mark.parametrize("exact", MODES)
def test_triple_equals_reference_build(ctx, case, exact):
    n, k = (int(x) for x in GOLD[f"{case}/h_shape"])
    h = ctx.csr_from_arrays(n, k, GOLD[f"{case}/h_indptr"], GOLD[f"{case}/h_indices"], GOLD[f"{case}/h_data"])
    q = ctx.csr_from_arrays(k, k, GOLD[f"{case}/q_indptr"], GOLD[f"{case}/q_indices"], GOLD[f"{case}/q_data"])
    try:
        _same(ctx.triple_host(h, q, exact=exact), GOLD[f"{case}/ref_triple_upper"], exact)
        _same(ctx.triple_host(h, q, full=True, exact=exact), GOLD[f"{case}/ref_triple_full"], exact)   # SURVEY F6 reproduced
    finally:
        h.close(); q.close()


@pytest.mark.parametrize("case", ["ref_CxD", "ref_CxF_square", "ref_AxB_8x8", "ref_zero_rows", "rand_120_d0.3"])
def test_python_api_equals_reference_build(case):
    """the drop-in entry point itself, dense ndarray inputs as in tests/test_matrix_multiply.py:89-112"""
    from sparse_matrix_mult_amd import set_exact, sparse_matrix_multiply

    def dense(tag):
        shape = tuple(int(x) for x in GOLD[f"{case}/{tag}_shape"])
        return sp.csr_matrix((GOLD[f"{case}/{tag}_data"], GOLD[f"{case}/{tag}_indices"], GOLD[f"{case}/{tag}_indptr"]),
                             shape=shape).toarray()
    A, B = dense("a"), dense("b")
    old = set_exact(True)
    try:
        D = sparse_matrix_multiply(A, B, output_format="dense")
        assert np.array_equal(D.view(np.int64), GOLD[f"{case}/ref_dense"].view(np.int64))
        C = sparse_matrix_multiply(A, B, output_format="sparse")
        # a dense ndarray input keeps its explicit structure through csr_matrix(): same pattern as the fixture
        assert np.array_equal(C.toarray().view(np.int64), GOLD[f"{case}/ref_dense"].view(np.int64))
        assert np.allclose(C.toarray(), GOLD[f"{case}/numpy_matmul"])            # what the reference's tests assert
    finally:
        set_exact(old)
