"""Pins the CPU oracle (oracle/smm_oracle.c) against tests/golden/ref_vectors.npz, i.e. against
outputs of the REFERENCE'S OWN SOURCES compiled in the build container
(tests/golden/make_golden.py), and against what the reference's tests assert (numpy's
product under np.allclose).  Runs on CPU; nothing here touches the GPU or /root/reference."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, rand_csr

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if "/a_indptr" in k})
TRIPLES = sorted({k.split("/")[0] for k in GOLD.files if "/h_indptr" in k})


def _operand(case, tag):
    return (GOLD[f"{case}/{tag}_indptr"], GOLD[f"{case}/{tag}_indices"], GOLD[f"{case}/{tag}_data"])


def test_fixture_inventory():
    assert len(CASES) >= 14 and len(TRIPLES) >= 3
    for name in ("ref_CxD", "ref_CxF_square", "ref_AxB_8x8", "ref_1x1", "ref_zero_rows", "ref_demo_4x4"):
        assert name in CASES          # the matrices the reference's own tests hold


@pytest.mark.parametrize("case", CASES)
def test_dense_matches_reference_build(oracle, case):
    a, b = _operand(case, "a"), _operand(case, "b")
    n = int(GOLD[f"{case}/b_shape"][1])
    got = oracle.dense(a, b, n)
    assert np.array_equal(got.view(np.int64), GOLD[f"{case}/ref_dense"].view(np.int64))   # bit-exact
    assert np.allclose(got, GOLD[f"{case}/numpy_matmul"])       # what the reference's tests assert
    if f"{case}/ref_dense_sym" in GOLD.files:
        got = oracle.dense(a, b, n, symmetric=True)
        assert np.array_equal(got.view(np.int64), GOLD[f"{case}/ref_dense_sym"].view(np.int64))
        assert np.all(np.tril(got, -1) == 0.0)


@pytest.mark.parametrize("case", CASES)
def test_sparse_values_and_pattern_match_reference_build(oracle, case):
    """sparse->sparse: HEAD's driver does not run, so the VALUES and the PATTERN are
    pinned by the reference-built dense result of the same inputs (identical products in
    identical order) and nnz/indptr by scipy's structural product."""
    a, b = _operand(case, "a"), _operand(case, "b")
    (m, k), n = GOLD[f"{case}/a_shape"], int(GOLD[f"{case}/b_shape"][1])
    ptr, idx, val = oracle.sparse(a, b, n)
    C = sp.csr_matrix((val, idx, ptr), shape=(m, n))
    # no duplicate columns inside a row
    for i in range(m):
        row = idx[ptr[i]:ptr[i + 1]]
        assert len(np.unique(row)) == len(row)
    assert np.array_equal(C.toarray().view(np.int64), GOLD[f"{case}/ref_dense"].view(np.int64))
    A = sp.csr_matrix((np.ones(len(a[1])), a[1], a[0]), shape=(m, k))
    B = sp.csr_matrix((np.ones(len(b[1])), b[1], b[0]), shape=(k, n))
    S = (A @ B).tocsr()
    assert np.array_equal(ptr, S.indptr)                                   # structural nnz per row
    if f"{case}/ref_dense_sym" in GOLD.files:
        ptr, idx, val = oracle.sparse(a, b, n, symmetric=True)
        Cs = sp.csr_matrix((val, idx, ptr), shape=(m, n))
        assert np.array_equal(Cs.toarray().view(np.int64), GOLD[f"{case}/ref_dense_sym"].view(np.int64))
        rows = np.repeat(np.arange(m), np.diff(ptr))
        assert np.all(rows <= idx)                                         # upper triangle only


@pytest.mark.parametrize("case", CASES)
def test_first_touch_order_matches_the_reference_loop(oracle, case):
    """`refloop_*` = output of the reference's own row kernel (src/sparsework.cpp, unedited, marker
    array initialised to -1 -- oracle/marker_init.c): per-row counts, colInd in first-touch order,
    values.  The oracle must reproduce all three bit for bit (sym: counts and order)."""
    a, b = _operand(case, "a"), _operand(case, "b")
    m, n = int(GOLD[f"{case}/a_shape"][0]), int(GOLD[f"{case}/b_shape"][1])
    cnt, idx, val = oracle.sparse_rows(a, b, n, 0, m)
    assert np.array_equal(cnt, GOLD[f"{case}/refloop_counts"])
    assert np.array_equal(idx, GOLD[f"{case}/refloop_indices"])
    assert np.array_equal(val.view(np.int64), GOLD[f"{case}/refloop_values"].view(np.int64))
    if f"{case}/refloop_sym_indices" in GOLD.files:
        cnt, idx, _ = oracle.sparse_rows(a, b, n, 0, m, symmetric=True)
        assert np.array_equal(cnt, GOLD[f"{case}/refloop_sym_counts"])
        assert np.array_equal(idx, GOLD[f"{case}/refloop_sym_indices"])


def test_first_touch_order_known_answer(oracle):
    """src/sparsework.cpp:59-110 by hand: row 0 of A visits B rows 2 then 0; columns appear in
    the order they are first produced, not sorted."""
    A = sp.csr_matrix((np.array([2.0, 3.0, 1.0]), np.array([2, 0, 1]), np.array([0, 2, 3])), shape=(2, 3))
    B = sp.csr_matrix(np.array([[0.0, 5.0, 0.0, 7.0], [1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 4.0, 6.0]]))
    ptr, idx, val = oracle.sparse(arrays(A), arrays(B), 4)
    assert ptr.tolist() == [0, 3, 4]
    assert idx.tolist() == [2, 3, 1, 0]                # B row 2 -> cols 2,3 ; B row 0 -> col 1 new, col 3 seen
    assert val.tolist() == [8.0, 2.0 * 6.0 + 3.0 * 7.0, 15.0, 1.0]


def test_structural_zero_is_kept(oracle):
    a, b = _operand("cancel", "a"), _operand("cancel", "b")
    ptr, idx, val = oracle.sparse(a, b, 2)
    assert ptr.tolist() == [0, 2, 4] and val[0] == 0.0     # 1*1 + (-1)*1 stays as an entry


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 64])
def test_partitioned_driver_is_partition_independent(oracle, nparts):
    """sparse_sparse_sparse.cpp:269-291: stitching limits()-partitions reproduces one pass."""
    A, B = rand_csr(97, 60, 0.1, 1), rand_csr(60, 83, 0.1, 2)
    one = oracle.sparse(arrays(A), arrays(B), 83, nparts=1)
    many = oracle.sparse(arrays(A), arrays(B), 83, nparts=nparts)
    for x, y in zip(one, many):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("case", TRIPLES)
def test_triple_matches_reference_build(oracle, case):
    h, q = _operand(case, "h"), _operand(case, "q")
    n, k = GOLD[f"{case}/h_shape"]
    up = oracle.triple(h, q, int(k), 0)
    assert np.array_equal(up.view(np.int64), GOLD[f"{case}/ref_triple_upper"].view(np.int64))
    full = oracle.triple(h, q, int(k), 1)
    assert np.array_equal(full.view(np.int64), GOLD[f"{case}/ref_triple_full"].view(np.int64))
    # what tests/test_computation_speed.py:76-87 asserts
    H = sp.csr_matrix((h[2], h[1], h[0]), shape=(n, k)); Q = sp.csr_matrix((q[2], q[1], q[0]), shape=(k, k))
    assert np.allclose(np.triu(up), np.triu((H @ Q @ H.T).toarray()))
    # compute_full_matrix=1 doubles every off-diagonal cell
    off = ~np.eye(n, dtype=bool)
    S = (H @ Q @ H.T).toarray()
    assert np.allclose(full[off], (S + S.T)[off]) and np.allclose(np.diag(full), np.diag(S))


def test_limits_matches_reference_build(oracle):
    for key in [k for k in GOLD.files if k.startswith("limits/")]:
        rows, procs = map(int, key.split("/")[1].split("_"))
        want = GOLD[key]
        p, arr = oracle.limits(rows, procs)
        assert p == want[0] and np.array_equal(arr, want[1:])


def test_reference_build_still_agrees_when_present(oracle):
    """In the build container (oracle/_ref present) re-check a fresh random case live."""
    from oracle import ref_binding as rb
    if not rb.available():
        pytest.skip("oracle/_ref not built here")
    A, B = rand_csr(70, 50, 0.2, 5), rand_csr(50, 70, 0.2, 6)
    assert np.array_equal(oracle.dense(arrays(A), arrays(B), 70), rb.dense(arrays(A), arrays(B), 70, 50, 70))
    assert np.array_equal(oracle.dense(arrays(A), arrays(B), 70, True), rb.dense(arrays(A), arrays(B), 70, 50, 70, True))
    if rb.m1_available():
        for sym in (False, True):
            cnt, idx, val = rb.sparsework(arrays(A), arrays(B), 70, 50, 70, 10, 61, sym)
            c2, i2, v2 = oracle.sparse_rows(arrays(A), arrays(B), 70, 10, 61, sym)
            assert np.array_equal(cnt, c2) and np.array_equal(idx, i2)
            if not sym:
                assert np.array_equal(val, v2)


def test_generator_reproduces_the_committed_fixtures(tmp_path):
    """tests/golden/make_golden.py, run against the reference sources compiled here, must rebuild
    ref_vectors.npz array for array (build container only: the GPU box has no /root/reference)."""
    import importlib.util
    import sys
    from oracle import ref_binding as rb
    if not (os.path.isdir("/root/reference/src") and rb.available()):
        pytest.skip("reference sources / oracle/_ref not present")
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "regen.npz")
    mod.main(out)
    new = np.load(out)
    assert sorted(new.files) == sorted(GOLD.files)
    for k in GOLD.files:
        assert new[k].dtype == GOLD[k].dtype and np.array_equal(new[k], GOLD[k]), k
