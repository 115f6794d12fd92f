"""Host-side behaviour of sparse_matrix_multiply() that needs no GPU: the argument checks, the
raised errors and the nnz==0 early returns of the reference (matrix_ops.py:288-322), plus the
shard planner of the multi-GPU driver."""
import numpy as np
import pytest
import scipy.sparse as sp

from sparse_matrix_mult_amd import sparse_matrix_multiply
from sparse_matrix_mult_amd.distributed import balanced_row_shards


def test_signature_is_the_references():
    import inspect
    sig = inspect.signature(sparse_matrix_multiply)
    assert list(sig.parameters) == ["matrix_a", "matrix_b", "output_format", "symmetric", "imem_size",
                                    "use_triple_product", "compute_full_matrix"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["output_format"] == "sparse" and d["symmetric"] is False and d["imem_size"] is None
    assert d["use_triple_product"] is False and d["compute_full_matrix"] is None


def test_import_name_shim():
    from sparse_matrix_mult import sparse_matrix_multiply as f
    assert f is sparse_matrix_multiply


def test_dimension_mismatch_raises():            # reference matrix_ops.py:312-313
    with pytest.raises(ValueError, match="incompatible"):
        sparse_matrix_multiply(np.ones((2, 3)), np.ones((4, 2)))


def test_symmetric_needs_square_result():        # reference matrix_ops.py:321-322
    with pytest.raises(ValueError, match="square"):
        sparse_matrix_multiply(np.ones((2, 3)), np.ones((3, 4)), symmetric=True)


def test_bad_imem_size_and_full_matrix():        # reference matrix_ops.py:291-304
    with pytest.raises(ValueError, match="imem_size"):
        sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), imem_size="x")
    with pytest.raises(ValueError, match="compute_full_matrix"):
        sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), compute_full_matrix=2)


def test_zero_operands_return_empty_without_touching_the_gpu():
    """reference tests/test_edge_case.py:54-71 and matrix_ops.py:315-319."""
    z33, z34 = np.zeros((3, 3)), np.zeros((3, 4))
    r = sparse_matrix_multiply(z33, z34, output_format="sparse")
    assert sp.isspmatrix_csr(r) and r.shape == (3, 4) and r.nnz == 0 and r.dtype == np.float64
    r = sparse_matrix_multiply(sp.csr_matrix(z33), sp.csr_matrix(z34), output_format="dense")
    assert isinstance(r, np.ndarray) and r.shape == (3, 4) and not r.any()
    r = sparse_matrix_multiply(z33, np.ones((3, 4)), use_triple_product=False, output_format="dense", symmetric=False)
    assert r.shape == (3, 4)
    # the zero check comes before the symmetric-square check, as in the reference (:315 before :321)
    assert sparse_matrix_multiply(z33, z34, symmetric=True).shape == (3, 4)


def test_unknown_output_format_prints_and_returns_zeros(capsys):   # reference :367-368, :377-387
    r = sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), output_format="coo")
    assert isinstance(r, np.ndarray) and r.shape == (2, 2) and not r.any()
    assert "Invalid output_format" in capsys.readouterr().out


def test_balanced_row_shards():
    work = np.array([1, 1, 1, 1, 100, 1, 1, 1], dtype=float)
    sh = balanced_row_shards(work, 2)
    assert sh[0][0] == 0 and sh[-1][1] == 8 and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
    assert all(e > b for b, e in sh)
    # more shards than rows: clamped like limits() (workdivision.cpp:26-29)
    assert balanced_row_shards(np.ones(3), 8) == [(0, 1), (1, 2), (2, 3)]
    # uniform work -> near-equal row counts
    sh = balanced_row_shards(np.ones(1000), 8)
    assert max(e - b for b, e in sh) - min(e - b for b, e in sh) <= 1
    # skewed work is balanced by work, not rows
    w = np.concatenate([np.full(100, 50.0), np.full(900, 1.0)])
    sh = balanced_row_shards(w, 4)
    loads = [w[b:e].sum() for b, e in sh]
    assert max(loads) / (sum(loads) / 4) < 1.1
    assert balanced_row_shards([], 4) == [(0, 0)]


# ---- operand cache key (no GPU involved: smm_host_hash64 is host code)
def test_operand_key_sees_every_single_element_edit():
    """The reference marshals the caller's CURRENT arrays on every call (matrix_ops.py:339-340, :187-202); the
    operand cache may therefore only recognise an operand whose every element is unchanged.  Its key is a
    full-content hash: an in-place edit of ANY one element of indptr / indices / data changes the part of the
    key it belongs to (the round-2 key, a strided sample, missed exactly these edits)."""
    from sparse_matrix_mult_amd.matrix_ops import _operand_key
    B = sp.random(5000, 5000, density=0.01, format="csr", random_state=np.random.default_rng(1))
    assert B.nnz == 250000
    pat0, dat0 = _operand_key(B)
    assert _operand_key(B.copy()) == (pat0, dat0)                 # content, not identity
    rng = np.random.default_rng(7)
    for k in [12345, 0, B.nnz - 1] + rng.integers(0, B.nnz, 40).tolist():
        old = B.data[k]
        B.data[k] = np.nextafter(old, 2.0)                        # one ulp
        pat, dat = _operand_key(B)
        assert pat == pat0 and dat != dat0, k
        B.data[k] = old
    for k in [777, 0, B.nnz - 1] + rng.integers(0, B.nnz, 40).tolist():
        old = B.indices[k]
        B.indices[k] = old ^ 1
        pat, dat = _operand_key(B)
        assert pat != pat0 and dat == dat0, k
        B.indices[k] = old
    i = 2500
    B.indptr[i] -= 1
    assert _operand_key(B)[0] != pat0
    B.indptr[i] += 1
    assert _operand_key(B) == (pat0, dat0)
    # swapping two elements (same multiset) is seen too: the hash is position-dependent
    B.data[[10, 11]] = B.data[[11, 10]]
    assert _operand_key(B)[1] != dat0


def test_host_hash_is_the_same_for_every_thread_count_and_length_aware(monkeypatch):
    import ctypes
    from sparse_matrix_mult_amd._lib import SmmLibrary
    lib = SmmLibrary().get_lib()
    a = np.random.default_rng(3).integers(0, 255, 40 << 20, dtype=np.uint8)     # 10 blocks of 4 MB
    h = lambda x: lib.smm_host_hash64(ctypes.c_void_p(x.ctypes.data), x.nbytes)
    monkeypatch.setenv("SMM_HASH_THREADS", "1")
    one = h(a)
    monkeypatch.setenv("SMM_HASH_THREADS", "5")
    assert h(a) == one
    assert h(a[:-1]) != one and h(a[:1000]) != h(a[:1001])
    z = np.zeros(64, np.uint8)
    assert len({h(z[:n]) for n in range(1, 65)}) == 64                          # zero padding of the tail is not a collision
