"""device-resident results through the PUBLIC entry point.  The reference copies every result
out into fresh host arrays (sparse_matrix_mult/matrix_ops.py:205-240); with set_result_device(True) the same call
leaves the result in HBM as torch tensors.  Default behaviour (host objects) is unchanged and tested everywhere else."""
import numpy as np
import pytest

from helpers import arrays, assert_csr_equal, rand_csr

pytestmark = pytest.mark.gpu


@pytest.fixture
def device_results():
    import sparse_matrix_mult_amd as pkg
    old = pkg.set_result_device(True)
    yield pkg
    pkg.set_result_device(old)
    pkg.set_exact(False); pkg.set_full_symmetric(False)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
def test_sparse_result_stays_in_hbm_and_matches_the_oracle(device_results, oracle, symmetric, exact):
    import torch
    pkg = device_results
    pkg.set_exact(exact)
    A = rand_csr(700, 500, 0.05, 1)
    B = rand_csr(500, 700, 0.05, 2)
    C = pkg.sparse_matrix_multiply(A, B, output_format="sparse", symmetric=symmetric)
    assert isinstance(C, pkg.DeviceCSRResult) and C.shape == (700, 700)
    assert C.indptr.is_cuda and C.indptr.dtype == torch.int64 and C.indices.dtype == torch.int32 and C.data.dtype == torch.float64
    want = oracle.sparse(arrays(A), arrays(B), 700, symmetric=symmetric)
    got = (C.indptr.cpu().numpy(), C.indices.cpu().numpy(), C.data.cpu().numpy())
    assert_csr_equal(got, want, values="bits" if exact else "tol")
    assert C.nnz == len(want[1])
    # the reference's return type on request, and torch's own CSR type over the same values
    S = C.to_scipy()
    assert S.indices.dtype == np.int32 and np.array_equal(S.indices, want[1]) and S.shape == (700, 700)
    T = C.to_torch_sparse_csr()
    x = torch.ones(700, 1, dtype=torch.float64, device=C.data.device)
    assert np.allclose((T @ x).cpu().numpy().ravel(), S @ np.ones(700), rtol=1e-12)


@pytest.mark.parametrize("exact", [False, True])
def test_dense_triple_and_mirror_results_stay_in_hbm(device_results, oracle, exact):
    import torch
    pkg = device_results
    pkg.set_exact(exact)
    same = (lambda x, y: np.array_equal(x, y)) if exact else (lambda x, y: np.allclose(x, y, rtol=1e-10, atol=0))
    A = rand_csr(300, 200, 0.05, 3)
    B = rand_csr(200, 300, 0.05, 4)
    for symmetric in (False, True):
        D = pkg.sparse_matrix_multiply(A, B, output_format="dense", symmetric=symmetric)
        assert isinstance(D, torch.Tensor) and D.is_cuda and D.dtype == torch.float64 and tuple(D.shape) == (300, 300)
        assert same(D.cpu().numpy(), oracle.dense(arrays(A), arrays(B), 300, symmetric=symmetric))
    S = rand_csr(200, 200, 0.02, 5)
    Q = (S + S.T).tocsr()
    T = pkg.sparse_matrix_multiply(A, Q, use_triple_product=True)
    assert isinstance(T, torch.Tensor) and T.is_cuda
    assert same(T.cpu().numpy(), oracle.triple(arrays(A), arrays(Q), 200, 0))
    # the CSR mirror epilogue, device-resident: A A^T as the full symmetric matrix
    At = A.T.tocsr(); At.sort_indices()
    pkg.set_full_symmetric(True)
    F = pkg.sparse_matrix_multiply(A, At, symmetric=True)
    pkg.set_full_symmetric(False)
    assert isinstance(F, pkg.DeviceCSRResult)
    want = oracle.sparse(arrays(A), arrays(At), 300, symmetric=False)
    Fs = F.to_scipy(); Fs.sort_indices()
    import scipy.sparse as sp
    W = sp.csr_matrix((want[2], want[1], want[0]), shape=(300, 300)); W.sort_indices()
    assert np.array_equal(Fs.indptr, W.indptr) and np.array_equal(Fs.indices, W.indices) and same(Fs.data, W.data)


def test_zero_operand_and_default_setting(device_results, capsys):
    import scipy.sparse as sp
    import torch
    pkg = device_results
    A = rand_csr(50, 40, 0.1, 6)
    Z = sp.csr_matrix((40, 30))
    C = pkg.sparse_matrix_multiply(A, Z)
    assert isinstance(C, pkg.DeviceCSRResult) and C.nnz == 0 and C.shape == (50, 30) and int(C.indptr.abs().sum()) == 0
    D = pkg.sparse_matrix_multiply(A, Z, output_format="dense")
    assert isinstance(D, torch.Tensor) and tuple(D.shape) == (50, 30) and not bool(D.any())
    pkg.set_result_device(False)                      # the default: the reference's host objects
    C = pkg.sparse_matrix_multiply(A, rand_csr(40, 30, 0.1, 7))
    assert sp.isspmatrix_csr(C)
