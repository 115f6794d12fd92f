"""ctypes front-end of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (sparse_matrix_mult_amd) must never do so.

Each helper mirrors one reference entry point (file:line under /root/reference):
  sparse()  -> sparse_nosym / sparse_sym   src/sparse_sparse_sparse.cpp:172-299 / :41-155
  dense()   -> dense_nosym / dense_sym     src/sparse_sparse_dense.cpp:79-131 / :13-74
  triple()  -> triple_product              src/sparse_sparse_dense.cpp:141-249
  limits()  -> limits                      src/workdivision.cpp:16-89
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    """Compile liboracle.so (and oracle/_ref when the reference sources are present)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        # SMM_ORACLE_LIB: another build of the same restatement (scripts/sanitize_cpu.sh: ASan/UBSan)
        path = os.environ.get("SMM_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_limits.argtypes = [ctypes.c_int, ctypes.c_int, _i32p]
        L.oracle_limits.restype = ctypes.c_int
        csr = [_i32p, _i32p, _f64p]
        L.oracle_sparse.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + csr + csr + [
            ctypes.c_int, _i64p, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_sparse.restype = ctypes.c_int64
        L.oracle_sparsework.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64] + csr + csr + [
            ctypes.c_int, _i64p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.oracle_sparsework.restype = ctypes.c_int64
        L.oracle_dense.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64] + csr + csr + [
            ctypes.c_int, _f64p]
        L.oracle_dense.restype = None
        L.oracle_triple.argtypes = [ctypes.c_int64, ctypes.c_int64] + csr + csr + [
            ctypes.c_int, ctypes.c_int64, ctypes.c_int64, _f64p]
        L.oracle_triple.restype = None
        _LIB = L
    return _LIB


def csr_arrays(m):
    """(indptr int32, indices int32, data float64) exactly as matrix_ops.py:196-198 casts them."""
    return (np.ascontiguousarray(m.indptr, dtype=np.int32),
            np.ascontiguousarray(m.indices, dtype=np.int32),
            np.ascontiguousarray(m.data, dtype=np.float64))


def limits(rows, nprocs):
    out = np.zeros(2 * max(1, min(rows, max(nprocs, 1))), dtype=np.int32)
    p = lib().oracle_limits(rows, nprocs, out)
    return p, out[:2 * max(p, 0)]


def sparse(a, b, n_cols, symmetric=False, nparts=1):
    """Returns (indptr int64, indices int32, data float64) in first-touch order."""
    ap, ai, av = a
    bp, bi, bv = b[:3]
    m = len(ap) - 1
    n = int(n_cols)
    c_ptr = np.zeros(m + 1, dtype=np.int64)
    L = lib()
    nnz = L.oracle_sparse(m, n, nparts, ap, ai, av, bp, bi, bv, int(bool(symmetric)), c_ptr, None, None)
    if nnz < 0:
        raise MemoryError("oracle_sparse failed")
    c_idx = np.empty(max(nnz, 1), dtype=np.int32)
    c_val = np.empty(max(nnz, 1), dtype=np.float64)
    got = L.oracle_sparse(m, n, nparts, ap, ai, av, bp, bi, bv, int(bool(symmetric)), c_ptr,
                          c_idx.ctypes.data, c_val.ctypes.data)
    assert got == nnz
    return c_ptr, c_idx[:nnz], c_val[:nnz]


def sparse_rows(a, b, n_cols, row_begin, row_end, symmetric=False):
    """Rows [row_begin,row_end) only: (rowcnt int64, indices, data).  Used on row subsets
    of the large BASELINE configs and as the timed cpu_baseline sample."""
    ap, ai, av = a
    bp, bi, bv = b[:3]
    L = lib()
    cnt = np.zeros(row_end - row_begin, dtype=np.int64)
    nnz = L.oracle_sparsework(row_begin, row_end, n_cols, ap, ai, av, bp, bi, bv,
                              int(bool(symmetric)), cnt, None, None, 0)
    c_idx = np.empty(max(nnz, 1), dtype=np.int32)
    c_val = np.empty(max(nnz, 1), dtype=np.float64)
    got = L.oracle_sparsework(row_begin, row_end, n_cols, ap, ai, av, bp, bi, bv,
                              int(bool(symmetric)), cnt, c_idx.ctypes.data, c_val.ctypes.data, nnz)
    assert got == nnz
    return cnt, c_idx[:nnz], c_val[:nnz]


def dense(a, b, n_cols, symmetric=False, row_begin=0, row_end=None):
    ap, ai, av = a
    bp, bi, bv = b[:3]
    m = len(ap) - 1
    row_end = m if row_end is None else row_end
    out = np.empty((row_end - row_begin, n_cols), dtype=np.float64)
    lib().oracle_dense(row_begin, row_end, n_cols, ap, ai, av, bp, bi, bv, int(bool(symmetric)), out)
    return out


def triple(h, q, k_dim, full=0, row_begin=0, row_end=None):
    hp, hi, hv = h
    qp, qi, qv = q[:3]
    n = len(hp) - 1
    row_end = n if row_end is None else row_end
    out = np.zeros((n, n), dtype=np.float64)
    lib().oracle_triple(n, k_dim, hp, hi, hv, qp, qi, qv, int(full), row_begin, row_end, out)
    return out
