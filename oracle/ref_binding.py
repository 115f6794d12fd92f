"""ctypes binding of oracle/_ref/libsparse_ref.so -- the reference's own sources compiled by
oracle/Makefile.  TEST INFRASTRUCTURE ONLY (same rule as oracle.py).

HEAD's include/matrix_def.h:17-31 declares size_t dims, so the structs here are the
48-byte / 24-byte size_t layouts -- NOT the int layouts of the reference's matrix_ops.py
(SURVEY F1).  Only dense_nosym, dense_sym, triple_product and limits are exposed: HEAD's
sparse_nosym / sparse_sym do not run (nnz==0 and free() of interior pointers),
so they are deliberately not bound.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libsparse_ref.so")


class SparseMatSz(ctypes.Structure):          # include/matrix_def.h:17-24
    _fields_ = [("nzmax", ctypes.c_size_t), ("rows", ctypes.c_size_t), ("cols", ctypes.c_size_t),
                ("rowPtr", ctypes.POINTER(ctypes.c_int)), ("colInd", ctypes.POINTER(ctypes.c_int)),
                ("values", ctypes.POINTER(ctypes.c_double))]


class DArraySz(ctypes.Structure):             # include/matrix_def.h:27-31
    _fields_ = [("array", ctypes.POINTER(ctypes.c_double)), ("rows", ctypes.c_size_t),
                ("cols", ctypes.c_size_t)]


class IArraySz(ctypes.Structure):             # include/matrix_def.h:34-38
    _fields_ = [("array", ctypes.POINTER(ctypes.c_int)), ("rows", ctypes.c_size_t),
                ("cols", ctypes.c_size_t)]


_LIB = None


def available():
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(PATH)
        sp, dp = ctypes.POINTER(SparseMatSz), ctypes.POINTER(DArraySz)
        L.dense_nosym.argtypes = [sp, sp, dp]; L.dense_nosym.restype = None
        L.dense_sym.argtypes = [sp, sp, dp]; L.dense_sym.restype = None
        L.triple_product.argtypes = [sp, sp, dp, ctypes.c_int]; L.triple_product.restype = None
        L.destroy_darray.argtypes = [dp]; L.destroy_darray.restype = None
        L.limits.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(IArraySz)]; L.limits.restype = None
        L.destroy_iarray.argtypes = [ctypes.POINTER(IArraySz)]; L.destroy_iarray.restype = None
        _LIB = L
    return _LIB


def _wrap(arrs, rows, cols):
    ptr, idx, val = arrs[:3]
    s = SparseMatSz()
    s.nzmax, s.rows, s.cols = len(idx), rows, cols
    s.rowPtr = ptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    s.colInd = idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    s.values = val.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    return s


def _take(d):
    out = np.ctypeslib.as_array(d.array, shape=(d.rows, d.cols)).copy()
    lib().destroy_darray(ctypes.byref(d))
    return out


def dense(a, b, m, k, n, symmetric=False):
    sa, sb, d = _wrap(a, m, k), _wrap(b, k, n), DArraySz()
    (lib().dense_sym if symmetric else lib().dense_nosym)(ctypes.byref(sa), ctypes.byref(sb), ctypes.byref(d))
    return _take(d)


def triple(h, q, n, k, full=0):
    sh, sq, d = _wrap(h, n, k), _wrap(q, k, k), DArraySz()
    lib().triple_product(ctypes.byref(sh), ctypes.byref(sq), ctypes.byref(d), int(full))
    return _take(d)


def limits(rows, nprocs):
    r = IArraySz()
    lib().limits(rows, nprocs, ctypes.byref(r))
    p = int(r.rows)
    out = np.ctypeslib.as_array(r.array, shape=(2 * p,)).copy()
    lib().destroy_iarray(ctypes.byref(r))
    return p, out


# ---- HEAD's row kernel with its marker array initialised to -1 (oracle/marker_init.c) ------------
PATH_M1 = os.path.join(_HERE, "_ref", "libsparsework_m1.so")
_LIB_M1 = None


def m1_available():
    return os.path.exists(PATH_M1)


def sparsework(a, b, m, k, n, row_begin, row_end, symmetric=False):
    """Runs the reference's sparsework_nosym/_sym (src/sparsework.cpp:12-149 / :156-300, unedited,
    marker initialised to -1) on rows [row_begin,row_end).  Returns (per-row counts, colInd, values)."""
    global _LIB_M1
    if _LIB_M1 is None:
        _LIB_M1 = ctypes.CDLL(PATH_M1)
        sp = ctypes.POINTER(SparseMatSz)
        for f in (_LIB_M1.sparsework_nosym, _LIB_M1.sparsework_sym):
            f.argtypes = [sp, sp, sp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            f.restype = None
    fn = _LIB_M1.sparsework_sym if symmetric else _LIB_M1.sparsework_nosym
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    nr = row_end - row_begin

    def run(mem_increase):
        sa, sb, out = _wrap(a, m, k), _wrap(b, k, n), SparseMatSz()
        fn(ctypes.byref(sa), ctypes.byref(sb), ctypes.byref(out), row_begin, row_end - 1, mem_increase)
        nnz = int(out.nzmax)
        cnt = np.ctypeslib.as_array(out.rowPtr, shape=(nr,)).copy().astype(np.int64)
        idx = np.ctypeslib.as_array(out.colInd, shape=(max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(out.values, shape=(max(nnz, 1),))[:nnz].copy()
        libc.free(ctypes.cast(out.rowPtr, ctypes.c_void_p))  # rowPtr/colInd/values share one malloc pool (:32-42)
        return nnz, cnt, idx, val

    # HEAD lays colInd and values out behind each other in one pool and does not move `values`
    # when the pool grows or shrinks (:81-103, :135-148), so the values come back intact only when
    # the initial capacity equals the final nnz: first call for nnz (counts and colInd are
    # position-stable), second call with exactly that capacity.
    nnz, cnt, idx, _ = run(max(n, 64))
    if nnz == 0:
        return cnt, idx, np.zeros(0)
    nnz2, cnt2, idx2, val = run(nnz)
    assert nnz2 == nnz and np.array_equal(idx, idx2) and np.array_equal(cnt, cnt2)
    return cnt, idx, val


def sparsework_once(a, b, m, k, n, row_begin, row_end, capacity, symmetric=False):
    """ONE run of the reference's row kernel (as in sparsework() above) with the initial capacity given -- the
    exact nnz of the rows, so that HEAD's pool never grows and `values` come back intact.  This is the call
    bench.py times as cpu_baseline kind "reference" (several row ranges on several host threads: ctypes
    releases the GIL and the kernel keeps no global state)."""
    global _LIB_M1
    if _LIB_M1 is None:
        sparsework(a, b, m, k, n, row_begin, min(row_begin + 1, row_end), symmetric)      # loads the library
    fn = _LIB_M1.sparsework_sym if symmetric else _LIB_M1.sparsework_nosym
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    nr = row_end - row_begin
    sa, sb, out = _wrap(a, m, k), _wrap(b, k, n), SparseMatSz()
    fn(ctypes.byref(sa), ctypes.byref(sb), ctypes.byref(out), row_begin, row_end - 1, int(max(capacity, 1)))
    nnz = int(out.nzmax)
    cnt = np.ctypeslib.as_array(out.rowPtr, shape=(nr,)).copy().astype(np.int64)
    idx = np.ctypeslib.as_array(out.colInd, shape=(max(nnz, 1),))[:nnz].copy()
    val = np.ctypeslib.as_array(out.values, shape=(max(nnz, 1),))[:nnz].copy()
    libc.free(ctypes.cast(out.rowPtr, ctypes.c_void_p))
    return cnt, idx, val
