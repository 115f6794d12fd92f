"""The C-ABI library loads on a CPU-only box and exports every symbol include/smm_hip.h declares;
struct layouts are the reference wrapper's; without a GPU every compute entry point fails
loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "smm_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^[ \t]*#.*$", "", text, flags=re.M)          # preprocessor lines (macros are not symbols)
    names = re.findall(r"\b(\w+)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if not n.startswith("__")))


def test_library_exports_every_declared_symbol():
    from sparse_matrix_mult_amd._lib import LEGACY_SYMBOLS, LIB_PATH, V2_PROTOTYPES
    assert os.path.exists(LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/smm_hip.h but not exported"
    # and the Python side binds exactly the declared set
    assert sorted(list(V2_PROTOTYPES) + LEGACY_SYMBOLS) == declared


def test_legacy_struct_layout_is_the_reference_wrappers():
    """reference matrix_ops.py:26-33 / :44-48: int dims -> 40-byte sparsemat, 16-byte darray."""
    class SparseMat(ctypes.Structure):
        _fields_ = [("nzmax", ctypes.c_int), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                    ("rowPtr", ctypes.POINTER(ctypes.c_int)), ("colInd", ctypes.POINTER(ctypes.c_int)),
                    ("values", ctypes.POINTER(ctypes.c_double))]

    class DArray(ctypes.Structure):
        _fields_ = [("array", ctypes.POINTER(ctypes.c_double)), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]

    assert ctypes.sizeof(SparseMat) == 40 and ctypes.sizeof(DArray) == 16
    from sparse_matrix_mult_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    lib.create_sparsemat.restype = ctypes.POINTER(SparseMat)
    lib.create_sparsemat.argtypes = [ctypes.c_int] * 3
    m = lib.create_sparsemat(3, 4, 5).contents
    assert (m.rows, m.cols, m.nzmax) == (3, 4, 5)
    assert [m.rowPtr[i] for i in range(4)] == [0, 0, 0, 0] and m.values[4] == 0.0
    lib.destroy_sparsemat.argtypes = [ctypes.POINTER(SparseMat)]
    lib.destroy_sparsemat(ctypes.byref(m))
    assert (m.rows, m.cols, m.nzmax) == (0, 0, 0) and not m.rowPtr
    lib.create_darray.restype = ctypes.POINTER(DArray)
    lib.create_darray.argtypes = [ctypes.c_int] * 2
    d = lib.create_darray(2, 3).contents
    assert (d.rows, d.cols) == (2, 3) and d.array[5] == 0.0
    lib.destroy_darray.argtypes = [ctypes.POINTER(DArray)]
    lib.destroy_darray(ctypes.byref(d))


def test_legacy_limits_matches_oracle(oracle):
    class IArray(ctypes.Structure):
        _fields_ = [("array", ctypes.POINTER(ctypes.c_int)), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]
    from sparse_matrix_mult_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    lib.limits.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(IArray)]
    lib.destroy_iarray.argtypes = [ctypes.POINTER(IArray)]
    for rows, procs in ((10, 3), (3, 8), (100, 7), (1, 1), (50000, 8), (17, 17)):
        r = IArray()
        lib.limits(rows, procs, ctypes.byref(r))
        p, want = oracle.limits(rows, procs)
        assert r.rows == p and r.cols == 2
        assert [r.array[i] for i in range(2 * p)] == want.tolist()
        lib.destroy_iarray(ctypes.byref(r))


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    from sparse_matrix_mult_amd._lib import SmmLibrary
    lib = SmmLibrary().get_lib()
    if lib.smm_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from sparse_matrix_mult_amd import sparse_matrix_multiply
    from sparse_matrix_mult_amd.engine import SmmError
    with pytest.raises(SmmError) as e:
        sparse_matrix_multiply(np.eye(3), np.eye(3))
    assert e.value.code == -1 and "no CPU path" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sparse_matrix_mult_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("oracle/ ", ""), f"{f} mentions the oracle"


def test_modifyalloc_resizes_and_frees_like_the_reference():
    """reference src/memfunctions.cpp:77-103: realloc colInd / values to new_size (contents kept), free both and
    NULL them for new_size <= 0.  Host-only: no GPU involved."""
    class SparseMat(ctypes.Structure):
        _fields_ = [("nzmax", ctypes.c_int), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                    ("rowPtr", ctypes.POINTER(ctypes.c_int)), ("colInd", ctypes.POINTER(ctypes.c_int)),
                    ("values", ctypes.POINTER(ctypes.c_double))]
    from sparse_matrix_mult_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    lib.create_sparsemat.restype = ctypes.POINTER(SparseMat)
    lib.create_sparsemat.argtypes = [ctypes.c_int] * 3
    lib.modifyalloc.argtypes = [ctypes.POINTER(SparseMat), ctypes.c_int]
    lib.destroy_sparsemat.argtypes = [ctypes.POINTER(SparseMat)]
    m = lib.create_sparsemat(2, 3, 4).contents
    for i in range(4):
        m.colInd[i], m.values[i] = i + 1, 0.5 * i
    lib.modifyalloc(ctypes.byref(m), 100000)                       # grow: the first entries survive
    assert [m.colInd[i] for i in range(4)] == [1, 2, 3, 4] and [m.values[i] for i in range(4)] == [0.0, 0.5, 1.0, 1.5]
    m.colInd[99999], m.values[99999] = 7, 7.0                      # the new capacity is writable
    lib.modifyalloc(ctypes.byref(m), 2)                            # shrink
    assert (m.colInd[0], m.colInd[1], m.values[1]) == (1, 2, 0.5)
    lib.modifyalloc(ctypes.byref(m), 0)                            # <= 0: free and NULL, rowPtr untouched
    assert not m.colInd and not m.values and bool(m.rowPtr)
    lib.destroy_sparsemat(ctypes.byref(m))
