"""ctypes loader of lib/libsmm_hip.so -- the only native code this package uses.

Mirrors MatrixOpsLibrary of the reference (sparse_matrix_mult/matrix_ops.py:51-184): a
process-wide singleton that loads the shared library once and declares the prototypes.
Unlike the reference it is silent at import (SURVEY F9 / section 8b "Side effects") and it
refuses to run without the HIP library: there is no CPU fallback in this package.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMM_LIB_PATH: load another build of the same library (diagnostic builds, e.g. -DSMM_STAMPS)
LIB_PATH = os.environ.get("SMM_LIB_PATH") or os.path.join(_HERE, "lib", "libsmm_hip.so")

SMM_SYMMETRIC = 1
SMM_FULL_MATRIX = 2
SMM_EXACT = 4
SMM_MIRROR = 8
SMM_ERR_ALLOC = -3
SMM_ERR_UNSUPPORTED = -6
SMM_ERR_INTERNAL = -7

_c_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); every v2 symbol declared in include/smm_hip.h
V2_PROTOTYPES = {
    "smm_device_count": (ctypes.c_int, []),
    "smm_last_error": (ctypes.c_char_p, []),
    "smm_ctx_create": (ctypes.c_int, [ctypes.c_int, _vp, _pp]),
    "smm_ctx_destroy": (None, [_vp]),
    "smm_ctx_synchronize": (ctypes.c_int, [_vp]),
    "smm_ctx_timing": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_timing_reset": (ctypes.c_int, [_vp]),
    "smm_ctx_kernel_time": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(_c_i64)]),
    "smm_ctx_tune": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "smm_ctx_tune_shared": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "smm_ctx_tune_hash": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "smm_ctx_tune_slab": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "smm_ctx_tune_narrow": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_exact_selftest": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_tune_symbolic": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_tune_dense_runs": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_tune_stage2": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_set_check": (ctypes.c_int, [_vp, ctypes.c_int]),
    "smm_ctx_release_pool": (ctypes.c_int, [_vp]),
    "smm_ctx_pool_bytes": (_c_i64, [_vp]),
    "smm_ctx_inject_alloc_failure": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "smm_ctx_alloc_retries": (_c_i64, [_vp]),
    "smm_plan_check": (ctypes.c_int, [_vp, _vp]),
    "smm_plan_inject_fault": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "smm_csr_from_host": (ctypes.c_int, [_vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _pp]),
    "smm_csr_from_device": (ctypes.c_int, [_vp, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp, _pp]),
    "smm_csr_destroy": (None, [_vp]),
    "smm_csr_rows": (_c_i64, [_vp]),
    "smm_csr_cols": (_c_i64, [_vp]),
    "smm_csr_nnz": (_c_i64, [_vp]),
    "smm_csr_is_canonical": (ctypes.c_int, [_vp, _vp]),
    "smm_csr_update_values": (ctypes.c_int, [_vp, _vp, _vp]),
    "smm_csr_update_values_device": (ctypes.c_int, [_vp, _vp, _vp]),
    "smm_csr_device_bytes": (_c_i64, [_vp]),
    "smm_host_hash64": (ctypes.c_uint64, [_vp, _c_i64]),
    "smm_row_products": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "smm_spgemm_symbolic": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _c_i64, _pp, ctypes.POINTER(_c_i64)]),
    "smm_spgemm_numeric": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "smm_spgemm_numeric_host": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "smm_spgemm_numeric_host_i64": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "smm_plan_indptr_host": (ctypes.c_int, [_vp, _vp, _vp]),
    "smm_plan_nnz": (_c_i64, [_vp]),
    "smm_plan_device_bytes": (_c_i64, [_vp]),
    "smm_plan_destroy": (None, [_vp]),
    "smm_csr_mirror_symbolic": (ctypes.c_int, [_vp, _c_i64, _vp, _vp, _vp, ctypes.POINTER(_c_i64)]),
    "smm_csr_mirror_fill": (ctypes.c_int, [_vp, _c_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "smm_spgemm_dense": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _c_i64, _vp]),
    "smm_spgemm_dense_host": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _c_i64, _vp]),
    "smm_triple_product": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _c_i64, _c_i64, _vp]),
    "smm_triple_product_host": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, _c_i64, _c_i64, _vp]),
    "smm_device_malloc": (ctypes.c_int, [_vp, _c_i64, _pp]),
    "smm_device_free": (ctypes.c_int, [_vp, _vp]),
    "smm_memcpy_d2h": (ctypes.c_int, [_vp, _vp, _vp, _c_i64]),
    "smm_memcpy_h2d": (ctypes.c_int, [_vp, _vp, _vp, _c_i64]),
}

# the legacy symbols the reference's own matrix_ops.py binds (matrix_ops.py:147-171) plus the
# rest of include/functions.h:43-84
LEGACY_SYMBOLS = [
    "create_sparsemat", "create_darray", "destroy_sparsemat", "destroy_darray", "destroy_iarray",
    "modifyalloc", "limits", "sparse_nosym", "sparse_sym", "sparsework_nosym", "sparsework_sym",
    "dense_nosym", "dense_sym", "triple_product",
]


class SmmError(RuntimeError):
    """A call into libsmm_hip.so failed; .code is the negative smm_status."""

    def __init__(self, code, message):
        super().__init__(f"libsmm_hip error {code}: {message}")
        self.code = code


def build(force=False):
    """Compile lib/libsmm_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.run(args, check=True)


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's, which libsmm_hip.so links).  When torch is imported FIRST the dynamic
    linker hands that copy to this library too and device pointers are interchangeable
    (csr_from_torch, spgemm_torch, bench.py).  In the other order the process would hold two
    runtimes and the second one finds no device -- so, when torch is installed but not yet
    imported, its copy is loaded here first (no `import torch`: only the shared object).
    SMM_HIP_RUNTIME=system keeps /opt/rocm's runtime (a process that will never import torch)."""
    if "torch" in sys.modules or os.environ.get("SMM_HIP_RUNTIME", "") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


class SmmLibrary:
    """Singleton holding the loaded library (reference MatrixOpsLibrary, matrix_ops.py:51-72)."""
    _instance = None
    _lock = threading.Lock()

    def __new__(cls):
        with cls._lock:
            if cls._instance is None:
                inst = super().__new__(cls)
                inst._lib = None
                cls._instance = inst
        return cls._instance

    def get_lib(self):
        if self._lib is None:
            if not os.path.exists(LIB_PATH):
                raise OSError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C sparse_matrix_mult_amd/csrc`. This package has no CPU fallback.")
            _share_hip_runtime_with_torch()
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in V2_PROTOTYPES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            self._lib = lib
        return self._lib


def check(lib, rc):
    if rc != 0:
        msg = lib.smm_last_error()
        raise SmmError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")
