"""sparse_matrix_multiply() -- the reference's public entry point, served by the MI355X engine.

Drop-in for reference sparse_matrix_mult/matrix_ops.py:271-387: same name, same arguments,
same argument meaning, same return types, same raised errors.  What differs:
  * the work is done by libsmm_hip.so (hand-written HIP, gfx950) through the v2 C ABI of
    include/smm_hip.h; operands go host->HBM once, results come back into numpy buffers this
    module owns (no create_sparsemat/memmove round trip, no leaked input structs:
    reference :187-202, :339-340);
  * nothing is printed at import (reference :89-90,133,137);
  * a failure inside the library RAISES (SmmError) instead of being printed and turned into
    an all-zero result (reference :377-387) -- silent zeros would hide a missing GPU.  The
    one swallowed case the reference's callers can observe on purpose, an unknown
    output_format, is kept: message printed, zeros returned.
"""
import os

import numpy as np
from scipy.sparse import csr_matrix, isspmatrix_csr

from .engine import default_context

_INT32_MAX = np.iinfo(np.int32).max

# SMM_EXACT=1 (or set_exact(True)): add products in exactly the reference's order, so float64
# values are bit-identical to the reference's loop; the default lets the waves of a workgroup
# add concurrently -- values agree to rounding (tested to 1e-10 relative), ~3x faster.
# indptr / indices are bit-exact either way.
_exact = os.environ.get("SMM_EXACT", "0") not in ("", "0")


def set_exact(flag):
    """Select bit-exact (reference-order) accumulation for later calls; returns the old setting."""
    global _exact
    old, _exact = _exact, bool(flag)
    return old


def _as_csr(x):
    # reference :307-310: anything csr_matrix() accepts; no sort, no dedup
    return x if isspmatrix_csr(x) else csr_matrix(x)


def _result_csr(indptr, indices, data, shape):
    """reference sparsemat_to_csr (:205-228): nzmax==0 -> empty matrix; int32 index arrays
    (widened to int64 only when nnz does not fit, which the reference cannot represent)."""
    if len(indices) == 0:
        return csr_matrix(shape)
    if indptr[-1] <= _INT32_MAX:
        indptr = indptr.astype(np.int32)
    out = csr_matrix(shape, dtype=np.float64)
    # assign the arrays directly: the constructor would be free to check/copy, and must not
    # sort -- the reference returns first-touch order
    out.data, out.indices, out.indptr = data, indices, indptr
    return out


def sparse_matrix_multiply(matrix_a, matrix_b, output_format='sparse', symmetric=False, imem_size=None,
                           use_triple_product=False, compute_full_matrix=None):
    """Multiply two sparse matrices on the GPU.

    matrix_a, matrix_b : scipy CSR, or anything csr_matrix() accepts.
    output_format      : 'sparse' -> scipy.sparse.csr_matrix, 'dense' -> numpy.ndarray.
    symmetric          : keep only the upper triangle (i <= j) of a square result.
    imem_size          : the reference's CPU scratch hint; validated, then ignored.
    use_triple_product : return matrix_a @ matrix_b @ matrix_a.T as a dense array (upper
                         triangle unless compute_full_matrix=1); output_format/symmetric are
                         then ignored, as in the reference (:325).
    compute_full_matrix: None/0/1, see above (1 reproduces the reference exactly).
    """
    if imem_size is None:                                    # reference :288-295
        imem_size = 5
    else:
        try:
            imem_size = int(imem_size)
        except ValueError:
            raise ValueError(f"imem_size must be an integer or None, got {type(imem_size)}")

    if compute_full_matrix is None:                          # reference :298-304
        compute_full_matrix = 0
    else:
        if compute_full_matrix not in (0, 1):
            raise ValueError("compute_full_matrix must be None, 0, or 1")
        compute_full_matrix = int(compute_full_matrix)

    matrix_a = _as_csr(matrix_a)
    matrix_b = _as_csr(matrix_b)

    if matrix_a.shape[1] != matrix_b.shape[0]:               # reference :312-313
        raise ValueError("Matrix dimensions are incompatible for multiplication.")

    out_shape = (matrix_a.shape[0], matrix_b.shape[1])
    if matrix_a.nnz == 0 or matrix_b.nnz == 0:               # reference :315-319
        return csr_matrix(out_shape) if output_format == 'sparse' else np.zeros(out_shape)

    if symmetric and out_shape[0] != out_shape[1]:           # reference :321-322
        raise ValueError("For symmetric output, the resulting matrix must be square.")

    if not use_triple_product and output_format not in ('sparse', 'dense'):
        # reference :367-368 raises inside its try and :377-387 prints and returns zeros
        print("An error occurred during matrix multiplication: Invalid output_format. Choose 'sparse' or 'dense'.")
        return np.zeros(out_shape)

    ctx = default_context()
    a = ctx.csr_from_scipy(matrix_a)
    b = ctx.csr_from_scipy(matrix_b)
    try:
        if use_triple_product:                               # reference :325-336
            result = ctx.triple_host(a, b, full=bool(compute_full_matrix), exact=_exact)
        elif output_format == 'sparse':                      # reference :338-351
            indptr, indices, data = ctx.spgemm_host(a, b, symmetric=bool(symmetric), exact=_exact)
            result = _result_csr(indptr, indices, data, out_shape)
        else:                                                # reference :353-365
            result = ctx.dense_host(a, b, symmetric=bool(symmetric), exact=_exact)
    finally:
        a.close()
        b.close()

    if isinstance(result, np.ndarray):                       # reference :370-373
        if not result.any():
            print("Multiplication resulted in a zero matrix.")
    elif result.nnz == 0:
        print("Multiplication resulted in a zero matrix.")
    return result
