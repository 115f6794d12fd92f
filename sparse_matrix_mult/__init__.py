"""Import-name shim: lets existing callers keep `from sparse_matrix_mult import
sparse_matrix_multiply` (reference sparse_matrix_mult/__init__.py:1-3) while the work is done
by sparse_matrix_mult_amd on the GPU."""
from sparse_matrix_mult_amd.matrix_ops import sparse_matrix_multiply

__all__ = ['sparse_matrix_multiply']
