"""sparse_matrix_mult_amd -- MI355X (gfx950) engine behind sparse_matrix_multiply().

    from sparse_matrix_mult_amd import sparse_matrix_multiply        # reference __init__.py:1-3

The package holds only the hot path named in SURVEY.md section 8: the ctypes mirror of the
reference's Python API (matrix_ops.py), an object layer over the C ABI (engine.py), the
row-sharded multi-GPU driver (distributed.py) and the HIP sources (csrc/).
"""
from .matrix_ops import (DeviceCSRResult, PinnedOperand, clear_cache, pin_operand, set_exact, set_full_symmetric,
                         set_operand_cache, set_result_device, sparse_matrix_multiply)

__all__ = ['sparse_matrix_multiply', 'set_exact', 'set_full_symmetric', 'set_result_device', 'clear_cache', 'set_operand_cache',
           'pin_operand', 'PinnedOperand', 'DeviceCSRResult']
