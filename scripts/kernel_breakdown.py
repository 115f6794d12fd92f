"""Per-kernel HIP-event times of one SpGEMM step on synthetic operands: python scripts/kernel_breakdown.py rows cols density"""
import sys
import time

import torch

sys.path.insert(0, ".")
from sparse_matrix_mult_amd.synthetic import gen_csr_device  # noqa: E402
from sparse_matrix_mult_amd.engine import Context  # noqa: E402

m, n, d = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
dev = torch.device("cuda", 0)
ctx = Context(0, torch.cuda.current_stream().cuda_stream)
A = ctx.csr_from_torch(m, n, *gen_csr_device(torch, m, n, d, 1, dev))
B = ctx.csr_from_torch(n, n, *gen_csr_device(torch, n, n, d, 2, dev))
for it in range(3):
    if it == 2:
        ctx.timing(True); ctx.timing_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ctx.spgemm_torch(A, B)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{m}x{n} d={d}: step {dt * 1e3:.2f} ms, nnz(C) {out[1].numel()}")
for k in ("smm_row_work", "smm_scan", "smm_bin_rows", "smm_symbolic", "smm_symbolic_hash", "smm_runs", "smm_numeric",
          "smm_numeric_hash", "smm_copy_lists", "smm_numeric_general", "smm_validate", "smm_segptr", "smm_loc16"):
    ms, calls = ctx.kernel_time(k)
    if calls:
        print(f"   {k:22s} {ms:8.3f} ms  x{calls}")
