"""Can the symbolic phase of one product run beside the numeric phase of another?  Two contexts (own streams, own
copies of the operands), two host threads: one repeats smm_spgemm_numeric on a finished plan, the other repeats
smm_spgemm_symbolic.  Prints the rate of each loop alone and together (BASELINE configs[1] operands)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_matrix_mult_amd.engine import Context
from sparse_matrix_mult_amd.synthetic import gen_csr_device

dev = torch.device("cuda", 0)
n, d = 50000, 0.01
a_t = gen_csr_device(torch, n, n, d, 1, dev); b_t = gen_csr_device(torch, n, n, d, 2, dev)
torch.cuda.synchronize()
c1, c2 = Context(0), Context(0)
A1, B1 = c1.csr_from_torch(n, n, *a_t), c1.csr_from_torch(n, n, *b_t)
A2, B2 = c2.csr_from_torch(n, n, *a_t), c2.csr_from_torch(n, n, *b_t)
plan = c1.spgemm_plan(A1, B1)
indptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
indices = torch.empty(plan.nnz, dtype=torch.int32, device=dev)
data = torch.empty(plan.nnz, dtype=torch.float64, device=dev)

def numeric_loop(k, out):
    t = time.perf_counter()
    for _ in range(k):
        plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
    c1.synchronize()
    out["numeric"] = (time.perf_counter() - t) / k * 1e3

def symbolic_loop(k, out):
    t = time.perf_counter()
    for _ in range(k):
        p = c2.spgemm_plan(A2, B2); p.close()
    c2.synchronize()
    out["symbolic"] = (time.perf_counter() - t) / k * 1e3

for f in (numeric_loop, symbolic_loop):
    f(2, {})
alone = {}
numeric_loop(10, alone); symbolic_loop(10, alone)
both = {}
t1 = threading.Thread(target=numeric_loop, args=(10, both)); t2 = threading.Thread(target=symbolic_loop, args=(30, both))
t = time.perf_counter(); t1.start(); t2.start(); t1.join(); t2.join(); wall = time.perf_counter() - t
print(f"alone: numeric {alone['numeric']:.2f} ms, symbolic(+runs) {alone['symbolic']:.2f} ms per product")
print(f"together: numeric {both['numeric']:.2f} ms per product (10), symbolic {both['symbolic']:.2f} ms per product (30), wall {wall*1e3:.1f} ms")
