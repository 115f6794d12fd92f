"""which call of the slab fuzz case 0 aborts?  each variant in a child process"""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASE = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from sparse_matrix_mult_amd.engine import Context
from test_gpu_fuzz_campaign import _rand_rows
from helpers import arrays
from oracle import oracle
v = sys.argv[1]
it = 0
r = np.random.default_rng(777000 + it)
cols = int(r.integers(64, 400)); waves = int(r.choice([4, 8, 16])); ws = int(r.integers(64, 1500))
m, k = int(r.integers(1, 600)), int(r.integers(1, 600))
n = m
A = _rand_rows(r, m, k, 10 ** r.uniform(0, 1.6), int(r.integers(0, 3)))
B = _rand_rows(r, k, n, 10 ** r.uniform(0, 2.0), int(r.integers(0, 3)))
c = Context(0)
if "notune" not in v:
    c.tune_shared(cols, waves); c.tune(cols, min(waves, 8))
if "ws" in v: c.tune_symbolic(ws)
if "nohash" in v: c.tune_hash(0, 0)
exact = "exact" in v
a, b = c.csr_from_scipy(A), c.csr_from_scipy(B)
print(v, "plan...", flush=True)
p = c.spgemm_plan(a, b, exact=exact)
want = oracle.sparse(arrays(A), arrays(B), n)
print(v, "nnz", p.nnz, "want", len(want[1]), "indptr ok", np.array_equal(p.indptr_host(), want[0]), flush=True)
got = p.numeric_host()
print(v, "numeric ok; indices ok", np.array_equal(got[1], want[1]), "values", np.array_equal(got[2], want[2]), flush=True)
''' % (ROOT, ROOT)
for v in ["exact_ws", "exact", "exact_notune", "default_ws", "exact_ws_nohash", "exact_nohash"]:
    try:
        r = subprocess.run([sys.executable, "-c", CASE, v], timeout=40, capture_output=True, text=True)
        print(v, "rc", r.returncode, "|", r.stdout.strip().replace("\n", " | "), "|", r.stderr.strip().replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory", "")[-400:], flush=True)
    except subprocess.TimeoutExpired as e:
        print(v, "TIMEOUT", e.stdout, flush=True)
