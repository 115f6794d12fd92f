"""Generates tests/golden/ref_vectors.npz -- run in the BUILD container only (needs
/root/reference sources compiled by `make -C oracle ref`; nothing here runs on the GPU box).

    python tests/golden/make_golden.py

What is recorded, per case: the inputs (CSR arrays, explicitly -- RNG streams are
version-dependent) and the outputs of the REFERENCE'S OWN CODE built from source
(oracle/_ref/libsparse_ref.so: dense_nosym, dense_sym, triple_product with
compute_full_matrix 0 and 1, limits).  HEAD's sparse_nosym/sparse_sym drivers do not run
 and the prebuilt binaries shipped inside the reference are never loaded.  For the
sparse->sparse routines the file records (1) what the reference's own tests assert: numpy's
product of the same inputs (tests/test_matrix_multiply.py:89-112 compare with np.matmul under
np.allclose); (2) the reference-built dense results, which pin the sparse VALUES bit for bit
(same products, same order: SURVEY F4); (3) `refloop_*`: per-row counts, colInd (the
first-touch ORDER) and values produced by the reference's own row kernel src/sparsework.cpp --
unedited, but linked so that its marker array starts at -1 (oracle/marker_init.c: HEAD's
zero-filled marker makes the loop treat every product as already present).

The matrices are the ones the reference's tests hold as data:
  tests/test_matrix_multiply.py:9-78  (A/B 8x8, C 9x12, D 12x6, F 12x9)
  tests/test_edge_case.py:9-24        (1x1, trailing zero rows, all-zero)
  sparse_matrix_mult/matrix_ops_test_script.py:28-54 (4x4 / 3x4 / 4x3 demos)
plus seeded random cases shaped like tests/test_computation_speed.py:10-15 (density 0.3,
uniform values) and tests/test_with_dense.py (non-square, identity), and cases for the
behaviours the reference's suite never checks (cancellation -> structural zero, unsorted and
duplicated inputs, empty rows).
"""
import os
import sys
import zlib

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle, ref_binding as rb  # noqa: E402


def csr(x):
    m = sp.csr_matrix(x)
    return (m.shape, np.ascontiguousarray(m.indptr, np.int32), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))


def rnd(m, n, d, seed):
    return csr(sp.random(m, n, density=d, format="csr", random_state=np.random.default_rng(seed)))


A8 = np.array([[0.64, 0.99, 0.89, 0.72, 0, 0, 0, 0], [0, 0.67, 0.54, 0, 0.81, 0, 0, 0], [0, 0.32, 0, 0, 0, 0.45, 0, 0],
               [0.1, 0, 0, 0, 0, 0, 0.23, 0], [0, 0, 0.78, 0, 0.55, 0, 0, 0.91], [0.43, 0, 0, 0.12, 0, 0, 0, 0],
               [0, 0, 0.33, 0, 0, 0.68, 0, 0], [0, 0.21, 0, 0, 0, 0, 0.39, 0]])
B8 = np.array([[0.23, 0, 0, 0, 0.51, 0, 0, 0], [0, 0.72, 0, 0, 0, 0.38, 0, 0], [0, 0, 0.99, 0, 0, 0, 0.84, 0],
               [0, 0.76, 0.87, 0.97, 0, 0, 0, 0.29], [0.15, 0, 0, 0, 0.62, 0, 0, 0], [0, 0.44, 0, 0, 0, 0.75, 0, 0],
               [0, 0, 0.58, 0, 0, 0, 0.93, 0], [0.36, 0, 0, 0.82, 0, 0, 0, 0.47]])
C9x12 = np.arange(1, 109, dtype=np.int64).reshape(9, 12)
D12x6 = (np.arange(1, 73, dtype=np.float64) / 10.0).reshape(12, 6)
F12x9 = np.arange(1, 109, dtype=np.int64).reshape(12, 9)
A4 = np.array([[0.64, 0.99, 0.89, 0.72], [0, 0.67, 0.54, 0], [0, 0.32, 0, 0], [0.1, 0, 0, 0]])
B4 = np.array([[0.23, 0, 0, 0.51], [0, 0.72, 0, 0], [0, 0, 0.99, 0], [0, 0.76, 0.87, 0.97]])
ZROWS = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [0, 0, 0], [0, 0, 0], [0, 0, 0]])


def unsorted(c, seed):
    shape, ip, ix, dv = c
    r = np.random.default_rng(seed)
    ix, dv = ix.copy(), dv.copy()
    for i in range(shape[0]):
        p = r.permutation(ip[i + 1] - ip[i]) + ip[i]
        ix[ip[i]:ip[i + 1]], dv[ip[i]:ip[i + 1]] = ix[p], dv[p]
    return shape, ip, ix, dv


def with_dups(c):
    shape, ip, ix, dv = c
    rep = np.repeat(np.arange(len(ix)), 1 + (np.arange(len(ix)) % 3 == 0))
    rows = np.searchsorted(ip, rep, side="right") - 1
    ip2 = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=shape[0]))]).astype(np.int32)
    return shape, ip2, ix[rep].copy(), (dv[rep] * 0.5).copy()


CASES = {
    "ref_AxB_8x8": (csr(A8), csr(B8)),
    "ref_CxD": (csr(C9x12), csr(D12x6)),
    "ref_CxF_square": (csr(C9x12), csr(F12x9)),
    "ref_demo_4x4": (csr(A4), csr(B4)),
    "ref_demo_3x4_4x3": (csr(A4[:3]), csr(A4.T[:, :3].copy())),
    "ref_1x1": (csr(np.array([[5]])), csr(np.array([[2]]))),
    "ref_zero_rows": (csr(ZROWS), csr(np.random.default_rng(5).random((3, 4)))),
    "rand_120_d0.3": (rnd(120, 120, 0.3, 42), rnd(120, 120, 0.3, 43)),
    "rand_nonsquare": (rnd(90, 70, 0.1, 1), rnd(70, 110, 0.1, 2)),
    "rand_identity": (rnd(64, 64, 0.1, 3), csr(sp.identity(64))),
    "cancel": (csr(np.array([[1.0, -1.0], [2.0, 0.0]])), csr(np.array([[1.0, 3.0], [1.0, 0.0]]))),
    "unsorted_inputs": (unsorted(rnd(50, 60, 0.2, 7), 70), unsorted(rnd(60, 80, 0.2, 8), 80)),
    "dup_inputs": (with_dups(rnd(40, 50, 0.2, 9)), with_dups(rnd(50, 45, 0.2, 10))),
    "empty_rows": (rnd(80, 90, 0.02, 11), rnd(90, 100, 0.02, 12)),
}
TRIPLE = {   # H (n x k), Q (k x k, symmetric as BASELINE config 4)
    "triple_small": (rnd(30, 45, 0.2, 21), None),
    "triple_d0.3": (rnd(60, 60, 0.3, 42), None),
    "triple_wide": (rnd(25, 200, 0.05, 23), None),
}


def main(path=None):
    assert rb.available(), "run `make -C oracle ref` first"
    out = {}
    for name, (a, b) in CASES.items():
        (m, k), (k2, n) = a[0], b[0]
        assert k == k2
        for tag, mat in (("a", a), ("b", b)):
            out[f"{name}/{tag}_shape"] = np.array(mat[0], np.int64)
            out[f"{name}/{tag}_indptr"], out[f"{name}/{tag}_indices"], out[f"{name}/{tag}_data"] = mat[1:]
        out[f"{name}/ref_dense"] = rb.dense(a[1:], b[1:], m, k, n, False)
        dense_np = sp.csr_matrix((a[3], a[2], a[1]), shape=a[0]).toarray() @ sp.csr_matrix((b[3], b[2], b[1]), shape=b[0]).toarray()
        out[f"{name}/numpy_matmul"] = dense_np
        if m == n:
            out[f"{name}/ref_dense_sym"] = rb.dense(a[1:], b[1:], m, k, n, True)
        # first-touch ORDER by execution: the reference's own src/sparsework.cpp loop, unedited, with
        # its marker array initialised to -1 (oracle/marker_init.c)
        cnt, idx, val = rb.sparsework(a[1:], b[1:], m, k, n, 0, m, False)
        out[f"{name}/refloop_counts"], out[f"{name}/refloop_indices"], out[f"{name}/refloop_values"] = cnt, idx, val
        if m == n:
            cnt, idx, _ = rb.sparsework(a[1:], b[1:], m, k, n, 0, m, True)
            out[f"{name}/refloop_sym_counts"], out[f"{name}/refloop_sym_indices"] = cnt, idx
        # consistency of the restatement, checked at generation time
        p, i, v = oracle.sparse(a[1:], b[1:], n)
        assert np.array_equal(sp.csr_matrix((v, i, p), shape=(m, n)).toarray(), out[f"{name}/ref_dense"]), name
    for name, (h, _) in TRIPLE.items():
        n, k = h[0]
        s = sp.random(k, k, density=0.05, format="csr", random_state=np.random.default_rng(zlib.crc32(name.encode()) % 1000))
        q = csr((s + s.T).tocsr())
        out[f"{name}/h_shape"] = np.array(h[0], np.int64)
        out[f"{name}/h_indptr"], out[f"{name}/h_indices"], out[f"{name}/h_data"] = h[1:]
        out[f"{name}/q_indptr"], out[f"{name}/q_indices"], out[f"{name}/q_data"] = q[1:]
        out[f"{name}/ref_triple_upper"] = rb.triple(h[1:], q[1:], n, k, 0)
        out[f"{name}/ref_triple_full"] = rb.triple(h[1:], q[1:], n, k, 1)
    for rows, procs in ((10, 3), (3, 8), (100, 7), (16, 16), (1, 1), (50000, 8)):
        p, arr = rb.limits(rows, procs)
        out[f"limits/{rows}_{procs}"] = np.concatenate([[p], arr]).astype(np.int64)
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
