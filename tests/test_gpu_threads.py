"""sparse_matrix_multiply() from several Python threads at once (SURVEY 8b "Threading": the
reference's library has no global state and ctypes releases the GIL, matrix_ops.py:136).
Here all threads share one device context; every entry point of the C ABI takes the context's
lock, so their calls interleave safely.  One process, run once."""
import threading

import numpy as np
import pytest

from helpers import arrays, rand_csr

pytestmark = pytest.mark.gpu


def test_four_threads_different_products_bit_exact(oracle):
    from sparse_matrix_mult_amd import set_exact, sparse_matrix_multiply
    jobs = [  # (m, k, n, dA, dB, output_format, symmetric)
        (300, 200, 300, 0.05, 0.05, "sparse", False),
        (257, 129, 257, 0.10, 0.08, "sparse", True),
        (180, 240, 150, 0.06, 0.07, "dense", False),
        (1000, 1000, 1000, 0.05, 0.05, "sparse", False),       # BASELINE configs[0]
    ]
    want, ops = [], []
    for t, (m, k, n, da, db, fmt, sym) in enumerate(jobs):
        A, B = rand_csr(m, k, da, 10 + t), rand_csr(k, n, db, 20 + t)
        ops.append((A, B))
        if fmt == "sparse":
            want.append(oracle.sparse(arrays(A), arrays(B), n, symmetric=sym))
        else:
            want.append(oracle.dense(arrays(A), arrays(B), n, symmetric=sym))
    errors, rounds = [], 6
    start = threading.Barrier(len(jobs))

    def work(t):
        try:
            m, k, n, da, db, fmt, sym = jobs[t]
            A, B = ops[t]
            start.wait()
            for _ in range(rounds):
                got = sparse_matrix_multiply(A, B, output_format=fmt, symmetric=sym)
                if fmt == "sparse":
                    ptr, idx, val = want[t]
                    assert np.array_equal(got.indptr, ptr) and np.array_equal(got.indices, idx)
                    assert np.array_equal(got.data.view(np.int64), val.view(np.int64))
                else:
                    assert np.array_equal(got.view(np.int64), want[t].view(np.int64))
        except BaseException as e:                               # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    old = set_exact(True)
    try:
        threads = [threading.Thread(target=work, args=(t,)) for t in range(len(jobs))]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=300)
        assert not any(th.is_alive() for th in threads), "a worker thread hung"
    finally:
        set_exact(old)
    assert not errors, errors
