"""Round 3: B wider than one slab of the chunked column stream is walked slab by slab (smm_symbolic_ccs per
(slab, row), smm_runs_slab, the slab epilogue of smm_numeric).  The path is forced onto small matrices with
smm_ctx_tune_symbolic + narrow tiles and must reproduce the reference's first-touch order exactly
(src/sparsework.cpp:56-129) -- the full-size case is tests/test_gpu_baseline_configs.py::test_config4_*
(200 000 columns take this path by default)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, assert_csr_equal, rand_csr, signed

pytestmark = pytest.mark.gpu

# (tile columns, waves, max slab columns): tiles per slab 1, 3, 8 (the cap; only the widest cases exceed the slab) and 2
GEOMS = [(64, 4, 64), (64, 4, 200), (64, 8, 520), (100, 8, 250)]


@pytest.fixture(params=GEOMS, ids=lambda g: f"tile{g[0]}-w{g[1]}-slab{g[2]}")
def slabbed(request, ctx):
    cols, waves, ws = request.param
    ctx.tune_shared(cols, waves)
    ctx.tune(cols, min(waves, 8))
    ctx.tune_symbolic(ws)
    ctx._test_slab_cols = ws
    ctx.tune_hash(256, 2048)               # the slab path sends every row to the tile kernel by itself
    yield ctx
    ctx.tune_shared(20000, 16); ctx.tune(18000, 8); ctx.tune_symbolic(0)


def _sparse(ctx, A, B, **kw):
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        return ctx.spgemm_host(a, b, **kw)
    finally:
        a.close(); b.close()


CASES = {
    "random": lambda: (rand_csr(120, 300, 0.05, 1), rand_csr(300, 500, 0.05, 2)),
    "sparse-with-empty-rows": lambda: (rand_csr(200, 150, 0.02, 5), rand_csr(150, 700, 0.01, 6)),
    "dense-rows": lambda: (rand_csr(40, 60, 0.5, 7), rand_csr(60, 333, 0.6, 8)),
    "signed": lambda: (signed(rand_csr(90, 200, 0.1, 9), 10), signed(rand_csr(200, 410, 0.08, 11), 12)),
    "one-row": lambda: (rand_csr(1, 50, 0.5, 13), rand_csr(50, 1000, 0.2, 14)),
    "long-rows-of-a": lambda: (rand_csr(30, 400, 0.6, 15), rand_csr(400, 300, 0.03, 16)),     # > 64 entries per row: several batches
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("exact", [False, True])
def test_slab_path_matches_oracle(slabbed, oracle, case, exact):
    ctx = slabbed
    A, B = CASES[case]()
    ctx.timing(True); ctx.timing_reset()
    got = _sparse(ctx, A, B, exact=exact)
    ran = ctx.kernel_time("smm_runs")[1]
    ctx.timing(False)
    want = oracle.sparse(arrays(A), arrays(B), B.shape[1])
    assert_csr_equal(got, want, values="bits" if exact else "tol")
    if B.shape[1] > ctx._test_slab_cols:         # wider than the slab limit: the slab path (it sends every row to the tile kernel)
        assert ran >= 1


@pytest.mark.parametrize("exact", [False, True])
def test_slab_path_symmetric_and_row_shards(slabbed, oracle, exact):
    ctx = slabbed
    A = rand_csr(260, 180, 0.06, 21)
    B = A.T.tocsr(); B.sort_indices()
    want = oracle.sparse(arrays(A), arrays(B), 260, symmetric=True)
    assert_csr_equal(_sparse(ctx, A, B, symmetric=True, exact=exact), want, values="bits" if exact else "tol")
    # a row shard of the symmetric product: the diagonal moves with a_row_offset
    r0, r1 = 100, 190
    got = _sparse(ctx, A[r0:r1], B, symmetric=True, row_offset=r0, exact=exact)
    lo, hi = int(want[0][r0]), int(want[0][r1])
    assert np.array_equal(got[0], want[0][r0:r1 + 1] - lo) and np.array_equal(got[1], want[1][lo:hi])
    assert np.array_equal(got[2], want[2][lo:hi]) if exact else np.allclose(got[2], want[2][lo:hi], rtol=1e-10, atol=0)


def test_slab_path_plan_replay_and_values_update(slabbed, oracle):
    ctx = slabbed
    A, B = rand_csr(100, 200, 0.1, 31), rand_csr(200, 450, 0.07, 32)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        plan = ctx.spgemm_plan(a, b, exact=True)
        assert_csr_equal(plan.numeric_host(), oracle.sparse(arrays(A), arrays(B), 450), values="bits")
        A2, B2 = signed(A, 33), signed(B, 34)
        a.update_values(A2.data); b.update_values(B2.data)
        assert_csr_equal(plan.numeric_host(), oracle.sparse(arrays(A2), arrays(B2), 450), values="bits")
        assert plan.device_bytes() > 0
        plan.close()
    finally:
        a.close(); b.close()


def test_default_settings_keep_narrow_products_on_one_slab(ctx, oracle):
    """Below 63 456 columns nothing changes: one slab, the hash classes for rows with few products."""
    A, B = rand_csr(150, 120, 0.1, 41), rand_csr(120, 400, 0.1, 42)
    assert_csr_equal(_sparse(ctx, A, B, exact=True), oracle.sparse(arrays(A), arrays(B), 400), values="bits")


@pytest.mark.parametrize("it", range(40))
def test_slab_path_fuzz(ctx, oracle, it):
    """Random tile width / waves / slab limit on random row-length distributions (the generator of the fuzz
    campaign), both modes, symmetric on square cases."""
    from test_gpu_fuzz_campaign import _rand_rows
    r = np.random.default_rng(777000 + it)
    cols = int(r.integers(64, 400)); waves = int(r.choice([4, 8, 16])); ws = int(r.integers(64, 1500))
    m, k = int(r.integers(1, 600)), int(r.integers(1, 600))
    n = m if it % 3 == 0 else int(r.integers(1, 6000))
    A = _rand_rows(r, m, k, 10 ** r.uniform(0, 1.6), int(r.integers(0, 3)))
    B = _rand_rows(r, k, n, 10 ** r.uniform(0, 2.0), int(r.integers(0, 3)))
    if A is None or B is None or np.diff(B.indptr)[A.indices].sum() > 2e7:
        pytest.skip("case larger than the budget")
    ctx.tune_shared(cols, waves); ctx.tune(cols, min(waves, 8)); ctx.tune_symbolic(ws)
    ctx.tune_dense_runs(2 if it % 3 == 1 else 1)          # every third case: the dense-run instantiation of the slab walk
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        for sym in ((False, True) if m == n else (False,)):
            want = oracle.sparse(arrays(A), arrays(B), n, symmetric=sym)
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=sym, exact=True), want, values="bits")
            gp, gi, gv = ctx.spgemm_host(a, b, symmetric=sym)
            assert np.array_equal(np.asarray(gp, np.int64), np.asarray(want[0], np.int64)) and np.array_equal(gi, want[1])
            mag = oracle.sparse(arrays(abs(A)), arrays(abs(B)), n, symmetric=sym)[2]
            assert np.all(np.abs(gv - want[2]) <= 1e-12 * mag)
    finally:
        a.close(); b.close()
        ctx.tune_shared(20000, 16); ctx.tune(18000, 8); ctx.tune_symbolic(0); ctx.tune_dense_runs(1)
