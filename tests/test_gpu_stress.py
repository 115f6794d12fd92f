"""Randomised parity sweep on the GPU: shapes, densities and row-length distributions the
BASELINE configs never produce -- skewed (power-law) rows, fully dense rows inside sparse
matrices (segments longer than a wave, sub-runs longer than a wave), single-row / single-column
operands, very sparse products (most tiles empty).  Every case: indptr / indices bit-exact in
both modes, values bit-exact with SMM_EXACT and <= 1e-10 relative in the default mode, dense
and symmetric variants included."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, assert_csr_equal, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["hash+tiles", "tiles-only", "small-hash", "slab-all", "slab-narrow", "idx32"], autouse=True)
def numeric_paths(request, ctx):
    """Every case runs with the default dispatch (rows with few nonzeros -> LDS hash kernels, the
    rest -> dense LDS tiles), with the hash kernels off, with only the one-wave hash kernel
    on and a low threshold (mixes all three kernels inside one product), and with the row-block x
    column-slab kernels forced (L2-sized slabs for every row; 50-column slabs next to the hash kernels); "idx32"
    switches the 16-bit column stream / lists of the symbolic phase off (every other configuration has them on
    wherever B has < 65535 columns)."""
    hash_cfg, slab_cfg = {"hash+tiles": ((256, 2048), (0, 0, 4)), "tiles-only": ((0, 0), (0, 0, 4)),
                          "small-hash": ((24, 150), (0, 0, 4)), "slab-all": ((0, 0), (2, 0, 4)),
                          "slab-narrow": ((24, 150), (2, 50, 2)), "idx32": ((24, 150), (0, 0, 4))}[request.param]
    ctx.tune_hash(*hash_cfg)
    ctx.tune_slab(*slab_cfg)
    ctx.tune_narrow(request.param != "idx32")
    yield
    ctx.tune_hash(256, 2048)
    ctx.tune_slab(0, 0, 4)
    ctx.tune_narrow(True)
RTOL = 1e-10


def _skewed(m, n, avg, seed, dense_rows=0):
    """Row lengths ~ Pareto (a few very long rows), plus `dense_rows` completely full rows."""
    r = np.random.default_rng(seed)
    lens = np.minimum(n, (r.pareto(1.2, size=m) * avg * 0.4).astype(np.int64))
    if dense_rows:
        lens[r.choice(m, size=min(dense_rows, m), replace=False)] = n
    lens[r.random(m) < 0.1] = 0                        # some empty rows
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    indices = np.concatenate([np.sort(r.choice(n, size=k, replace=False)) for k in lens] + [np.zeros(0, np.int64)])
    data = r.uniform(0.0, 1.0, size=indices.size)
    return sp.csr_matrix((data, indices.astype(np.int32), indptr), shape=(m, n))


CASES = [
    # name, A, B builders
    ("skew_small", lambda: (_skewed(200, 300, 6, 1), _skewed(300, 250, 8, 2))),
    ("skew_dense_rows", lambda: (_skewed(150, 400, 10, 3, dense_rows=3), _skewed(400, 900, 12, 4, dense_rows=4))),
    ("wide_tiles", lambda: (_skewed(120, 200, 8, 5, dense_rows=1), _skewed(200, 45000, 60, 6, dense_rows=2))),
    ("very_sparse", lambda: (sp.random(500, 4000, 0.0008, format="csr", random_state=np.random.default_rng(7)),
                             sp.random(4000, 60000, 0.0003, format="csr", random_state=np.random.default_rng(8)))),
    ("one_row", lambda: (sp.random(1, 500, 0.5, format="csr", random_state=np.random.default_rng(9)),
                         sp.random(500, 700, 0.2, format="csr", random_state=np.random.default_rng(10)))),
    ("one_col", lambda: (sp.random(300, 200, 0.1, format="csr", random_state=np.random.default_rng(11)),
                         sp.random(200, 1, 0.6, format="csr", random_state=np.random.default_rng(12)))),
    ("inner_one", lambda: (sp.random(300, 1, 0.7, format="csr", random_state=np.random.default_rng(13)),
                           sp.random(1, 400, 0.7, format="csr", random_state=np.random.default_rng(14)))),
    ("dense_small", lambda: (sp.csr_matrix(np.random.default_rng(15).random((70, 130))),
                             sp.csr_matrix(np.random.default_rng(16).random((130, 90))))),
    ("long_a_rows", lambda: (sp.random(30, 5000, 0.5, format="csr", random_state=np.random.default_rng(17)),
                             sp.random(5000, 300, 0.01, format="csr", random_state=np.random.default_rng(18)))),
    ("square_sym", lambda: (_skewed(350, 350, 9, 19, dense_rows=2), _skewed(350, 350, 9, 20, dense_rows=2))),
    ("square_sym_wide", lambda: (sp.random(600, 600, 0.02, format="csr", random_state=np.random.default_rng(21)),
                                 sp.random(600, 600, 0.02, format="csr", random_state=np.random.default_rng(22)))),
    # rows of B of 20 000 entries: 313 chunks for one A entry, several 64-chunk descriptor groups per batch
    ("long_b_rows", lambda: (sp.random(40, 60, 0.3, format="csr", random_state=np.random.default_rng(23)),
                             _skewed(60, 20000, 50, 24, dense_rows=5))),
    # 300 000 columns, Pareto row lengths: the rows of one product spread over both hash-marker classes and the bitmap
    ("skew_wide_markers", lambda: (_skewed(400, 300, 6, 25, dense_rows=2), _skewed(300, 300000, 10, 26))),
    # more than 64 A entries per row with empty rows of B in between (entries without chunks inside a batch)
    ("gappy_b", lambda: (sp.random(50, 900, 0.4, format="csr", random_state=np.random.default_rng(27)),
                         _skewed(900, 2000, 3, 28))),
]


@pytest.mark.parametrize("name,build", CASES, ids=[c[0] for c in CASES])
def test_random_sweep(ctx, oracle, name, build):
    A, B = build()
    A, B = A.tocsr(), B.tocsr()
    A.sort_indices(); B.sort_indices()
    m, n = A.shape[0], B.shape[1]
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        for symmetric in ((False, True) if m == n else (False,)):
            want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=symmetric, exact=True), want, values="bits")
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=symmetric), want, values="tol", rtol=RTOL)
            if m * n <= 8_000_000:
                wd = oracle.dense(arrays(A), arrays(B), n, symmetric=symmetric)
                got = ctx.dense_host(a, b, symmetric=symmetric, exact=True)
                assert np.array_equal(got.view(np.int64), wd.view(np.int64))
                assert rel_err(ctx.dense_host(a, b, symmetric=symmetric), wd) <= RTOL
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("lds_cols,waves", [(128, 4), (1024, 8), (20000, 16)])
def test_random_sweep_small_tiles(ctx, oracle, lds_cols, waves):
    """The same skewed case under tiny tiles: hundreds of tiles per row, sub-runs of one entry."""
    A, B = _skewed(150, 400, 10, 3, dense_rows=3), _skewed(400, 900, 12, 4, dense_rows=4)
    want = oracle.sparse(arrays(A), arrays(B), 900)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    ctx.tune_shared(lds_cols, waves)
    ctx.tune(lds_cols, min(waves, 8))
    try:
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), want, values="bits")
        assert_csr_equal(ctx.spgemm_host(a, b), want, values="tol", rtol=RTOL)
    finally:
        ctx.tune_shared(20000, 16); ctx.tune(18000, 8)
        a.close(); b.close()


def test_triple_skewed(ctx, oracle):
    H = _skewed(130, 700, 15, 31, dense_rows=2)
    S = sp.random(700, 700, 0.01, format="csr", random_state=np.random.default_rng(32))
    Q = (S + S.T).tocsr(); Q.sort_indices()
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        for full in (0, 1):
            want = oracle.triple(arrays(H), arrays(Q), 700, full=full)
            got = ctx.triple_host(h, q, full=bool(full), exact=True)
            assert np.array_equal(got.view(np.int64), want.view(np.int64))
            assert rel_err(ctx.triple_host(h, q, full=bool(full)), want) <= RTOL
    finally:
        h.close(); q.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzz(ctx, oracle, seed):
    """Random shapes / densities / value signs, both modes, sparse + dense + (square) symmetric."""
    r = np.random.default_rng(1000 + seed)
    m, k, n = (int(x) for x in r.integers(1, 400, size=3))
    if seed % 3 == 0:
        n = m                                              # square: exercises the symmetric variants
    if seed % 5 == 0:
        n = int(r.integers(3000, 30000))                   # several coarse tiles
    da, db = (float(x) for x in 10 ** r.uniform(-2.5, -0.3, size=2))
    A = sp.random(m, k, density=da, format="csr", random_state=r)
    B = sp.random(k, n, density=db, format="csr", random_state=r)
    A.data = r.uniform(-1, 1, size=A.nnz); B.data = r.uniform(-1, 1, size=B.nnz)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        for symmetric in ((False, True) if m == n else (False,)):
            want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
            got = ctx.spgemm_host(a, b, symmetric=symmetric, exact=True)
            assert_csr_equal(got, want, values="bits")
            got = ctx.spgemm_host(a, b, symmetric=symmetric)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            assert np.allclose(got[2], want[2], rtol=1e-9, atol=1e-12)     # signed data: cancellation
            if m * n <= 4_000_000:
                wd = oracle.dense(arrays(A), arrays(B), n, symmetric=symmetric)
                assert np.array_equal(ctx.dense_host(a, b, symmetric=symmetric, exact=True).view(np.int64), wd.view(np.int64))
    finally:
        a.close(); b.close()
