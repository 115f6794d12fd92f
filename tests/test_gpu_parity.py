"""Parity of the HIP path against the CPU oracle, through the C ABI (run on the MI355X box).

Bar (north star / SURVEY 8a): indptr and indices bit-exact in the reference's first-touch
order; float64 values within 1e-10 relative.  Every case runs in both modes: the default
(waves add concurrently; values held to 1e-10) and SMM_EXACT (reference order; values asserted
bit for bit, for operands whose rows are sorted -- every scipy-built CSR).  Only the general
path for unsorted B (global atomics) is held to 1e-10 in both modes.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import wide_csr, arrays, assert_csr_equal, rand_csr, rel_err, shuffle_rows, signed

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["hash+tiles", "tiles-only", "small-hash", "slab-all", "slab-narrow", "idx32", "dense-runs"], autouse=True)
def numeric_paths(request, ctx):
    """Every case runs with the default dispatch (rows with few nonzeros -> LDS hash kernels, the
    rest -> dense LDS tiles), with the hash kernels off, with only the one-wave hash kernel
    on and a low threshold (mixes all three kernels inside one product), and with the row-block x
    column-slab kernels forced (L2-sized slabs for every row; 50-column slabs next to the hash kernels); "idx32"
    switches the 16-bit column stream / lists of the symbolic phase off (every other configuration has them on
    wherever B has < 65535 columns); "dense-runs" forces the symbolic walk's instantiation for operands with dense runs of
    columns (plain reads + merged ORs instead of returning atomics, round 4) on every operand, hash kernels off."""
    hash_cfg, slab_cfg = {"hash+tiles": ((256, 2048), (0, 0, 4)), "tiles-only": ((0, 0), (0, 0, 4)),
                          "small-hash": ((24, 150), (0, 0, 4)), "slab-all": ((0, 0), (2, 0, 4)),
                          "slab-narrow": ((24, 150), (2, 50, 2)), "idx32": ((24, 150), (0, 0, 4)),
                          "dense-runs": ((0, 0), (0, 0, 4))}[request.param]
    ctx.tune_hash(*hash_cfg)
    ctx.tune_slab(*slab_cfg)
    ctx.tune_narrow(request.param != "idx32")
    ctx.tune_dense_runs(2 if request.param == "dense-runs" else 1)
    yield
    ctx.tune_hash(256, 2048)
    ctx.tune_slab(0, 0, 4)
    ctx.tune_narrow(True)
    ctx.tune_dense_runs(1)

RTOL = 1e-10   # north star: "float64 values within 1e-10 relative"

CASES = [
    # (m, k, n, dA, dB)
    (1, 1, 1, 1.0, 1.0),
    (7, 5, 9, 0.5, 0.5),
    (64, 64, 64, 0.1, 0.1),
    (130, 70, 257, 0.08, 0.05),
    (500, 500, 500, 0.3, 0.3),          # reference tests/test_computation_speed.py:10-15
    (500, 400, 500, 0.1, 0.1),          # reference tests/test_with_dense.py non-square case
    (1000, 1000, 1000, 0.05, 0.05),     # BASELINE config 1
    (300, 2000, 20000, 0.01, 0.004),    # several coarse tiles, long rows
    (2000, 300, 40000, 0.02, 0.002),    # many tiles, short B rows
]


# default mode: values to rounding (held to the north star's 1e-10); SMM_EXACT: bit for bit
MODES = [pytest.param(False, id="default"), pytest.param(True, id="exact")]


def _check_values(exact):
    return "bits" if exact else "tol"


def _gpu_sparse(ctx, A, B, **kw):
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        return ctx.spgemm_host(a, b, **kw)
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("m,k,n,da,db", CASES)
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("exact", MODES)
def test_sparse_matches_oracle(ctx, oracle, m, k, n, da, db, symmetric, exact):
    if symmetric and m != n:
        pytest.skip("symmetric needs a square result")
    A, B = rand_csr(m, k, da, 1), rand_csr(k, n, db, 2)
    want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
    got = _gpu_sparse(ctx, A, B, symmetric=symmetric, exact=exact)
    assert_csr_equal(got, want, values=_check_values(exact), rtol=RTOL)


@pytest.mark.parametrize("n,ka,kb", [(200000, 10, 10), (200000, 16, 16), (200000, 45, 45), (70000, 12, 20), (1000000, 8, 12),
                                     (300000, 3, 700), (1200000, 50, 50)])       # the last: bitmap in global memory
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("exact", MODES)
def test_sparse_wide_matrices_hash_marker(ctx, oracle, n, ka, kb, symmetric, exact):
    """Very wide B: the symbolic phase marks columns in an LDS hash set for rows with <= 256 / <= 2048
    products and in the bitmap for the rest; the cases sit inside the classes and on their boundaries."""
    m = 600
    A, B = wide_csr(m, 500, ka, 5), wide_csr(500, n, kb, 6)
    if symmetric:                       # square result needed: pad A's rows to n
        A = sp.vstack([A, sp.csr_matrix((n - m, 500))]).tocsr()
    want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
    got = _gpu_sparse(ctx, A, B, symmetric=symmetric, exact=exact)
    assert_csr_equal(got, want, values=_check_values(exact), rtol=RTOL)


@pytest.mark.parametrize("m", [32768, 32769, 100003])
def test_sparse_many_rows(ctx, oracle, m):
    """Row counts around and past the single-workgroup scan's limit (the row pointer comes from a
    three-launch tiled scan above 32 768 rows), with empty rows in between."""
    A, B = wide_csr(m, 300, 3, 7), wide_csr(300, 5000, 6, 8)
    A = A.tolil(); A[5:900] = 0; A = A.tocsr(); A.eliminate_zeros()
    want = oracle.sparse(arrays(A), arrays(B), 5000)
    assert_csr_equal(_gpu_sparse(ctx, A, B, exact=True), want, values="bits")


@pytest.mark.parametrize("lds_cols,waves", [(64, 1), (256, 4), (512, 8), (1000, 2), (5000, 1), (16384, 4), (20000, 8), (17000, 16), (300, 16)])
def test_sparse_tile_geometries(ctx, oracle, lds_cols, waves):
    """Every tile geometry must give the same bits (tiles only change who adds, not the order)."""
    A, B = signed(rand_csr(300, 400, 0.05, 3), 30), signed(rand_csr(400, 3000, 0.03, 4), 40)
    want = oracle.sparse(arrays(A), arrays(B), 3000)
    ctx.tune(lds_cols, waves)
    try:
        assert_csr_equal(_gpu_sparse(ctx, A, B, exact=True), want, values="bits")
    finally:
        ctx.tune(18000, 8)


@pytest.mark.parametrize("lds_cols,waves", [(64, 4), (300, 8), (1000, 16), (16384, 16), (20000, 8)])
def test_sparse_shared_tile_geometries(ctx, oracle, lds_cols, waves):
    """Default walk: signed values (cancellation), so compare against the magnitude of the sums."""
    A, B = signed(rand_csr(300, 400, 0.05, 3), 30), signed(rand_csr(400, 3000, 0.03, 4), 40)
    want = oracle.sparse(arrays(A), arrays(B), 3000)
    ctx.tune_shared(lds_cols, waves)
    try:
        got = _gpu_sparse(ctx, A, B)
    finally:
        ctx.tune_shared(20000, 16)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.allclose(got[2], want[2], rtol=1e-10, atol=1e-13)


def test_sparse_structural_zeros_and_signed_zero(ctx, oracle):
    """cancelled sums stay as explicit entries; -0.0 products keep their sign."""
    A = sp.csr_matrix(np.array([[1.0, -1.0, 0.0], [0.0, 0.0, -2.0], [3.0, 0.0, 0.0]]))
    B = sp.csr_matrix((np.array([1.0, 1.0, 0.0, 5.0]), np.array([0, 0, 1, 2]), np.array([0, 1, 2, 4])), shape=(3, 3))
    want = oracle.sparse(arrays(A), arrays(B), 3)
    got = _gpu_sparse(ctx, A, B, exact=True)
    assert_csr_equal(got, want, values="bits")
    assert_csr_equal(_gpu_sparse(ctx, A, B), want, values="bits")      # single products: exact in any order
    assert got[0][-1] == want[0][-1] == 4          # (0,0) cancels to 0.0 and is kept as an entry
    assert got[2][0] == 0.0 and got[1][0] == 0
    assert np.signbit(got[2][1]) and got[2][1] == 0.0   # -2 * 0.0 = -0.0 keeps its sign (first-touch store)


def test_sparse_empty_rows_and_trailing_zero_rows(ctx, oracle):
    """reference tests/test_edge_case.py:14-21,62-66: trailing all-zero rows."""
    A = sp.csr_matrix(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9], [0, 0, 0], [0, 0, 0], [0, 0, 0]], dtype=float))
    B = sp.csr_matrix(np.random.default_rng(0).random((3, 4)))
    want = oracle.sparse(arrays(A), arrays(B), 4)
    assert_csr_equal(_gpu_sparse(ctx, A, B, exact=True), want, values="bits")
    assert_csr_equal(_gpu_sparse(ctx, A, B), want, values="tol", rtol=RTOL)
    # empty rows in the middle of A and empty rows of B
    A2 = rand_csr(200, 150, 0.02, 5); B2 = rand_csr(150, 180, 0.02, 6)
    want = oracle.sparse(arrays(A2), arrays(B2), 180)
    assert_csr_equal(_gpu_sparse(ctx, A2, B2, exact=True), want, values="bits")
    assert_csr_equal(_gpu_sparse(ctx, A2, B2), want, values="tol", rtol=RTOL)


def test_sparse_unsorted_and_duplicate_inputs(ctx, oracle):
    """Non-canonical operands (reference matrix_ops.py:307-310 neither sorts nor dedups).
    Unsorted B takes the general numeric path (global atomics): indices exact, values to
    rounding.  Unsorted A only changes the order of the steps and stays bit-exact."""
    A, B = rand_csr(150, 120, 0.1, 7), rand_csr(120, 400, 0.1, 8)
    Au, Bu = shuffle_rows(A, 70), shuffle_rows(B, 80)
    want = oracle.sparse(arrays(Au), arrays(B), 400)
    assert_csr_equal(_gpu_sparse(ctx, Au, B, exact=True), want, values="bits")
    assert_csr_equal(_gpu_sparse(ctx, Au, B), want, values="tol", rtol=RTOL)
    want = oracle.sparse(arrays(Au), arrays(Bu), 400)
    assert_csr_equal(_gpu_sparse(ctx, Au, Bu), want, values="tol", rtol=RTOL)
    # round 4: SMM_EXACT keeps its promise for unsorted B too (ordered read-modify-write instead of atomics)
    assert_csr_equal(_gpu_sparse(ctx, Au, Bu, exact=True), want, values="bits")
    want = oracle.sparse(arrays(Au), arrays(Bu), 400, symmetric=False)
    # duplicates: repeat some entries of B's rows (sorted, repeated columns)
    ip, ix, dv = arrays(B)
    rep = np.repeat(np.arange(len(ix)), 1 + (np.arange(len(ix)) % 3 == 0))
    cnt = np.bincount(np.searchsorted(ip, rep, side="right") - 1, minlength=B.shape[0])
    ip2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    Bd = (ip2, ix[rep].copy(), dv[rep].copy())
    want = oracle.sparse(arrays(A), Bd, 400)
    a = ctx.csr_from_scipy(A); b = ctx.csr_from_arrays(120, 400, *Bd)
    try:
        assert not b.is_canonical()
        got = ctx.spgemm_host(a, b)
        got_exact = ctx.spgemm_host(a, b, exact=True)
    finally:
        a.close(); b.close()
    assert_csr_equal(got, want, values="tol", rtol=RTOL)
    assert_csr_equal(got_exact, want, values="bits")          # repeated columns: same-accumulator lanes of one step, in lane order
    # unsorted AND repeated: rows of B shuffled after the duplication (the ordered path serialises the lanes that meet)
    r = np.random.default_rng(81)
    ix2, dv2 = Bd[1].copy(), Bd[2].copy()
    for i in range(120):
        s, e = ip2[i], ip2[i + 1]
        o = r.permutation(e - s)
        ix2[s:e], dv2[s:e] = ix2[s:e][o], dv2[s:e][o]
    Bs = (ip2, ix2, dv2)
    a = ctx.csr_from_scipy(signed(A, 82)); b = ctx.csr_from_arrays(120, 400, *Bs)
    try:
        for symmetric in (False,):
            want = oracle.sparse(arrays(signed(A, 82)), Bs, 400, symmetric=symmetric)
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=symmetric, exact=True), want, values="bits")
            wd = oracle.dense(arrays(signed(A, 82)), Bs, 400, symmetric=symmetric)
            gd = ctx.dense_host(a, b, symmetric=symmetric, exact=True)
            assert np.array_equal(gd.view(np.int64), wd.view(np.int64))
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("m,k,n,da,db", CASES)
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("exact", MODES)
def test_dense_matches_oracle(ctx, oracle, m, k, n, da, db, symmetric, exact):
    if symmetric and m != n:
        pytest.skip("symmetric needs a square result")
    # exact mode: signed values (cancellation, signed zeros); default mode: uniform[0,1) as BASELINE
    A, B = rand_csr(m, k, da, 11), rand_csr(k, n, db, 12)
    if exact:
        A, B = signed(A, 1), signed(B, 2)
    want = oracle.dense(arrays(A), arrays(B), n, symmetric=symmetric)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        got = ctx.dense_host(a, b, symmetric=symmetric, exact=exact)
    finally:
        a.close(); b.close()
    if exact:
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), f"max rel {rel_err(got, want):.3e}"
    else:
        assert rel_err(got, want) <= RTOL


def test_dense_unsorted_b(ctx, oracle):
    A, B = rand_csr(90, 80, 0.1, 13), shuffle_rows(rand_csr(80, 300, 0.1, 14), 15)
    want = oracle.dense(arrays(A), arrays(B), 300)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        got = ctx.dense_host(a, b)
    finally:
        a.close(); b.close()
    assert rel_err(got, want) <= RTOL


@pytest.mark.parametrize("n,k,dh,dq", [(1, 1, 1.0, 1.0), (60, 90, 0.1, 0.1), (500, 500, 0.3, 0.3),
                                       (300, 9000, 0.02, 0.004), (257, 5000, 0.05, 0.01),
                                       (1500, 2500, 0.02, 0.004)])      # two k-groups of 1024 rows, three column chunks
@pytest.mark.parametrize("full", [0, 1])
@pytest.mark.parametrize("exact", MODES)
def test_triple_matches_oracle(ctx, oracle, n, k, dh, dq, full, exact):
    H = rand_csr(n, k, dh, 21)
    S = rand_csr(k, k, dq / 2, 22)
    Q = (S + S.T).tocsr()
    want = oracle.triple(arrays(H), arrays(Q), k, full=full)
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        got = ctx.triple_host(h, q, full=bool(full), exact=exact)
    finally:
        h.close(); q.close()
    if exact:
        assert np.array_equal(got.view(np.int64), want.view(np.int64)), f"max rel {rel_err(got, want):.3e}"
    else:
        assert rel_err(got, want) <= RTOL


def test_triple_step_order_edge_cases(ctx, oracle):
    """Default mode deals the entries of a row to the ELL steps so that a wave's LDS reads do not collide
    (smm_ell_fill<true>).  Edge cases of that scheduler: every column of H in ONE residue class mod 16 (no
    collision-free order exists: it must still place every entry), rows of very different length inside one
    64-row slice (lanes with and without steps to spare), and the same handle used in default, exact and default
    mode again (the ELL copy is rebuilt for the order each mode needs)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(77)
    n, k = 200, 1600
    # (a) columns = 3 mod 16 only
    cols = np.arange(3, k, 16)
    Ha = sp.random(n, len(cols), density=0.4, format="csr", random_state=rng)
    Ha = sp.csr_matrix((Ha.data, cols[Ha.indices], Ha.indptr), shape=(n, k))
    # (b) ragged: row r has about r % 64 entries
    rows = [np.sort(rng.choice(k, size=(r % 64) + (3 if r % 7 == 0 else 0), replace=False)) for r in range(n)]
    ptr = np.concatenate([[0], np.cumsum([len(x) for x in rows])])
    Hb = sp.csr_matrix((rng.standard_normal(ptr[-1]), np.concatenate(rows), ptr), shape=(n, k))
    S = rand_csr(k, k, 0.004, 78); Q = (S + S.T).tocsr()
    for H in (Ha, Hb):
        H.sort_indices()
        want = oracle.triple(arrays(H), arrays(Q), k, full=0)
        h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
        try:
            d1 = ctx.triple_host(h, q, exact=False)
            e1 = ctx.triple_host(h, q, exact=True)
            d2 = ctx.triple_host(h, q, exact=False)
        finally:
            h.close(); q.close()
        assert np.array_equal(e1.view(np.int64), want.view(np.int64))
        assert rel_err(d1, want) <= RTOL
        assert rel_err(d2, want) <= RTOL


@pytest.mark.parametrize("exact", MODES)
def test_triple_many_k_groups(ctx, oracle, exact):
    """7 k-groups of 1024 rows = two super-groups of the stage-2 block order (5 + 2), a row-block count that is
    not a multiple of 8 and a last k-group that is mostly past n; the whole result against the oracle."""
    n, k = 6200, 320
    H = rand_csr(n, k, 0.02, 91); S = rand_csr(k, k, 0.03, 92); Q = (S + S.T).tocsr()
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        for full in (0, 1):
            want = oracle.triple(arrays(H), arrays(Q), k, full=full)
            got = ctx.triple_host(h, q, full=bool(full), exact=exact)
            if exact:
                assert np.array_equal(got.view(np.int64), want.view(np.int64)), f"full={full}: max rel {rel_err(got, want):.3e}"
            else:
                assert rel_err(got, want) <= RTOL
            del want, got
    finally:
        h.close(); q.close()


@pytest.mark.parametrize("group", [1, 2, 3, 7])
def test_triple_block_order_variants(oracle, group, monkeypatch):
    """SMM_S2_GROUP (k-groups per super-group of the stage-2 block order) only permutes workgroups: every value
    gives the same bits, also when it does not divide the number of k-groups (3 here) or exceeds it."""
    from sparse_matrix_mult_amd.engine import Context
    monkeypatch.setenv("SMM_S2_GROUP", str(group))
    n, k = 2200, 300
    H = rand_csr(n, k, 0.03, 95); S = rand_csr(k, k, 0.03, 96); Q = (S + S.T).tocsr()
    want = oracle.triple(arrays(H), arrays(Q), k, full=0)
    c2 = Context(0)
    try:
        h, q = c2.csr_from_scipy(H), c2.csr_from_scipy(Q)
        got = c2.triple_host(h, q, exact=True)
        h.close(); q.close()
    finally:
        c2.close()
    assert np.array_equal(got.view(np.int64), want.view(np.int64))


def test_triple_row_range(ctx, oracle):
    H = rand_csr(200, 300, 0.05, 23); S = rand_csr(300, 300, 0.02, 24); Q = (S + S.T).tocsr()
    want = oracle.triple(arrays(H), arrays(Q), 300, full=0)
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        top = ctx.triple_host(h, q, row_begin=0, row_end=77, exact=True)
        bot = ctx.triple_host(h, q, row_begin=77, row_end=200, exact=True)
    finally:
        h.close(); q.close()
    assert np.array_equal(np.vstack([top, bot]), want)


def test_triple_row_range_across_k_groups(ctx, oracle):
    """Row ranges that start inside a 16-row block and end past the first 1024-row k-group."""
    H = rand_csr(1300, 1100, 0.02, 25); S = rand_csr(1100, 1100, 0.004, 26); Q = (S + S.T).tocsr()
    want = oracle.triple(arrays(H), arrays(Q), 1100, full=0)
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        parts = [ctx.triple_host(h, q, row_begin=r0, row_end=r1, exact=True) for r0, r1 in ((0, 1003), (1003, 1030), (1030, 1300))]
    finally:
        h.close(); q.close()
    assert np.array_equal(np.vstack(parts), want)


def test_row_shards_concatenate_to_single_result(ctx, oracle):
    """contiguous row shards, concatenated in order, ARE the single-device CSR."""
    A, B = rand_csr(400, 300, 0.05, 31), rand_csr(300, 400, 0.05, 32)
    for symmetric in (False, True):
        want = oracle.sparse(arrays(A), arrays(B), 400, symmetric=symmetric)
        b = ctx.csr_from_scipy(B)
        ptrs, idxs, vals, base = [np.zeros(1, np.int64)], [], [], 0
        try:
            for r0, r1 in ((0, 90), (90, 250), (250, 400)):
                a = ctx.csr_from_scipy(A[r0:r1])
                p, i, v = ctx.spgemm_host(a, b, symmetric=symmetric, row_offset=r0, exact=True)
                a.close()
                ptrs.append(p[1:] + base); base += p[-1]; idxs.append(i); vals.append(v)
        finally:
            b.close()
        got = (np.concatenate(ptrs), np.concatenate(idxs), np.concatenate(vals))
        assert_csr_equal(got, want, values="bits")


def test_very_wide_b_uses_the_global_bitmap(ctx, oracle):
    """More than 262 144 columns: the first-touch marker no longer fits LDS and lives in HBM
    (smm_symbolic<LDSBM=false>); many coarse tiles, mostly empty segments."""
    n = 300_000
    A, B = rand_csr(40, 200, 0.2, 41), rand_csr(200, n, 0.0005, 42)
    for symmetric in (False,):
        want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
        a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
        try:
            assert_csr_equal(ctx.spgemm_host(a, b, exact=True), want, values="bits")
            assert_csr_equal(ctx.spgemm_host(a, b), want, values="tol", rtol=RTOL)
        finally:
            a.close(); b.close()


def test_unsorted_b_symmetric_and_dense_general_paths(ctx, oracle):
    A, B = rand_csr(100, 90, 0.1, 43), shuffle_rows(rand_csr(90, 100, 0.1, 44), 45)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        for symmetric in (False, True):
            want = oracle.sparse(arrays(A), arrays(B), 100, symmetric=symmetric)
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=symmetric), want, values="tol", rtol=RTOL)
            wd = oracle.dense(arrays(A), arrays(B), 100, symmetric=symmetric)
            assert rel_err(ctx.dense_host(a, b, symmetric=symmetric), wd) <= RTOL
            # round 4: bit-exact under SMM_EXACT on the general paths as well
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=symmetric, exact=True), want, values="bits")
            assert np.array_equal(ctx.dense_host(a, b, symmetric=symmetric, exact=True).view(np.int64), wd.view(np.int64))
    finally:
        a.close(); b.close()


def test_malformed_operand_is_rejected(ctx):
    from sparse_matrix_mult_amd.engine import SmmError
    indptr = np.array([0, 2, 3], dtype=np.int32)
    with pytest.raises(SmmError):
        ctx.csr_from_arrays(2, 3, indptr, np.array([0, 7, 1], dtype=np.int32), np.ones(3))      # column out of range
    with pytest.raises(SmmError):
        ctx.csr_from_arrays(2, 3, np.array([0, 3, 2], dtype=np.int32), np.array([0, 1, 2], dtype=np.int32), np.ones(3))
    A = ctx.csr_from_scipy(sp.identity(3, format="csr")); B = ctx.csr_from_scipy(sp.identity(4, format="csr"))
    with pytest.raises(SmmError):
        ctx.spgemm_host(A, B)
    A.close(); B.close()


def test_plans_of_both_modes_on_one_b_run_in_any_order(ctx, oracle):
    """the tile index lives on the operand; a plan keeps ITS geometry's index, so an
    SMM_EXACT plan and a default plan on the same B (plus a dense product and a triple product
    with other geometries in between) may run their numeric phases in any order."""
    import ctypes
    A, B = rand_csr(400, 300, 0.05, 1), rand_csr(300, 30000, 0.01, 2)
    want = oracle.sparse(arrays(A), arrays(B), 30000)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    ctx.tune_hash(0, 0)                                   # every row through the tile kernels
    try:
        p_exact = ctx.spgemm_plan(a, b, exact=True)
        p_deflt = ctx.spgemm_plan(a, b, exact=False)
        ctx.dense_host(a, b, exact=True)                  # both geometries touched again
        ctx.tune(9000, 4)                                 # a third one for later plans
        p_third = ctx.spgemm_plan(a, b, exact=True)
        ctx.tune(18000, 8)
        for plan, values in ((p_deflt, "tol"), (p_exact, "bits"), (p_third, "bits")):   # not the creation order
            indptr = np.empty(a.rows + 1, np.int64)
            indices, data = np.empty(plan.nnz, np.int32), np.empty(plan.nnz, np.float64)
            from sparse_matrix_mult_amd._lib import check
            check(ctx.lib, ctx.lib.smm_spgemm_numeric_host(ctx.handle, plan.handle, ctypes.c_void_p(indptr.ctypes.data),
                                                           ctypes.c_void_p(indices.ctypes.data),
                                                           ctypes.c_void_p(data.ctypes.data)))
            assert_csr_equal((indptr, indices, data), want, values=values, rtol=RTOL)
        for p in (p_exact, p_deflt, p_third):
            p.close()
    finally:
        ctx.tune_hash(256, 2048)
        a.close(); b.close()


@pytest.mark.parametrize("full", [False, True])
def test_triple_unsorted_h_and_narrow_q(ctx, oracle, full):
    """The reference's triple_product sums in H's stored order whatever it is and only needs Q's
    columns to fit temp_values[K] (src/sparse_sparse_dense.cpp:178-212): an H with unsorted rows
    and a K x c Q with c < K are legal."""
    H = shuffle_rows(rand_csr(90, 150, 0.08, 3), 4)
    S = sp.random(150, 150, density=0.04, format="csr", random_state=np.random.default_rng(5))
    Q = (S + S.T).tocsr()
    Qn = Q[:, :100].tocsr()                                # 150 x 100
    for q in (Q, Qn):
        want = oracle.triple(arrays(H), arrays(q), 150, int(full))
        h, qd = ctx.csr_from_scipy(H), ctx.csr_from_scipy(q)
        try:
            got = ctx.triple_host(h, qd, full=full, exact=True)
        finally:
            h.close(); qd.close()
        assert np.array_equal(got.view(np.int64), want.view(np.int64))
    Hs = rand_csr(90, 150, 0.08, 3)                        # sorted H, narrow Q: the ELL path
    want = oracle.triple(arrays(Hs), arrays(Qn), 150, int(full))
    h, qd = ctx.csr_from_scipy(Hs), ctx.csr_from_scipy(Qn)
    try:
        got = ctx.triple_host(h, qd, full=full, exact=True)
    finally:
        h.close(); qd.close()
    assert np.array_equal(got.view(np.int64), want.view(np.int64))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 200, 777])
def test_mirror_epilogue_dense_and_triple(ctx, oracle, n):
    """SURVEY 8f-2 (opt-in, not in the reference): the lower triangle becomes the mirror image of the
    upper one on the device.  For C = A A^T both triangles hold the same products in the same order, so
    the mirrored symmetric product equals the non-symmetric one bit for bit."""
    A = rand_csr(n, max(1, n // 2 + 3), 0.2, 31)
    if A.nnz == 0:
        A = sp.csr_matrix(np.ones((n, max(1, n // 2 + 3))))
    B = A.T.tocsr()
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        got = ctx.dense_host(a, b, symmetric=True, exact=True, mirror=True)
        upper = ctx.dense_host(a, b, symmetric=True, exact=True)
    finally:
        a.close(); b.close()
    want = oracle.dense(arrays(A), arrays(B), n, symmetric=False)
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    assert np.array_equal(np.triu(got).view(np.int64), upper.view(np.int64))
    # triple product: upper triangle untouched, lower = its transpose
    S = sp.random(A.shape[1], A.shape[1], density=0.1, format="csr", random_state=np.random.default_rng(32))
    Q = (S + S.T).tocsr()
    h, q = ctx.csr_from_scipy(A), ctx.csr_from_scipy(Q)
    try:
        got = ctx.triple_host(h, q, exact=True, mirror=True)
    finally:
        h.close(); q.close()
    up = oracle.triple(arrays(A), arrays(Q), A.shape[1], 0)
    assert np.array_equal(np.triu(got).view(np.int64), up.view(np.int64))
    assert np.array_equal(got, got.T)
    assert np.allclose(got, (A @ Q @ A.T).toarray(), rtol=1e-10, atol=1e-13)


def test_mirror_through_the_python_api(oracle):
    import sparse_matrix_mult_amd as pkg
    A = rand_csr(150, 90, 0.15, 41)
    S = sp.random(90, 90, density=0.1, format="csr", random_state=np.random.default_rng(42))
    Q = (S + S.T).tocsr()
    old_exact = pkg.set_exact(True)                      # bitwise comparisons between separate calls below
    try:
        _mirror_api_checks(pkg, A, Q)
    finally:
        pkg.set_exact(old_exact)


def _mirror_api_checks(pkg, A, Q):
    up = pkg.sparse_matrix_multiply(A, Q, use_triple_product=True)
    full = pkg.sparse_matrix_multiply(A, Q, use_triple_product=True, compute_full_matrix='mirror')
    assert np.array_equal(np.triu(full), up) and np.array_equal(full, full.T) and np.all(np.tril(up, -1) == 0.0)
    with pytest.raises(ValueError):
        pkg.sparse_matrix_multiply(A, A.T.tocsr(), compute_full_matrix='mirror')       # only for the triple product
    old = pkg.set_full_symmetric(True)
    try:
        D = pkg.sparse_matrix_multiply(A, A.T.tocsr(), output_format='dense', symmetric=True)
        assert np.array_equal(D, D.T) and np.allclose(D, (A @ A.T).toarray(), rtol=1e-10, atol=0)
        assert np.array_equal(pkg.sparse_matrix_multiply(A, Q, use_triple_product=True), full)
        C = pkg.sparse_matrix_multiply(A, A.T.tocsr(), output_format='sparse', symmetric=True)   # round 3: CSR is mirrored too
        assert abs(C - C.T).nnz == 0 and np.allclose(C.toarray(), (A @ A.T).toarray(), rtol=1e-10, atol=0)
        assert (sp.tril(C, -1)).nnz == (sp.triu(C, 1)).nnz > 0
    finally:
        pkg.set_full_symmetric(old)
    D = pkg.sparse_matrix_multiply(A, A.T.tocsr(), output_format='dense', symmetric=True)
    assert np.all(np.tril(D, -1) == 0.0)                                                # default: reference behaviour
