"""The tiny-row kernels (round 4: rows with <= 16 products from <= 16 entries of A, four rows to a wave,
smm_symbolic_tiny / smm_numeric_tiny) against the CPU oracle: indptr / indices bit for bit in the reference's first-touch
order (sparsework.cpp:59-110), values BIT FOR BIT in both modes (a tiny row is always added in the reference's order).
The launch counters prove which kernels ran."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, assert_csr_equal, signed

pytestmark = pytest.mark.gpu


def _rows(lens, n, rng):
    """CSR with (at most) the given row lengths, sorted distinct columns (vectorised: a draw per row, repeats dropped)."""
    lens = np.asarray(lens, dtype=np.int64)
    kmax = int(lens.max()) if len(lens) else 0
    if kmax == 0:
        return sp.csr_matrix((len(lens), n))
    if kmax > 64:                                               # a few long rows: one row at a time
        rows = [np.sort(rng.choice(n, int(k), replace=False)) for k in lens]
        indptr = np.concatenate(([0], np.cumsum([len(r) for r in rows]))).astype(np.int32)
        cols = np.concatenate(rows)
    else:
        draw = np.sort(rng.integers(0, n, size=(len(lens), kmax)), axis=1)
        keep = np.arange(kmax)[None, :] < lens[:, None]
        keep[:, 1:] &= draw[:, 1:] != draw[:, :-1]
        indptr = np.concatenate(([0], np.cumsum(keep.sum(axis=1)))).astype(np.int32)
        cols = draw[keep]
    return sp.csr_matrix((rng.uniform(-1, 1, int(indptr[-1])), cols.astype(np.int32), indptr), shape=(len(lens), n))


def _gpu(ctx, A, B, **kw):
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        ctx.timing(True); ctx.timing_reset()
        out = ctx.spgemm_host(a, b, **kw)
        launches = {k: ctx.kernel_time(k)[1] for k in ("smm_symbolic_tiny", "smm_numeric_tiny", "smm_symbolic_hash", "smm_numeric_hash",
                                                       "smm_symbolic", "smm_numeric")}
        ctx.timing(False)
        return out, launches
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("n,per_row", [(40, 2), (3000, 3), (100_000, 4), (70_000, 1), (50_000, 5), (9000, 8), (20_000, 31)])
def test_every_row_tiny(ctx, oracle, n, per_row, symmetric, exact):
    """per_row <= 4: at most 16 products (16 lanes per row); 5, 8, 31: at most 32 (32 lanes per row, two DPP rows)."""
    rng = np.random.default_rng(n + per_row)
    A = _rows(np.full(n, per_row), n, rng)
    cap = 16 if per_row <= 4 else 32
    B = _rows(rng.integers(0, min(per_row + 2, cap // per_row + 1), n), n, rng)     # some rows of B empty; <= cap products per row
    want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
    got, launches = _gpu(ctx, A, B, symmetric=symmetric, exact=exact)
    assert_csr_equal(got, want, values="bits")
    assert 1 <= launches["smm_symbolic_tiny"] <= 2 and launches["smm_numeric_tiny"] == launches["smm_symbolic_tiny"]
    assert launches["smm_symbolic_hash"] == 0 and launches["smm_numeric_hash"] == 0 and launches["smm_numeric"] == 0


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
def test_rows_around_the_class_limit(ctx, oracle, symmetric, exact):
    """Products 0 ... 40 per row in one matrix: 15, 16 are tiny, 17 is not; 17 entries of A with 16 products are not."""
    rng = np.random.default_rng(5)
    n = 600
    lensB = rng.integers(0, 6, n)
    lensB[:20] = 0                                              # rows of B without entries
    lensB[100:104] = 4; lensB[110] = 1; lensB[111] = 3
    B = _rows(lensB, n, rng)
    lensB = np.diff(B.indptr)                                   # (the generator may drop a repeated draw)
    rowsA = []
    for i in range(n):
        k = int(rng.integers(0, 12))
        rowsA.append(np.sort(rng.choice(n, k, replace=False)))
    rowsA[7] = np.concatenate([np.arange(17), [300]])           # 17 rows of B without entries + one: few products, 18 entries
    rowsA[8] = np.array([100, 101, 102, 103])                   # 16 products (if no draw was dropped)
    rowsA[9] = np.array([100, 101, 102, 103, 110])              # 17
    rowsA[10] = np.array([100, 101, 102, 111])                  # 15
    indptr = np.concatenate(([0], np.cumsum([len(r) for r in rowsA]))).astype(np.int32)
    A = sp.csr_matrix((rng.uniform(-1, 1, int(indptr[-1])), np.concatenate(rowsA).astype(np.int32), indptr), shape=(n, n))
    prods = np.array([int(lensB[r].sum()) for r in rowsA])
    assert (prods > 16).any() and (prods < 16).any() and (prods[8:11] >= 12).all()
    want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
    got, launches = _gpu(ctx, A, B, symmetric=symmetric, exact=exact)
    assert_csr_equal(got, want, values="bits" if exact else "tol")
    assert launches["smm_symbolic_tiny"] == 2 and launches["smm_numeric_tiny"] == 2       # the 16-lane and the 32-lane class
    # the tiny rows alone, value for value (they are added in the reference's order in both modes)
    tiny = np.flatnonzero((prods <= 32) & (np.diff(A.indptr) <= 32))
    gp, gi, gv = got
    wp, wi, wv = want
    for r in tiny:
        assert np.array_equal(gv[gp[r]:gp[r + 1]].view(np.int64), wv[wp[r]:wp[r + 1]].view(np.int64)), r


@pytest.mark.parametrize("exact", [False, True])
def test_tiny_rows_of_an_unsorted_operand_with_repeated_columns(ctx, oracle, exact):
    """B non-canonical: rows shuffled and columns repeated inside a row -- the earliest product wins (sparsework.cpp:70-76)."""
    rng = np.random.default_rng(11)
    n = 2000
    A = _rows(np.full(n, 3), n, rng)
    lens = rng.integers(1, 11, n)                               # up to 30 products: both tiny classes
    indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cols = np.concatenate([rng.integers(0, 6, int(k)) + int(rng.integers(0, n - 6)) for k in lens]).astype(np.int32)   # repeats likely
    B = sp.csr_matrix((rng.uniform(-1, 1, int(indptr[-1])), cols, indptr), shape=(n, n))
    B.has_sorted_indices = False
    B.has_canonical_format = False
    want = oracle.sparse(arrays(A), arrays(B), n)
    got, launches = _gpu(ctx, A, B, exact=exact)
    assert_csr_equal(got, want, values="bits")
    assert launches["smm_symbolic_tiny"] == 2 and launches["smm_numeric_tiny"] == 2


def test_tiny_rows_next_to_long_rows_and_new_values_on_the_same_plan(ctx, oracle):
    rng = np.random.default_rng(13)
    n = 5000
    lens = np.where(rng.random(n) < 0.9, rng.integers(0, 4, n), rng.integers(50, 400, n))
    A = _rows(lens, n, rng)
    B = _rows(np.where(rng.random(n) < 0.9, rng.integers(0, 4, n), rng.integers(50, 400, n)), n, rng)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        plan = ctx.spgemm_plan(a, b)
        plan.check()
        for it in range(2):
            want = oracle.sparse(arrays(A), arrays(B), n)
            assert_csr_equal(plan.numeric_host(), want, values="tol")
            A = signed(A, 100 + it); B = signed(B, 200 + it)
            a.update_values(A.data); b.update_values(B.data)
        plan.close()
    finally:
        a.close(); b.close()


def test_switch_off(ctx, oracle, monkeypatch):
    """SMM_TINY=0 (read when a context is created): the hash kernels take the rows, same result."""
    from sparse_matrix_mult_amd import engine
    monkeypatch.setenv("SMM_TINY", "0")
    c2 = engine.Context(0)
    try:
        rng = np.random.default_rng(17)
        n = 100_000                                         # wide enough for the hash classes of the symbolic phase
        A = _rows(np.full(n, 3), n, rng); B = _rows(np.full(n, 3), n, rng)
        want = oracle.sparse(arrays(A), arrays(B), n)
        got, launches = _gpu(c2, A, B)
        assert_csr_equal(got, want, values="bits")
        assert launches["smm_symbolic_tiny"] == 0 and launches["smm_numeric_tiny"] == 0 and launches["smm_numeric_hash"] >= 1
    finally:
        c2.close()


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("family", ["band", "blocks"])
def test_dense_runs_are_detected_and_walked_by_the_merged_or_kernel(ctx, oracle, family, symmetric):
    """Operands with dense runs of columns (round 4): chosen automatically (>= 80 % of the neighbouring entries of B share a
    bitmap word), same first-touch order as the oracle; a uniform random operand keeps the returning-atomic walk."""
    rng = np.random.default_rng(23)
    n = 3000
    if family == "band":
        offs = np.arange(-60, 61)
        B = sp.diags([rng.uniform(-1, 1, n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
    else:
        B = sp.block_diag([sp.random(150, 150, density=0.6, format="csr", random_state=rng) for _ in range(n // 150)], format="csr")
    B.sort_indices()
    A = sp.random(n, n, density=0.01, format="csr", random_state=rng)
    want = oracle.sparse(arrays(A), arrays(B), n, symmetric=symmetric)
    ctx.tune_hash(0, 0)                                     # every row through the bitmap walk
    try:
        for exact in (False, True):
            got, _ = _gpu(ctx, A, B, symmetric=symmetric, exact=exact)
            assert_csr_equal(got, want, values="bits" if exact else "tol")
    finally:
        ctx.tune_hash(256, 2048)


def test_short_rows_with_a_few_hubs(ctx, oracle):
    """nnz(A) <= 8 rows: the per-lane row-work kernel; rows beyond 32 entries take the listed per-wave kernel."""
    rng = np.random.default_rng(29)
    n = 50_000
    lens = np.full(n, 2)
    lens[[5, 777, 49_999]] = [5000, 33, 20000]
    A = _rows(lens, n, rng)
    B = _rows(rng.integers(0, 5, n), n, rng)
    want = oracle.sparse(arrays(A), arrays(B), n)
    got, launches = _gpu(ctx, A, B, exact=True)
    assert_csr_equal(got, want, values="bits")
    assert launches["smm_symbolic_tiny"] == 1
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        assert np.array_equal(ctx.row_products(a, b), np.diff(B.indptr)[A.indices].astype(np.int64).cumsum()[A.indptr[1:] - 1] -
                              np.concatenate(([0], np.diff(B.indptr)[A.indices].astype(np.int64).cumsum()))[A.indptr[:-1]])
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("narrow", [True, False], ids=["chunked-walk", "plain-bitmap-walk"])
@pytest.mark.parametrize("symmetric", [False, True])
def test_many_short_rows_take_units_sixteen_at_a_time(ctx, oracle, symmetric, narrow):
    """600 000 rows with ~50 ... 400 products each against a narrow B: the chunked symbolic walk with 16 units per counter
    round trip (>= 64 units per wave), same lists and counts as the oracle."""
    rng = np.random.default_rng(31)
    m, k = 600_000, 600
    A = _rows(rng.integers(2, 9, m), k, rng)
    B = _rows(rng.integers(20, 60, k), k if symmetric else 900, rng)
    if symmetric:
        A = sp.csr_matrix((A.data[:A.indptr[k]], A.indices[:A.indptr[k]], A.indptr[:k + 1]), shape=(k, k))       # square: 600 rows only
        m = k
    want = oracle.sparse(arrays(A), arrays(B), B.shape[1], symmetric=symmetric)
    ctx.tune_narrow(narrow)                                 # False: 32-bit lists -> the plain bitmap walk (smm_symbolic)
    try:
        got, launches = _gpu(ctx, A, B, symmetric=symmetric, exact=True)
    finally:
        ctx.tune_narrow(True)
    assert_csr_equal(got, want, values="bits")
    assert launches["smm_symbolic"] >= 1


def test_tiny_rows_of_a_row_shard_keep_the_global_diagonal(ctx, oracle):
    """Symmetric product of a row block (multi-GPU sharding: row_offset): the triangle test uses the global row index."""
    rng = np.random.default_rng(37)
    n = 4000
    A = _rows(np.full(n, 3), n, rng)
    B = _rows(rng.integers(0, 6, n), n, rng)
    wp, wi, wv = oracle.sparse(arrays(A), arrays(B), n, symmetric=True)
    b = ctx.csr_from_scipy(B)
    try:
        for r0, r1 in ((0, 1500), (1500, 1501), (1501, 4000)):
            As = A[r0:r1]
            a = ctx.csr_from_scipy(As)
            try:
                gp, gi, gv = ctx.spgemm_host(a, b, symmetric=True, row_offset=r0)
            finally:
                a.close()
            assert np.array_equal(np.asarray(gp, np.int64), np.asarray(wp[r0:r1 + 1], np.int64) - int(wp[r0]))
            assert np.array_equal(gi, wi[wp[r0]:wp[r1]])
            assert np.array_equal(gv.view(np.int64), wv[wp[r0]:wp[r1]].view(np.int64))
    finally:
        b.close()
