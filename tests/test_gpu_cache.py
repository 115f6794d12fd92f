"""SURVEY 8f-3 on the GPU: the operand cache of sparse_matrix_multiply() under its DEFAULT settings can never
serve a stale operand (the reference marshals the caller's current arrays on every call,
matrix_ops.py:339-340, :187-202), a repeated sparsity pattern with new values re-runs only the numeric phase
(README.md:5,13: covariance products), and operands can be pinned explicitly."""
import numpy as np
import pytest

from helpers import arrays, assert_csr_equal, rand_csr, signed

pytestmark = pytest.mark.gpu

SYMBOLIC_KERNELS = ("smm_symbolic", "smm_symbolic_hash", "smm_runs", "smm_segptr", "smm_validate", "smm_row_work", "smm_bin_rows",
                    "smm_pack_count", "smm_idx16", "smm_loc16")


@pytest.fixture()
def pkg():
    import sparse_matrix_mult_amd as p
    p.clear_cache()
    from sparse_matrix_mult_amd import matrix_ops
    assert matrix_ops._cache_entries == 4, "these tests run under the default cache setting"
    old = p.set_exact(True)                      # reference-order sums: every comparison below is bit for bit
    yield p
    p.set_exact(old)
    p.clear_cache()


def _check(C, A, B, oracle, symmetric=False):
    want = oracle.sparse(arrays(A), arrays(B), B.shape[1], symmetric=symmetric)
    assert_csr_equal((C.indptr, C.indices, C.data), want, values="bits")


def test_in_place_edit_of_either_operand_is_never_served_stale(pkg, oracle):
    """The round-2 cache keyed on a strided sample: B.data[12345] = 1e6 on a 250 000-nnz operand left the key
    unchanged and the call returned A * B_old.  Default settings, no clear_cache() in between."""
    smm = pkg.sparse_matrix_multiply
    A, B = rand_csr(400, 5000, 0.01, 1), rand_csr(5000, 5000, 0.01, 2)
    assert B.nnz == 250000
    _check(smm(A, B), A, B, oracle)
    B.data[12345] = 1e6                                           # an index the old sample stride did not visit
    _check(smm(A, B), A, B, oracle)
    A.data[1234] = -3.5                                           # the same for A
    _check(smm(A, B), A, B, oracle)
    k = next(k for k in range(1000, B.nnz - 1) if B.indices[k] + 1 < B.indices[k + 1])
    B.indices[k] += 1                                             # a structural edit in place (row stays sorted)
    _check(smm(A, B), A, B, oracle)
    r = 100                                                       # and of indptr: move one entry to the next row
    B.indptr[r + 1] -= 1
    B.has_sorted_indices = False
    _check(smm(A, B), A, B, oracle)
    D = smm(A, B, output_format="dense")
    assert np.array_equal(D, oracle.dense(arrays(A), arrays(B), 5000))


def test_same_pattern_new_values_reruns_only_the_numeric_phase(pkg, oracle):
    """Three products on one pair of patterns with new values each time: each bit-exact against the oracle; from
    the second on no kernel of the symbolic side is launched -- the operands' values are rewritten in place
    (smm_csr_update_values) and the cached plan's numeric phase is replayed."""
    from sparse_matrix_mult_amd import matrix_ops
    from sparse_matrix_mult_amd.engine import default_context
    ctx = default_context()
    smm = pkg.sparse_matrix_multiply
    for hash_cfg in ((0, 0), (256, 2048)):                        # tile kernels / LDS-hash kernels for the result's rows
        pkg.clear_cache()
        ctx.tune_hash(*hash_cfg)
        try:
            A, B = rand_csr(300, 600, 0.05, 11), rand_csr(600, 700, 0.04, 12)
            ctx.timing(True); ctx.timing_reset()
            matrix_ops.cache_stats.clear()
            _check(smm(A, B), A, B, oracle)
            base = {k: ctx.kernel_time(k)[1] for k in SYMBOLIC_KERNELS}
            assert base["smm_validate"] == 2 and base["smm_symbolic"] + base["smm_symbolic_hash"] >= 1
            for step in range(2):
                A.data[:] = np.random.default_rng(100 + step).uniform(-1, 1, A.nnz)      # in place, same arrays
                B = signed(B, 200 + step)                                                  # new arrays, same pattern
                _check(smm(A, B), A, B, oracle)
                assert {k: ctx.kernel_time(k)[1] for k in SYMBOLIC_KERNELS} == base, "symbolic side ran again"
            assert matrix_ops.cache_stats["plan_hit"] == 2 and matrix_ops.cache_stats["values_update"] == 4
            assert matrix_ops.cache_stats["upload"] == 2
            # a symmetric product on the same operands is another plan; the operands stay
            S = rand_csr(600, 300, 0.05, 13)
            _check(smm(A, S, symmetric=True), A, S, oracle, symmetric=True)
            # default (rounding) mode on the cached pair as well
            pkg.set_exact(False)
            C = smm(A, B)
            want = oracle.sparse(arrays(A), arrays(B), 700)
            assert_csr_equal((C.indptr, C.indices, C.data), want, values="tol")
            A.data[:] *= 0.5
            C = smm(A, B)
            want = oracle.sparse(arrays(A), arrays(B), 700)
            assert_csr_equal((C.indptr, C.indices, C.data), want, values="tol")
            pkg.set_exact(True)
        finally:
            ctx.timing(False)
            ctx.tune_hash(256, 2048)


def test_values_update_through_the_c_abi_refreshes_every_cached_copy(ctx, oracle):
    """smm_csr_update_values + a second smm_spgemm_numeric on the same plan, under every dispatch that keeps a
    copy of B's values (packed payload: default walk; CSR-order values: exact walk; slab-major copy: slab
    kernels), then the dense product and the triple product (sliced-ELL copy of H) on the updated handles."""
    A, B = rand_csr(260, 500, 0.06, 21), rand_csr(500, 640, 0.05, 22)
    A2, B2 = signed(A, 23), signed(B, 24)
    for exact, slab in ((False, 1), (True, 1), (False, 2), (True, 2)):
        ctx.tune_hash(0, 0); ctx.tune_slab(slab, 0, 0)
        a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
        try:
            plan = ctx.spgemm_plan(a, b, exact=exact)
            first = plan.numeric_host()
            want = oracle.sparse(arrays(A), arrays(B), 640)
            assert_csr_equal(first, want, values="bits" if (exact or slab == 2) else "tol")
            a.update_values(A2.data); b.update_values(B2.data)
            again = plan.numeric_host()
            want = oracle.sparse(arrays(A2), arrays(B2), 640)
            assert_csr_equal(again, want, values="bits" if (exact or slab == 2) else "tol")
            D = ctx.dense_host(a, b, exact=exact)
            wd = oracle.dense(arrays(A2), arrays(B2), 640)
            assert np.array_equal(D, wd) if (exact or slab == 2) else np.allclose(D, wd, rtol=1e-10, atol=1e-13)
            plan.close()
        finally:
            a.close(); b.close(); ctx.tune_hash(256, 2048); ctx.tune_slab(0, 0, 0)
    # triple product: H's sliced-ELL copy carries values too
    H = rand_csr(200, 2300, 0.03, 31)
    S = rand_csr(2300, 2300, 0.004, 32); Q = (S + S.T).tocsr()
    for exact in (False, True):
        h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
        try:
            T1 = ctx.triple_host(h, q, exact=exact)
            w1 = oracle.triple(arrays(H), arrays(Q), 2300, 0)
            assert np.array_equal(T1, w1) if exact else np.allclose(T1, w1, rtol=1e-10, atol=1e-12)
            H2, Q2 = signed(H, 33), signed(Q, 34)
            h.update_values(H2.data); q.update_values(Q2.data)
            T2 = ctx.triple_host(h, q, exact=exact)
            w2 = oracle.triple(arrays(H2), arrays(Q2), 2300, 0)
            assert np.array_equal(T2, w2) if exact else np.allclose(T2, w2, rtol=1e-10, atol=1e-12)
        finally:
            h.close(); q.close()


def test_borrowed_operand_values_rewritten_in_place(ctx, oracle):
    import torch
    A, B = rand_csr(200, 300, 0.05, 41), rand_csr(300, 400, 0.05, 42)
    dev = torch.device("cuda", ctx.device)
    tb = [torch.from_numpy(x).to(dev) for x in arrays(B)]
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_torch(300, 400, *tb)
    try:
        plan = ctx.spgemm_plan(a, b)
        plan.numeric_host()
        B2 = signed(B, 43)
        tb[2].copy_(torch.from_numpy(B2.data).to(dev))
        torch.cuda.synchronize()
        b.values_changed()
        got = plan.numeric_host()
        assert_csr_equal(got, oracle.sparse(arrays(A), arrays(B2), 400), values="tol")
        with pytest.raises(Exception):
            b.update_values(B2.data)                  # host-source update needs an operand that owns its arrays
        plan.close()
    finally:
        a.close(); b.close()


def test_pinned_operands(pkg, oracle):
    from sparse_matrix_mult_amd import matrix_ops
    from sparse_matrix_mult_amd.engine import default_context
    ctx = default_context()
    smm = pkg.sparse_matrix_multiply
    A, B = rand_csr(300, 500, 0.05, 51), rand_csr(500, 500, 0.05, 52)
    pb = pkg.pin_operand(B)
    assert pb.shape == (500, 500) and pb.nnz == B.nnz
    matrix_ops.cache_stats.clear()
    _check(smm(A, pb), A, B, oracle)
    pa = pkg.pin_operand(A)
    _check(smm(pa, pb), A, B, oracle)
    ctx.timing(True); ctx.timing_reset()
    B2 = signed(B, 53)
    pb.update_values(B2.data)
    _check(smm(pa, pb), A, B2, oracle)                 # plan of (pa, pb) replayed on the new values
    assert ctx.kernel_time("smm_symbolic")[1] + ctx.kernel_time("smm_symbolic_hash")[1] == 0
    ctx.timing(False)
    D = smm(pa, pb, output_format="dense", symmetric=False)
    assert np.array_equal(D, oracle.dense(arrays(A), arrays(B2), 500))
    with pytest.raises(ValueError, match="incompatible"):
        smm(pb, pa)
    pb.unpin()
    with pytest.raises(ValueError, match="unpinned"):
        smm(pa, pb)
    pa.unpin()


def test_cache_limits_and_orphans(pkg, oracle):
    """Entries whose arrays were garbage-collected go first; the entry limit holds; real HBM (derived copies
    included) is what the byte cap counts."""
    import gc
    from sparse_matrix_mult_amd import matrix_ops
    smm = pkg.sparse_matrix_multiply
    B = rand_csr(400, 400, 0.05, 61)
    for i in range(6):
        A = rand_csr(100, 400, 0.05, 70 + i)
        _check(smm(A, B), A, B, oracle)
        del A
        gc.collect()
    assert len(matrix_ops._cache) <= 2                 # B and at most the last A (orphans are purged at the next call)
    ent = max(matrix_ops._cache.values(), key=lambda e: e.handle.nnz)      # B: tile index, packed payload, 16-bit columns
    assert ent.handle.device_bytes() >= 14 * ent.handle.nnz     # the three CSR arrays (12 B per entry) AND the cached copies
    mats = [rand_csr(100, 400, 0.05, 80 + i) for i in range(6)]
    for A in mats:
        smm(A, B)
    assert len(matrix_ops._cache) <= 4 and len(matrix_ops._plans) <= 2
    old = matrix_ops._cache_max_bytes
    matrix_ops._cache_max_bytes = 1                    # nothing fits: every call marshals afresh, results stay right
    try:
        _check(smm(mats[0], B), mats[0], B, oracle)
        assert len(matrix_ops._cache) == 0 and len(matrix_ops._plans) == 0
    finally:
        matrix_ops._cache_max_bytes = old


def test_exact_guard(ctx):
    """Run-time guard of SMM_EXACT (smm_ctx_exact_selftest): passes on gfx950; with the injected fault (the
    expectation reversed to descending lane order) it fails loudly with SMM_ERR_UNSUPPORTED."""
    from sparse_matrix_mult_amd.engine import SmmError
    ctx.exact_selftest()
    with pytest.raises(SmmError) as e:
        ctx.exact_selftest(inject_fault=True)
    assert e.value.code == -6 and "ascending lane order" in str(e.value)


def test_exact_guard_blocks_exact_products_when_it_fails(oracle, monkeypatch):
    from sparse_matrix_mult_amd.engine import Context, SmmError
    monkeypatch.setenv("SMM_EXACT_INJECT_FAULT", "1")
    c = Context(0)
    A, B = rand_csr(50, 60, 0.1, 1), rand_csr(60, 70, 0.1, 2)
    a, b = c.csr_from_scipy(A), c.csr_from_scipy(B)
    try:
        for _ in range(2):                                         # the verdict is remembered by the context
            with pytest.raises(SmmError) as e:
                c.spgemm_host(a, b, exact=True)
            assert e.value.code == -6
        with pytest.raises(SmmError):
            c.dense_host(a, b, exact=True)
        got = c.spgemm_host(a, b)                                  # the default mode does not depend on the property
        assert_csr_equal(got, oracle.sparse(arrays(A), arrays(B), 70), values="tol")
    finally:
        a.close(); b.close(); c.close()


def test_a_failed_allocation_is_retried_after_the_pool_went_back_to_the_device(pkg, oracle):
    """destroyed plans leave their multi-GB lists in the context's pool, which only pool allocations used to
    flush.  Now every device allocation that fails flushes the pool and retries once (no sticky HIP error is left
    behind), clear_cache() releases the pool, and a hard failure reaches sparse_matrix_multiply()'s own retry
    (SMM_ERR_ALLOC -> clear_cache -> once more)."""
    from sparse_matrix_mult_amd.engine import SmmError, default_context
    ctx = default_context()
    smm = pkg.sparse_matrix_multiply
    A, B = rand_csr(300, 400, 0.05, 31), rand_csr(400, 500, 0.05, 32)
    _check(smm(A, B), A, B, oracle)
    pkg.clear_cache()
    assert ctx.pool_bytes() == 0                                   # the closed plan's lists left the pool as well
    # (1) the first attempt of an allocation of a fresh operand's upload fails: the library retries by itself
    before = ctx.alloc_retries()
    ctx.inject_alloc_failure(2)
    A2, B2 = rand_csr(300, 400, 0.05, 33), rand_csr(400, 500, 0.05, 34)
    _check(smm(A2, B2), A2, B2, oracle)
    assert ctx.alloc_retries() == before + 1
    # (2) a cached copy built during the symbolic phase (packed payload / tile index) fails the same way
    ctx.inject_alloc_failure(1)
    _check(smm(A2, B2, symmetric=False), A2, B2, oracle)           # operands cached; plan cached -> force a new geometry instead:
    a, b = ctx.csr_from_scipy(A2), ctx.csr_from_scipy(B2)
    try:
        ctx.inject_alloc_failure(3)
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A2), arrays(B2), 500), values="bits")
        # (3) both attempts fail: SMM_ERR_ALLOC through the C ABI, nothing sticky behind it
        ctx.inject_alloc_failure(1, hard=True)
        a3 = None
        with pytest.raises(SmmError) as e:
            a3 = ctx.csr_from_scipy(A)
        assert e.value.code == -3 and a3 is None
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A2), arrays(B2), 500), values="bits")
    finally:
        a.close(); b.close()
    # (4) ... and through the public entry point: the call drops what is resident and succeeds on its second try
    A4, B4 = rand_csr(300, 400, 0.05, 35), rand_csr(400, 500, 0.05, 36)
    _check(smm(A4, B2), A4, B2, oracle)                            # something is resident now
    ctx.inject_alloc_failure(1, hard=True)
    _check(smm(A4, B4), A4, B4, oracle)
    ctx.inject_alloc_failure(0)
