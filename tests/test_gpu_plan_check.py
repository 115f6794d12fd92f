"""Round 4: inconsistent plan metadata is an ERROR, not a hang or a fault.

The symbolic side of a product (capacities -> ordered lists -> start slots -> sub-run tables) is produced by one set
of kernels and trusted by the next.  Two defects of round 3 had the shape "symbolic side wrong -> numeric side trusts
it" and surfaced as a SIGABRT and a hang.  Now (1) every consumer clamps what it reads, always, and records the fact
in the context's error word, (2) smm_plan_check verifies the whole plan (SMM_CHECK=1: the whole GPU test-suite runs
with it, tests/conftest.py), and both come back as SMM_ERR_INTERNAL through smm_last_error() -- the reference's
"message + early return, never crash the caller" convention (src/sparsework.cpp:33-36,
src/sparse_sparse_sparse.cpp:257-262).  smm_plan_inject_fault damages one table the way a kernel defect would."""
import numpy as np
import pytest

from helpers import arrays, assert_csr_equal, rand_csr

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(120)]

SMM_ERR_INTERNAL = -7
FAULTS = {1: "reversed sub-run", 2: "sub-run beyond the row", 3: "tail descriptor", 4: "row count", 5: "list entry",
          6: "slab sub-run", 7: "start slot", 8: "negative capacity"}


@pytest.fixture
def tiles_only(ctx):
    ctx.tune_hash(0, 0)                   # every row takes the dense-tile kernel: the tables exist for all of them
    ctx.tune_shared(96, 4); ctx.tune(96, 4)
    yield ctx
    ctx.tune_hash(256, 2048); ctx.tune_shared(20000, 16); ctx.tune(18000, 8); ctx.tune_symbolic(0)


def _operands(ctx, seed=1, m=150, k=180, n=600):
    # rows of B hold ~120 entries: the first steps of every row of C append >= 64 columns, so they are sub-run
    # steps with table entries (steps that append fewer form the tail, which has no table)
    A, B = rand_csr(m, k, 0.1, seed), rand_csr(k, n, 0.2, seed + 1)
    return A, B, ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)


def _expect_internal(fn):
    from sparse_matrix_mult_amd.engine import SmmError
    with pytest.raises(SmmError) as e:
        fn()
    assert e.value.code == SMM_ERR_INTERNAL, str(e.value)
    assert "inconsistent plan metadata" in str(e.value)


def test_a_sound_plan_passes_the_checker(tiles_only, oracle):
    ctx = tiles_only
    A, B, a, b = _operands(ctx)
    try:
        for exact in (False, True):
            for sym in (False,):
                plan = ctx.spgemm_plan(a, b, symmetric=sym, exact=exact)
                plan.check()
                assert_csr_equal(plan.numeric_host(), oracle.sparse(arrays(A), arrays(B), 600), values="bits" if exact else "tol")
                ctx.synchronize()             # nothing was recorded by the kernels' clamps either
                plan.close()
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("kind", [1, 2, 3, 4, 5, 7, 8], ids=lambda k: FAULTS[k].replace(" ", "-"))
def test_checker_reports_every_injected_fault(tiles_only, oracle, kind):
    ctx = tiles_only
    A, B, a, b = _operands(ctx, seed=10 + kind)
    try:
        plan = ctx.spgemm_plan(a, b)
        plan.check()
        plan.inject_fault(kind)
        _expect_internal(plan.check)
        plan.close()
        # the context is clean again and still usable
        ctx.synchronize()
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A), arrays(B), 600), values="bits")
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("kind", [1, 2, 3], ids=lambda k: FAULTS[k].replace(" ", "-"))
def test_numeric_phase_reports_a_damaged_table_instead_of_hanging_or_faulting(tiles_only, oracle, kind, exact):
    """What round 3's aborts looked like from the caller: numeric_host() on a plan whose tables are wrong.  The
    epilogue's clamps keep every store inside its row, bound every loop, and the call returns SMM_ERR_INTERNAL."""
    ctx = tiles_only
    A, B, a, b = _operands(ctx, seed=30 + kind)
    try:
        plan = ctx.spgemm_plan(a, b, exact=exact)
        plan.inject_fault(kind)
        _expect_internal(plan.numeric_host)
        plan.close()
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A), arrays(B), 600), values="bits")
    finally:
        a.close(); b.close()


def test_device_resident_numeric_reports_through_synchronize(tiles_only):
    """smm_spgemm_numeric is asynchronous: what its kernels record comes back from the next smm_ctx_synchronize."""
    import torch
    ctx = tiles_only
    A, B, a, b = _operands(ctx, seed=50)
    try:
        plan = ctx.spgemm_plan(a, b)
        plan.inject_fault(2)
        dev = torch.device("cuda", 0)
        indptr = torch.empty(a.rows + 1, dtype=torch.int64, device=dev)
        indices = torch.full((plan.nnz + 64,), -7, dtype=torch.int32, device=dev)
        data = torch.zeros(plan.nnz + 64, dtype=torch.float64, device=dev)
        plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        _expect_internal(ctx.synchronize)
        assert bool((indices[plan.nnz:] == -7).all())          # nothing was written behind the result
        ctx.synchronize()                                       # reported once
        plan.close()
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("exact", [False, True])
def test_slab_table_faults_are_reported(ctx, oracle, exact):
    ctx.tune_shared(64, 4); ctx.tune(64, 4); ctx.tune_symbolic(128); ctx.tune_hash(256, 2048)
    A, B, a, b = _operands(ctx, seed=60, n=900)
    try:
        plan = ctx.spgemm_plan(a, b, exact=exact)
        plan.check()
        plan.inject_fault(6)
        _expect_internal(plan.check)
        _expect_internal(plan.numeric_host)
        plan.close()
        plan = ctx.spgemm_plan(a, b, exact=exact)
        plan.inject_fault(7)                                    # a start slot of slab 0 beyond its list
        _expect_internal(plan.check)
        plan.close()
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A), arrays(B), 900), values="bits")
    finally:
        a.close(); b.close()
        ctx.tune_shared(20000, 16); ctx.tune(18000, 8); ctx.tune_symbolic(0)


def test_hash_kernels_do_not_spin_on_a_column_missing_from_the_list(ctx, oracle):
    """The LDS-hash kernels look every product's column up in the row's list: a list that lacks one (kind 5
    overwrites an entry) used to be an endless probe loop."""
    ctx.tune_hash(256, 2048)
    A, B = rand_csr(60, 80, 0.05, 71), rand_csr(80, 3000, 0.01, 72)
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        plan = ctx.spgemm_plan(a, b)
        plan.inject_fault(5)
        _expect_internal(plan.numeric_host)
        plan.close()
        assert_csr_equal(ctx.spgemm_host(a, b, exact=True), oracle.sparse(arrays(A), arrays(B), 3000), values="bits")
    finally:
        a.close(); b.close()


def test_check_mode_is_on_for_the_whole_gpu_suite(ctx):
    """tests/conftest.py sets SMM_CHECK=1 before any context exists: every plan any GPU test makes -- the fuzz
    campaigns included -- goes through smm_plan_check."""
    import os
    assert os.environ.get("SMM_CHECK") == "1"
    A, B, a, b = _operands(ctx, seed=80)
    try:
        ctx.timing(True); ctx.timing_reset()
        ctx.spgemm_plan(a, b).close()
        assert ctx.kernel_time("smm_plan_check")[1] >= 1
        ctx.timing(False)
    finally:
        a.close(); b.close()
