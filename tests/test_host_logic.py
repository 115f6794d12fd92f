"""Host-side behaviour of sparse_matrix_multiply() that needs no GPU: the argument checks, the
raised errors and the nnz==0 early returns of the reference (matrix_ops.py:288-322), plus the
shard planner of the multi-GPU driver."""
import numpy as np
import pytest
import scipy.sparse as sp

from sparse_matrix_mult_amd import sparse_matrix_multiply
from sparse_matrix_mult_amd.distributed import balanced_row_shards


def test_signature_is_the_references():
    import inspect
    sig = inspect.signature(sparse_matrix_multiply)
    assert list(sig.parameters) == ["matrix_a", "matrix_b", "output_format", "symmetric", "imem_size",
                                    "use_triple_product", "compute_full_matrix"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["output_format"] == "sparse" and d["symmetric"] is False and d["imem_size"] is None
    assert d["use_triple_product"] is False and d["compute_full_matrix"] is None


def test_import_name_shim():
    from sparse_matrix_mult import sparse_matrix_multiply as f
    assert f is sparse_matrix_multiply


def test_dimension_mismatch_raises():            # reference matrix_ops.py:312-313
    with pytest.raises(ValueError, match="incompatible"):
        sparse_matrix_multiply(np.ones((2, 3)), np.ones((4, 2)))


def test_symmetric_needs_square_result():        # reference matrix_ops.py:321-322
    with pytest.raises(ValueError, match="square"):
        sparse_matrix_multiply(np.ones((2, 3)), np.ones((3, 4)), symmetric=True)


def test_bad_imem_size_and_full_matrix():        # reference matrix_ops.py:291-304
    with pytest.raises(ValueError, match="imem_size"):
        sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), imem_size="x")
    with pytest.raises(ValueError, match="compute_full_matrix"):
        sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), compute_full_matrix=2)


def test_zero_operands_return_empty_without_touching_the_gpu():
    """reference tests/test_edge_case.py:54-71 and matrix_ops.py:315-319."""
    z33, z34 = np.zeros((3, 3)), np.zeros((3, 4))
    r = sparse_matrix_multiply(z33, z34, output_format="sparse")
    assert sp.isspmatrix_csr(r) and r.shape == (3, 4) and r.nnz == 0 and r.dtype == np.float64
    r = sparse_matrix_multiply(sp.csr_matrix(z33), sp.csr_matrix(z34), output_format="dense")
    assert isinstance(r, np.ndarray) and r.shape == (3, 4) and not r.any()
    r = sparse_matrix_multiply(z33, np.ones((3, 4)), use_triple_product=False, output_format="dense", symmetric=False)
    assert r.shape == (3, 4)
    # the zero check comes before the symmetric-square check, as in the reference (:315 before :321)
    assert sparse_matrix_multiply(z33, z34, symmetric=True).shape == (3, 4)


def test_unknown_output_format_prints_and_returns_zeros(capsys):   # reference :367-368, :377-387
    r = sparse_matrix_multiply(np.ones((2, 2)), np.ones((2, 2)), output_format="coo")
    assert isinstance(r, np.ndarray) and r.shape == (2, 2) and not r.any()
    assert "Invalid output_format" in capsys.readouterr().out


def test_balanced_row_shards():
    work = np.array([1, 1, 1, 1, 100, 1, 1, 1], dtype=float)
    sh = balanced_row_shards(work, 2)
    assert sh[0][0] == 0 and sh[-1][1] == 8 and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
    assert all(e > b for b, e in sh)
    # more shards than rows: clamped like limits() (workdivision.cpp:26-29)
    assert balanced_row_shards(np.ones(3), 8) == [(0, 1), (1, 2), (2, 3)]
    # uniform work -> near-equal row counts
    sh = balanced_row_shards(np.ones(1000), 8)
    assert max(e - b for b, e in sh) - min(e - b for b, e in sh) <= 1
    # skewed work is balanced by work, not rows
    w = np.concatenate([np.full(100, 50.0), np.full(900, 1.0)])
    sh = balanced_row_shards(w, 4)
    loads = [w[b:e].sum() for b, e in sh]
    assert max(loads) / (sum(loads) / 4) < 1.1
    assert balanced_row_shards([], 4) == [(0, 0)]


# ---- operand cache key (no GPU involved: smm_host_hash64 is host code)
def test_operand_key_sees_every_single_element_edit():
    """The reference marshals the caller's CURRENT arrays on every call (matrix_ops.py:339-340, :187-202); the
    operand cache may therefore only recognise an operand whose every element is unchanged.  Its key is a
    full-content hash: an in-place edit of ANY one element of indptr / indices / data changes the part of the
    key it belongs to (the round-2 key, a strided sample, missed exactly these edits)."""
    from sparse_matrix_mult_amd.matrix_ops import _operand_key
    B = sp.random(5000, 5000, density=0.01, format="csr", random_state=np.random.default_rng(1))
    assert B.nnz == 250000
    pat0, dat0 = _operand_key(B)
    assert _operand_key(B.copy()) == (pat0, dat0)                 # content, not identity
    rng = np.random.default_rng(7)
    for k in [12345, 0, B.nnz - 1] + rng.integers(0, B.nnz, 40).tolist():
        old = B.data[k]
        B.data[k] = np.nextafter(old, 2.0)                        # one ulp
        pat, dat = _operand_key(B)
        assert pat == pat0 and dat != dat0, k
        B.data[k] = old
    for k in [777, 0, B.nnz - 1] + rng.integers(0, B.nnz, 40).tolist():
        old = B.indices[k]
        B.indices[k] = old ^ 1
        pat, dat = _operand_key(B)
        assert pat != pat0 and dat == dat0, k
        B.indices[k] = old
    i = 2500
    B.indptr[i] -= 1
    assert _operand_key(B)[0] != pat0
    B.indptr[i] += 1
    assert _operand_key(B) == (pat0, dat0)
    # swapping two elements (same multiset) is seen too: the hash is position-dependent
    B.data[[10, 11]] = B.data[[11, 10]]
    assert _operand_key(B)[1] != dat0


def test_host_hash_is_the_same_for_every_thread_count_and_length_aware(monkeypatch):
    import ctypes
    from sparse_matrix_mult_amd._lib import SmmLibrary
    lib = SmmLibrary().get_lib()
    a = np.random.default_rng(3).integers(0, 255, 40 << 20, dtype=np.uint8)     # 10 blocks of 4 MB
    h = lambda x: lib.smm_host_hash64(ctypes.c_void_p(x.ctypes.data), x.nbytes)
    monkeypatch.setenv("SMM_HASH_THREADS", "1")
    one = h(a)
    monkeypatch.setenv("SMM_HASH_THREADS", "5")
    assert h(a) == one
    assert h(a[:-1]) != one and h(a[:1000]) != h(a[:1001])
    z = np.zeros(64, np.uint8)
    assert len({h(z[:n]) for n in range(1, 65)}) == 64                          # zero padding of the tail is not a collision


# ---- operand / plan cache policy with a stub engine (the same code paths sparse_matrix_multiply() takes; no GPU)
class _FakeHandle:
    def __init__(self, ctx, m):
        self.ctx, self.handle, self.nnz, self.rows, self.cols = ctx, object(), int(m.nnz), m.shape[0], m.shape[1]
        self.values = np.array(m.data[:m.nnz], dtype=np.float64)
        self.updates = 0

    def update_values(self, data):
        assert len(data) == self.nnz
        self.values = np.array(data, dtype=np.float64); self.updates += 1

    def device_bytes(self):
        return 12 * self.nnz + 1000 if self.handle else 0

    def close(self):
        self.handle = None


class _FakePlan:
    def __init__(self, a, b):
        self.a, self.b, self.handle, self.nnz = a, b, object(), 1

    def device_bytes(self):
        return 5000 if self.handle else 0

    def close(self):
        self.handle = None


class _FakeCtx:
    def __init__(self):
        self.uploads, self.plans = 0, 0

    def csr_from_scipy(self, m):
        self.uploads += 1
        return _FakeHandle(self, m)

    def spgemm_plan(self, a, b, symmetric=False, exact=False):
        self.plans += 1
        return _FakePlan(a, b)


@pytest.fixture()
def cache():
    from sparse_matrix_mult_amd import matrix_ops as mo
    mo.clear_cache()
    saved = (mo._cache_entries, mo._cache_max_bytes, mo._plan_entries)
    mo._cache_entries, mo._cache_max_bytes, mo._plan_entries = 4, 1 << 40, 2
    mo.cache_stats.clear()
    yield mo
    mo.clear_cache()
    mo._cache_entries, mo._cache_max_bytes, mo._plan_entries = saved


def _rand(m, n, seed):
    return sp.random(m, n, density=0.1, format="csr", random_state=np.random.default_rng(seed))


def test_cache_policy_hit_update_and_plan_replay(cache):
    mo, ctx = cache, _FakeCtx()
    A, B = _rand(30, 40, 1), _rand(40, 50, 2)

    def call(a, b):
        ka, kb = mo._operand_key(a), mo._operand_key(b)
        la = mo._acquire(ctx, a, ka, protect=(kb[0],)); lb = mo._acquire(ctx, b, kb)
        plan, release = mo._plan_for(ctx, la, lb, False, False)
        vals = (la.handle.values.copy(), lb.handle.values.copy())
        release(); lb.release(); la.release()
        return plan, vals

    p1, _ = call(A, B)
    assert (ctx.uploads, ctx.plans) == (2, 1) and len(mo._cache) == 2 and len(mo._plans) == 1
    p2, _ = call(A, B)                                           # unchanged: both hit, plan replayed
    assert p2 is p1 and (ctx.uploads, ctx.plans) == (2, 1) and mo.cache_stats["hit"] == 2 and mo.cache_stats["plan_hit"] == 1
    B.data[7] += 1.0                                             # in-place edit of one value: seen, values only
    p3, (va, vb) = call(A, B)
    assert p3 is p1 and ctx.uploads == 2 and mo.cache_stats["values_update"] == 1 and np.array_equal(vb, B.data)
    B2 = B.copy(); B2.data *= 2                                  # equal pattern in other arrays: same entry, values replaced
    p4, (va, vb) = call(A, B2)
    assert p4 is p1 and ctx.uploads == 2 and np.array_equal(vb, B2.data)
    B.indices[3] = (B.indices[3] + 1) % 50                       # structural edit in place: a new operand, a new plan
    p5, _ = call(A, B)
    assert p5 is not p1 and ctx.uploads == 3 and ctx.plans == 2


def test_cache_policy_limits_orphans_and_leases(cache):
    import gc
    mo, ctx = cache, _FakeCtx()
    B = _rand(40, 50, 2)
    mats = [_rand(30, 40, 10 + i) for i in range(6)]
    for A in mats:
        la = mo._acquire(ctx, A); lb = mo._acquire(ctx, B); lb.release(); la.release()
    assert len(mo._cache) == 4 and mo._operand_key(B)[0] in mo._cache          # LRU: B was used every time and stays
    # an entry in use is neither evicted nor updated in place
    held = mo._acquire(ctx, mats[5])
    old_values = held.handle.values.copy()
    C = mats[5].copy(); C.data += 1.0
    other = mo._acquire(ctx, C)                                                 # same pattern, other values, entry busy
    assert other.entry is None and np.array_equal(held.handle.values, old_values) and held.handle.updates == 0
    other.release(); held.release()
    # orphans (their arrays are gone) are purged at the next miss, not the operand about to be looked up
    mo.clear_cache()
    T = _rand(30, 40, 99)
    lt = mo._acquire(ctx, T); lt.release()
    pat = mo._operand_key(T)[0]
    T2 = T.copy()                                                               # equal content in other arrays
    del T; gc.collect()
    assert mo._cache[pat].orphaned()
    lx = mo._acquire(ctx, mats[0], protect=(pat,))                             # a miss that protects T's pattern (the call's other operand)
    assert pat in mo._cache
    l2 = mo._acquire(ctx, T2); assert l2.entry is mo._cache[pat] and not l2.entry.orphaned()
    l2.release(); lx.release()
    del T2; gc.collect()
    ly = mo._acquire(ctx, mats[1]); ly.release()                                # no protection now: the orphan goes
    assert pat not in mo._cache
    # byte cap: plans go first, then idle entries
    mo._cache_max_bytes = 1
    lz = mo._acquire(ctx, mats[2]); lz.release()
    assert len(mo._cache) == 0 and len(mo._plans) == 0


# ---- round 4: lock hygiene of the operand cache, unpin while leased, pool release
def test_values_upload_runs_outside_the_cache_lock_and_clear_cache_releases_the_pool(cache):
    """A values-only update (a 200 MB host-to-device copy at BASELINE configs[1]) must not stall callers that need no
    cache state of that entry: while one thread is inside update_values, another thread's acquire / release of a
    different operand completes.  clear_cache() hands the library's pooled scratch back (smm_ctx_release_pool)."""
    import threading
    mo = cache
    gate, inside = threading.Event(), threading.Event()

    class SlowHandle(_FakeHandle):
        def update_values(self, data):
            inside.set()
            assert gate.wait(20), "the other thread never got through: the cache lock was held across the upload"
            super().update_values(data)

    class Ctx(_FakeCtx):
        released = 0

        def csr_from_scipy(self, m):
            self.uploads += 1
            return SlowHandle(self, m)

        def release_pool(self):
            Ctx.released += 1

    ctx = Ctx()
    ctx.handle = object()
    B, other = _rand(40, 50, 2), _rand(30, 40, 3)
    mo._acquire(ctx, B).release()
    B2 = B.copy(); B2.data += 1.0
    done = []

    def updater():
        lease = mo._acquire(ctx, B2)                  # same pattern, new values: update path
        done.append(np.array_equal(lease.handle.values, B2.data))
        lease.release()

    t = threading.Thread(target=updater); t.start()
    assert inside.wait(20)
    # the entry is busy: the same pattern with yet other values gets a private upload instead of waiting
    B3 = B.copy(); B3.data += 2.0
    l3 = mo._acquire(ctx, B3); assert l3.entry is None; l3.release()
    lo = mo._acquire(ctx, other); lo.release()        # an unrelated operand goes straight through
    gate.set(); t.join(20)
    assert done == [True] and mo.cache_stats["values_update"] == 1
    mo.clear_cache()
    assert Ctx.released >= 1 and len(mo._cache) == 0


def test_unpin_while_leased_defers_the_close_and_gc_only_leaves_a_note(cache):
    import gc
    mo = cache
    ctx = _FakeCtx()
    P = mo.PinnedOperand(ctx, _rand(30, 40, 5))
    B = _rand(40, 50, 6)
    lp = mo._acquire(ctx, P); lb = mo._acquire(ctx, B)
    handle = lp.handle
    P.unpin()                                         # a product is running with it
    assert handle.handle is not None                  # ... so the device handle is still alive
    lb.release(); lp.release()
    assert handle.handle is None                      # closed by the lease that held it
    with pytest.raises(ValueError):
        mo._acquire(ctx, P)
    # garbage collection of a pinned operand never touches the cache structures itself
    Q = mo.PinnedOperand(ctx, _rand(30, 40, 7))
    qh = Q._handle
    del Q; gc.collect()
    assert qh.handle is not None and len(mo._graveyard) == 1
    mo._acquire(ctx, B).release()                     # the next trim buries it
    assert qh.handle is None and not mo._graveyard
