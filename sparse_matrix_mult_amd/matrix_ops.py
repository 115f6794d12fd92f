"""sparse_matrix_multiply() -- the reference's public entry point, served by the MI355X engine.

Drop-in for reference sparse_matrix_mult/matrix_ops.py:271-387: same name, same arguments,
same argument meaning, same return types, same raised errors.  What differs:
  * the work is done by libsmm_hip.so (hand-written HIP, gfx950) through the v2 C ABI of
    include/smm_hip.h; operands go host->HBM once, results come back into numpy buffers this
    module owns (no create_sparsemat/memmove round trip, no leaked input structs:
    reference :187-202, :339-340);
  * nothing is printed at import (reference :89-90,133,137);
  * a failure inside the library RAISES (SmmError) instead of being printed and turned into
    an all-zero result (reference :377-387) -- silent zeros would hide a missing GPU.  The
    one swallowed case the reference's callers can observe on purpose, an unknown
    output_format, is kept: message printed, zeros returned.
"""
import collections
import os
import threading
import weakref
import ctypes

import numpy as np
from scipy.sparse import csr_matrix, isspmatrix_csr

from ._lib import SMM_ERR_ALLOC, SmmError, SmmLibrary
from .engine import default_context

_INT32_MAX = np.iinfo(np.int32).max

# SMM_EXACT=1 (or set_exact(True)): add products in exactly the reference's order, so float64
# values are bit-identical to the reference's loop; the default lets the waves of a workgroup
# add concurrently -- values agree to rounding (tested to 1e-10 relative), ~3x faster.
# indptr / indices are bit-exact either way.
_exact = os.environ.get("SMM_EXACT", "0") not in ("", "0")


# Mirror epilogue, opt-in: with set_full_symmetric(True) the DENSE results that hold only
# the upper triangle -- output_format='dense' with symmetric=True, and the triple product with
# compute_full_matrix None/0 -- come back as the full symmetric matrix (lower triangle = mirror image of
# the upper one, filled on the device).  compute_full_matrix='mirror' asks for the same for one triple
# product.  The default stays the reference's behaviour (lower triangle 0.0), and compute_full_matrix=1
# keeps reproducing the reference exactly (off-diagonal doubled).  CSR results (output_format='sparse',
# symmetric=True) are mirrored too: row i of the full matrix holds the mirrored entries (columns < i) in ascending
# column order, then the reference's upper-triangle row in its first-touch order; any row length (segments of up to
# 8192 mirrored entries are sorted in LDS, longer ones placed by rank).
_full_symmetric = False


def set_full_symmetric(flag):
    """Return dense upper-triangle results as full symmetric matrices; returns the old setting."""
    global _full_symmetric
    old, _full_symmetric = _full_symmetric, bool(flag)
    return old


# Device-resident results, opt-in: the reference copies every result out of the library into fresh host
# arrays (sparsemat_to_csr / darray_to_numpy, matrix_ops.py:205-240); at BASELINE configs[1] that copy is 40 GB and
# 0.8 s behind 32 ms of GPU work.  With set_result_device(True) (or SMM_RESULT_DEVICE=1) the same call leaves the
# result in HBM: 'dense' and the triple product return a torch.Tensor (float64, cuda), 'sparse' a DeviceCSRResult
# (indptr int64 / indices int32 / data float64 torch tensors, first-touch order as ever) with .to_scipy() for the
# reference's return type.  The default is the reference's behaviour: caller-owned host objects.
_result_device = os.environ.get("SMM_RESULT_DEVICE", "0") not in ("", "0")


def set_result_device(flag):
    """Leave results in HBM as torch tensors (see DeviceCSRResult); returns the old setting."""
    global _result_device
    old, _result_device = _result_device, bool(flag)
    return old


class DeviceCSRResult:
    """A CSR result resident in HBM: .indptr (int64), .indices (int32), .data (float64) are torch CUDA tensors,
    .shape / .nnz as scipy's.  Columns inside a row are in the reference's first-touch order (not sorted)."""
    __slots__ = ("indptr", "indices", "data", "shape")

    def __init__(self, indptr, indices, data, shape):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, tuple(shape)

    @property
    def nnz(self):
        return int(self.indices.numel())

    def to_scipy(self):
        """The reference's return type (copies the result to the host)."""
        return _result_csr(self.indptr.cpu().numpy(), self.indices.cpu().numpy(), self.data.cpu().numpy(), self.shape)

    def to_torch_sparse_csr(self):
        """torch.sparse_csr_tensor over the same values (torch wants one index dtype: both int64; columns unsorted)."""
        import torch
        return torch.sparse_csr_tensor(self.indptr, self.indices.to(torch.int64), self.data, size=self.shape,
                                       check_invariants=False)


def set_exact(flag):
    """Select bit-exact (reference-order) accumulation for later calls; returns the old setting."""
    global _exact
    old, _exact = _exact, bool(flag)
    return old


# ---------------------------------------------------------------------------------------------
# Operand cache.  The reference re-marshals both operands from the caller's CURRENT arrays on
# every call (matrix_ops.py:339-340, :187-202); its README's use case is many products on one sparsity
# pattern (README.md:5,13: covariance matrices).  Here an uploaded operand stays resident in HBM together
# with what the kernels derive from it (validation, tile index, packed payload, 16-bit columns, the
# sliced-ELL copy of H), and the symbolic phase of a product stays resident as a plan:
#   * an operand is recognised by the CONTENT of its three arrays -- a 64-bit hash of every byte
#     (smm_host_hash64: a chain of bijective steps, so a change of any one element changes it), taken on
#     every call.  An operand edited in place between two calls is therefore never served stale, and an
#     equal matrix in other arrays is a hit.  Nothing is assumed from object identity;
#   * same indptr/indices hashes, other data hash: only the values travel (smm_csr_update_values: the value
#     array and the value halves of the cached copies are rewritten in place), and
#   * a product whose two patterns and mode were seen before re-runs only the numeric phase of its cached
#     plan: no smm_symbolic / smm_runs / smm_segptr launch at all.
# Entries are dropped when the arrays they were made from have been garbage-collected, least recently used
# first beyond SMM_OPERAND_CACHE entries (default 4; 0 = marshal afresh on every call, exactly as the
# reference) or SMM_OPERAND_CACHE_GB of HBM (default 16; what the library reports for each handle and plan,
# derived copies included), and all at once when an allocation fails (the product is then retried once).
# pin_operand() is the explicit alternative: a handle the caller owns, no hashing at all.
_cache_lock = threading.RLock()
_cache = collections.OrderedDict()          # pattern key -> _Entry; most recently used last
_plans = collections.OrderedDict()          # (pattern key A, pattern key B, symmetric, exact) -> _PlanEntry
_cache_entries = int(os.environ.get("SMM_OPERAND_CACHE", "4"))
_cache_max_bytes = int(float(os.environ.get("SMM_OPERAND_CACHE_GB", "16")) * (1 << 30))
_plan_entries = int(os.environ.get("SMM_PLAN_CACHE", "2"))
cache_stats = collections.Counter()         # 'upload', 'hit', 'values_update', 'plan_hit', 'plan_miss' (tests, diagnostics)


def _hash(a):
    a = np.ascontiguousarray(a)
    return int(SmmLibrary().get_lib().smm_host_hash64(ctypes.c_void_p(a.ctypes.data), a.nbytes))


def _operand_key(m):
    """(pattern key, data key) of a scipy CSR matrix: full-content hashes of indptr / indices and of data.
    Any in-place edit of any element changes the part it belongs to."""
    nnz = int(m.indptr[-1]) if len(m.indptr) else 0
    idx, dat = m.indices[:nnz], m.data[:nnz]
    pattern = (tuple(m.shape), nnz, m.indptr.dtype.str, idx.dtype.str, _hash(m.indptr), _hash(idx))
    return pattern, (dat.dtype.str, _hash(dat))


class _Entry:
    # nbytes: HBM of the handle and its cached copies as of its last use (refreshed OUTSIDE _cache_lock: the query takes
    # the context's lock and would wait for a running product); dead: a PinnedOperand that was unpinned while leased
    __slots__ = ("handle", "pattern", "data_key", "users", "refs", "nbytes", "dead")

    def __init__(self, handle, pattern, data_key):
        self.handle, self.pattern, self.data_key, self.users, self.refs = handle, pattern, data_key, 0, []
        self.nbytes, self.dead = 0, False

    def remember(self, arr):
        self.refs = [r for r in self.refs if r() is not None]
        if not any(r() is arr for r in self.refs):
            try:
                self.refs.append(weakref.ref(arr))
            except TypeError:
                pass

    def orphaned(self):
        return bool(self.refs) and all(r() is None for r in self.refs)


class _PlanEntry:
    __slots__ = ("plan", "a", "b", "users", "nbytes")

    def __init__(self, plan, a, b, nbytes=0):
        self.plan, self.a, self.b, self.users, self.nbytes = plan, a, b, 0, nbytes


class _Lease:
    """One call's hold on a device operand: .handle for the engine, .entry when it is a cache entry."""
    __slots__ = ("handle", "entry", "_transient")

    def __init__(self, handle, entry=None, transient=False):
        self.handle, self.entry, self._transient = handle, entry, transient

    def release(self):
        if self.entry is not None:
            ent = self.entry
            nbytes = ent.handle.device_bytes() if ent.handle is not None else 0      # (library call: before taking _cache_lock)
            close = None
            with _cache_lock:
                ent.nbytes = nbytes
                ent.users -= 1
                if ent.users == 0 and ent.dead:
                    close = ent.handle               # unpinned while this call was using it
                elif ent.users == 0 and ent.pattern in _cache:
                    _trim_locked()                   # what had to stay while it was in use may go now
            if close is not None:
                close.close()
            self.entry = None
        elif self._transient and self.handle is not None:
            self.handle.close()
        self.handle = None


def _drop_entry_locked(key):
    ent = _cache.pop(key, None)
    if ent is None:
        return
    for pk in [pk for pk, pe in _plans.items() if pe.a is ent or pe.b is ent]:
        pe = _plans.pop(pk)
        if pe.users == 0:
            pe.plan.close()                      # (else: _release_plan closes it when the call that runs it ends)
    if ent.users == 0:
        ent.handle.close()
    else:
        ent.dead = True                          # leased by a running call: its release closes the handle


def _cached_bytes_locked():
    # (byte counts recorded on the entries: no library call -- it would take the context's lock -- under _cache_lock)
    return sum(e.nbytes for e in _cache.values()) + sum(pe.nbytes for pe in _plans.values())


_graveyard = []          # PinnedOperands dropped by the garbage collector: closed by the next _trim_locked (list.append is atomic)


def _bury_locked():
    while _graveyard:
        ent = _graveyard.pop()
        for pk in [pk for pk, pe in _plans.items() if pe.a is ent or pe.b is ent]:
            pe = _plans.pop(pk)
            if pe.users == 0:
                pe.plan.close()                  # (a plan in use is closed by _release_plan when its call ends)
        if ent.users == 0 and ent.handle is not None:
            ent.handle.close()
        else:
            ent.dead = True                      # closed by the lease that still holds it


def _trim_locked(keep=(), reserve=0, protect=()):
    """Enforce the limits (entries idle and not in `keep` only); reserve = entries about to be added; protect =
    pattern keys this call is about to look up (its other operand)."""
    _bury_locked()
    for key in [k for k, e in _cache.items() if e.users == 0 and e.orphaned() and e not in keep and k not in protect]:
        _drop_entry_locked(key)                  # the arrays it was made from are gone

    def idle_plan():
        return next((k for k, pe in _plans.items() if pe.users == 0), None)

    def idle_entry():
        return next((k for k, e in _cache.items() if e.users == 0 and e not in keep and k not in protect), None)

    while len(_plans) > max(_plan_entries, 0) and idle_plan() is not None:
        _plans.pop(idle_plan()).plan.close()
    while len(_cache) + reserve > _cache_entries and idle_entry() is not None:
        _drop_entry_locked(idle_entry())
    while _cached_bytes_locked() > _cache_max_bytes:
        if idle_plan() is not None:              # plans (the symbolic phase's lists) go first
            _plans.pop(idle_plan()).plan.close()
        elif idle_entry() is not None:
            _drop_entry_locked(idle_entry())
        else:
            break


def _acquire(ctx, m, key=None, protect=()):
    """Device operand for one call (a _Lease).  m: scipy CSR matrix, or a PinnedOperand; key = its
    _operand_key when the caller has it already."""
    if isinstance(m, PinnedOperand):
        return m._lease(ctx)
    if _cache_entries <= 0:
        cache_stats["upload"] += 1
        return _Lease(ctx.csr_from_scipy(m), transient=True)
    pattern, data_key = key if key is not None else _operand_key(m)
    with _cache_lock:
        ent = _cache.get(pattern)
        if ent is not None and (not ent.handle.handle or ent.handle.ctx is not ctx):
            _drop_entry_locked(pattern)
            ent = None
        update = False
        if ent is not None:
            if ent.data_key != data_key:
                if ent.users > 0:                # another call is multiplying with (or uploading) other values right now
                    ent = None
                else:
                    update = True
                    ent.data_key = None          # in flux: nobody else may take it for a hit until the values are in HBM
            else:
                cache_stats["hit"] += 1
            if ent is not None:
                ent.users += 1
                ent.remember(m.data)
                _cache.move_to_end(pattern)
                if not update:
                    _trim_locked(keep=(ent,), protect=protect)
                    return _Lease(ent.handle, entry=ent)
        elif pattern not in _cache:
            _trim_locked(reserve=1, protect=protect)     # make room before the upload
        busy = ent is None and pattern in _cache
    if update:
        # the host-to-device copy (200 MB at BASELINE configs[1]) runs WITHOUT _cache_lock: the entry is marked in use
        try:
            ent.handle.update_values(m.data[:ent.handle.nnz])
        except BaseException:
            with _cache_lock:
                ent.users -= 1
                if _cache.get(pattern) is ent:
                    _drop_entry_locked(pattern)
            raise
        with _cache_lock:
            ent.data_key = data_key
            cache_stats["values_update"] += 1
            _trim_locked(keep=(ent,), protect=protect)
        return _Lease(ent.handle, entry=ent)
    h = ctx.csr_from_scipy(m)
    cache_stats["upload"] += 1
    if busy:
        return _Lease(h, transient=True)
    nbytes = h.device_bytes()
    with _cache_lock:
        if pattern in _cache:                    # another thread was faster: use ours for this call only
            return _Lease(h, transient=True)
        ent = _Entry(h, pattern, data_key)
        ent.nbytes = nbytes
        ent.users = 1
        ent.remember(m.data)
        _cache[pattern] = ent
        _trim_locked(keep=(ent,), protect=protect)
        if pattern not in _cache:                # does not fit the cache at all
            ent.users = 0
            return _Lease(h, transient=True)
    return _Lease(h, entry=ent)


def _plan_for(ctx, la, lb, symmetric, exact):
    """(plan, release) for the product of two leased operands: the cached plan of this pair of patterns when
    there is one (numeric phase only), else a fresh symbolic phase, cached when both operands are."""
    key = None
    if la.entry is not None and lb.entry is not None and _plan_entries > 0:
        key = (la.entry.pattern, lb.entry.pattern, bool(symmetric), bool(exact))
        with _cache_lock:
            pe = _plans.get(key)
            if pe is not None and pe.a is la.entry and pe.b is lb.entry and pe.plan.handle:
                pe.users += 1
                _plans.move_to_end(key)
                cache_stats["plan_hit"] += 1
                return pe.plan, lambda: _release_plan(pe)
    cache_stats["plan_miss"] += 1
    plan = ctx.spgemm_plan(la.handle, lb.handle, symmetric=symmetric, exact=exact)
    if key is None:
        return plan, plan.close
    nbytes = plan.device_bytes()
    with _cache_lock:
        old = _plans.pop(key, None)
        if old is not None and old.users == 0:
            old.plan.close()
        pe = _PlanEntry(plan, la.entry, lb.entry, nbytes)
        pe.users = 1
        _plans[key] = pe
        _trim_locked(keep=(la.entry, lb.entry))
    return plan, lambda: _release_plan(pe)


def _release_plan(pe):
    with _cache_lock:
        pe.users -= 1
        if pe.users == 0 and not any(v is pe for v in _plans.values()):
            pe.plan.close()                      # evicted while in use


def clear_cache():
    """Forget every cached operand and plan and return their HBM to the device: the handles are destroyed, and the
    scratch of the closed plans -- which the library keeps pooled for reuse -- is released too (smm_ctx_release_pool).
    Handles in use by a running call are released when that call ends."""
    ctxs = {}
    with _cache_lock:
        _bury_locked()
        for pk in list(_plans):
            pe = _plans.pop(pk)
            ctxs[id(getattr(pe.plan, "ctx", None))] = getattr(pe.plan, "ctx", None)
            if pe.users == 0:
                pe.plan.close()
        for key in list(_cache):
            ctxs[id(getattr(_cache[key].handle, "ctx", None))] = getattr(_cache[key].handle, "ctx", None)
            _drop_entry_locked(key)              # (an entry still leased is closed by that lease's release)
    for c in ctxs.values():
        release = getattr(c, "release_pool", None)
        if release is not None and getattr(c, "handle", None):
            release()


def set_operand_cache(entries):
    """Number of operands kept resident between calls (0 switches the cache off: every call marshals both
    operands afresh, as the reference does); returns the old value."""
    global _cache_entries
    old, _cache_entries = _cache_entries, int(entries)
    if _cache_entries <= 0:
        clear_cache()
    return old


class PinnedOperand:
    """An operand the caller keeps resident explicitly (no hashing, no look-up): pass it to
    sparse_matrix_multiply() in place of the matrix.  update_values() replaces the values on the same
    sparsity pattern; unpin() (or garbage collection) releases the HBM."""

    def __init__(self, ctx, matrix):
        matrix = _as_csr(matrix)
        self.shape, self.nnz = tuple(matrix.shape), int(matrix.nnz)
        self._ctx = ctx
        self._handle = ctx.csr_from_scipy(matrix) if self.nnz else None
        self._entry = _Entry(self._handle, ("pinned", id(self)), None)

    def update_values(self, data):
        if self._handle is not None:
            self._handle.update_values(data)

    def _lease(self, ctx):
        if ctx is not self._ctx or self._handle is None or not self._handle.handle:
            raise ValueError("PinnedOperand: unpinned, or pinned on another context")
        with _cache_lock:
            self._entry.users += 1
        return _Lease(self._handle, entry=self._entry)

    def unpin(self):
        """Release the operand.  Plans made on it leave the cache (one that a running call is using is closed when that
        call ends); the handle is closed now, or -- while a call still multiplies with it -- by that call's release."""
        close = None
        with _cache_lock:
            for pk in [pk for pk, pe in _plans.items() if pe.a is self._entry or pe.b is self._entry]:
                pe = _plans.pop(pk)
                if pe.users == 0:
                    pe.plan.close()
            if self._entry.users > 0:
                self._entry.dead = True
            else:
                close = self._handle
            self._handle = None
        if close is not None:
            close.close()

    def __del__(self):
        # (may run inside any allocation, also while _trim_locked walks _plans: only leave a note; the next
        # _trim_locked / clear_cache closes the plans and the handle)
        if getattr(self, "_handle", None) is not None:
            _graveyard.append(self._entry)


def pin_operand(matrix):
    """Upload `matrix` once and keep it (and everything derived from it) in HBM until unpin()."""
    return PinnedOperand(default_context(), matrix)


def _device_zeros(ctx, shape, sparse):
    import torch
    dev = torch.device("cuda", ctx.device)
    if not sparse:
        return torch.zeros(shape, dtype=torch.float64, device=dev)
    return DeviceCSRResult(torch.zeros(shape[0] + 1, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int32, device=dev),
                           torch.empty(0, dtype=torch.float64, device=dev), shape)


def _product_on_device(ctx, la, lb, triple, output_format, symmetric, compute_full_matrix):
    """The product of two leased operands with the result left in HBM (torch tensors).  The library works on the
    context's own stream: torch's current stream is drained first (its allocator may hand out memory that kernels
    queued there still use) and the context is synchronised before the tensors are returned -- which also reports
    anything the kernels' bounds clamps recorded (SMM_ERR_INTERNAL)."""
    import torch
    dev = torch.device("cuda", ctx.device)
    a, b = la.handle, lb.handle
    torch.cuda.current_stream(dev).synchronize()
    if triple:
        mirror = compute_full_matrix == 'mirror' or (_full_symmetric and compute_full_matrix == 0)
        out = torch.empty((a.rows, a.rows), dtype=torch.float64, device=dev)
        ctx.triple_into(a, b, out.data_ptr(), full=(compute_full_matrix == 1), exact=_exact, mirror=mirror)
    elif output_format == 'dense':
        out = torch.empty((a.rows, b.cols), dtype=torch.float64, device=dev)
        ctx.dense_into(a, b, out.data_ptr(), symmetric=symmetric, exact=_exact, mirror=symmetric and _full_symmetric)
    elif symmetric and _full_symmetric:
        out = DeviceCSRResult(*ctx.spgemm_mirrored_torch(a, b, exact=_exact), (a.rows, b.cols))
    else:
        plan, release = _plan_for(ctx, la, lb, symmetric, _exact)
        try:
            indptr = torch.empty(a.rows + 1, dtype=torch.int64, device=dev)
            indices = torch.empty(plan.nnz, dtype=torch.int32, device=dev)
            data = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        finally:
            release()
        out = DeviceCSRResult(indptr, indices, data, (a.rows, b.cols))
    ctx.synchronize()
    return out


def _as_csr(x):
    # reference :307-310: anything csr_matrix() accepts; no sort, no dedup
    return x if (isspmatrix_csr(x) or isinstance(x, PinnedOperand)) else csr_matrix(x)


def _result_csr(indptr, indices, data, shape):
    """reference sparsemat_to_csr (:205-228): nzmax==0 -> empty matrix; int32 index arrays.  A result
    with nnz >= 2^31 -- which the reference's int32 structs cannot represent -- carries int64
    indptr AND indices (engine.spgemm_host widens the indices while it copies them out): scipy's kernels
    take one index dtype per matrix."""
    if len(indices) == 0:
        return csr_matrix(shape)
    if indices.dtype == np.int32:
        indptr = indptr.astype(np.int32)
    out = csr_matrix(shape, dtype=np.float64)
    # assign the arrays directly: the constructor would be free to check/copy, and must not
    # sort -- the reference returns first-touch order
    out.data, out.indices, out.indptr = data, indices, indptr
    return out


def sparse_matrix_multiply(matrix_a, matrix_b, output_format='sparse', symmetric=False, imem_size=None,
                           use_triple_product=False, compute_full_matrix=None):
    """Multiply two sparse matrices on the GPU.

    matrix_a, matrix_b : scipy CSR, or anything csr_matrix() accepts.
    output_format      : 'sparse' -> scipy.sparse.csr_matrix, 'dense' -> numpy.ndarray.
    symmetric          : keep only the upper triangle (i <= j) of a square result.
    imem_size          : the reference's CPU scratch hint; validated, then ignored.
    use_triple_product : return matrix_a @ matrix_b @ matrix_a.T as a dense array (upper
                         triangle unless compute_full_matrix=1); output_format/symmetric are
                         then ignored, as in the reference (:325).
    compute_full_matrix: None/0/1, see above (1 reproduces the reference exactly).
    """
    if imem_size is None:                                    # reference :288-295
        imem_size = 5
    else:
        try:
            imem_size = int(imem_size)
        except ValueError:
            raise ValueError(f"imem_size must be an integer or None, got {type(imem_size)}")

    if compute_full_matrix is None:                          # reference :298-304
        compute_full_matrix = 0
    else:
        if isinstance(compute_full_matrix, str) and compute_full_matrix == 'mirror' and use_triple_product:
            compute_full_matrix = 'mirror'                  # extension: S itself, lower triangle mirrored
        elif compute_full_matrix not in (0, 1):
            raise ValueError("compute_full_matrix must be None, 0, or 1")
        else:
            compute_full_matrix = int(compute_full_matrix)

    matrix_a = _as_csr(matrix_a)
    matrix_b = _as_csr(matrix_b)

    if matrix_a.shape[1] != matrix_b.shape[0]:               # reference :312-313
        raise ValueError("Matrix dimensions are incompatible for multiplication.")

    out_shape = (matrix_a.shape[0], matrix_b.shape[1])
    if matrix_a.nnz == 0 or matrix_b.nnz == 0:               # reference :315-319
        if _result_device:
            return _device_zeros(default_context(), out_shape, output_format == 'sparse')
        return csr_matrix(out_shape) if output_format == 'sparse' else np.zeros(out_shape)

    if symmetric and out_shape[0] != out_shape[1]:           # reference :321-322
        raise ValueError("For symmetric output, the resulting matrix must be square.")

    if not use_triple_product and output_format not in ('sparse', 'dense'):
        # reference :367-368 raises inside its try and :377-387 prints and returns zeros
        print("An error occurred during matrix multiplication: Invalid output_format. Choose 'sparse' or 'dense'.")
        return np.zeros(out_shape)

    ctx = default_context()

    def product():
        # both keys first: looking A up must not evict the entry B is about to hit
        cached = _cache_entries > 0
        key_a = _operand_key(matrix_a) if cached and not isinstance(matrix_a, PinnedOperand) else None
        key_b = _operand_key(matrix_b) if cached and not isinstance(matrix_b, PinnedOperand) else None
        la = _acquire(ctx, matrix_a, key_a, protect=(key_b[0],) if key_b else ())
        try:
            lb = _acquire(ctx, matrix_b, key_b)
            try:
                a, b = la.handle, lb.handle
                if _result_device:
                    return _product_on_device(ctx, la, lb, use_triple_product, output_format, bool(symmetric), compute_full_matrix)
                if use_triple_product:                           # reference :325-336
                    mirror = compute_full_matrix == 'mirror' or (_full_symmetric and compute_full_matrix == 0)
                    return ctx.triple_host(a, b, full=(compute_full_matrix == 1), exact=_exact, mirror=mirror)
                if output_format == 'sparse' and symmetric and _full_symmetric:
                    # opt-in mirror epilogue: the full symmetric CSR (row i = mirrored entries in ascending column
                    # order, then the reference's upper-triangle row in its first-touch order)
                    return _result_csr(*ctx.spgemm_host_mirrored(a, b, exact=_exact), out_shape)
                if output_format == 'sparse':                    # reference :338-351
                    plan, release = _plan_for(ctx, la, lb, bool(symmetric), _exact)
                    try:
                        indptr, indices, data = plan.numeric_host()
                    finally:
                        release()
                    return _result_csr(indptr, indices, data, out_shape)
                return ctx.dense_host(a, b, symmetric=bool(symmetric), exact=_exact,  # reference :353-365
                                      mirror=bool(symmetric) and _full_symmetric)
            finally:
                lb.release()
        finally:
            la.release()

    try:
        result = product()
    except SmmError as e:
        if e.code != SMM_ERR_ALLOC or not (_cache or _plans):
            raise
        clear_cache()                                            # the resident operands / plans were in the way
        result = product()

    if _result_device and not isinstance(result, (np.ndarray, csr_matrix)):
        empty = result.nnz == 0 if isinstance(result, DeviceCSRResult) else not bool(result.any())
        if empty:
            print("Multiplication resulted in a zero matrix.")
        return result
    if isinstance(result, np.ndarray):                       # reference :370-373
        # (the first row decides almost always; a full scan of a 20 GB result costs 0.7 s)
        if not (result[:1].any() or result.any()):
            print("Multiplication resulted in a zero matrix.")
    elif result.nnz == 0:
        print("Multiplication resulted in a zero matrix.")
    return result
