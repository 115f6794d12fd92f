"""Thin object layer over the v2 C ABI (include/smm_hip.h): context, HBM-resident CSR
operands, and the three products.  All arithmetic happens in libsmm_hip.so on the GPU; this
module only moves pointers and sizes (numpy for host buffers, torch tensors -- optional --
for device buffers and streams).
"""
import ctypes
import os
import threading

import numpy as np

from ._lib import SMM_EXACT, SMM_FULL_MATRIX, SMM_MIRROR, SMM_SYMMETRIC, SmmError, SmmLibrary, check

__all__ = ["Context", "DeviceCSR", "default_context", "SmmError",
           "SMM_SYMMETRIC", "SMM_FULL_MATRIX", "SMM_EXACT", "SMM_MIRROR"]


def _flags(symmetric=False, exact=False, full=False, mirror=False):
    return ((SMM_SYMMETRIC if symmetric else 0) | (SMM_EXACT if exact else 0) | (SMM_FULL_MATRIX if full else 0) |
            (SMM_MIRROR if mirror else 0))


def _ptr(arr):
    return ctypes.c_void_p(arr.ctypes.data) if arr is not None and arr.size else ctypes.c_void_p(0)


class Context:
    """One GPU + one stream + pooled workspace (smm_ctx).  `stream`: None lets the library create
    (and own) a non-blocking stream; a raw hipStream_t (int) makes it launch there, e.g.
    torch.cuda.current_stream().cuda_stream -- whose value 0 means torch's default stream, i.e.
    the device's null stream (SMM_STREAM_DEFAULT), NOT "create one".  Calls from several host
    threads on one context are safe: the library serialises them on the context's lock."""

    def __init__(self, device=0, stream=None):
        self.lib = SmmLibrary().get_lib()
        h = ctypes.c_void_p()
        if stream is None:
            raw = 0                                   # NULL: the context creates its own stream
        elif int(stream) == 0:
            raw = ctypes.c_void_p(-1).value           # SMM_STREAM_DEFAULT: the null stream
        else:
            raw = int(stream)
        check(self.lib, self.lib.smm_ctx_create(int(device), ctypes.c_void_p(raw), ctypes.byref(h)))
        self.handle = h
        self.device = int(device)
        self.stream = None if stream is None else int(stream)     # None = private stream

    def close(self):
        if getattr(self, "handle", None):
            self.lib.smm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib, self.lib.smm_ctx_synchronize(self.handle))

    def tune(self, lds_cols=0, waves=0):
        """Geometry of the SMM_EXACT walk (waves 1/2/4/8)."""
        check(self.lib, self.lib.smm_ctx_tune(self.handle, int(lds_cols), int(waves)))

    def tune_shared(self, lds_cols=0, waves=0):
        """Geometry of the default shared-tile walk (waves 4/8/16)."""
        check(self.lib, self.lib.smm_ctx_tune_shared(self.handle, int(lds_cols), int(waves)))

    def tune_hash(self, small_max=256, medium_max=2048):
        """Rows of C with at most small_max / medium_max nonzeros use the LDS-hash kernels
        (0, 0 = dense LDS tiles for every row)."""
        check(self.lib, self.lib.smm_ctx_tune_hash(self.handle, int(small_max), int(medium_max)))

    def tune_slab(self, mode=0, ws=0, rows_per_wave=0):
        """Row-block x column-slab kernels: mode 0 auto / 1 off / 2 force; ws = slab width (0 = L2-sized);
        rows_per_wave 2 or 4."""
        check(self.lib, self.lib.smm_ctx_tune_slab(self.handle, int(mode), int(ws), int(rows_per_wave)))

    def tune_narrow(self, enable=True):
        """uint16 column stream / lists in the symbolic phase for operands with < 65535 columns (default on)."""
        check(self.lib, self.lib.smm_ctx_tune_narrow(self.handle, 1 if enable else 0))

    def tune_symbolic(self, max_slab_cols=0):
        """Widest column slab of the symbolic walk (0 = default 63456); a wider B is walked slab by slab."""
        check(self.lib, self.lib.smm_ctx_tune_symbolic(self.handle, int(max_slab_cols)))

    def tune_dense_runs(self, mode=1):
        """Symbolic walk for operands with dense runs of columns: 0 never, 1 chosen per operand (default), 2 always."""
        check(self.lib, self.lib.smm_ctx_tune_dense_runs(self.handle, int(mode)))

    def tune_stage2(self, ring=False):
        """Triple product, stage 2: the ring kernel of round 4 (True) or the chunk kernel (False, default)."""
        check(self.lib, self.lib.smm_ctx_tune_stage2(self.handle, 1 if ring else 0))

    def exact_selftest(self, inject_fault=False):
        """Run the SMM_EXACT guard now (every context runs it by itself before its first exact product):
        raises SmmError (code SMM_ERR_UNSUPPORTED) where the device does not add same-address lanes of one
        ds_add_f64 in ascending lane order.  inject_fault=True exercises that failure path."""
        check(self.lib, self.lib.smm_ctx_exact_selftest(self.handle, 1 if inject_fault else 0))

    def release_pool(self):
        """Return the pooled scratch of destroyed plans / finished calls to the device (smm_ctx_release_pool)."""
        check(self.lib, self.lib.smm_ctx_release_pool(self.handle))

    def pool_bytes(self):
        return int(self.lib.smm_ctx_pool_bytes(self.handle))

    def inject_alloc_failure(self, nth, hard=False):
        """TEST HOOK: the nth device allocation from now fails (first attempt only, or both with hard=True)."""
        check(self.lib, self.lib.smm_ctx_inject_alloc_failure(self.handle, int(nth), 1 if hard else 0))

    def alloc_retries(self):
        return int(self.lib.smm_ctx_alloc_retries(self.handle))

    def set_check(self, enable=True):
        """Run the plan checker at the end of every symbolic phase (also env SMM_CHECK=1): inconsistent plan
        metadata raises SmmError (SMM_ERR_INTERNAL).  The kernels' own bounds clamps are always on."""
        check(self.lib, self.lib.smm_ctx_set_check(self.handle, 1 if enable else 0))

    def timing(self, enable=True):
        check(self.lib, self.lib.smm_ctx_timing(self.handle, 1 if enable else 0))

    def timing_reset(self):
        check(self.lib, self.lib.smm_ctx_timing_reset(self.handle))

    def kernel_time(self, name):
        """(total milliseconds, launches) of the named kernel since the last reset."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        check(self.lib, self.lib.smm_ctx_kernel_time(self.handle, name.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    # ------------------------------------------------------------------ operands
    def csr_from_scipy(self, m):
        """Upload a scipy CSR matrix with the reference's casts (matrix_ops.py:196-198)."""
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.ascontiguousarray(m.data, dtype=np.float64)
        return self.csr_from_arrays(m.shape[0], m.shape[1], indptr, indices, data)

    def csr_from_arrays(self, rows, cols, indptr, indices, data):
        assert indptr.dtype == np.int32 and indices.dtype == np.int32 and data.dtype == np.float64
        nnz = int(indptr[-1]) if len(indptr) else 0
        h = ctypes.c_void_p()
        check(self.lib, self.lib.smm_csr_from_host(self.handle, rows, cols, nnz, _ptr(indptr), _ptr(indices),
                                                   _ptr(data), ctypes.byref(h)))
        return DeviceCSR(self, h, rows, cols, nnz)

    def csr_from_torch(self, rows, cols, indptr, indices, data):
        """Borrow int32/int32/float64 CUDA tensors already in HBM (kept alive by the handle).  The
        tensors must be complete before the library reads them: when the context does not launch on
        torch's current stream, that stream is synchronised here."""
        import torch
        cur = torch.cuda.current_stream(indices.device)
        if self.stream is None or int(cur.cuda_stream) != self.stream:
            cur.synchronize()
        nnz = int(indices.numel())
        h = ctypes.c_void_p()
        check(self.lib, self.lib.smm_csr_from_device(self.handle, rows, cols, nnz, ctypes.c_void_p(indptr.data_ptr()),
                                                     ctypes.c_void_p(indices.data_ptr()),
                                                     ctypes.c_void_p(data.data_ptr()), ctypes.byref(h)))
        out = DeviceCSR(self, h, rows, cols, nnz)
        out._keep = (indptr, indices, data)
        return out

    def row_products(self, a, b):
        out = np.zeros(a.rows, dtype=np.int64)
        check(self.lib, self.lib.smm_row_products(self.handle, a.handle, b.handle, _ptr(out)))
        return out

    # ------------------------------------------------------------------ CSR x CSR -> CSR
    def spgemm_plan(self, a, b, symmetric=False, row_offset=0, exact=False):
        flags = _flags(symmetric, exact)
        plan, nnz = ctypes.c_void_p(), ctypes.c_int64()
        check(self.lib, self.lib.smm_spgemm_symbolic(self.handle, a.handle, b.handle, flags, int(row_offset),
                                                     ctypes.byref(plan), ctypes.byref(nnz)))
        return Plan(self, plan, a, b, nnz.value)

    def spgemm_host(self, a, b, symmetric=False, row_offset=0, exact=False, index_dtype=None):
        """(indptr int64, indices int32 -- int64 when nnz >= 2^31 or index_dtype says so --, data float64)
        numpy arrays, reference (first-touch) order."""
        plan = self.spgemm_plan(a, b, symmetric, row_offset, exact)
        try:
            # nnz >= 2^31 (BASELINE configs[1]: 2.48e9): int64 column indices, like the row pointer -- scipy's
            # kernels take ONE index dtype per matrix; below that int32, as the reference returns
            return plan.numeric_host(index_dtype)
        finally:
            plan.close()

    def spgemm_torch(self, a, b, symmetric=False, row_offset=0, exact=False):
        """Same product with the result left in HBM as torch tensors (indptr int64)."""
        import torch
        plan = self.spgemm_plan(a, b, symmetric, row_offset, exact)
        try:
            dev = torch.device("cuda", self.device)
            indptr = torch.empty(a.rows + 1, dtype=torch.int64, device=dev)
            indices = torch.empty(plan.nnz, dtype=torch.int32, device=dev)
            data = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        finally:
            plan.close()
        return indptr, indices, data

    def spgemm_mirrored_torch(self, a, b, exact=False):
        """spgemm_host_mirrored with the full symmetric CSR left in HBM: (indptr int64, indices int32, data float64)
        torch tensors."""
        import torch
        if a.rows != b.cols:
            raise ValueError("For symmetric output, the resulting matrix must be square.")
        n = a.rows
        dev = torch.device("cuda", self.device)
        plan = self.spgemm_plan(a, b, symmetric=True, exact=exact)
        try:
            up = torch.empty(n + 1, dtype=torch.int64, device=dev)
            ui = torch.empty(plan.nnz, dtype=torch.int32, device=dev)
            uv = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            plan.numeric_into(up.data_ptr(), ui.data_ptr(), uv.data_ptr())
        finally:
            plan.close()
        fp = torch.empty(n + 1, dtype=torch.int64, device=dev)
        nnz = ctypes.c_int64()
        vp = ctypes.c_void_p
        check(self.lib, self.lib.smm_csr_mirror_symbolic(self.handle, n, vp(up.data_ptr()), vp(ui.data_ptr()), vp(fp.data_ptr()), ctypes.byref(nnz)))
        fi = torch.empty(nnz.value, dtype=torch.int32, device=dev)
        fv = torch.empty(nnz.value, dtype=torch.float64, device=dev)
        check(self.lib, self.lib.smm_csr_mirror_fill(self.handle, n, vp(up.data_ptr()), vp(ui.data_ptr()), vp(uv.data_ptr()), vp(fp.data_ptr()),
                                                     vp(fi.data_ptr()), vp(fv.data_ptr())))
        self.synchronize()
        return fp, fi, fv

    def _dmalloc(self, nbytes):
        p = ctypes.c_void_p()
        check(self.lib, self.lib.smm_device_malloc(self.handle, int(nbytes), ctypes.byref(p)))
        return p

    def spgemm_host_mirrored(self, a, b, exact=False):
        """The symmetric product's upper triangle (symmetric=True), mirrored on the device to the full symmetric
        CSR (smm_csr_mirror_*).  Row i: mirrored entries (columns < i) in ascending column order, then the row's
        own entries in first-touch order; any row length (long segments are placed by rank on the device).
        (indptr int64, indices int32 -- int64 when nnz >= 2^31 --, data float64) numpy arrays."""
        if a.rows != b.cols:
            raise ValueError("For symmetric output, the resulting matrix must be square.")
        n = a.rows
        plan = self.spgemm_plan(a, b, symmetric=True, exact=exact)
        bufs = []
        try:
            up, ui, uv = (self._dmalloc(8 * (n + 1)), self._dmalloc(4 * max(plan.nnz, 1)), self._dmalloc(8 * max(plan.nnz, 1)))
            bufs += [up, ui, uv]
            plan.numeric_into(up.value, ui.value, uv.value)
            fp = self._dmalloc(8 * (n + 1)); bufs.append(fp)
            nnz = ctypes.c_int64()
            check(self.lib, self.lib.smm_csr_mirror_symbolic(self.handle, n, up, ui, fp, ctypes.byref(nnz)))
            fi, fv = self._dmalloc(4 * max(nnz.value, 1)), self._dmalloc(8 * max(nnz.value, 1))
            bufs += [fi, fv]
            check(self.lib, self.lib.smm_csr_mirror_fill(self.handle, n, up, ui, uv, fp, fi, fv))
            indptr = np.empty(n + 1, dtype=np.int64)
            indices = np.empty(nnz.value, dtype=np.int32)
            data = np.empty(nnz.value, dtype=np.float64)
            for dst, src in ((indptr, fp), (indices, fi), (data, fv)):
                check(self.lib, self.lib.smm_memcpy_d2h(self.handle, _ptr(dst), src, dst.nbytes))
            if nnz.value > np.iinfo(np.int32).max:            # scipy wants one index dtype per matrix (as spgemm_host)
                indices = indices.astype(np.int64)
            return indptr, indices, data
        finally:
            plan.close()
            for p in bufs:
                self.lib.smm_device_free(self.handle, p)

    # ------------------------------------------------------------------ CSR x CSR -> dense
    def dense_host(self, a, b, symmetric=False, row_offset=0, exact=False, mirror=False):
        flags = _flags(symmetric, exact, mirror=mirror)
        out = np.empty((a.rows, b.cols), dtype=np.float64)
        check(self.lib, self.lib.smm_spgemm_dense_host(self.handle, a.handle, b.handle, flags, int(row_offset),
                                                       _ptr(out)))
        return out

    def dense_into(self, a, b, d_ptr, symmetric=False, row_offset=0, exact=False, mirror=False):
        flags = _flags(symmetric, exact, mirror=mirror)
        check(self.lib, self.lib.smm_spgemm_dense(self.handle, a.handle, b.handle, flags, int(row_offset),
                                                  ctypes.c_void_p(d_ptr)))

    # ------------------------------------------------------------------ H Q H^T
    def triple_host(self, h, q, full=False, row_begin=0, row_end=None, exact=False, mirror=False):
        row_end = h.rows if row_end is None else row_end
        out = np.empty((row_end - row_begin, h.rows), dtype=np.float64)
        check(self.lib, self.lib.smm_triple_product_host(self.handle, h.handle, q.handle, _flags(False, exact, full, mirror),
                                                         int(row_begin), int(row_end), _ptr(out)))
        return out

    def triple_into(self, h, q, d_ptr, full=False, row_begin=0, row_end=None, exact=False, mirror=False):
        row_end = h.rows if row_end is None else row_end
        check(self.lib, self.lib.smm_triple_product(self.handle, h.handle, q.handle, _flags(False, exact, full, mirror),
                                                    int(row_begin), int(row_end), ctypes.c_void_p(d_ptr)))


class DeviceCSR:
    def __init__(self, ctx, handle, rows, cols, nnz):
        self.ctx, self.handle, self.rows, self.cols, self.nnz = ctx, handle, int(rows), int(cols), int(nnz)
        self._keep = None

    def is_canonical(self):
        return bool(self.ctx.lib.smm_csr_is_canonical(self.ctx.handle, self.handle))

    def update_values(self, data):
        """New values on the same sparsity pattern (host array of nnz float64): the operand and every cached
        copy are rewritten in place, plans made on it stay valid (smm_csr_update_values)."""
        data = np.ascontiguousarray(data, dtype=np.float64)
        if data.size != self.nnz:
            raise ValueError(f"update_values: {data.size} values for an operand with {self.nnz} nonzeros")
        check(self.ctx.lib, self.ctx.lib.smm_csr_update_values(self.ctx.handle, self.handle, _ptr(data)))

    def values_changed(self, d_ptr=None):
        """The values in HBM were rewritten by the caller (borrowed operands: csr_from_torch), or are at device
        pointer d_ptr (copied over the operand's own array): refresh the cached copies."""
        check(self.ctx.lib, self.ctx.lib.smm_csr_update_values_device(self.ctx.handle, self.handle,
                                                                      ctypes.c_void_p(d_ptr or 0)))

    def device_bytes(self):
        return int(self.ctx.lib.smm_csr_device_bytes(self.handle)) if self.handle else 0

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.smm_csr_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Symbolic phase of one product (smm_plan): nnz and row pointer are known, the numeric
    phase fills caller-owned buffers."""

    def __init__(self, ctx, handle, a, b, nnz):
        self.ctx, self.handle, self.a, self.b, self.nnz = ctx, handle, a, b, int(nnz)

    def indptr_host(self):
        out = np.empty(self.a.rows + 1, dtype=np.int64)
        check(self.ctx.lib, self.ctx.lib.smm_plan_indptr_host(self.ctx.handle, self.handle, _ptr(out)))
        return out

    def device_bytes(self):
        return int(self.ctx.lib.smm_plan_device_bytes(self.handle)) if self.handle else 0

    def check(self):
        """Verify the plan's metadata on the device (smm_plan_check); raises SmmError (SMM_ERR_INTERNAL)."""
        check(self.ctx.lib, self.ctx.lib.smm_plan_check(self.ctx.handle, self.handle))

    def inject_fault(self, kind):
        """TEST HOOK: damage one piece of the plan's metadata (smm_plan_inject_fault)."""
        check(self.ctx.lib, self.ctx.lib.smm_plan_inject_fault(self.ctx.handle, self.handle, int(kind)))

    def numeric_host(self, index_dtype=None):
        """Run the numeric phase of this plan (again, after update_values on its operands if wanted) and return
        (indptr int64, indices, data) as numpy arrays."""
        ctx = self.ctx
        wide = self.nnz > np.iinfo(np.int32).max if index_dtype is None else np.dtype(index_dtype) == np.int64
        indptr = np.empty(self.a.rows + 1, dtype=np.int64)
        indices = np.empty(self.nnz, dtype=np.int64 if wide else np.int32)
        data = np.empty(self.nnz, dtype=np.float64)
        fn = ctx.lib.smm_spgemm_numeric_host_i64 if wide else ctx.lib.smm_spgemm_numeric_host
        check(ctx.lib, fn(ctx.handle, self.handle, _ptr(indptr), _ptr(indices), _ptr(data)))
        return indptr, indices, data

    def numeric_into(self, d_indptr, d_indices, d_data):
        check(self.ctx.lib, self.ctx.lib.smm_spgemm_numeric(self.ctx.handle, self.handle, ctypes.c_void_p(d_indptr),
                                                            ctypes.c_void_p(d_indices), ctypes.c_void_p(d_data)))

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.smm_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None
_default_lock = threading.Lock()


def default_context():
    """Process-wide context on the device of this rank (LOCAL_RANK, else SMM_DEVICE, else 0)."""
    global _default
    with _default_lock:
        if _default is None:
            dev = int(os.environ.get("SMM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            _default = Context(dev)
        return _default
