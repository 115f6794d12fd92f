/*
 * smm_hip.h -- C ABI of libsmm_hip.so, the MI355X (gfx950) engine behind
 * sparse_matrix_multiply().  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Two layers are exported by the same library:
 *
 *  (1) The LEGACY entry points the reference's Python wrapper binds by name
 *      (reference sparse_matrix_mult/matrix_ops.py:147-171, call sites :195,333,346,348,
 *      360,362,336,351,365; declarations include/functions.h:43-84).  Struct layout is the
 *      one that wrapper declares -- `int` dims (matrix_ops.py:26-33: 40-byte sparsemat,
 *      :44-48: 16-byte darray) -- NOT the size_t layout of include/matrix_def.h at HEAD
 *      (SURVEY F1).  Host arrays in, libc-malloc'd host arrays out, freed by destroy_*.
 *
 *  (2) The v2 API our own matrix_ops.py, bench.py and the multi-GPU driver call: operands
 *      live in HBM behind opaque handles, results are written into caller-owned DEVICE (or
 *      host) buffers, row pointers / nnz are int64 (nnz(C) of the 50k x 50k d=0.01 config
 *      is 2.48e9 > INT32_MAX), every call returns 0 or a negative smm_status and
 *      smm_last_error() gives the message.  Nothing here falls back to a CPU path: with no
 *      usable GPU every compute entry point fails with SMM_ERR_NO_DEVICE.
 */
#ifndef SMM_HIP_H
#define SMM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status / flags */
enum smm_status {
    SMM_OK = 0,
    SMM_ERR_NO_DEVICE = -1,   /* no gfx950 device visible / HIP init failed            */
    SMM_ERR_INVALID   = -2,   /* bad argument, shape mismatch, malformed CSR            */
    SMM_ERR_ALLOC     = -3,   /* hipMalloc / malloc failed                              */
    SMM_ERR_HIP       = -4,   /* a HIP runtime call or kernel launch failed             */
    SMM_ERR_OVERFLOW  = -5,   /* result does not fit the legacy int32 ABI               */
    SMM_ERR_UNSUPPORTED = -6, /* the device does not behave as SMM_EXACT needs (smm_ctx_exact_selftest) */
    SMM_ERR_INTERNAL  = -7    /* inconsistent plan metadata caught by a kernel's bounds clamp or by the plan
                                 checker: a defect of this library, reported instead of a hang or a fault
                                 (reference convention: message + early return, src/sparsework.cpp:33-36,
                                 src/sparse_sparse_sparse.cpp:257-262)                                   */
};

enum smm_flags {
    SMM_SYMMETRIC   = 1,  /* keep only i <= col   (sparsework.cpp:217, sparse_sparse_dense.cpp:59) */
    SMM_FULL_MATRIX = 2,  /* triple_product compute_full_matrix=1 (sparse_sparse_dense.cpp:201,213) */
    SMM_MIRROR      = 8,  /* mirror epilogue (not in the reference): after an upper-triangle result -- dense with
                             SMM_SYMMETRIC, or the triple product without SMM_FULL_MATRIX -- the lower triangle
                             is filled with the mirror image of the upper one, i.e. the full symmetric matrix
                             (what compute_full_matrix=1 was meant to give; SMM_FULL_MATRIX reproduces its
                             doubling bug instead).  Whole square results only.                          */
    SMM_EXACT       = 4   /* add every product in exactly the reference's order: float64 values
                             are then bit-identical to the CPU loop, for any legal CSR operand (B with
                             unsorted rows or repeated columns takes an ordered read-modify-write
                             path instead of the tile kernels: exact, not fast).  Without it the numeric phase lets the waves of a workgroup
                             add concurrently (LDS atomics): values agree to rounding -- tested
                             to the north star's 1e-10 relative -- and the kernel is ~3x faster.
                             indptr / indices are bit-exact in both modes.                     */
};

typedef struct smm_ctx  smm_ctx;   /* one device + one stream + a workspace arena        */
typedef struct smm_csr  smm_csr;   /* a CSR operand resident in HBM (+ cached tile index) */
typedef struct smm_plan smm_plan;  /* result of the symbolic phase of one product         */

/* ------------------------------------------------------------------ context
 * A context owns one device, one stream and a workspace pool.  Every entry point takes the
 * context's lock, so host threads may share one context (their calls serialise; the reference's
 * library has no global state and is called with the GIL released, matrix_ops.py:136).
 * Handles (smm_csr, smm_plan) belong to the context that made them; a plan borrows its two
 * operands, which must stay alive until the plan is destroyed. */
int         smm_device_count(void);              /* number of usable devices, 0 if none  */
const char *smm_last_error(void);                /* thread-local message of the last failure */
/* hip_stream: a hipStream_t to launch on, NULL for a non-blocking stream the context creates and
 * owns, or SMM_STREAM_DEFAULT for the device's null stream (what torch's default stream is:
 * torch.cuda.current_stream().cuda_stream == 0 -- pass SMM_STREAM_DEFAULT for it, not NULL).
 * Device buffers handed to smm_csr_from_device must be complete on that stream (or the
 * caller synchronises first): the library orders its work only against its own stream. */
#define SMM_STREAM_DEFAULT ((void *)(intptr_t)-1)
int  smm_ctx_create(int device, void *hip_stream, smm_ctx **out);
void smm_ctx_destroy(smm_ctx *ctx);
/* Waits for the context's stream and reports what the kernels recorded since the last report: SMM_ERR_INTERNAL when
 * a numeric kernel met inconsistent plan metadata (see smm_plan_check) -- the way an error of the asynchronous
 * smm_spgemm_numeric reaches the caller. */
int  smm_ctx_synchronize(smm_ctx *ctx);
/* 1: every smm_spgemm_symbolic ends with smm_plan_check (also: env SMM_CHECK=1 when the context is created).  Off by
 * default: the checker streams the ordered lists once more (~2 ms at 50k x 50k); the kernels' own clamps are always on. */
int  smm_ctx_set_check(smm_ctx *ctx, int enable);
/* The context keeps freed scratch (the lists and tables of destroyed plans, result staging) in a pool for reuse.
 * smm_ctx_release_pool returns the pool's free blocks to the device (matrix_ops.clear_cache() calls it);
 * smm_ctx_pool_bytes says how much is held.  Every allocation that fails flushes the pool and retries once by itself. */
int     smm_ctx_release_pool(smm_ctx *ctx);
int64_t smm_ctx_pool_bytes(smm_ctx *ctx);
/* TEST HOOK: the nth device allocation from now fails -- its first attempt only (the library's own flush-and-retry
 * must make the call succeed), or with hard != 0 both attempts (the call returns SMM_ERR_ALLOC).  smm_ctx_alloc_retries
 * counts the allocations that needed the retry. */
int     smm_ctx_inject_alloc_failure(smm_ctx *ctx, int nth, int hard);
int64_t smm_ctx_alloc_retries(smm_ctx *ctx);
/* Per-kernel timing with HIP events on the context's stream (bench.py's roofline leg).
 * enable=1 starts recording; smm_ctx_kernel_time returns the accumulated milliseconds and
 * launch count of the named kernel since the last reset (name as printed by rocprofv3,
 * without template arguments: "smm_numeric", "smm_symbolic", ...). */
int  smm_ctx_timing(smm_ctx *ctx, int enable);
int  smm_ctx_timing_reset(smm_ctx *ctx);
int  smm_ctx_kernel_time(smm_ctx *ctx, const char *kernel, double *ms_total, int64_t *launches);
/* Tuning knobs (0 keeps the default): LDS accumulator columns per workgroup (x 8 bytes of
 * LDS; default 20000 = one workgroup per CU) and waves per workgroup of the numeric kernels.
 * In SMM_EXACT mode results do not depend on them, bit for bit. */
int  smm_ctx_tune(smm_ctx *ctx, int lds_cols, int waves);             /* SMM_EXACT walk          */
int  smm_ctx_tune_shared(smm_ctx *ctx, int lds_cols, int waves);      /* default walk (4/8/16)   */
/* Rows of C with at most small_max (<= 256) / medium_max (<= 2048) nonzeros are accumulated in an
 * LDS hash table (one wave / one workgroup per row) instead of dense LDS tiles; 0, 0 turns the
 * hash kernels off.  Defaults 256 / 2048. */
int  smm_ctx_tune_hash(smm_ctx *ctx, int small_max, int medium_max);
/* Row-block x column-slab numeric kernels (B's gather served by the XCD's L2; values always in the
 * reference's order).  mode 0 = used where they pay (default), 1 = never, 2 = wherever they can run;
 * ws = slab width in columns (0 = sized so that one slab of B is L2-resident); rows_per_wave 2 or 4
 * (0 keeps the setting).  Results do not depend on any of it beyond default-mode rounding. */
int  smm_ctx_tune_slab(smm_ctx *ctx, int mode, int ws, int rows_per_wave);
/* 1 (default): products whose B has < 65535 columns run the symbolic phase on uint16 -- a 16-bit copy of
 * B's column indices (cached on the operand) and 16-bit ordered lists: half the bytes of that phase's gather
 * and of the list traffic.  0: always int32.  Results are identical. */
int  smm_ctx_tune_narrow(smm_ctx *ctx, int enable);
/* The symbolic phase walks B as a chunk-padded stream of 16-bit columns, one column slab of at most
 * max_slab_cols columns at a time (0 = the default, 63 456: products whose B has fewer columns are one slab).  A
 * wider B is walked slab by slab -- every (slab, row) has its own ordered list, the numeric phase reads them
 * through a table of (source, length, destination) sub-runs -- which keeps the marker bitmaps small (many waves
 * per CU) and the lists 16-bit at any width.  Results do not depend on it, bit for bit. */
int  smm_ctx_tune_symbolic(smm_ctx *ctx, int max_slab_cols);
/* Symbolic walk for operands B with dense runs of columns (bands, blocks): 0 never, 1 (default) chosen per operand from the
   share of neighbouring entries that fall into one 32-column word of the marker bitmap, 2 always.  Results never depend on it. */
int  smm_ctx_tune_dense_runs(smm_ctx *ctx, int mode);
/* Triple product, stage 2: ring != 0 selects the ring kernel of round 4 (the tile of T as a ring of column pieces, the
 * waves of a workgroup synchronised by progress words in LDS instead of barriers: csrc/smm_ring.hpp), 0 (default) the
 * chunk kernel.  Results are identical (bit for bit under SMM_EXACT); the ring is the slower of the two on MI355X
 *  and is kept as a tested alternative.  Also: env SMM_S2_RING=1 when the context is created. */
int  smm_ctx_tune_stage2(smm_ctx *ctx, int ring);
/* Run-time guard of SMM_EXACT.  The exact walk adds the products of one wave-instruction that fall on the same
 * accumulator in ascending lane order (the reference's order, src/sparsework.cpp:59-76) -- a property of
 * gfx950's ds_add_f64 that was measured, not one the ISA promises.  This runs a sub-millisecond kernel that
 * checks exactly that (random groupings of lanes on addresses, values of many magnitudes, compared bit for
 * bit with the sum taken lane by lane) and fails with SMM_ERR_UNSUPPORTED where it does not hold.  Every
 * context runs it by itself before its first SMM_EXACT product; inject_fault != 0 makes the check compare
 * against the DESCENDING lane order instead, i.e. exercises the failure path (tests). */
int  smm_ctx_exact_selftest(smm_ctx *ctx, int inject_fault);

/* ------------------------------------------------------------------ operands
 * Replaces create_sparsemat + the three memmoves of csr_to_sparsemat
 * (matrix_ops.py:187-202; src/memfunctions.cpp:117-131).  indptr/indices are int32 as in
 * the reference (`int* rowPtr, colInd`, include/matrix_def.h:21-22); nnz < 2^31.
 * The CSR is validated on the device (monotone indptr, indices in range); a malformed
 * operand is rejected with SMM_ERR_INVALID instead of faulting a kernel. */
int  smm_csr_from_host(smm_ctx *ctx, int64_t rows, int64_t cols, int64_t nnz,
                       const int32_t *indptr, const int32_t *indices, const double *data,
                       smm_csr **out);
/* Borrow arrays already in HBM (no copy; they must outlive the handle). */
int  smm_csr_from_device(smm_ctx *ctx, int64_t rows, int64_t cols, int64_t nnz,
                         const int32_t *d_indptr, const int32_t *d_indices, const double *d_data,
                         smm_csr **out);
void smm_csr_destroy(smm_csr *m);
int64_t smm_csr_rows(const smm_csr *m);
int64_t smm_csr_cols(const smm_csr *m);
int64_t smm_csr_nnz(const smm_csr *m);
/* Values-only update of an operand whose sparsity pattern stays (the reference re-marshals all three arrays
 * on every call, matrix_ops.py:187-202,339-340; the README's use case -- covariance products, README.md:5,13 --
 * repeats one pattern with new values).  `data`: nnz doubles in HOST memory, copied over the operand's values
 * in HBM; every cached copy that holds values (packed tile payload, slab-major copy, sliced-ELL copy of H) is
 * refreshed in place, so plans made on this operand stay valid and smm_spgemm_numeric can simply be run
 * again on them: a product with an unchanged pattern costs the numeric phase only.
 * _device: the new values are already in HBM at d_data (copied, for operands made by smm_csr_from_host), or
 * d_data is NULL / the operand's own borrowed array, which the caller has rewritten in place (operands made
 * by smm_csr_from_device): only the cached copies are refreshed. */
int  smm_csr_update_values(smm_ctx *ctx, smm_csr *m, const double *data);
int  smm_csr_update_values_device(smm_ctx *ctx, smm_csr *m, const double *d_data);
/* HBM held by the handle: its arrays (when owned) plus every cached copy. */
int64_t smm_csr_device_bytes(const smm_csr *m);
/* 64-bit content hash of a HOST buffer (no GPU involved): a chain of bijective mixing steps, so changing any
 * one 8-byte word always changes the result; large buffers are hashed by several threads.  matrix_ops.py keys
 * its operand cache on it -- a full-content check, so an operand edited in place is never served stale. */
uint64_t smm_host_hash64(const void *p, int64_t bytes);
/* 1 when every row has strictly increasing column indices (scipy "canonical" CSR). */
int  smm_csr_is_canonical(smm_ctx *ctx, smm_csr *m);
/* products[i] = sum over nonzeros (i,r) of A of nnz(B[r,:]) -- the work measure used to
 * balance contiguous row shards across GPUs (replaces limits(), src/workdivision.cpp:16-89,
 * which balances row counts).  Host output, a.rows entries. */
int  smm_row_products(smm_ctx *ctx, const smm_csr *a, const smm_csr *b, int64_t *products_host);

/* ------------------------------------------------------------------ CSR x CSR -> CSR
 * Replaces sparse_nosym / sparse_sym (src/sparse_sparse_sparse.cpp:172-299 / :41-155) and
 * the row kernels sparsework_nosym / sparsework_sym (src/sparsework.cpp:12-149 / :156-300).
 * Two phases, as the reference's count-then-stitch driver: symbolic produces per-row counts
 * and the first-touch-ordered column lists and returns nnz(C); the caller allocates
 * c_indices / c_data (device) of that size; numeric fills indptr (int64, a.rows+1),
 * indices (int32, reference order) and data.
 * a_row_offset: global index of A's row 0 when A is one contiguous row shard of a larger
 * matrix (only the SMM_SYMMETRIC filter i <= col looks at it). */
int  smm_spgemm_symbolic(smm_ctx *ctx, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset,
                         smm_plan **plan, int64_t *nnz_out);
int  smm_spgemm_numeric(smm_ctx *ctx, smm_plan *plan,
                        int64_t *d_c_indptr, int32_t *d_c_indices, double *d_c_data);
/* Same, results copied into host buffers (numpy arrays owned by the caller). */
int  smm_spgemm_numeric_host(smm_ctx *ctx, smm_plan *plan,
                             int64_t *c_indptr, int32_t *c_indices, double *c_data);
/* The same with int64 column indices in the host array (widened while the chunks are copied out): for
 * results with nnz >= 2^31, where a scipy CSR needs indptr and indices of one (64-bit) dtype -- which
 * the reference's int32 structs cannot represent at all. */
int  smm_spgemm_numeric_host_i64(smm_ctx *ctx, smm_plan *plan,
                                 int64_t *c_indptr, int64_t *c_indices, double *c_data);
/* Only the int64 row pointer of the planned product (device->host, a.rows+1 entries). */
int  smm_plan_indptr_host(smm_ctx *ctx, smm_plan *plan, int64_t *c_indptr);
/* Verify every invariant of the plan that the numeric phase relies on (list capacities non-negative and not
 * overlapping, row counts = row pointer, start slots monotone, list entries in range, sub-run tables consistent with
 * the lists, tile by tile) on the device.  SMM_OK, or SMM_ERR_INTERNAL with the first offending row in
 * smm_last_error().  The numeric kernels additionally clamp everything they read from a plan, always: a table that
 * slipped through can cost a row its values, never a store outside the row, a hang or a fault. */
int  smm_plan_check(smm_ctx *ctx, smm_plan *plan);
/* TEST HOOK: damage one piece of the plan's metadata in HBM (kind 1..8: reversed / out-of-row sub-run, tail
 * descriptor, row count, list entry, slab sub-run, start slot, negative capacity) so that the error path above can
 * be exercised; the plan is to be destroyed afterwards. */
int  smm_plan_inject_fault(smm_ctx *ctx, smm_plan *plan, int kind);
int64_t smm_plan_nnz(const smm_plan *plan);
int64_t smm_plan_device_bytes(const smm_plan *plan);   /* HBM scratch the plan holds (lists, sub-run table, ...) */
void smm_plan_destroy(smm_plan *plan);

/* CSR mirror epilogue (not in the reference, opt-in): callers of symmetric=True get only i <= col
 * (src/sparsework.cpp:217).  These two calls turn such an n x n upper-triangle CSR, resident in HBM, into the full
 * symmetric matrix, in HBM: _symbolic writes the full row pointer (n+1 entries) and returns the full nnz, the
 * caller allocates, _fill writes indices and values.  Order inside row i of the full matrix: first the mirrored
 * entries (columns j < i) in ascending column order, then the row's own entries in the order the input holds them
 * (the reference's first-touch order).  Any row length: mirrored segments of up to 8192 entries are sorted in LDS,
 * longer ones (results as full as the BASELINE configs') are placed by rank -- the columns of a segment are distinct,
 * so an entry's sorted position is the number of set bits below its column in a bitmap of the segment.  Entries left
 * of the diagonal in the input are refused with SMM_ERR_INVALID. */
int  smm_csr_mirror_symbolic(smm_ctx *ctx, int64_t n, const int64_t *d_indptr, const int32_t *d_indices,
                             int64_t *d_full_indptr, int64_t *nnz_full);
int  smm_csr_mirror_fill(smm_ctx *ctx, int64_t n, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                         const int64_t *d_full_indptr, int32_t *d_full_indices, double *d_full_data);

/* ------------------------------------------------------------------ CSR x CSR -> dense
 * Replaces dense_nosym / dense_sym (src/sparse_sparse_dense.cpp:79-131 / :13-74).
 * d_c: a.rows x b.cols row-major float64 in HBM; every element is written (cells the
 * reference leaves at calloc's 0.0 are written as 0.0). */
int  smm_spgemm_dense(smm_ctx *ctx, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset,
                      double *d_c);
int  smm_spgemm_dense_host(smm_ctx *ctx, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset,
                           double *c);

/* ------------------------------------------------------------------ H * Q * H^T
 * Replaces triple_product (src/sparse_sparse_dense.cpp:141-249).  Rows
 * [row_begin,row_end) of the n x n result are computed; d_c points at row row_begin of a
 * row-major buffer with leading dimension n.  Without SMM_FULL_MATRIX only k >= i is
 * computed and the rest of each row is written as 0.0.  With SMM_FULL_MATRIX the whole
 * range must be [0,n) and the reference's behaviour is reproduced exactly: every
 * off-diagonal cell holds S[i,k] + S[k,i]. */
int  smm_triple_product(smm_ctx *ctx, smm_csr *h, smm_csr *q, int flags,
                        int64_t row_begin, int64_t row_end, double *d_c);
int  smm_triple_product_host(smm_ctx *ctx, smm_csr *h, smm_csr *q, int flags,
                             int64_t row_begin, int64_t row_end, double *c);

/* ------------------------------------------------------------------ device memory helpers
 * (so that hosts without torch can still hold results in HBM) */
int  smm_device_malloc(smm_ctx *ctx, int64_t bytes, void **d_ptr);
int  smm_device_free(smm_ctx *ctx, void *d_ptr);
int  smm_memcpy_d2h(smm_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);
int  smm_memcpy_h2d(smm_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes);

/* ================================================================== LEGACY ABI
 * Binary drop-in for the library the reference's unmodified matrix_ops.py loads
 * (lib/libsparse*.so, matrix_ops.py:118-136).  Layouts as that file declares them. */
struct sparsemat {                 /* matrix_ops.py:26-33 */
    int nzmax, rows, cols;
    int *rowPtr;
    int *colInd;
    double *values;
};
struct darray {                    /* matrix_ops.py:44-48 */
    double *array;
    int rows, cols;
};
struct iarray {                    /* include/matrix_def.h:34-38 with int dims */
    int *array;
    int rows, cols;
};

struct sparsemat *create_sparsemat(int rows, int cols, int nzmax);   /* memfunctions.cpp:117-131 */
struct darray    *create_darray(int rows, int cols);                 /* memfunctions.cpp:143-154 */
void destroy_sparsemat(struct sparsemat *m);                         /* memfunctions.cpp:22-33   */
void destroy_darray(struct darray *m);                               /* memfunctions.cpp:51-57   */
void destroy_iarray(struct iarray *m);                               /* memfunctions.cpp:37-43   */
void modifyalloc(struct sparsemat *m, int new_size);                 /* memfunctions.cpp:77-103  */
void limits(int tcov_rows, int numprocs, struct iarray *result);     /* workdivision.cpp:16-89   */
/* imemSize is the reference's CPU scratch hint (sparse_sparse_sparse.cpp:204-217); it has
 * no meaning on the GPU and is accepted and ignored.  On failure the output struct is left
 * with nzmax==0 / NULL arrays and a message goes to stderr, as the reference does. */
void sparse_nosym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int imemSize);
void sparse_sym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int imemSize);
void sparsework_nosym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c,
                      int startIndex, int endIndex, int memIncrease);   /* sparsework.cpp:12-149  */
void sparsework_sym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c,
                    int startIndex, int endIndex, int memIncrease);     /* sparsework.cpp:156-300 */
void dense_nosym(const struct sparsemat *a, const struct sparsemat *b, struct darray *c);
void dense_sym(const struct sparsemat *a, const struct sparsemat *b, struct darray *c);
void triple_product(struct sparsemat *h, struct sparsemat *q, struct darray *c, int compute_full_matrix);

#ifdef __cplusplus
}
#endif
#endif /* SMM_HIP_H */
