/*
 * smm_oracle.h -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Nothing in the product path (sparse_matrix_mult_amd/, the C-ABI library) may include,
 * link or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (under /root/reference) it restates.
 * Index arrays are int32 on input (the reference's `int* rowPtr/colInd`,
 * include/matrix_def.h:21-22); row pointers of the OUTPUT are int64 so that the
 * BASELINE configs with nnz(C) > INT32_MAX can still be checked on row subsets.
 */
#ifndef SMM_ORACLE_H
#define SMM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* src/workdivision.cpp:16-89.  out[i] = start, out[i+p] = inclusive end, p = min(nprocs, rows).
 * Returns p (the reference stores it in result->rows), or -1 where the reference exit(0)s. */
int oracle_limits(int rows, int nprocs, int32_t *out);

/* src/sparsework.cpp:12-149 (nosym) / :156-300 (sym), with the marker semantics of the
 * working revision (marker initialised to -1, "== -1 => new").
 * Rows [row_begin,row_end) of C = A*B.  Pass 1 (idx==NULL): fills rowcnt[0..nrows) only.
 * Pass 2: appends first-touch-ordered (col,val) into idx/val (capacity cap), returns nnz
 * or -1 on overflow.  symmetric!=0 keeps only i <= col (sparsework.cpp:217). */
int64_t oracle_sparsework(int64_t row_begin, int64_t row_end, int64_t n_cols_b,
                          const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                          const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                          int symmetric, int64_t *rowcnt,
                          int32_t *idx, double *val, int64_t cap);

/* src/sparse_sparse_sparse.cpp:172-299 / :41-155: partition rows with limits(), run
 * sparsework per partition, stitch (rowPtr = running sum of per-row counts, :272-276;
 * colInd/values concatenated in partition order, :281-286).  c_ptr has m+1 int64 entries.
 * Two-call protocol: call with c_idx==NULL to get nnz and c_ptr, then with buffers. */
int64_t oracle_sparse(int64_t m, int64_t n, int nparts,
                      const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                      const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                      int symmetric, int64_t *c_ptr, int32_t *c_idx, double *c_val);

/* src/sparse_sparse_dense.cpp:79-131 (nosym) / :13-74 (sym).  c is m*n row-major and is
 * zero-filled here (the reference calloc()s it, :97/:30).  Rows [row_begin,row_end) only
 * are computed; c points at row row_begin. */
void oracle_dense(int64_t row_begin, int64_t row_end, int64_t n,
                  const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                  const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                  int symmetric, double *c);

/* src/sparse_sparse_dense.cpp:141-249, single-thread semantics (one thread-local copy,
 * so the final reduction :229-242 is 0.0 + x).  full!=0 reproduces the reference's
 * compute_full_matrix=1 behaviour bit for bit, including SURVEY F6 (every off-diagonal
 * cell receives S[i,k] + S[k,i]).  Rows [row_begin,row_end) of the n x n result; c points
 * at row 0 of a zero-filled n*n buffer when full!=0 (mirror writes land in other rows),
 * and at row row_begin otherwise is NOT supported: c is always the whole n*n buffer. */
void oracle_triple(int64_t n, int64_t k,
                   const int32_t *h_ptr, const int32_t *h_idx, const double *h_val,
                   const int32_t *q_ptr, const int32_t *q_idx, const double *q_val,
                   int full, int64_t row_begin, int64_t row_end, double *c);

#ifdef __cplusplus
}
#endif
#endif
