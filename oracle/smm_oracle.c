/*
 * smm_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY:
 * see smm_oracle.h.  Build with -ffp-contract=off: the reference's default build
 * (setup.py:142, `g++ -O3 -fPIC -std=c++11`, x86-64 baseline) has no FMA, so every
 * product is rounded before it is added.
 *
 * Parity status: dense / dense-sym / triple / limits are pinned bit for bit against
 * oracle/_ref/libsparse_ref.so (the reference's own sources compiled here, unmodified).  The
 * sparse->sparse routine is pinned three ways: values and pattern against the same reference-built
 * dense results and the reference tests' matrices; per-row counts, the first-touch ORDER of
 * colInd and the values against oracle/_ref/libsparsework_m1.so -- the reference's own
 * src/sparsework.cpp, unedited, executed with its marker array initialised to -1
 * (oracle/marker_init.c explains why HEAD needs that one value to run at all; DESIGN.md section 2
 * states the caveat).
 */
#include "smm_oracle.h"
#include <stdlib.h>
#include <string.h>

/* src/workdivision.cpp:26-86 */
int oracle_limits(int rows, int nprocs, int32_t *out)
{
    if (nprocs <= 0) return -1;                 /* :17-23 exit(0) in the reference */
    int p = nprocs > rows ? rows : nprocs;      /* :26-29 */
    if (p <= 0) return 0;
    int rem = rows % p, base = (rows - rem) / p, next = 0;
    for (int i = 0; i < p; ++i) {
        int len = base + (i < rem ? 1 : 0);     /* :59-75: first `rem` blocks get one extra */
        out[i] = next;
        out[i + p] = next + len - 1;            /* inclusive end, column-major [start|end] */
        next += len;
    }
    return p;
}

/* src/sparsework.cpp:56-129 (nosym), :201-280 (sym) */
int64_t oracle_sparsework(int64_t row_begin, int64_t row_end, int64_t n_cols_b,
                          const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                          const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                          int symmetric, int64_t *rowcnt,
                          int32_t *idx, double *val, int64_t cap)
{
    const int fill = idx != NULL;
    /* marker[col] = slot of col in the current row, or -1 (workArray, :45 / working rev.) */
    int64_t *marker = (int64_t *)malloc((size_t)(n_cols_b > 0 ? n_cols_b : 1) * sizeof(int64_t));
    int32_t *touched = NULL;
    if (!marker) return -1;
    for (int64_t c = 0; c < n_cols_b; ++c) marker[c] = -1;
    if (!fill) {
        touched = (int32_t *)malloc((size_t)(n_cols_b > 0 ? n_cols_b : 1) * sizeof(int32_t));
        if (!touched) { free(marker); return -1; }
    }
    int64_t nnz = 0;
    for (int64_t i = row_begin; i < row_end; ++i) {
        const int64_t row_start = nnz;
        for (int32_t j = a_ptr[i]; j < a_ptr[i + 1]; ++j) {          /* :59 */
            const double a = a_val[j];
            const int32_t r = a_idx[j];
            for (int32_t k = b_ptr[r]; k < b_ptr[r + 1]; ++k) {      /* :65 */
                const int32_t c = b_idx[k];
                if (symmetric && i > c) continue;                    /* :217 */
                if (marker[c] >= 0) {                                /* :70-76 */
                    if (fill) val[marker[c]] += a * b_val[k];
                } else {                                             /* :105-110 */
                    if (fill) {
                        if (nnz >= cap) { free(marker); return -1; }
                        idx[nnz] = c;
                        val[nnz] = a * b_val[k];
                    } else {
                        touched[nnz - row_start] = c;
                    }
                    marker[c] = nnz++;
                }
            }
        }
        if (rowcnt) rowcnt[i - row_begin] = nnz - row_start;         /* :116 per-row COUNT */
        for (int64_t s = row_start; s < nnz; ++s)                    /* :120-128 reset */
            marker[fill ? idx[s] : touched[s - row_start]] = -1;
    }
    free(marker);
    free(touched);
    return nnz;
}

/* src/sparse_sparse_sparse.cpp:181-185 (zero early-out), :196 limits, :235-241 work,
 * :269-291 stitch */
int64_t oracle_sparse(int64_t m, int64_t n, int nparts,
                      const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                      const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                      int symmetric, int64_t *c_ptr, int32_t *c_idx, double *c_val)
{
    c_ptr[0] = 0;
    if (m == 0) return 0;
    if (nparts < 1) nparts = 1;
    int32_t *lim = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(nparts > m ? m : nparts));
    if (!lim) return -1;
    const int p = oracle_limits((int)m, nparts, lim);
    int64_t total = 0;
    for (int part = 0; part < p; ++part) {
        const int64_t r0 = lim[part], r1 = (int64_t)lim[part + p] + 1;
        int64_t *cnt = c_ptr + 1 + r0;      /* per-row counts land where the prefix will be */
        int64_t got;
        if (c_idx == NULL) {
            got = oracle_sparsework(r0, r1, n, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val,
                                    symmetric, cnt, NULL, NULL, 0);
        } else {
            /* capacity: whatever the first call reported for this partition */
            int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r1 - r0));
            if (!tmp) { free(lim); return -1; }
            const int64_t cap = c_ptr[r1] - c_ptr[r0];
            got = oracle_sparsework(r0, r1, n, a_ptr, a_idx, a_val, b_ptr, b_idx, b_val,
                                    symmetric, tmp, c_idx + c_ptr[r0], c_val + c_ptr[r0], cap);
            free(tmp);
        }
        if (got < 0) { free(lim); return -1; }
        total += got;
    }
    if (c_idx == NULL)                       /* :272-276 rowPtr[r+1] = rowPtr[r] + count[r] */
        for (int64_t r = 0; r < m; ++r) c_ptr[r + 1] += c_ptr[r];
    free(lim);
    return total;
}

/* src/sparse_sparse_dense.cpp:108-129 (nosym), :40-73 (sym) */
void oracle_dense(int64_t row_begin, int64_t row_end, int64_t n,
                  const int32_t *a_ptr, const int32_t *a_idx, const double *a_val,
                  const int32_t *b_ptr, const int32_t *b_idx, const double *b_val,
                  int symmetric, double *c)
{
    memset(c, 0, sizeof(double) * (size_t)(row_end - row_begin) * (size_t)n);  /* calloc :97 */
    for (int64_t i = row_begin; i < row_end; ++i) {
        double *row = c + (size_t)(i - row_begin) * (size_t)n;
        for (int32_t j = a_ptr[i]; j < a_ptr[i + 1]; ++j) {
            const double a = a_val[j];
            const int32_t r = a_idx[j];
            for (int32_t k = b_ptr[r]; k < b_ptr[r + 1]; ++k) {
                const int32_t cb = b_idx[k];
                const double prod = a * b_val[k];                    /* :124 / :56 */
                if (!symmetric || i <= cb) row[cb] += prod;          /* :127 / :59-61 */
            }
        }
    }
}

/* src/sparse_sparse_dense.cpp:185-220 */
void oracle_triple(int64_t n, int64_t kdim,
                   const int32_t *h_ptr, const int32_t *h_idx, const double *h_val,
                   const int32_t *q_ptr, const int32_t *q_idx, const double *q_val,
                   int full, int64_t row_begin, int64_t row_end, double *c)
{
    double *temp = (double *)calloc((size_t)(kdim > 0 ? kdim : 1), sizeof(double));  /* :178 */
    if (!temp) return;
    for (int64_t i = row_begin; i < row_end; ++i) {
        for (int32_t jp = h_ptr[i]; jp < h_ptr[i + 1]; ++jp) {       /* stage 1 :187-198 */
            const int32_t j = h_idx[jp];
            const double h = h_val[jp];
            for (int32_t kp = q_ptr[j]; kp < q_ptr[j + 1]; ++kp)
                temp[q_idx[kp]] += h * q_val[kp];
        }
        for (int64_t kk = full ? 0 : i; kk < n; ++kk) {              /* stage 2 :201-216 */
            double sum = 0.0;
            for (int32_t jp = h_ptr[kk]; jp < h_ptr[kk + 1]; ++jp)
                sum += temp[h_idx[jp]] * h_val[jp];
            c[(size_t)i * (size_t)n + (size_t)kk] += sum;            /* :212 */
            if (full && i != kk)
                c[(size_t)kk * (size_t)n + (size_t)i] += sum;        /* :213-215 (F6) */
        }
        memset(temp, 0, sizeof(double) * (size_t)kdim);              /* :219 */
    }
    free(temp);
}
