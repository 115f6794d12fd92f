// smm_legacy.cpp -- the nine (+4) symbols the reference's unmodified ctypes wrapper binds
// (reference sparse_matrix_mult/matrix_ops.py:147-171), implemented on the v2 API.
// Host arrays in, libc-malloc'd host arrays out (the caller frees them with destroy_*,
// matrix_ops.py:336,351,365), `int` struct layout (matrix_ops.py:26-33,44-48).
// Errors follow the reference's convention: message on stderr, void return, output struct
// left empty (src/sparsework.cpp:33-36, src/sparse_sparse_sparse.cpp:257-262) -- but never
// exit() (src/workdivision.cpp:19-23 does; a shared library should not).
#include "../../include/smm_hip.h"

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static smm_ctx *g_ctx = nullptr;
static std::mutex g_mu;          // guards creation of the shared context
static std::mutex g_call_mu;     // a context is single-threaded: legacy calls are serialised on it

static smm_ctx *legacy_ctx()
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ctx) {
        int dev = 0;
        if (const char *e = getenv("SMM_DEVICE")) dev = atoi(e);
        if (smm_ctx_create(dev, nullptr, &g_ctx) != SMM_OK) {
            fprintf(stderr, "libsmm_hip: %s\n", smm_last_error());
            g_ctx = nullptr;
        }
    }
    return g_ctx;
}

// SMM_EXACT=1 in the environment selects reference-order accumulation (bit-identical values) for
// the legacy symbols too, as sparse_matrix_mult_amd.set_exact() does for the Python API.
static int legacy_mode_flags()
{
    const char *e = getenv("SMM_EXACT");
    return (e && *e && strcmp(e, "0") != 0) ? SMM_EXACT : 0;
}

extern "C" {

struct sparsemat *create_sparsemat(int rows, int cols, int nzmax)
{
    struct sparsemat *m = (struct sparsemat *)calloc(1, sizeof(struct sparsemat));
    if (!m) return nullptr;
    m->rows = rows; m->cols = cols; m->nzmax = nzmax;
    m->rowPtr = (int *)calloc((size_t)(rows > 0 ? rows : 0) + 1, sizeof(int));
    m->colInd = (int *)calloc((size_t)(nzmax > 0 ? nzmax : 0), sizeof(int));
    m->values = (double *)calloc((size_t)(nzmax > 0 ? nzmax : 0), sizeof(double));
    return m;
}

struct darray *create_darray(int rows, int cols)
{
    struct darray *m = (struct darray *)calloc(1, sizeof(struct darray));
    if (!m) return nullptr;
    m->rows = rows; m->cols = cols;
    m->array = (double *)calloc((size_t)(rows > 0 ? rows : 0) * (size_t)(cols > 0 ? cols : 0), sizeof(double));
    return m;
}

void destroy_sparsemat(struct sparsemat *m)
{
    if (!m) return;
    free(m->rowPtr); free(m->colInd); free(m->values);
    m->rowPtr = m->colInd = nullptr; m->values = nullptr;
    m->nzmax = m->rows = m->cols = 0;
}
void destroy_darray(struct darray *m)
{
    if (!m) return;
    free(m->array); m->array = nullptr; m->rows = m->cols = 0;
}
void destroy_iarray(struct iarray *m)
{
    if (!m) return;
    free(m->array); m->array = nullptr; m->rows = m->cols = 0;
}

void modifyalloc(struct sparsemat *m, int new_size)
{
    if (!m) return;
    if (new_size <= 0) {
        free(m->colInd); free(m->values);
        m->colInd = nullptr; m->values = nullptr;
        return;
    }
    int *ci = (int *)realloc(m->colInd, (size_t)new_size * sizeof(int));
    if (ci) m->colInd = ci;
    double *vv = (double *)realloc(m->values, (size_t)new_size * sizeof(double));
    if (vv) m->values = vv;
    if (!ci || !vv) fprintf(stderr, "Reallocation failed.\n");
}

void limits(int tcov_rows, int numprocs, struct iarray *result)
{
    if (!result) return;
    result->array = nullptr; result->rows = 0; result->cols = 2;
    if (numprocs <= 0) { fprintf(stderr, "limits: numprocs must be positive\n"); return; }
    const int p = numprocs > tcov_rows ? tcov_rows : numprocs;
    if (p <= 0) return;
    result->array = (int *)calloc((size_t)p * 2, sizeof(int));
    if (!result->array) { fprintf(stderr, "limits: allocation failed\n"); return; }
    result->rows = p;
    const int extra = tcov_rows % p, base = tcov_rows / p;
    for (int i = 0, at = 0; i < p; ++i) {
        const int len = base + (i < extra);
        result->array[i] = at;
        result->array[i + p] = at + len - 1;
        at += len;
    }
}

}  // extern "C"

namespace {

struct Operand {
    smm_csr *h = nullptr;
    ~Operand() { smm_csr_destroy(h); }
};

bool upload(smm_ctx *c, const struct sparsemat *m, Operand &o, int row0 = 0, int row1 = -1)
{
    if (!m || !m->rowPtr) { fprintf(stderr, "libsmm_hip: NULL operand\n"); return false; }
    if (row1 < 0) row1 = m->rows;
    const int nr = row1 - row0;
    int rc;
    if (row0 == 0 && row1 == m->rows) {
        rc = smm_csr_from_host(c, m->rows, m->cols, m->rowPtr[m->rows], m->rowPtr, m->colInd, m->values, &o.h);
    } else {
        std::vector<int> ptr((size_t)nr + 1);
        const int base = m->rowPtr[row0];
        for (int i = 0; i <= nr; ++i) ptr[i] = m->rowPtr[row0 + i] - base;
        rc = smm_csr_from_host(c, nr, m->cols, ptr[nr], ptr.data(), m->colInd + base, m->values + base, &o.h);
    }
    if (rc != SMM_OK) { fprintf(stderr, "libsmm_hip: %s\n", smm_last_error()); return false; }
    return true;
}

// counts_only_ptr: sparsework_* return per-row COUNTS in rowPtr (sparsework.cpp:116),
// sparse_* return the running sum (sparse_sparse_sparse.cpp:272-276).
void sparse_impl(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *out, int flags, int row0,
                 int row1, bool counts)
{
    if (!out) return;
    out->nzmax = 0; out->rowPtr = out->colInd = nullptr; out->values = nullptr;
    if (!a || !b) return;
    const int nr = row1 - row0;
    out->rows = nr; out->cols = b->cols;
    if (a->cols != b->rows) { fprintf(stderr, "Error: Matrix dimensions are incompatible for multiplication.\n"); return; }
    out->rowPtr = (int *)calloc((size_t)nr + 1, sizeof(int));
    if (!out->rowPtr) { fprintf(stderr, "Memory allocation failed for matrixC\n"); return; }
    if (nr <= 0 || a->nzmax == 0 || b->nzmax == 0) return;     // sparse_sparse_sparse.cpp:181-185
    smm_ctx *c = legacy_ctx();
    if (!c) return;
    std::lock_guard<std::mutex> call_lock(g_call_mu);
    Operand A, B;
    if (!upload(c, a, A, row0, row1) || !upload(c, b, B)) return;
    smm_plan *plan = nullptr;
    int64_t nnz = 0;
    if (smm_spgemm_symbolic(c, A.h, B.h, flags | legacy_mode_flags(), row0, &plan, &nnz) != SMM_OK) {
        fprintf(stderr, "libsmm_hip: %s\n", smm_last_error());
        return;
    }
    if (nnz > INT_MAX) {
        fprintf(stderr, "libsmm_hip: result has %lld nonzeros, more than the int32 ABI of struct sparsemat can hold; "
                        "use the v2 API (smm_spgemm_*)\n", (long long)nnz);
        smm_plan_destroy(plan);
        return;
    }
    std::vector<int64_t> ptr64((size_t)nr + 1);
    int *ci = (int *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    double *cv = (double *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double));
    if (!ci || !cv) {
        fprintf(stderr, "Memory allocation failed for matrixC\n");
        free(ci); free(cv); smm_plan_destroy(plan);
        return;
    }
    if (smm_spgemm_numeric_host(c, plan, ptr64.data(), ci, cv) != SMM_OK) {
        fprintf(stderr, "libsmm_hip: %s\n", smm_last_error());
        free(ci); free(cv); smm_plan_destroy(plan);
        return;
    }
    smm_plan_destroy(plan);
    if (counts) for (int i = 0; i < nr; ++i) out->rowPtr[i] = (int)(ptr64[i + 1] - ptr64[i]);
    else        for (int i = 0; i <= nr; ++i) out->rowPtr[i] = (int)ptr64[i];
    out->colInd = ci; out->values = cv; out->nzmax = (int)nnz;
}

void dense_impl(const struct sparsemat *a, const struct sparsemat *b, struct darray *out, int flags)
{
    if (!out) return;
    out->array = nullptr;
    if (!a || !b) return;
    if (a->cols != b->rows) { fprintf(stderr, "Error: Matrix dimensions are incompatible for multiplication.\n"); return; }
    out->rows = a->rows; out->cols = b->cols;
    const size_t total = (size_t)a->rows * (size_t)b->cols;
    out->array = (double *)calloc(total ? total : 1, sizeof(double));
    if (!out->array) { fprintf(stderr, "Error: Memory allocation failed for matrixc->array.\n"); return; }
    if (total == 0 || a->nzmax == 0 || b->nzmax == 0) return;
    smm_ctx *c = legacy_ctx();
    if (!c) { free(out->array); out->array = nullptr; return; }
    std::lock_guard<std::mutex> call_lock(g_call_mu);
    Operand A, B;
    if (!upload(c, a, A) || !upload(c, b, B) ||
        smm_spgemm_dense_host(c, A.h, B.h, flags | legacy_mode_flags(), 0, out->array) != SMM_OK) {
        fprintf(stderr, "libsmm_hip: %s\n", smm_last_error());
        free(out->array); out->array = nullptr;
    }
}

}  // namespace

extern "C" {

void sparse_nosym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int)
{
    sparse_impl(a, b, c, 0, 0, a ? a->rows : 0, false);
}
void sparse_sym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int)
{
    sparse_impl(a, b, c, SMM_SYMMETRIC, 0, a ? a->rows : 0, false);
}
void sparsework_nosym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int startIndex,
                      int endIndex, int)
{
    if (!a || startIndex < 0 || endIndex >= a->rows || startIndex > endIndex + 1) {
        fprintf(stderr, "sparsework_nosym: bad row range\n");
        return;
    }
    sparse_impl(a, b, c, 0, startIndex, endIndex + 1, true);
}
void sparsework_sym(const struct sparsemat *a, const struct sparsemat *b, struct sparsemat *c, int startIndex,
                    int endIndex, int)
{
    if (!a || startIndex < 0 || endIndex >= a->rows || startIndex > endIndex + 1) {
        fprintf(stderr, "sparsework_sym: bad row range\n");
        return;
    }
    sparse_impl(a, b, c, SMM_SYMMETRIC, startIndex, endIndex + 1, true);
}
void dense_nosym(const struct sparsemat *a, const struct sparsemat *b, struct darray *c) { dense_impl(a, b, c, 0); }
void dense_sym(const struct sparsemat *a, const struct sparsemat *b, struct darray *c)
{
    dense_impl(a, b, c, SMM_SYMMETRIC);
}

void triple_product(struct sparsemat *h, struct sparsemat *q, struct darray *out, int compute_full_matrix)
{
    if (!out) return;
    out->array = nullptr;
    if (!h || !q) return;
    const int n = h->rows;
    out->rows = n; out->cols = n;
    const size_t total = (size_t)n * (size_t)n;
    out->array = (double *)calloc(total ? total : 1, sizeof(double));
    if (!out->array) { fprintf(stderr, "Memory allocation failed for C->array\n"); return; }
    if (total == 0 || h->nzmax == 0 || q->nzmax == 0) return;
    smm_ctx *c = legacy_ctx();
    if (!c) { free(out->array); out->array = nullptr; return; }
    std::lock_guard<std::mutex> call_lock(g_call_mu);
    Operand H, Q;
    if (!upload(c, h, H) || !upload(c, q, Q) ||
        smm_triple_product_host(c, H.h, Q.h, (compute_full_matrix ? SMM_FULL_MATRIX : 0) | legacy_mode_flags(), 0, n,
                                out->array) != SMM_OK) {
        fprintf(stderr, "libsmm_hip: %s\n", smm_last_error());
        free(out->array); out->array = nullptr;
    }
}

}  // extern "C"
