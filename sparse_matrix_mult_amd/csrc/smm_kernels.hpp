// smm_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the CSR x CSR engine.
//
// Path restated (reference file:line under /root/reference):
//   sparsework_nosym / _sym   src/sparsework.cpp:56-129 / :201-280   -> smm_symbolic + smm_numeric
//   sparse_nosym / _sym       src/sparse_sparse_sparse.cpp:269-291   -> smm_scan + smm_compact (no stitch copy)
//   dense_nosym / _sym        src/sparse_sparse_dense.cpp:108-129 / :40-73 -> smm_numeric<OUT_DENSE>
//   triple_product            src/sparse_sparse_dense.cpp:185-220    -> smm_numeric<OUT_DENSE> + smm_triple_stage2
//
// Design notes (DESIGN.md has the long form):
//  * A row of C is accumulated in LDS as a dense f64 tile of <= lds_cols columns ("coarse
//    tile"); a workgroup owns one (row, coarse tile) unit.  In ORDERED mode every wave owns
//    a contiguous "fine tile" of that coarse tile and walks the row's A entries in stored
//    order, so each accumulator receives its products in exactly the reference's order
//    (one wave's LDS atomics execute in issue order): values are bit-identical.
//  * The first-touch column order of the reference is produced by smm_symbolic:
//    one wave per row, a bitmap of B's columns in LDS, test-and-set + ballot/mbcnt ordered
//    compaction.  The numeric epilogue then moves accumulators to their first-touch slots
//    in contiguous sub-runs (for each A entry j, the new columns of step j that fall in the
//    tile are contiguous in the row because B's rows are sorted).
//  * No MFMA anywhere: this is an indexing / HBM path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smm {

constexpr int WAVE = 64;
enum { OUT_SPARSE = 0, OUT_DENSE = 1 };

// flag bits written by smm_validate
enum { CSR_BAD = 1, CSR_UNSORTED = 2, CSR_HAS_EQUAL = 4 };

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ int rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned rl(unsigned v, int l) {
    return (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ double rl(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ void lds_add(double *p, double x) {
    // ds_add_f64 (no return): fire-and-forget, executed by the LDS in issue order
    (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double *p, double x) {
    (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------
// Operand validation: one wave per row.  Sets CSR_BAD for a non-monotone indptr or an index
// outside [0, cols); CSR_UNSORTED / CSR_HAS_EQUAL describe the column order inside rows.
// A malformed operand must never reach the compute kernels (an out-of-range column would be
// an out-of-bounds LDS / HBM write).
__global__ __launch_bounds__(256) void smm_validate(int rows, int cols, int nnz,
                                                    const int *__restrict__ ptr,
                                                    const int *__restrict__ idx,
                                                    unsigned *__restrict__ flags)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    unsigned f = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (ptr[0] != 0 || ptr[rows] != nnz) f |= CSR_BAD;
    }
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const int s = ptr[row], e = ptr[row + 1];
        if (s > e || s < 0 || e > nnz) { f |= CSR_BAD; continue; }
        for (int k = s + lane; k < e; k += WAVE) {
            const int c = idx[k];
            if (c < 0 || c >= cols) f |= CSR_BAD;
            if (k > s) {
                const int p = idx[k - 1];
                if (p > c) f |= CSR_UNSORTED;
                if (p == c) f |= CSR_HAS_EQUAL;
            }
        }
    }
    if (f) atomicOr(flags, f);
}

// ---------------------------------------------------------------------------------------
// Tile index of a sorted operand: seg[row*(n_ft+1) + t] = first position in the row whose
// column is >= t*wf (t = n_ft -> row end).  One thread per (row, t).  Built once per operand
// and tile width, cached on the handle.
__global__ __launch_bounds__(256) void smm_segptr(int rows, int n_ft, int wf,
                                                  const int *__restrict__ ptr,
                                                  const int *__restrict__ idx,
                                                  int *__restrict__ seg)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = n_ft + 1;
    if (gid >= (int64_t)rows * per) return;
    const int row = (int)(gid / per), t = (int)(gid % per);
    int lo = ptr[row], hi = ptr[row + 1];
    if (t < n_ft) {
        const int64_t bound = (int64_t)t * wf;
        while (lo < hi) {                       // lower_bound(idx[lo..hi), bound)
            const int mid = lo + ((hi - lo) >> 1);
            if ((int64_t)idx[mid] < bound) lo = mid + 1; else hi = mid;
        }
    } else {
        lo = hi;
    }
    seg[gid] = lo;
}

// ---------------------------------------------------------------------------------------
// Per-row work: products[i] = sum_j nnz(B[a_idx[j],:]); ub[i] = min(products, columns that
// can appear) -- the capacity of the row's first-touch list.  One wave per row.
__global__ __launch_bounds__(256) void smm_row_work(int m, int ncols, int64_t row_offset, int sym,
                                                    const int *__restrict__ a_ptr,
                                                    const int *__restrict__ a_idx,
                                                    const int *__restrict__ b_ptr,
                                                    int64_t *__restrict__ products,
                                                    int64_t *__restrict__ ub)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        int64_t s = 0;
        for (int e = a_ptr[row] + lane; e < a_ptr[row + 1]; e += WAVE) {
            const int r = a_idx[e];
            s += b_ptr[r + 1] - b_ptr[r];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (lane == 0) {
            products[row] = s;
            int64_t cap = ncols;
            if (sym) { const int64_t gi = row + row_offset; cap = gi < ncols ? ncols - gi : 0; }
            if (ub) ub[row] = s < cap ? s : cap;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Exclusive scan, one 1024-thread workgroup walking the array (rows are at most a few 1e5).
// out has n+1 entries; out[n] = total.
template <typename T>
__global__ __launch_bounds__(1024) void smm_scan(int n, const T *__restrict__ in, int64_t *__restrict__ out)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int64_t v = i < n ? (int64_t)in[i] : 0;
        int64_t x = v;                                   // inclusive wave scan
        for (int o = 1; o < WAVE; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int64_t carry = carry_s;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

// ---------------------------------------------------------------------------------------
// Symbolic phase = the reference's first-touch rule (sparsework.cpp:70-110 with the marker
// of the working revision): walk A's row in stored order, inside it B's row in stored
// order, append a column the first time it is seen.  One wave per row; the marker is a
// bitmap of B's columns (LDS when it fits, else a global slab per wave).
//   tmp_idx + ub_off[row] : the row's ordered column list (capacity ub[row])
//   P[e]                  : number of columns already in the list when A entry e starts
//   rowcnt[row]           : final length = nnz of the row of C
// SAFE resolves two lanes of one wave-instruction hitting the same column (possible only
// when a row of B repeats a column): the LOWEST lane must win, whatever the LDS picks.
constexpr int SYM_UNROLL = 4;
template <bool SYM, bool SAFE, bool LDSBM>
__global__ __launch_bounds__(256) void smm_symbolic(int m, int64_t row_offset, int bm_words,
                                                    const int *__restrict__ a_ptr,
                                                    const int *__restrict__ a_idx,
                                                    const int *__restrict__ b_ptr,
                                                    const int *__restrict__ b_idx,
                                                    const int64_t *__restrict__ ub_off,
                                                    int *__restrict__ tmp_idx,
                                                    unsigned *__restrict__ P,
                                                    int *__restrict__ rowcnt,
                                                    unsigned *__restrict__ gbitmap)
{
    extern __shared__ unsigned lds_bm[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x / WAVE;
    unsigned *bm = LDSBM ? lds_bm + (size_t)wave * bm_words
                         : gbitmap + ((size_t)blockIdx.x * wpb + wave) * bm_words;
    for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0;
    if (!LDSBM) __threadfence();

    for (int row = blockIdx.x * wpb + wave; row < m; row += gridDim.x * wpb) {
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const int64_t gi = row + row_offset;
        int *__restrict__ out = tmp_idx + ub_off[row];
        int n = 0;
        for (int jb = a0; jb < a1; jb += WAVE) {
            const int e = jb + lane;
            int bs = 0, be = 0;
            if (e < a1) { const int r = a_idx[e]; bs = b_ptr[r]; be = b_ptr[r + 1]; }
            unsigned myP = 0;
            const int nb = (a1 - jb) < WAVE ? (a1 - jb) : WAVE;
            for (int jj = 0; jj < nb; ++jj) {
                const int s = rl(bs, jj), en = rl(be, jj);
                if (lane == jj) myP = (unsigned)n;
                for (int base = s; base < en; base += WAVE * SYM_UNROLL) {
                    int c[SYM_UNROLL];
                    bool act[SYM_UNROLL];
#pragma unroll
                    for (int u = 0; u < SYM_UNROLL; ++u) {      // all loads first (MLP)
                        const int k = base + u * WAVE + lane;
                        act[u] = k < en;
                        c[u] = act[u] ? b_idx[k] : 0;
                    }
#pragma unroll
                    for (int u = 0; u < SYM_UNROLL; ++u) {
                        if (base + u * WAVE >= en) break;        // wave-uniform
                        bool a = act[u];
                        if (SYM) a = a && ((int64_t)c[u] >= gi);
                        const unsigned bit = 1u << (c[u] & 31);
                        unsigned *wp = bm + (c[u] >> 5);
                        bool isnew = false;
                        if (SAFE) {
                            bool pre = true;
                            if (a) pre = (*(volatile unsigned *)wp & bit) != 0;
                            bool hw = false;
                            if (a) hw = (atomicOr(wp, bit) & bit) == 0;
                            isnew = hw;
                            unsigned long long losers = __ballot(a && !pre && !hw);
                            while (losers) {                     // rare: duplicate column in a B row
                                const int x = __ffsll((long long)losers) - 1;
                                const int cx = rl(c[u], x);
                                const bool ingrp = a && c[u] == cx;
                                const unsigned long long grp = __ballot(ingrp);
                                const int first = __ffsll((long long)grp) - 1;
                                if (ingrp) isnew = (lane == first);
                                losers &= ~grp;
                            }
                        } else {
                            if (a) isnew = (atomicOr(wp, bit) & bit) == 0;
                        }
                        const unsigned long long mask = __ballot(isnew);
                        if (isnew) out[n + mbcnt(mask)] = c[u];
                        n += __popcll(mask);
                    }
                }
            }
            if (e < a1) P[e] = myP;
        }
        if (lane == 0) rowcnt[row] = n;
        // reset the marker for the next row (sparsework.cpp:120-128: memset when the row is
        // long, per-entry otherwise)
        if (!LDSBM) __threadfence();
        if (n >= bm_words) {
            for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0;
        } else {
            for (int s2 = lane; s2 < n; s2 += WAVE) bm[out[s2] >> 5] = 0;
        }
        if (!LDSBM) __threadfence();
    }
}

// ---------------------------------------------------------------------------------------
// Move the ordered lists from their capacity-strided slots to the final CSR index array.
// One workgroup per row chunk; pure streaming copy.
__global__ __launch_bounds__(256) void smm_compact(int m, const int64_t *__restrict__ ub_off,
                                                   const int64_t *__restrict__ c_ptr,
                                                   const int *__restrict__ tmp_idx,
                                                   int *__restrict__ c_idx)
{
    for (int row = blockIdx.x; row < m; row += gridDim.x) {
        const int64_t src = ub_off[row], dst = c_ptr[row];
        const int n = (int)(c_ptr[row + 1] - dst);
        for (int s = threadIdx.x; s < n; s += blockDim.x) c_idx[dst + s] = tmp_idx[src + s];
    }
}

// ---------------------------------------------------------------------------------------
// Sub-run table.  Step e of a row appended the columns list[P[e] .. P[e+1]) in ascending
// order (B's rows are sorted), so the part that falls into coarse tile t is the contiguous
// slot range [runs[e][t], runs[e][t+1]).  One lane per A entry, nct-1 lower_bounds each.
__global__ __launch_bounds__(256) void smm_runs(int m, int nct, int wc,
                                                const int *__restrict__ a_ptr,
                                                const int64_t *__restrict__ ub_off,
                                                const int *__restrict__ rowcnt,
                                                const unsigned *__restrict__ P,
                                                const int *__restrict__ tmp_idx,
                                                unsigned *__restrict__ runs)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const int *__restrict__ list = tmp_idx + ub_off[row];
        const unsigned total = (unsigned)rowcnt[row];
        for (int e = a0 + lane; e < a1; e += WAVE) {
            const unsigned p0 = P[e];
            const unsigned p1 = (e + 1 < a1) ? P[e + 1] : total;
            unsigned *r = runs + (size_t)e * (nct + 1);
            r[0] = p0;
            unsigned lo = p0;
            for (int t = 1; t < nct; ++t) {
                const int64_t bound = (int64_t)t * wc;
                unsigned hi = p1;
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    if ((int64_t)list[mid] < bound) lo = mid + 1; else hi = mid;
                }
                r[t] = lo;
            }
            r[nct] = p1;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Numeric phase.  Workgroup = one (row, coarse tile) unit; NW waves; LDS = wc doubles.
//   OUT_SPARSE : accumulators start at -0.0 (the additive identity: -0.0 + p == p bit for
//                bit, which reproduces `values[index] = p` of sparsework.cpp:108-109), the
//                epilogue writes them to their first-touch slots.
//   OUT_DENSE  : accumulators start at +0.0 (calloc, sparse_sparse_dense.cpp:97), the
//                epilogue writes the tile row to C[row, lo..hi).
//   ORDERED    : wave w owns fine tile w (columns [lo_c + w*wf, +wf)) and visits every A
//                entry in order -> bit-exact sums.  Otherwise the waves split the A entries
//                and add concurrently (any order, LDS atomics).
struct NumericArgs {
    int m, ncols, nct, wc, wf, n_ft;
    int64_t row_offset;
    const int *a_ptr, *a_idx; const double *a_val;
    const int *b_idx; const double *b_val;
    const int *seg;                 // [rowsB][n_ft+1]
    // sparse output
    const int64_t *c_ptr; const int *c_idx; double *c_val; const unsigned *runs;
    // dense output
    double *c_dense; int64_t ldc;
};

constexpr int NUM_UNROLL = 8;

template <int OUT, bool SYM, bool ORDERED, int NW>
__global__ __launch_bounds__(NW * 64) void smm_numeric(const NumericArgs A)
{
    extern __shared__ double acc[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int tc = blockIdx.x / A.m;            // tile-major: concurrent units share B's slab
    const int row = blockIdx.x - tc * A.m;
    const int a0 = A.a_ptr[row], a1 = A.a_ptr[row + 1];
    const int lo_c = tc * A.wc;
    const int64_t gi = row + A.row_offset;
    int64_t rs = 0;
    if (OUT == OUT_SPARSE) {
        rs = A.c_ptr[row];
        if (A.c_ptr[row + 1] == rs) return;     // empty row of C (workgroup-uniform)
    }
    const bool below = SYM && ((int64_t)lo_c + A.wc <= gi);   // tile entirely left of the diagonal
    if (OUT == OUT_SPARSE && below) return;

    const double zero = OUT == OUT_SPARSE ? -0.0 : 0.0;
    for (int x = threadIdx.x; x < A.wc; x += NW * 64) acc[x] = zero;
    __syncthreads();

    if (!below) {
        const int ft0 = tc * NW;
        const int per = A.n_ft + 1;
        for (int jb = a0; jb < a1; jb += WAVE) {
            const int e = jb + lane;
            int s_l = 0, e_l = 0;
            double av = 0.0;
            if (e < a1) {
                const int r = A.a_idx[e];
                av = A.a_val[e];
                const int *sp = A.seg + (size_t)r * per + ft0;
                if (ORDERED) { s_l = sp[wave]; e_l = sp[wave + 1]; }
                else         { s_l = sp[0];    e_l = sp[NW]; }
            }
            const int nb = (a1 - jb) < WAVE ? (a1 - jb) : WAVE;
            const int jstep = ORDERED ? 1 : NW;
            for (int jj = ORDERED ? 0 : wave; jj < nb; jj += jstep * NUM_UNROLL) {
                int s[NUM_UNROLL], en[NUM_UNROLL], c[NUM_UNROLL];
                double a[NUM_UNROLL], v[NUM_UNROLL];
                bool p[NUM_UNROLL];
#pragma unroll
                for (int u = 0; u < NUM_UNROLL; ++u) {      // issue every first-chunk load
                    const int j = jj + u * jstep;
                    const int jc = j < WAVE ? j : WAVE - 1;
                    s[u] = rl(s_l, jc); en[u] = rl(e_l, jc); a[u] = rl(av, jc);
                    if (j >= nb) en[u] = s[u];
                    const int k = s[u] + lane;
                    p[u] = k < en[u];
                    c[u] = p[u] ? A.b_idx[k] : 0;
                    v[u] = p[u] ? A.b_val[k] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < NUM_UNROLL; ++u) {      // then add, in A-entry order
                    bool ok = p[u];
                    if (SYM) ok = ok && ((int64_t)c[u] >= gi);
                    if (ok) lds_add(&acc[c[u] - lo_c], a[u] * v[u]);
                    for (int base = s[u] + WAVE; base < en[u]; base += WAVE) {   // long segments
                        const int k = base + lane;
                        if (k < en[u]) {
                            const int c2 = A.b_idx[k];
                            const double v2 = A.b_val[k];
                            if (!SYM || (int64_t)c2 >= gi) lds_add(&acc[c2 - lo_c], a[u] * v2);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    if (OUT == OUT_DENSE) {
        const int w = (A.ncols - lo_c) < A.wc ? (A.ncols - lo_c) : A.wc;
        double *dst = A.c_dense + (int64_t)row * A.ldc + lo_c;
        for (int x = threadIdx.x; x < w; x += NW * 64) dst[x] = acc[x];
    } else {
        const int per = A.nct + 1;
        for (int jb = a0; jb < a1; jb += WAVE) {
            const int e = jb + lane;
            unsigned r0 = 0, r1 = 0;
            if (e < a1) { const unsigned *r = A.runs + (size_t)e * per + tc; r0 = r[0]; r1 = r[1]; }
            const int nb = (a1 - jb) < WAVE ? (a1 - jb) : WAVE;
            for (int jj = wave; jj < nb; jj += NW) {
                const unsigned s0 = rl(r0, jj), s1 = rl(r1, jj);
                for (unsigned sl = s0 + lane; sl < s1; sl += WAVE) {
                    const int c = A.c_idx[rs + sl];
                    A.c_val[rs + sl] = acc[c - lo_c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// General numeric fallbacks for an operand B whose rows are NOT sorted (tile segments are
// then not contiguous).  One wave per row, the reference's own structure: a column -> slot
// map in HBM (workArray, sparsework.cpp:45) and one global f64 atomic per product.  Values
// agree to rounding (atomics), indices are untouched.  Correctness path, not a fast path.
template <bool SYM>
__global__ __launch_bounds__(256) void smm_numeric_general(int m, int ncols, int64_t row_offset,
                                                           const int *__restrict__ a_ptr,
                                                           const int *__restrict__ a_idx,
                                                           const double *__restrict__ a_val,
                                                           const int *__restrict__ b_ptr,
                                                           const int *__restrict__ b_idx,
                                                           const double *__restrict__ b_val,
                                                           const int64_t *__restrict__ c_ptr,
                                                           const int *__restrict__ c_idx,
                                                           double *__restrict__ c_val,
                                                           int *__restrict__ slotmap)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    int *map = slotmap + ((size_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * (size_t)ncols;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        const int64_t rs = c_ptr[row];
        const int n = (int)(c_ptr[row + 1] - rs);
        const int64_t gi = row + row_offset;
        for (int s = lane; s < n; s += WAVE) { map[c_idx[rs + s]] = s; c_val[rs + s] = -0.0; }
        __threadfence();
        for (int j = a_ptr[row]; j < a_ptr[row + 1]; ++j) {
            const int r = a_idx[j];
            const double a = a_val[j];
            for (int k = b_ptr[r] + lane; k < b_ptr[r + 1]; k += WAVE) {
                const int c = b_idx[k];
                if (SYM && (int64_t)c < gi) continue;
                glb_add(&c_val[rs + map[c]], a * b_val[k]);
            }
        }
        __threadfence();
    }
}

template <bool SYM>
__global__ __launch_bounds__(256) void smm_dense_general(int m, int ncols, int64_t row_offset,
                                                         const int *__restrict__ a_ptr,
                                                         const int *__restrict__ a_idx,
                                                         const double *__restrict__ a_val,
                                                         const int *__restrict__ b_ptr,
                                                         const int *__restrict__ b_idx,
                                                         const double *__restrict__ b_val,
                                                         double *__restrict__ c, int64_t ldc)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        double *dst = c + (int64_t)row * ldc;
        const int64_t gi = row + row_offset;
        for (int x = lane; x < ncols; x += WAVE) dst[x] = 0.0;
        __threadfence();
        for (int j = a_ptr[row]; j < a_ptr[row + 1]; ++j) {
            const int r = a_idx[j];
            const double a = a_val[j];
            for (int k = b_ptr[r] + lane; k < b_ptr[r + 1]; k += WAVE) {
                const int cb = b_idx[k];
                if (SYM && (int64_t)cb < gi) continue;
                glb_add(&dst[cb], a * b_val[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Triple product, stage 2 (sparse_sparse_dense.cpp:201-216): C[i,k] = sum over row k of H of
// T[i, col] * val, for k >= i (or all k), where T = H*Q is the dense n x K matrix stage 1
// (smm_numeric<OUT_DENSE>) left in HBM.  The sum runs in H's stored order starting from
// 0.0, exactly as the reference's scalar loop.
// Workgroup = R rows i of T x one chunk of K columns in LDS; the running sums of a row live
// in C itself between chunks (chunks are visited in ascending column order = stored order
// of a sorted H, so the order of additions is unchanged).  Each lane owns one k and walks
// its H segment; hseg is H's tile index with tile width = chunk.
struct TripleArgs {
    int n, K, nchunks, chunk;
    int64_t row_begin, row_end;
    int full;
    const int *h_ptr, *h_idx; const double *h_val;
    const int *hseg;                  // [n][nchunks+1]
    const double *T;                  // (row_end-row_begin) x K
    double *C; int64_t ldc;           // row row_begin at C
};

template <int R, int NW>
__global__ __launch_bounds__(NW * 64) void smm_triple_stage2(const TripleArgs A)
{
    extern __shared__ double tl[];                     // [R][chunk]
    const int64_t i0 = A.row_begin + (int64_t)blockIdx.x * R;
    const int nr = (A.row_end - i0) < R ? (int)(A.row_end - i0) : R;
    const int per = A.nchunks + 1;
    const int64_t kfirst = A.full ? 0 : i0;            // lanes below their own row's diagonal are masked
    for (int q = 0; q < A.nchunks; ++q) {
        const int lo = q * A.chunk;
        const int w = (A.K - lo) < A.chunk ? (A.K - lo) : A.chunk;
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            const double *src = A.T + (int64_t)(i0 - A.row_begin + r) * A.K + lo;
            for (int x = threadIdx.x; x < w; x += NW * 64) tl[r * A.chunk + x] = src[x];
        }
        __syncthreads();
        for (int64_t k = kfirst + threadIdx.x; k < A.n; k += NW * 64) {
            const int s = A.hseg[k * per + q], e = A.hseg[k * per + q + 1];
            double sum[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                sum[r] = 0.0;
                if (q > 0 && r < nr && (A.full || k >= i0 + r))
                    sum[r] = A.C[(int64_t)(i0 - A.row_begin + r) * A.ldc + k];
            }
            for (int jp = s; jp < e; ++jp) {
                const int col = A.h_idx[jp] - lo;
                const double hv = A.h_val[jp];
#pragma unroll
                for (int r = 0; r < R; ++r) sum[r] += tl[r * A.chunk + col] * hv;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < nr && (A.full || k >= i0 + r))
                    A.C[(int64_t)(i0 - A.row_begin + r) * A.ldc + k] = sum[r];
        }
    }
    // cells left of the diagonal: the reference's calloc'd zeros
    if (!A.full) {
        for (int r = 0; r < nr; ++r) {
            double *dst = A.C + (int64_t)(i0 - A.row_begin + r) * A.ldc;
            for (int64_t k = threadIdx.x; k < i0 + r && k < A.n; k += NW * 64) dst[k] = 0.0;
        }
    }
}

// compute_full_matrix=1 (sparse_sparse_dense.cpp:212-215): cell (a,b), a != b, receives
// S[min,max] first and S[max,min] second; the diagonal receives S[a,a] once.  In place on
// the full S: each thread owns one unordered pair.
__global__ __launch_bounds__(256) void smm_triple_mirror(int n, double *__restrict__ C, int64_t ldc)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)n * n) return;
    const int a = (int)(gid / n), b = (int)(gid % n);
    if (a >= b) return;
    const double v = C[(int64_t)a * ldc + b] + C[(int64_t)b * ldc + a];
    C[(int64_t)a * ldc + b] = v;
    C[(int64_t)b * ldc + a] = v;
}

}  // namespace smm
