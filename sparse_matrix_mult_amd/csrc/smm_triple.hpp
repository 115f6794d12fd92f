// smm_triple.hpp -- stage 2 of the triple product with LANES = ROWS OF T ("row-lane" kernel).
//
// Reference: src/sparse_sparse_dense.cpp:201-216 -- for every row i and every k >= i,
//     C[i,k] = sum over row k of H, in stored order, of temp[i, col] * val      (temp = row i of T = H Q).
//
// The first kernel (smm_triple_stage2, smm_kernels.hpp) gives one k to every lane and keeps 16 rows of T per
// lane in registers; each lane gathers its own column of the LDS tile.  That is 8 bytes of LDS per
// multiply-add at the rate of a RANDOM gather: 16 lanes of a ds_read_b128 group fall on 16 bank slots like
// balls into bins (~2.9 deep), and the sliced-ELL steps run to the longest of 64 segments (67 % of the lanes
// busy): 74.6 ms at BASELINE configs[3], against a floor of ~50 ms for that formulation.
//
// Here a wave takes ONE entry of H at a time -- wave-uniform, fetched with scalar loads -- and its 64 lanes are
// 64 rows of T: the tile sits in LDS column-major, so the 64 lanes read one contiguous 512-byte column
// (conflict-free, the LDS's full rate) and no lane ever idles on padding.  Each lane keeps the running sums of
// its row for the NKW rows k its wave owns (static register indices: the loop over those k is unrolled, the
// loop over a row's entries inside it is dynamic).  Entries are stored per (column chunk, k) in blocks of four
// (offset of the column inside the LDS tile, value), padded with (zero column, 0.0): a sum starts at +0.0 and
// can never be -0.0, so adding +0.0 * 0.0 leaves it bit for bit.
//
// Traffic: a workgroup = 64 rows of T x KG rows k.  The tile of (row block, chunk) is needed by every k-group
// of that row block; those workgroups are placed on ONE XCD and run side by side (unit = row block * n_kg +
// k-group, block b -> unit (b % 8) * cpx + b / 8), so the tile comes through the fabric about once and out
// of that XCD's L2 for the others.  H's blocks are streamed once per row block (60 GB at configs[3]).
#pragma once
#include "smm_kernels.hpp"

namespace smm {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int TR_COLSTRIDE = 65;        // doubles between two columns of the LDS tile (64 rows + 1: the transposing
                                        // store of the staging pass then walks the banks instead of hitting one)

// blocks per (chunk, k): cnt[q*kpad + k] = ceil(len / 4) (0 for the padding rows k >= n)
__global__ __launch_bounds__(256) void smm_kell_count(int n, int kpad, int nchunks, const int *__restrict__ seg, int *__restrict__ cnt)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)nchunks * kpad) return;
    const int q = (int)(gid / kpad), k = (int)(gid % kpad);
    int len = 0;
    if (k < n) { const int *sp = seg + (size_t)k * (nchunks + 1) + q; len = sp[1] - sp[0]; }
    cnt[gid] = (len + 3) >> 2;
}

// payload: one thread per (chunk, k)
__global__ __launch_bounds__(256) void smm_kell_fill(int n, int kpad, int nchunks, int cw, const int *__restrict__ seg,
                                                     const int *__restrict__ h_idx, const double *__restrict__ h_val,
                                                     const int *__restrict__ blkptr, int *__restrict__ boff, double *__restrict__ bval)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)nchunks * kpad) return;
    const int q = (int)(gid / kpad), k = (int)(gid % kpad);
    if (k >= n) return;
    const int *sp = seg + (size_t)k * (nchunks + 1) + q;
    const int s = sp[0], len = sp[1] - s;
    const int64_t base = (int64_t)blkptr[gid] * 4;
    const int padded = ((len + 3) >> 2) << 2;
    const int zoff = cw * TR_COLSTRIDE * 8;
    for (int e = 0; e < padded; ++e) {
        int off = zoff; double v = 0.0;
        if (e < len) { off = (h_idx[s + e] - q * cw) * TR_COLSTRIDE * 8; v = h_val[s + e]; }
        boff[base + e] = off;
        bval[base + e] = v;
    }
}

struct RowLaneArgs {
    int n, K, cw, nchunks, kpad;
    int n_rb, n_kg, cpx;               // row blocks of 64, k-groups of KG, units per XCD
    int64_t row_begin, row_end;
    int full;
    const int *blkptr;                 // [nchunks*kpad + 1]
    const int *boff; const double *bval;   // blocks of 4 entries (+ one spare block at the end)
    int last_block;                    // index of the spare block (prefetches past the end land there)
    const double *T;                   // (row_end-row_begin) x K
    double *C; int64_t ldc;            // row row_begin at C
};

// NW waves; wave w owns the NKW rows k = kg*KG + w*NKW .. +NKW of its k-group (KG = NW*NKW); FMA: fused multiply-add
// (default mode; values to rounding) instead of the reference's separate multiply and add (SMM_EXACT).
template <int NW, int NKW, bool FMA>
__global__ __launch_bounds__(NW * 64) void smm_triple_rows(const RowLaneArgs A)
{
    extern __shared__ double tile[];                   // [cw + 1][TR_COLSTRIDE]; column cw = zeros
    constexpr int KG = NW * NKW;
    constexpr int RPP = NW / 4;                        // rows staged per pass: 4 waves cover 256 columns of one row
    constexpr int NV = 64 / RPP;                       // staging loads per thread and chunk
    static_assert(NW % 4 == 0 && 64 % RPP == 0, "staging geometry");
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t unit = (int64_t)(blockIdx.x & 7u) * A.cpx + (blockIdx.x >> 3);
    if (unit >= (int64_t)A.n_rb * A.n_kg) return;
    const int rb = (int)(unit / A.n_kg), kg = (int)(unit - (int64_t)rb * A.n_kg);
    const int64_t i0 = A.row_begin + (int64_t)rb * 64;
    const int nr = (A.row_end - i0) < 64 ? (int)(A.row_end - i0) : 64;
    const int64_t k0 = (int64_t)kg * KG + wave * NKW;  // this wave's first k
    double *crow = A.C + (i0 - A.row_begin + lane) * A.ldc;
    // the whole k-group lies left of the diagonal: the reference's calloc'd zeros
    if (!A.full && (int64_t)(kg + 1) * KG <= i0) {
        if (lane < nr)
            for (int kk = 0; kk < NKW; ++kk)
                if (k0 + kk < A.n) crow[k0 + kk] = 0.0;
        return;
    }
    double sum[NKW];
#pragma unroll
    for (int kk = 0; kk < NKW; ++kk) sum[kk] = 0.0;

    // staging: thread -> (row sr + RPP*t, column sc) of the 64 x cw tile, t = 0..NV-1; the next chunk's values
    // travel in registers while the current chunk is consumed
    const int sc = (wave & 3) * 64 + lane, sr = wave >> 2;
    const double *trow = A.T + (i0 - A.row_begin) * A.K;
    double v[NV];
    auto stage_load = [&](int q) {
        const int lo = q * A.cw;
        const int w = (A.K - lo) < A.cw ? (A.K - lo) : A.cw;
        const int c = sc < w ? sc : w - 1;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int r = sr + RPP * t;
            v[t] = trow[(int64_t)(r < nr ? r : nr - 1) * A.K + lo + c];
        }
    };
    if (threadIdx.x < TR_COLSTRIDE) tile[A.cw * TR_COLSTRIDE + threadIdx.x] = 0.0;     // the zero column
    stage_load(0);
    const unsigned lane8 = lds_addr(tile) + 8u * (unsigned)lane;
    // block pointers of this wave's k range for the chunk: lane kk holds blkptr[q*kpad + k0 + kk]
    const int *bpq = A.blkptr + k0 + (lane <= NKW ? lane : NKW);
    int bp_n = bpq[0];
    for (int q = 0; q < A.nchunks; ++q) {
        const int bp = bp_n;
        __syncthreads();                               // nobody reads the previous tile any more
        if (sc < A.cw) {
#pragma unroll
            for (int t = 0; t < NV; ++t) tile[sc * TR_COLSTRIDE + sr + RPP * t] = v[t];
        }
        const int qn = q + 1 < A.nchunks ? q + 1 : q;
        bp_n = bpq[(size_t)qn * A.kpad];
        __syncthreads();
        stage_load(qn);
        // The wave's blocks of this chunk are contiguous: [readlane(bp,0), readlane(bp,NKW)).  One asm statement
        // per block issues the scalar loads of the NEXT block (offsets + values: the compiler chose vector loads
        // for these uniform addresses, which ride vmcnt behind the 16 staging loads), the four column reads of
        // THIS block, and the single s_waitcnt that covers both -- nothing asynchronous leaves the statement, so
        // the compiler may copy its results as it likes (with separate statements it copied the SGPR tuples of a
        // scalar load before the wait).
        int p = rl(bp, 0);
        v4i o_n; v8i h_n;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(o_n), "=&s"(h_n) : "s"(A.boff + (size_t)4 * p), "s"(A.bval + (size_t)4 * p));
#pragma unroll
        for (int kk = 0; kk < NKW; ++kk) {
            const int pe = rl(bp, kk + 1);
            double s = sum[kk];
            while (p < pe) {
                const v4i o = o_n;
                const v8i h = h_n;
                ++p;
                const int pn = p < A.last_block ? p : A.last_block;
                double x0, x1, x2, x3;
                asm volatile("s_load_dwordx4 %0, %6, 0x0\n\t"
                             "s_load_dwordx8 %1, %7, 0x0\n\t"
                             "ds_read_b64 %2, %8\n\t"
                             "ds_read_b64 %3, %9\n\t"
                             "ds_read_b64 %4, %10\n\t"
                             "ds_read_b64 %5, %11\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&s"(o_n), "=&s"(h_n), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                             : "s"(A.boff + (size_t)4 * pn), "s"(A.bval + (size_t)4 * pn), "v"(lane8 + (unsigned)o.x),
                               "v"(lane8 + (unsigned)o.y), "v"(lane8 + (unsigned)o.z), "v"(lane8 + (unsigned)o.w));
                const double h0 = __hiloint2double(h.s1, h.s0), h1 = __hiloint2double(h.s3, h.s2);
                const double h2 = __hiloint2double(h.s5, h.s4), h3 = __hiloint2double(h.s7, h.s6);
                if (FMA) {
                    s = __builtin_fma(x0, h0, s); s = __builtin_fma(x1, h1, s);
                    s = __builtin_fma(x2, h2, s); s = __builtin_fma(x3, h3, s);
                } else {
                    s += x0 * h0; s += x1 * h1; s += x2 * h2; s += x3 * h3;
                }
            }
            sum[kk] = s;
        }
    }
    if (lane < nr) {
        const int64_t i = i0 + lane;
#pragma unroll
        for (int kk = 0; kk < NKW; ++kk)
            if (k0 + kk < A.n) crow[k0 + kk] = (A.full || k0 + kk >= i) ? sum[kk] : 0.0;
    }
}

}  // namespace smm
