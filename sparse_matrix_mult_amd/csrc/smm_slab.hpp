// smm_slab.hpp -- "row block x column slab" numeric kernels: the gather of B is served by the XCD's L2.
//
// Why.  Row-wise Gustavson (sparsework.cpp:56-76, sparse_sparse_dense.cpp:108-129) reads 10 bytes of B
// per product and never re-uses them inside a row.  smm_numeric (one row x a 16 667-column tile per
// workgroup) therefore fetches every product's bytes through the fabric: 32 rows in flight per XCD x
// d(A) = 0.01 share almost nothing, L2 hit rate 22 % and all of it line-pairing, 196 GB of fabric
// traffic for 30 GB of result (profiles/traffic_r1.json).  What an L2 can give is re-use ACROSS rows:
// row j of B is needed by d(A) x (rows in flight on the XCD) rows at about the same time.  LDS holds
// 20 000 f64 accumulators per CU whatever their shape, so the shape is chosen for that product:
// R rows x a NARROW column slab of ws columns per workgroup, and every workgroup resident on one XCD
// works on the SAME slab.  At ws ~ 600-1300 that is 500-900 rows per XCD sharing one slab of B
// (3-6 MB for 25 M entries): the slab is fetched from the fabric about once per round of row blocks
// and hit in L2 by everyone else.
//
// Layout.  B is re-laid once per slab width (cached on the operand handle): slab-major payload --
// scol[] (int16 slab-local column) + sval[] (f64), rows of B in order inside a slab -- and
// soff[t*rowsB + j] = position of the first entry of (slab t, row j); (t,j)'s entries end where the
// next pair's start.  A segment (row of B inside a slab) holds only ws*d(B) ~ 6-13 entries, so the walk
// is the flattened stream of smm_accumulate: the segments of 64 entries of A are concatenated and
// consumed 64 lanes at a time.
//
// Order of additions.  A wave owns RW whole rows of the block and nobody else adds into them; it
// walks A's entries in stored order and inside each the slab's piece of B's row in stored order, so
// every accumulator receives its products in exactly the reference's order (same two hardware facts
// as smm_accumulate: one wave's ds_add_f64 execute in issue order, same-address lanes of one
// instruction in lane order).  There is no separate "default" walk here: the exact one is the fast one.
//
// Output.  The accumulators are written in COLUMN order: straight to C for the dense routines
// (dense_nosym/_sym, stage 1 of triple_product), to a dense scratch for CSR output, which smm_numeric's
// epilogue (SRC_SCRATCH) then re-reads tile by tile to emit the reference's first-touch order.  A narrow
// slab cannot emit that order itself: step e's new columns inside one slab are ~2 entries -- 1e9 partial-
// line writes at 50k x 50k.
#pragma once
#include "smm_kernels.hpp"

namespace smm {

// cnt[t*rows + j] = entries of row j of B inside slab t, from the tile index seg[rows][n_slabs+1]
__global__ __launch_bounds__(256) void smm_slab_count(int rows, int n_slabs, const int *__restrict__ seg, int *__restrict__ cnt)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)rows * n_slabs) return;
    const int t = (int)(gid / rows), j = (int)(gid % rows);
    const int *sp = seg + (size_t)j * (n_slabs + 1) + t;
    cnt[gid] = sp[1] - sp[0];
}

__global__ __launch_bounds__(256) void smm_narrow32(int64_t n, const int64_t *__restrict__ in, int *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int)in[i];
}

// payload: one wave per row of B, entries read coalesced, written to their slab's block
__global__ __launch_bounds__(256) void smm_slab_fill(int rows, int n_slabs, int ws, const int *__restrict__ ptr,
                                                     const int *__restrict__ idx, const double *__restrict__ val,
                                                     const int *__restrict__ seg, const int *__restrict__ soff,
                                                     short *__restrict__ scol, double *__restrict__ sval)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int j = blockIdx.x * wpb + (threadIdx.x >> 6); j < rows; j += gridDim.x * wpb) {
        const int *sp = seg + (size_t)j * (n_slabs + 1);
        for (int k = ptr[j] + lane; k < ptr[j + 1]; k += WAVE) {
            const int c = idx[k];
            const int t = c / ws;
            const int dst = soff[(size_t)t * rows + j] + (k - sp[t]);
            scol[dst] = (short)(c - t * ws);
            sval[dst] = val[k];
        }
    }
}

struct SlabArgs {
    int m;                          // rows handled by this launch (length of rowlist, or rows of A)
    int ncols, ws, n_slabs, n_rb, rowsB;
    int cpx;                        // units per XCD = ceil(n_rb*n_slabs / 8)
    int kmax;                       // last valid position of scol / sval (lanes past a stream's end read it)
    int64_t row_offset;
    const int *rowlist;             // NULL = rows 0..m-1
    const int *a_ptr, *a_idx; const double *a_val;
    const int *soff;                // [n_slabs*rowsB + 1]
    const short *scol; const double *sval;
    const int *dummy_idx; const double *dummy_val;
    double *out; int64_t ldo;       // row number `ridx` of this launch starts at out + ridx*ldo
};

constexpr int SLAB_UNROLL = 8;      // chunk loads in flight per wave
struct SlabScratch {                // per wave, in LDS behind the accumulators
    int4 tab[WAVE];                 // per entry of the batch: .x = segment start - stream start, .y = row << 16 | threshold, .zw = a
    unsigned char head[WAVE * SLAB_UNROLL];   // one round of the stream: entry index + 1 where a segment starts
    int4 rows[4];                   // the wave's rows
};
struct SlabShared { double sink[WAVE]; };   // per workgroup: lanes with nothing to add put -0.0 here (branch-free adds)

// Block index -> unit -> (slab, row block).  Units are numbered slab-major (u = slab*n_rb + row block).
// Workgroups are dealt round-robin over the 8 XCDs (observed, speed only: blocks b and b+8 share an
// L2), so block b takes unit (b % 8)*cpx + b / 8: every XCD walks ONE contiguous eighth of the unit
// list in order -- at any time the workgroups resident on it sit on one slab (two at a slab change),
// whose share of B stays in that L2 -- and all eight XCDs stream through A's row blocks together.
template <bool SYM, int NW, int RW, bool NEGZERO>
__global__ __launch_bounds__(NW * 64) void smm_dense_slab(const SlabArgs A)
{
    extern __shared__ double acc[];
    constexpr int R = NW * RW, NT = NW * 64;
    static_assert(RW >= 1 && RW <= 4 && R <= 64, "rows per wave / per block");
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t unit = (int64_t)(blockIdx.x & 7u) * A.cpx + (blockIdx.x >> 3);
    if (unit >= (int64_t)A.n_rb * A.n_slabs) return;
    const int t = (int)(unit / A.n_rb);
    const int rb = (int)(unit - (int64_t)t * A.n_rb);
    const int lo_c = t * A.ws;
    const int w = (A.ncols - lo_c) < A.ws ? (A.ncols - lo_c) : A.ws;
    const int wsp = (A.ws + 1) & ~1;
    const double zero = NEGZERO ? -0.0 : 0.0;
    SlabScratch *sc = (SlabScratch *)(acc + (size_t)R * wsp) + wave;
    double *sink = ((SlabShared *)((SlabScratch *)(acc + (size_t)R * wsp) + NW))->sink;

    // this wave's rows: numbers rb*R + wave*RW + i of the launch; their entries of A form one stream.
    // Row i's (first entry of A, first stream position, column threshold) sit in a 4-entry LDS table of the
    // wave: kept in scalar registers they pushed the kernel past its SGPR budget (spills to scratch memory).
    if (lane < 4) {
        const int i = lane;
        const int ridx = rb * R + wave * RW + i;
        int s = 0, e = 0, th = 0;
        if (i < RW && ridx < A.m) {
            const int row = A.rowlist ? A.rowlist[ridx] : ridx;
            s = A.a_ptr[row]; e = A.a_ptr[row + 1];
            if (SYM) {                                  // keep col >= i (sparsework.cpp:217): slab-local threshold
                const int64_t d = row + A.row_offset - lo_c;
                th = d < 0 ? 0 : (d > 32767 ? 32767 : (int)d);
                if (d >= w) e = s;                      // the whole slab lies left of the diagonal
            }
        }
        int incl = e - s;                               // inclusive scan over the 4 rows
        int y = __shfl_up(incl, 1); if (lane >= 1) incl += y;
        y = __shfl_up(incl, 2); if (lane >= 2) incl += y;
        sc->rows[i] = make_int4(s, incl - (e - s), th, incl);
    }
    wave_sync();
    // Touch every line of this wave's rows of A now: they come through the fabric (A is streamed once per
    // slab), and the walk below must wait for whatever it requested BEFORE its chunk loads (vmcnt retires in
    // order).  Requested here, their latency passes while the tile is cleared, and the walk's own loads of A
    // are L2 hits like everything else it touches.
    // (Plain loads, clamped instead of predicated, folded into one value that is "used" behind the barrier:
    // the compiler issues them back to back and waits once.  The first 2048 entries of a row are touched;
    // a longer row's tail simply is not pre-fetched.)
    int warm = 0;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int4 ri = sc->rows[i];
        const int cnt = ri.w - ri.y;
        const char *pi = (const char *)(A.a_idx + ri.x), *pv = (const char *)(A.a_val + ri.x);
        const int o = lane * 128;
        auto clampo = [](int off, int last) { off = off < last ? off : last; return off > 0 ? off : 0; };
        warm ^= *(const int *)(pi + clampo(o, cnt * 4 - 4));
        warm ^= *(const int *)(pv + clampo(o, cnt * 8 - 8));
        warm ^= *(const int *)(pv + clampo(o + WAVE * 128, cnt * 8 - 8));
    }
    for (int x = threadIdx.x; x < R * wsp; x += NT) acc[x] = zero;
    if (threadIdx.x < WAVE) sink[threadIdx.x] = -0.0;
    const int E = sc->rows[3].w;
    __syncthreads();
    asm volatile("" ::"v"(warm));                           // the touches are "used" here: one wait, behind the clear
    const unsigned acc_a = lds_addr(acc), sink_a = lds_addr(sink) + 8u * (unsigned)lane;

    if (E > 0) {
        const int *__restrict__ soff = A.soff + (size_t)t * A.rowsB;
        const short *__restrict__ bi = A.scol;
        const double *__restrict__ bv = A.sval;
        const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= this one

        // entry number x of the stream: which row (i), which entry of A
        auto load_a = [&](int xb, int &r, double &av, int &pk) {
            int x = xb + lane;
            x = x < E ? x : E - 1;
            // row of the entry (.x first entry of A, .y first stream position, .z threshold)
            int4 ri = sc->rows[0];
            int i = 0;
            if constexpr (RW > 1) { const int4 q = sc->rows[1]; if (x >= q.y) { i = 1; ri = q; } }
            if constexpr (RW > 2) { const int4 q = sc->rows[2]; if (x >= q.y) { i = 2; ri = q; } }
            if constexpr (RW > 3) { const int4 q = sc->rows[3]; if (x >= q.y) { i = 3; ri = q; } }
            const int base = ri.x, off = ri.y, th = ri.z;
            const int e = base + (x - off);
            r = A.a_idx[e];
            av = A.a_val[e];
            pk = ((wave * RW + i) << 16) | th;
        };
        auto load_seg = [&](int r, int &s, int &en) { s = soff[r]; en = soff[r + 1]; };

        int r_c, r_n, s_c, e_c, pk_c, pk_n;
        double a_c, a_n;
        load_a(0, r_c, a_c, pk_c);
        load_a(WAVE, r_n, a_n, pk_n);
        load_seg(r_c, s_c, e_c);
        for (int xb = 0; xb < E; xb += WAVE) {
            const int rem = E - xb;
            const int nb = rem < WAVE ? rem : WAVE;
            int s_n, e_n, r_nn, pk_nn;
            double a_nn;
            load_seg(r_n, s_n, e_n);                       // segments of the NEXT 64 entries
            load_a(xb + 2 * WAVE, r_nn, a_nn, pk_nn);      // entries of A two batches ahead
            // stream of this batch: entry j owns positions [first_j, first_j + len_j)
            const int len = lane < nb ? e_c - s_c : 0;
            const int incl = wave_scan_incl(len);
            const int first = incl - len;
            const int total = rl(incl, WAVE - 1);
            wave_sync();
            sc->tab[lane] = make_int4(s_c - first, pk_c, __double2loint(a_c), __double2hiint(a_c));
            wave_sync();
            for (int g0 = 0; g0 < total; g0 += WAVE * SLAB_UNROLL) {
                int c[SLAB_UNROLL], pk[SLAB_UNROLL], kk[SLAB_UNROLL];
                double v[SLAB_UNROLL], a[SLAB_UNROLL];
                // heads of this round: entries whose segment starts inside it
                wave_sync();
                ((unsigned long long *)sc->head)[lane] = 0ull;
                wave_sync();
                if (len > 0 && first >= g0 && first < g0 + WAVE * SLAB_UNROLL) sc->head[first - g0] = (unsigned char)(lane + 1);
                wave_sync();
#pragma unroll
                for (int u = 0; u < SLAB_UNROLL; ++u) {     // owner of every stream position of the round
                    const int gbase = g0 + u * WAVE;        // wave-uniform
                    const int hd = sc->head[u * WAVE + lane];
                    const unsigned long long heads = __ballot(hd != 0);
                    // entry that owns the chunk's first position when no head precedes a lane
                    const int carry = (int)__popcll(__ballot(incl <= gbase));
                    const unsigned long long mine = heads & le_mask;
                    const int hl = 63 - __clzll((long long)(mine | 1ull));          // nearest head at or below
                    int j = __shfl(hd, hl) - 1;
                    if (mine == 0ull) j = carry;
                    j = j < WAVE ? j : WAVE - 1;
                    const int4 tb = sc->tab[j];
                    const int k = tb.x + gbase + lane;      // positions past the end read the payload's last entry
                    kk[u] = k < A.kmax ? k : A.kmax;
                    pk[u] = tb.y;
                    a[u] = __hiloint2double(tb.w, tb.z);
                }
#pragma unroll
                for (int u = 0; u < SLAB_UNROLL; ++u) {     // all loads of the round, back to back ...
                    gload_sshort(c[u], bi + kk[u]);
                    gload_f64(v[u], bv + kk[u]);
                }
                // ... then add, lane order = stream order.  Branch-free: a lane with nothing to add (past the
                // end of the stream, or left of the diagonal) adds -0.0 -- neutral for every accumulator value --
                // into its slot of the sink.  (With `if (keep) add` the compiler sank the value load of the
                // round's first chunk into the branch: a second, dependent round trip per round.)
#pragma unroll
                for (int u = 0; u < SLAB_UNROLL; ++u) {
                    wait_vm_pair(c[u], v[u], 2 * (SLAB_UNROLL - 1 - u));
                    const bool keep = (g0 + u * WAVE + lane) < total && c[u] >= (pk[u] & 0xffff);
                    const double prod = a[u] * v[u];
                    const unsigned at = acc_a + 8u * (unsigned)((pk[u] >> 16) * wsp + c[u]);
                    lds_add_asm(keep ? at : sink_a, keep ? prod : -0.0);
                }
            }
            s_c = s_n; e_c = e_n; a_c = a_n; pk_c = pk_n;
            r_n = r_nn; a_n = a_nn; pk_n = pk_nn;
        }
    }
    wait_lgkm0();                                           // the hand-issued ds_add's (the compiler does not count them)
    __syncthreads();

    // the block's R x w tile, in column order (each wave writes whole rows: 512 contiguous bytes per instruction)
    for (int r = wave; r < R; r += NW) {
        const int ridx = rb * R + r;
        if (ridx >= A.m) break;
        double *__restrict__ dst = A.out + (int64_t)ridx * A.ldo + lo_c;
        const double *src = acc + (size_t)r * wsp;
        for (int x = lane; x < w; x += WAVE) dst[x] = src[x];
    }
}

}  // namespace smm
