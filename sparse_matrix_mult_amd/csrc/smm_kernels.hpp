// smm_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the CSR x CSR engine.
//
// Path restated (reference file:line under /root/reference):
//   sparsework_nosym / _sym   src/sparsework.cpp:56-129 / :201-280   -> smm_symbolic + smm_numeric
//   sparse_nosym / _sym       src/sparse_sparse_sparse.cpp:269-291   -> smm_scan (rows land in place; no stitch copy)
//   dense_nosym / _sym        src/sparse_sparse_dense.cpp:108-129 / :40-73 -> smm_numeric<OUT_DENSE>
//   triple_product            src/sparse_sparse_dense.cpp:185-220    -> smm_numeric<OUT_DENSE> + smm_triple_stage2
//
// Design notes (DESIGN.md has the long form):
//  * A row of C is accumulated in LDS as dense f64 tiles of <= lds_cols columns ("coarse
//    tile").  Default mode: all waves of the workgroup add into the tile concurrently (LDS atomics;
//    values to rounding).  SMM_EXACT: every wave owns a contiguous "fine tile" of it and walks the
//    row's A entries in stored order, so each accumulator receives its products in exactly the
//    reference's order (one wave's LDS atomics execute in issue order): values are bit-identical
//    to the CPU loop.  Rows with few nonzeros take LDS hash kernels instead of tiles; rows with at most 16
//    products go four to a wave with no marker at all (smm_symbolic_tiny / smm_numeric_tiny, round 4).
//  * The first-touch column order of the reference is produced by smm_symbolic:
//    one wave per row, a marker of B's columns in LDS (bitmap, or a hash set for rows with few
//    products on wide matrices), test-and-set + ballot/mbcnt ordered compaction.  The numeric
//    kernel then emits indices/values in that order straight from the tile it has just
//    accumulated, as contiguous sub-runs (smm_runs).  Sorted operands take the walk over a chunk-padded
//    16-bit column stream (smm_symbolic_ccs), with a second instantiation for operands with dense runs
//    of columns (plain reads + merged ORs instead of same-word returning atomics).
//  * No MFMA anywhere: this is an indexing / HBM path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace smm {

constexpr int WAVE = 64;
enum { OUT_SPARSE = 0, OUT_DENSE = 1 };

// flag bits written by smm_validate
enum { CSR_BAD = 1, CSR_UNSORTED = 2, CSR_HAS_EQUAL = 4 };

// Plan metadata is produced by one kernel and trusted by the next (capacities -> lists -> P -> sub-run tables ->
// epilogue).  An inconsistency there must come back to the caller as SMM_ERR_INTERNAL, never as a hang or a fault
// (the reference's convention: message + early return, sparsework.cpp:33-36, sparse_sparse_sparse.cpp:257-262).
// Two layers: (1) always on -- every consumer clamps what it reads (unsigned differences, positions against the row
// length) so that a wrong table can only produce a wrong row, and records the fact in the context's error word;
// (2) smm_plan_check_* below verify the whole plan (SMM_CHECK=1 / smm_ctx_set_check, and every test).
// err[0] = OR of these bits, err[1] = lowest row (local index of A) that tripped one.
enum {
    PLAN_ERR_RUNS = 1,      // sub-run table: r1 < r0, or beyond the row
    PLAN_ERR_TAIL = 2,      // tail descriptor outside the row's steps / positions
    PLAN_ERR_RUNS2 = 4,     // slab table: source / length / destination outside the list or the row
    PLAN_ERR_P = 8,         // start slots not monotone, or beyond the list length
    PLAN_ERR_HASH = 16,     // a column of the product is not in the row's list (hash kernels)
    PLAN_ERR_CAP = 32,      // negative / overlapping list capacity, or a list longer than its capacity
    PLAN_ERR_COUNT = 64,    // row counts do not add up to the row pointer
    PLAN_ERR_LIST = 128,    // a list entry is not a column of the (slab of the) result, or not in its sub-run's tile
};
#ifndef SMM_CLAMPS
#define SMM_CLAMPS 1       // 0: a diagnostic build without the always-on clamps (A/B of their cost; never shipped)
#endif
__device__ __forceinline__ void plan_err(unsigned *err, unsigned bit, int row) {
    atomicOr(err, bit);
    atomicMin(err + 1, (unsigned)row);
}

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
// Lanes of ONE wave that exchange data through LDS need no hardware barrier (the LDS executes a
// wave's instructions in order), but the COMPILER must be told that other lanes may have
// written: without a fence it optimises per thread (it once kept a lane's own `= 0` instead of
// re-reading a word another lane had overwritten).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Load idx[lane] into v for the lanes of `mask` (uniform), whose other lanes keep their value: EXEC is
// narrowed around one global_load with a scalar base and the constant per-lane offset 4*lane, so the
// address costs no vector instruction and idle lanes issue no request.  Must be called with all 64
// lanes active (EXEC is restored to -1).  The compiler does not see the load: the caller waits
// with wait_vm() before it reads v.  (An `if (lane < cnt)` around an ordinary load made hipcc
// branch around it and wait for vmcnt(0) at every join; a select between the real and a dummy
// address cost 8 vector instructions per load.)
__device__ __forceinline__ void load_masked(int &v, const int *idx, unsigned long long mask, int lane4) {
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "global_load_dword %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(v)
                 : "v"(lane4), "s"(idx), "s"(mask));
}
// Inclusive prefix sum over the 64 lanes with DPP (no LDS traffic; __shfl_up is a ds_bpermute -- one LDS
// round trip per step): shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then the last lane of row 0 / 2
// into row 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31) -- gfx9 wave64 forms.
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
// EXEC convention of every hand-issued block in this file: the block narrows EXEC and restores it to -1, i.e. it must
// be reached with ALL 64 lanes active.  That holds by construction: every kernel is launched with a block size that
// is a multiple of 64 (__launch_bounds__ + the host's launch code), and the blocks sit in wave-uniform control flow
// (loop bounds and conditions are readfirstlane'd / ballot results), never under a per-lane branch.
// Hand-issued memory operations for loops whose order of issue matters more than the compiler's view of
// them (it sinks loads into the branch of their first use, or turns selects back into branches).  The
// compiler does not know these are in flight: a value loaded by gload_* must be defined ONCE and read only behind
// wait_vm_pair() (a re-defined destination frees its register while the first load is still in flight), and asm ds_add's are drained with wait_lgkm0() before a barrier.  Its own s_waitcnt's
// stay correct -- counters retire in order, and it can only under-estimate what is outstanding.
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ void gload_sshort(int &dst, const short *p) {
    asm volatile("global_load_sshort %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload_f64(double &dst, const double *p) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void lds_add_asm(unsigned addr, double x) {
    asm volatile("ds_add_f64 %0, %1" ::"v"(addr), "v"(x) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm_pair(int &c, double &v, int n) {
#define SMM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(c), "+v"(v) : : "memory"); break;
    switch (n) {
        SMM_W(0) SMM_W(1) SMM_W(2) SMM_W(3) SMM_W(4) SMM_W(5) SMM_W(6) SMM_W(7)
        SMM_W(8) SMM_W(9) SMM_W(10) SMM_W(11) SMM_W(12) SMM_W(13) SMM_W(14) SMM_W(15)
        SMM_W(16) SMM_W(17) SMM_W(18) SMM_W(19) SMM_W(20) SMM_W(21) SMM_W(22) SMM_W(23)
        SMM_W(24) SMM_W(25) SMM_W(26) SMM_W(27) SMM_W(28) SMM_W(29) SMM_W(30) SMM_W(31)
    }
#undef SMM_W
}
// the same for a 16-bit column stream (zero-extended): per-lane offset 2*lane
__device__ __forceinline__ void load_masked(int &v, const unsigned short *idx, unsigned long long mask, int lane4) {
    const int lane2 = lane4 >> 1;
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "global_load_ushort %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(v)
                 : "v"(lane2), "s"(idx), "s"(mask));
}
// two neighbouring 16-bit columns per lane (one dword at a 2-byte aligned address): per-lane offset 4*lane
__device__ __forceinline__ void load_masked_pair(int &v, const unsigned short *idx, unsigned long long mask, int lane4) {
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "global_load_dword %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(v)
                 : "v"(lane4), "s"(idx), "s"(mask));
}
// two neighbouring 32-bit columns per lane (one dwordx2): per-lane offset 8*lane
__device__ __forceinline__ void load_masked_pair(long long &v, const int *idx, unsigned long long mask, int lane8) {
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "global_load_dwordx2 %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(v)
                 : "v"(lane8), "s"(idx), "s"(mask));
}
__device__ __forceinline__ void wait_vm(long long &v, int n) {
#define SMM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(v)); break;
    switch (n) {
        SMM_W(0) SMM_W(1) SMM_W(2) SMM_W(3) SMM_W(4) SMM_W(5) SMM_W(6) SMM_W(7)
        SMM_W(8) SMM_W(9) SMM_W(10) SMM_W(11) SMM_W(12) SMM_W(13) SMM_W(14) SMM_W(15)
        SMM_W(16) SMM_W(17) SMM_W(18) SMM_W(19) SMM_W(20) SMM_W(21) SMM_W(22) SMM_W(23)
        SMM_W(24) SMM_W(25) SMM_W(26) SMM_W(27) SMM_W(28) SMM_W(29) SMM_W(30) SMM_W(31)
    }
#undef SMM_W
}
// n is a constant after unrolling; the switch folds to one s_waitcnt
__device__ __forceinline__ void wait_vm(int &v, int n) {
#define SMM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(v)); break;
    switch (n) {
        SMM_W(0) SMM_W(1) SMM_W(2) SMM_W(3) SMM_W(4) SMM_W(5) SMM_W(6) SMM_W(7)
        SMM_W(8) SMM_W(9) SMM_W(10) SMM_W(11) SMM_W(12) SMM_W(13) SMM_W(14) SMM_W(15)
        SMM_W(16) SMM_W(17) SMM_W(18) SMM_W(19) SMM_W(20) SMM_W(21) SMM_W(22) SMM_W(23)
        SMM_W(24) SMM_W(25) SMM_W(26) SMM_W(27) SMM_W(28) SMM_W(29) SMM_W(30) SMM_W(31)
    }
#undef SMM_W
}
template <typename T, int N>
__device__ __forceinline__ void wait_vm_all(T (&v)[N]) {
#pragma unroll
    for (int u = 0; u < N; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[u]));
}

__device__ __forceinline__ void lds_add(double *p, double x) {
    // ds_add_f64 (no return): fire-and-forget, executed by the LDS in issue order
    (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double *p, double x) {
    (void)__hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------
// Run-time guard of SMM_EXACT (smm_ctx_exact_selftest).  The exact walk relies on two things: one wave's
// ds_add_f64 instructions execute in issue order (the LDS is in order per wave), and inside ONE ds_add_f64 the
// lanes that hit the same address are applied in ASCENDING LANE ORDER -- measured on gfx950
// (scripts/ubench/lds_order.hip), not promised by the ISA.  Every trial throws the 64 lanes of a wave onto 1..32
// accumulators at random (some lanes idle), twice in a row, with values spanning 60 binary orders of magnitude,
// and compares each accumulator bit for bit with the sum taken lane by lane, instruction by instruction
// (reference order: src/sparsework.cpp:59-76).  `descending` makes the expectation the reverse lane order:
// the injected fault that exercises the failure path.  bad[0] counts the accumulators that differ.
__device__ __forceinline__ unsigned st_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ double st_value(unsigned h) {
    const double u = (double)(h & 0xfffffu) / 1048576.0 - 0.5;
    return ldexp(u, (int)((h >> 20) % 60u) - 30);
}
__global__ __launch_bounds__(64) void smm_lds_order_selftest(int ntrial, int descending, unsigned *__restrict__ bad)
{
    __shared__ double acc[32];
    __shared__ double vals[2][WAVE];
    __shared__ int addrs[2][WAVE];
    const int lane = lane_id();
    unsigned nbad = 0;
    for (int t = blockIdx.x; t < ntrial; t += gridDim.x) {
        const unsigned ht = st_hash(0x9e3779b9u * (unsigned)(t + 1));
        const int spread = 1 + (int)(ht % 32u);
        const double init = st_value(st_hash(ht ^ (0x1000u + (unsigned)lane)));
        if (lane < 32) acc[lane] = init;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned h = st_hash(ht + 0x85ebca6bu * (unsigned)(2 * lane + k + 1));
            addrs[k][lane] = (h % 8u == 0u) ? -1 : (int)((h >> 3) % (unsigned)spread);
            vals[k][lane] = st_value(st_hash(h ^ 0xc2b2ae35u));
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 2; ++k)                       // the instruction under test, as the exact walk issues it
            lds_add_asm(lds_addr(acc) + 8u * (unsigned)(addrs[k][lane] >= 0 ? addrs[k][lane] : 0),
                        addrs[k][lane] >= 0 ? vals[k][lane] : -0.0);
        wait_lgkm0();
        wave_sync();
        if (lane < 32) {
            double w = init;
            for (int k = 0; k < 2; ++k)
                for (int l = 0; l < WAVE; ++l) {
                    const int src = descending ? WAVE - 1 - l : l;
                    if (addrs[k][src] == lane) w += vals[k][src];
                    else if (addrs[k][src] < 0 && lane == 0) w += -0.0;      // idle lanes add -0.0 to slot 0
                }
            if (__double_as_longlong(w) != __double_as_longlong(acc[lane])) ++nbad;
        }
        wave_sync();
    }
    if (nbad) atomicAdd(bad, nbad);
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
// Tile-local column of every entry of a sorted operand: loc[k] = idx[k] mod wc as int16 (a
// coarse tile is at most 20 000 columns wide).  The numeric walks read these 2 bytes instead
// of the 4-byte global column: 10 instead of 12 bytes per product on the dominant gather.
__global__ __launch_bounds__(256) void smm_loc16(int nnz, int wc, const int *__restrict__ idx,
                                                 short *__restrict__ loc)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += gridDim.x * blockDim.x)
        loc[k] = (short)(idx[k] % wc);
}

// Packed tile-major payload of a sorted operand for the shared-tile walk: the piece of row j inside coarse
// tile t is ONE contiguous block -- its values (f64) followed by its tile-local columns (int16, padded to 8
// bytes) -- and the blocks of a tile follow each other in row order.  desc[t*rows + j] = {first 8-byte unit,
// entries}.  Against the separate loc16[] / val[] arrays in CSR order a piece touches two partial cache lines
// instead of four (-12 GB of fabric reads per 50k x 50k product), and its bounds are one 8-byte load.
// Pieces of the packed payload start on 128-byte lines (16 units of 8 bytes): a piece of ~1.7 KB then touches 14 lines instead
// of 14.1 + 1 (round 3, interleaved A/B on one box: configs[1] smm_numeric 25.67 -> 25.43 ms, configs[2] 20.22 -> 19.87,
// configs[4] share 59.75 -> 58.28; stage 1 of configs[3], 1 KB pieces, 18.40 -> 18.86; 64-byte alignment: no gain).
#ifndef SMM_PACK_ALIGN
#define SMM_PACK_ALIGN 16
#endif
__global__ __launch_bounds__(256) void smm_pack_count(int rows, int nct, const int *__restrict__ seg, int *__restrict__ units,
                                                      int *__restrict__ maxlen)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (gid < (int64_t)rows * nct) {
        const int t = (int)(gid / rows), j = (int)(gid % rows);
        const int *sp = seg + (size_t)j * (nct + 1) + t;
        len = sp[1] - sp[0];
        units[gid] = (len + ((len + 3) >> 2) + (SMM_PACK_ALIGN - 1)) & ~(SMM_PACK_ALIGN - 1);   // pieces start on SMM_PACK_ALIGN x 8 bytes (>= 16: value pairs are loaded)
    }
    int mx = len;                                          // the longest piece decides which accumulate walk can run
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    if (lane_id() == 0 && mx > 0) atomicMax(maxlen, mx);
}
__global__ __launch_bounds__(256) void smm_pack_desc(int rows, int nct, const int *__restrict__ seg, const int64_t *__restrict__ off,
                                                     int2 *__restrict__ desc)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)rows * nct) return;
    const int t = (int)(gid / rows), j = (int)(gid % rows);
    const int *sp = seg + (size_t)j * (nct + 1) + t;
    desc[gid] = make_int2((int)off[gid], sp[1] - sp[0]);
}
// one wave per row of the operand
__global__ __launch_bounds__(256) void smm_pack_fill(int rows, int nct, int wc, const int *__restrict__ ptr, const int *__restrict__ idx,
                                                     const double *__restrict__ val, const int *__restrict__ seg,
                                                     const int2 *__restrict__ desc, double *__restrict__ pay)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int j = blockIdx.x * wpb + (threadIdx.x >> 6); j < rows; j += gridDim.x * wpb) {
        const int *sp = seg + (size_t)j * (nct + 1);
        for (int k = ptr[j] + lane; k < ptr[j + 1]; k += WAVE) {
            const int c = idx[k];
            const int t = c / wc;
            const int2 d = desc[(size_t)t * rows + j];
            const int pos = k - sp[t];
            pay[d.x + pos] = val[k];
            short *cols = (short *)(pay + d.x + d.y);
            cols[pos] = (short)(c - t * wc);
            if (pos == d.y - 1)                            // the rest of the last 8-byte unit of columns: the sink accumulator
                for (int q = d.y; q < ((d.y + 3) & ~3); ++q) cols[q] = (short)((wc + 1) & ~1);
        }
    }
}

// 16-bit copy of an operand's column indices (operands with < 65 535 columns): the symbolic phase gathers
// these 2 bytes per product instead of 4 -- half of its fabric traffic.
__global__ __launch_bounds__(256) void smm_idx16(int nnz, const int *__restrict__ idx, unsigned short *__restrict__ out)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += gridDim.x * blockDim.x) out[k] = (unsigned short)idx[k];
}

// The ordered lists of smm_symbolic hold int32 columns, or uint16 when B has < 65 535 columns (l16): they are
// written once and read twice (smm_runs, the numeric epilogue) -- 10 GB each way at 50k x 50k as int32.
__device__ __forceinline__ int list_at(const void *list, int64_t i, bool l16) {
    return l16 ? (int)((const unsigned short *)list)[i] : ((const int *)list)[i];
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

// The same for operands whose rows are short on average (round 4: 30 M rows x 2 entries spent 5.7 ms in the kernel above,
// one row per wave): one row per LANE.  A row with more than ROW_WORK_SHORT_MAX entries (a hub of a power-law operand
// would keep its one lane busy for 1e6 dependent iterations) is put on a list instead and gets a wave of its own in
// smm_row_work_listed.
constexpr int ROW_WORK_SHORT_MAX = 32;
__global__ __launch_bounds__(256) void smm_row_work_short(int m, int ncols, int64_t row_offset, int sym,
                                                          const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                                          const int *__restrict__ b_ptr, int64_t *__restrict__ products,
                                                          int64_t *__restrict__ ub, int *__restrict__ long_rows,
                                                          int *__restrict__ n_long)
{
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < m; row += gridDim.x * blockDim.x) {
        int64_t s = 0;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        if (a1 - a0 > ROW_WORK_SHORT_MAX) { long_rows[atomicAdd(n_long, 1)] = row; continue; }
        for (int e = a0; e < a1; ++e) {
            const int r = a_idx[e];
            s += b_ptr[r + 1] - b_ptr[r];
        }
        products[row] = s;
        int64_t cap = ncols;
        if (sym) { const int64_t gi = row + row_offset; cap = gi < ncols ? ncols - gi : 0; }
        if (ub) ub[row] = s < cap ? s : cap;
    }
}
__global__ __launch_bounds__(256) void smm_row_work_listed(const int *__restrict__ n_rows, const int *__restrict__ rows, int ncols,
                                                           int64_t row_offset, int sym, const int *__restrict__ a_ptr,
                                                           const int *__restrict__ a_idx, const int *__restrict__ b_ptr,
                                                           int64_t *__restrict__ products, int64_t *__restrict__ ub)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    const int n = *n_rows;
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < n; ri += gridDim.x * wpb) {
        const int row = rows[ri];
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

// The same scan for long arrays (1e6 rows: the single workgroup above takes 1.5 ms), three launches:
// sums of 4096-element tiles, the scan above over those sums, then every tile scans itself
// on top of its offset.  tile_off has tiles+1 entries (the last one is the total).
constexpr int SCAN_TILE = 4096;
template <typename T>
__global__ __launch_bounds__(1024) void smm_scan_tile_sums(int n, const T *__restrict__ in, int64_t *__restrict__ sums)
{
    __shared__ int64_t wsum[16];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int base = blockIdx.x * SCAN_TILE;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_TILE / 1024; ++k) {
        const int i = base + k * 1024 + threadIdx.x;
        if (i < n) s += (int64_t)in[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        sums[blockIdx.x] = t;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void smm_scan_tiles(int n, const T *__restrict__ in, const int64_t *__restrict__ tile_off,
                                                       int64_t *__restrict__ out)
{
    __shared__ int64_t wsum[16];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int base = blockIdx.x * SCAN_TILE;
    int64_t carry = tile_off[blockIdx.x];
    for (int k = 0; k < SCAN_TILE / 1024; ++k) {
        const int i = base + k * 1024 + threadIdx.x;
        const int64_t v = i < n ? (int64_t)in[i] : 0;
        int64_t x = v;                                   // inclusive wave scan
        for (int o = 1; o < WAVE; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int64_t woff = 0, all = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; all += wsum[w]; }
        if (i < n) out[i] = carry + woff + x - v;
        carry += all;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = tile_off[gridDim.x];
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
// The walk is over 64-lane CHUNKS of B's rows.  The kernel is bound by instruction issue (one
// scalar and one vector instruction per cycle and CU), not by bytes, so the per-chunk work is kept
// minimal: for the rows of 64 A entries the chunk list is built with vector code (scan of the
// chunk counts, one descriptor per lane: start offset + lane mask), 64 chunks at a time; the hot
// loop reads a descriptor with three readlanes, issues SYM_UNROLL EXEC-masked loads together (idle
// lanes keep the guard column and issue no request), then all test-and-sets, then consumes them
// strictly in order.  The B-row pointers of the next 64 entries and the A indices of the 64 after
// those are prefetched.
// SYM_UNROLL = chunk loads in flight per wave: 16 when many waves fit a CU (8 and 32 measured slower at
// 50 000 columns, 24 waves); 32 when the bitmap leaves few waves per CU and a round is pure latency.
enum { MARK_LDS_BITMAP = 0, MARK_GLOBAL_BITMAP = 1, MARK_LDS_HASH = 2 };
// I16: B's columns are read from the 16-bit copy (smm_idx16) and the lists are written as uint16.
// WIDE (bitmap marker, no repeated columns): chunks of 128 entries, a lane loads two neighbouring columns with
// one dword / dwordx2 load -- the per-chunk work (descriptor, EXEC set-up, load, wait) is spent once per 128
// products instead of once per 64.  With 16-bit columns the idle column must fit 16 bits (columns <= 65 504).
template <bool SYM, bool SAFE, int MARK, int SYM_UNROLL, bool I16 = false, bool WIDE = false>
__global__ __launch_bounds__(256) void smm_symbolic(int m, const int *__restrict__ rowlist,
                                                    const int *__restrict__ nrows_p, int64_t row_offset, int bm_words,
                                                    const int *__restrict__ a_ptr,
                                                    const int *__restrict__ a_idx,
                                                    const int *__restrict__ b_ptr,
                                                    const void *__restrict__ b_idx_v,
                                                    const int64_t *__restrict__ ub_off,
                                                    void *__restrict__ tmp_idx_v,
                                                    unsigned *__restrict__ P,
                                                    int *__restrict__ rowcnt,
                                                    unsigned *__restrict__ gbitmap,
                                                    int *__restrict__ row_counter, int rbatch)
{
    extern __shared__ unsigned lds_bm[];
    using IT = std::conditional_t<I16, unsigned short, int>;
    const IT *__restrict__ b_idx = (const IT *)b_idx_v;
    IT *__restrict__ tmp_idx = (IT *)tmp_idx_v;
    constexpr bool HASH = MARK == MARK_LDS_HASH;
    static_assert(!(HASH && SAFE), "rows of B with repeated columns take the bitmap kernels");
    static_assert(!WIDE || (!SAFE && !HASH), "wide chunks: bitmap marker, no repeated columns");
    constexpr int CSH = WIDE ? 7 : 6;                  // log2 of the entries per chunk
    constexpr int CHN = 1 << CSH;
    const int lane = lane_id();
    const int lane4 = lane * 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x / WAVE;
    const int nrows = nrows_p ? *nrows_p : m;
    // Marker of the columns seen in the current row.
    //  bitmaps: one guard word behind each wave's bitmap, all ones: idle lanes carry the column `idle` =
    //           bit 31 of that word, which is therefore never "new" -- no per-lane predicate in the hot loop.
    //  hash:    for rows with few products (known before this kernel runs) an open-addressing set of
    //           `bm_words` slots (a power of two >= 2 x products) replaces the bitmap, whose ncols/8
    //           bytes per wave leave one wave per CU at 1e6 columns; idle lanes carry -1.
    unsigned *bm = MARK == MARK_GLOBAL_BITMAP ? gbitmap + ((size_t)blockIdx.x * wpb + wave) * (bm_words + 1)
                                              : lds_bm + (size_t)wave * (bm_words + (HASH ? 0 : 1));
    int *tab = (int *)bm;
    const int idle = HASH ? -1 : bm_words * 32 + 31;
    const unsigned hmask = (unsigned)bm_words - 1u;
    const int hshift = __builtin_clz((unsigned)bm_words) + 1;
    for (int w = lane; w < bm_words; w += WAVE) bm[w] = HASH ? 0xffffffffu : 0u;
    if (!HASH && lane == 0) bm[bm_words] = 0xffffffffu;
    if (MARK == MARK_GLOBAL_BITMAP) __threadfence();

    // Bitmap kernels: rows are handed out by a global counter, in order: a static split leaves the waves
    // of the last, partial round of workgroups running alone (2048 workgroups on 1536 resident ones
    // cost 25 %), and rows differ in work anyway.  Every wave leaves when the counter passes the
    // last row.  Hash classes: the rows are tiny and there are up to 1e6 of them -- that many atomics
    // on one word cost more than they balance (measured), so those keep the interleaved static split.
    // (round 4: rbatch rows per counter round trip -- 16 when the host sees >= 64 rows per wave, see smm_symbolic_ccs)
    for (int rs = blockIdx.x * wpb + wave;; rs += gridDim.x * wpb) {
        int r0 = rs, r1 = rs + 1;
        if (!HASH) {
            if (lane == 0) r0 = atomicAdd(row_counter, rbatch);
            r0 = rl(r0, 0);
            r1 = r0 + rbatch;
        }
        if (r0 >= nrows) break;
        r1 = r1 < nrows ? r1 : nrows;
      for (int ri = r0; ri < r1; ++ri) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        int thresh = 0;
        if (SYM) { const int64_t gi = row + row_offset; thresh = gi > 0x7fffffff ? 0x7fffffff : (int)gi; }
        IT *__restrict__ out = tmp_idx + ub_off[row];
        int n = 0;
        if (a1 > a0) {
            auto load_r = [&](int jb) { int e = jb + lane; e = e < a1 ? e : a1 - 1; return a_idx[e]; };
            int r_c = load_r(a0), r_n = load_r(a0 + WAVE);
            int bs = b_ptr[r_c], be = b_ptr[r_c + 1];
            for (int jb = a0; jb < a1; jb += WAVE) {
                const int rem = a1 - jb;
                const int nb = rem < WAVE ? rem : WAVE;
                const int bs_n = b_ptr[r_n], be_n = b_ptr[r_n + 1];      // next 64 entries' rows of B
                const int r_nn = load_r(jb + 2 * WAVE);                    // A indices two batches ahead
                unsigned myP = 0xffffffffu;                             // unset: entry without chunks
                // Re-define bs/be here so that the compiler's wait for their loads sits HERE, once: a
                // vmcnt wait inside the chunk loop would also drain the chunk loads of load_masked
                // (vmcnt retires in order).
                asm volatile("" : "+v"(bs), "+v"(be));
                // entry `lane` of the batch owns the chunks [excl, incl) of the batch's chunk list
                const int nch = lane < nb ? (be - bs + CHN - 1) >> CSH : 0;
                const int incl = wave_scan_incl(nch);
                const int excl = incl - nch;
                const int T = rl(incl, WAVE - 1);
                for (int tg = 0; tg < T; tg += WAVE) {
                    // descriptor of chunk t = tg + lane: its entry is the number of entries whose chunks
                    // all precede t (binary search over the scan), then start offset and lane mask
                    const int t = tg + lane;
                    int ej = 0;
#pragma unroll
                    for (int sft = WAVE / 2; sft > 0; sft >>= 1)
                        if (__shfl(incl, ej + sft - 1) <= t) ej += sft;
                    const int start = __shfl(bs, ej) + (t - __shfl(excl, ej)) * CHN;
                    const int cnt = __shfl(be, ej) - start;             // >= 1 for t < T
                    const int nl = WIDE ? (cnt + 1) >> 1 : cnt;         // lanes that load
                    const unsigned long long lm = t >= T ? 0ull : nl >= WAVE ? ~0ull : (1ull << nl) - 1ull;
                    const int d_cnt = cnt;                              // WIDE: lane l holds entries 2l, 2l+1 < cnt
                    const unsigned d_off = t < T ? (unsigned)start : 0u, d_lo = (unsigned)lm, d_hi = (unsigned)(lm >> 32);
                    const int G = T - tg < WAVE ? T - tg : WAVE;
                    int nrec = 0;                                       // lane t: length of the list before chunk t
                    for (int rb = 0; rb < G; rb += SYM_UNROLL) {
                        constexpr bool W32 = WIDE && !I16;              // pairs of 32-bit columns travel in 64-bit registers
                        int c[SYM_UNROLL];
                        long long cq[W32 ? SYM_UNROLL : 1];
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {          // idle lanes / dead slots
                            if constexpr (W32) cq[u] = (long long)(((unsigned long long)(unsigned)idle << 32) | (unsigned)idle);
                            else c[u] = WIDE ? (idle | (idle << 16)) : idle;
                        }
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {                  // all loads first (MLP)
                            const int tt = rb + u;
                            const unsigned long long mk = ((unsigned long long)rl(d_hi, tt) << 32) | rl(d_lo, tt);
                            if constexpr (W32) load_masked_pair(cq[u], (const int *)b_idx + rl(d_off, tt), mk, lane4 * 2);
                            else if constexpr (WIDE) load_masked_pair(c[u], (const unsigned short *)b_idx + rl(d_off, tt), mk, lane4);
                            else load_masked(c[u], b_idx + rl(d_off, tt), mk, lane4);
                        }
                        // chunk u of a full round has SYM_UNROLL-1-u younger loads behind it; in a partial
                        // round the dead slots may not count at all (EXEC = 0), so it simply drains
                        if (rb + SYM_UNROLL > G) { if constexpr (W32) wait_vm_all(cq); else wait_vm_all(c); }
                        if constexpr (HASH) {
                            // chunk after chunk: look the columns up in the set, insert the unseen ones
                            // (linear probing; the columns of one chunk are distinct, so two lanes
                            // compete for a slot only with different keys and the loser moves on)
#pragma unroll
                            for (int u = 0; u < SYM_UNROLL; ++u) {
                                wait_vm(c[u], SYM_UNROLL - 1 - u);
                                if (SYM) c[u] = c[u] >= thresh ? c[u] : -1;
                                if (lane == rb + u) nrec = n;
                                bool pend = c[u] >= 0, isnew = false;
                                unsigned h = ((unsigned)c[u] * 0x9E3779B1u) >> hshift;
                                while (__ballot(pend)) {
                                    if (pend) {
                                        const int was = atomicCAS(&tab[h], -1, c[u]);
                                        if (was == -1) { isnew = true; pend = false; }
                                        else if (was == c[u]) pend = false;
                                        else h = (h + 1) & hmask;
                                    }
                                }
                                const unsigned long long mask = __ballot(isnew);
                                if (isnew)
                                    out[__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)n))] = (IT)c[u];
                                n += __popcll(mask);
                            }
                        } else if constexpr (WIDE) {
                        // two columns per lane: entry 2l (low half) and 2l+1 (high half; the last lane of an
                        // odd chunk read one column too many: made idle).  Columns of one chunk are distinct, so
                        // the order of the two test-and-sets inside a chunk does not matter; the list order
                        // does: entry 2l, then 2l+1, lane after lane.
                        unsigned oldl[SYM_UNROLL], oldh[SYM_UNROLL], bitl[SYM_UNROLL], bith[SYM_UNROLL];
                        int ch[SYM_UNROLL];
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {
                            int lo, hi;
                            if constexpr (W32) {
                                wait_vm(cq[u], SYM_UNROLL - 1 - u);
                                lo = (int)(unsigned)cq[u]; hi = (int)(unsigned)((unsigned long long)cq[u] >> 32);
                            } else {
                                wait_vm(c[u], SYM_UNROLL - 1 - u);
                                lo = c[u] & 0xffff; hi = (int)((unsigned)c[u] >> 16);
                            }
                            const int cn = rl(d_cnt, (rb + u) & (WAVE - 1));
                            if (2 * lane + 1 >= cn) hi = idle;
                            if (SYM) { lo = lo >= thresh ? lo : idle; hi = hi >= thresh ? hi : idle; }
                            c[u] = lo; ch[u] = hi;
                            bitl[u] = 1u << (lo & 31); bith[u] = 1u << (hi & 31);
                            oldl[u] = atomicOr(bm + (lo >> 5), bitl[u]);
                            oldh[u] = atomicOr(bm + (hi >> 5), bith[u]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {
                            if (lane == rb + u) nrec = n;
                            const bool nl_ = (bitl[u] & ~oldl[u]) != 0, nh_ = (bith[u] & ~oldh[u]) != 0;
                            const unsigned long long ml = __ballot(nl_), mh = __ballot(nh_);
                            const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(mh >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mh,
                                                __builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, (unsigned)n))));
                            if (nl_) out[at] = (IT)c[u];
                            if (nh_) out[at + (nl_ ? 1u : 0u)] = (IT)ch[u];
                            n += __popcll(ml) + __popcll(mh);
                        }
                        } else {
                        // test-and-set every chunk's columns (LDS atomics of one wave execute in issue
                        // order, so chunk u sees the bits of chunks < u whenever its result is read).  All
                        // 16 atomics are issued before the first result is consumed: one LDS round trip
                        // per round instead of one per chunk.
                        unsigned old[SYM_UNROLL], bit[SYM_UNROLL], pre[SYM_UNROLL];
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {
                            wait_vm(c[u], SYM_UNROLL - 1 - u);
                            if (SYM) c[u] = c[u] >= thresh ? c[u] : idle;   // left of the diagonal: as idle
                            bit[u] = 1u << (c[u] & 31);
                            unsigned *wp = bm + (c[u] >> 5);                // idle -> the guard word
                            if (SAFE) pre[u] = *(volatile unsigned *)wp;
                            old[u] = atomicOr(wp, bit[u]);
                        }
                        __builtin_amdgcn_sched_barrier(0);                  // no result is consumed between the atomics
#pragma unroll
                        for (int u = 0; u < SYM_UNROLL; ++u) {              // then consume in order
                            if (lane == rb + u) nrec = n;
                            bool isnew = (bit[u] & ~old[u]) != 0;
                            if (SAFE) {
                                // two lanes of this instruction with the same new column: the LOWEST wins
                                const bool a = c[u] != idle;
                                unsigned long long losers = __ballot(a && !(pre[u] & bit[u]) && !isnew);
                                while (losers) {                     // rare: duplicate column in a B row
                                    const int x = __ffsll((long long)losers) - 1;
                                    const int cx = rl(c[u], x);
                                    const bool ingrp = a && c[u] == cx;
                                    const unsigned long long grp = __ballot(ingrp);
                                    const int firstl = __ffsll((long long)grp) - 1;
                                    if (ingrp) isnew = (lane == firstl);
                                    losers &= ~grp;
                                }
                            }
                            const unsigned long long mask = __ballot(isnew);
                            if (isnew)
                                out[__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mask, (unsigned)n))] = (IT)c[u];
                            n += __popcll(mask);
                        }
                        }
                    }
                    // P of the entries whose first chunk was in this group
                    const int tl = excl - tg;
                    const bool has = nch > 0 && tl >= 0 && tl < G;
                    const int pv = __shfl(nrec, has ? tl : 0);
                    if (has) myP = (unsigned)pv;
                }
                // an entry whose row of B is empty starts where the next one starts
                {
                    unsigned v = myP;
#pragma unroll
                    for (int o = 1; o < WAVE; o <<= 1) {
                        const unsigned y = __shfl_down(v, o);
                        if (lane + o < WAVE && y < v) v = y;
                    }
                    if (v == 0xffffffffu) v = (unsigned)n;
                    if (lane < nb) P[jb + lane] = v;
                }
                bs = bs_n; be = be_n; r_n = r_nn;
            }
        }
        if (lane == 0) rowcnt[row] = n;
        // reset the marker for the next row (sparsework.cpp:120-128: memset when the row is
        // long, per-entry otherwise)
        if (MARK == MARK_GLOBAL_BITMAP) __threadfence();
        if (HASH) {
            // (round 4) a row with few distinct columns empties the slots it filled instead of the whole set: every listed
            // column is in the set, so its probe sequence ends at its slot whatever the other lanes have emptied meanwhile
            // (an empty slot on the way does not stop it).  A band of half-width 8 puts 33 columns into 4096 slots.
            if (n * 8 >= bm_words) {
                for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0xffffffffu;
            } else {
                for (int s2 = lane; s2 < n; s2 += WAVE) {
                    const int cc = (int)out[s2];
                    unsigned h = ((unsigned)cc * 0x9E3779B1u) >> hshift;
                    for (unsigned tries = 0; tab[h] != cc && tries < hmask; ++tries) h = (h + 1) & hmask;
                    tab[h] = -1;
                }
            }
        } else if (n >= bm_words) {
            for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0;
        } else {
            for (int s2 = lane; s2 < n; s2 += WAVE) bm[out[s2] >> 5] = 0;
        }
        if (MARK == MARK_GLOBAL_BITMAP) __threadfence();
      }
    }
}

// ---------------------------------------------------------------------------------------
// Chunked column stream (CCS) of a sorted operand without repeated columns, and the symbolic walk over it
// (round 3).  The columns of B are cut into n_slabs column slabs of ws <= 63 456 columns (one slab when B has
// fewer columns than that); the piece of row j inside slab s is stored as slab-LOCAL 16-bit columns, padded to
// whole chunks of 128 entries, chunk after chunk, 256-byte aligned:
//     cptr[s * (rows + 1) + j]   first chunk of piece (s, j); the pieces of a slab follow each other in row order
//     stream[chunk * 128 + p]    local column of entry p, or -- past the piece's end -- the guard column of lane p / 2
// and ONE more chunk of guard columns only stands behind the last piece (the walk's dead slots load it),
// so that a chunk is ONE unmasked 256-byte load (two whole cache lines, a dword = two columns per lane) and
// needs no lane mask, no entry count and no EXEC set-up: the padding lanes carry guard columns, which live in 64
// words of their own behind the wave's bitmap (all ones: never "new"; one word per lane: they do not collide on
// one address as the single guard word of smm_symbolic's idle lanes does).
#ifndef SMM_CCS_ASM_STORES
#define SMM_CCS_ASM_STORES 1
#endif
constexpr int CCS_CHUNK = 128;
constexpr int CCS_MAX_WS = 63456;            // (ws / 32 + 64 guard words) * 32 + 31 must fit 16 bits
__host__ __device__ __forceinline__ int ccs_guard_col(int bm_words, int p) { return (bm_words + (p >> 1)) * 32 + 31; }

__global__ __launch_bounds__(256) void smm_ccs_count(int rows, int n_slabs, const int *__restrict__ seg, int *__restrict__ chunks)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)rows * n_slabs) return;
    const int s = (int)(gid / rows), j = (int)(gid % rows);
    const int *sp = seg + (size_t)j * (n_slabs + 1) + s;
    chunks[gid] = (sp[1] - sp[0] + CCS_CHUNK - 1) / CCS_CHUNK;
}
// cptr[s * (rows + 1) + j] from the exclusive scan of chunks[] (off64 has rows * n_slabs + 1 entries)
__global__ __launch_bounds__(256) void smm_ccs_ptr(int rows, int n_slabs, const int64_t *__restrict__ off64, int *__restrict__ cptr)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)(rows + 1) * n_slabs) return;
    const int s = (int)(gid / (rows + 1)), j = (int)(gid % (rows + 1));
    cptr[gid] = (int)off64[(int64_t)s * rows + j];            // j == rows: the first chunk of the next slab = this slab's end
}
// one wave per (slab, row) piece
// stat[0] += neighbouring entries of a piece that share a 32-column word of the marker bitmap, stat[1] += neighbouring
// entries: the share tells the symbolic walk whether B has dense runs of columns (see smm_symbolic_ccs<.., DR>)
__global__ __launch_bounds__(256) void smm_ccs_fill(int rows, int n_slabs, int ws, int bm_words, const int *__restrict__ idx,
                                                    const int *__restrict__ seg, const int *__restrict__ cptr,
                                                    unsigned short *__restrict__ stream, unsigned long long *__restrict__ stat)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    const int64_t npieces = (int64_t)rows * n_slabs;
    unsigned n_same = 0, n_pairs = 0;
    if (blockIdx.x == 0 && threadIdx.x < CCS_CHUNK)              // the all-guard chunk behind the last piece
        stream[(int64_t)cptr[(size_t)n_slabs * (rows + 1) - 1] * CCS_CHUNK + threadIdx.x] = (unsigned short)ccs_guard_col(bm_words, threadIdx.x);
    for (int64_t it = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); it < npieces; it += (int64_t)gridDim.x * wpb) {
        const int s = (int)(it / rows), j = (int)(it % rows);
        const int *sp = seg + (size_t)j * (n_slabs + 1) + s;
        const int k0 = sp[0], len = sp[1] - sp[0];
        const int *cp = cptr + (size_t)s * (rows + 1) + j;
        const int64_t base = (int64_t)cp[0] * CCS_CHUNK;
        const int total = (cp[1] - cp[0]) * CCS_CHUNK;
        for (int p = lane; p < total; p += WAVE) {
            const int col = p < len ? idx[k0 + p] - s * ws : ccs_guard_col(bm_words, p & (CCS_CHUNK - 1));
            stream[base + p] = (unsigned short)col;
            if (p + 1 < len) { ++n_pairs; n_same += ((idx[k0 + p + 1] - s * ws) >> 5) == (col >> 5) ? 1u : 0u; }
        }
    }
    if (stat) {
        for (int o = 32; o > 0; o >>= 1) { n_same += __shfl_xor(n_same, o); n_pairs += __shfl_xor(n_pairs, o); }
        if (lane == 0 && n_pairs) { atomicAdd(&stat[0], (unsigned long long)n_same); atomicAdd(&stat[1], (unsigned long long)n_pairs); }
    }
}
// products of every (slab, row of A) unit: the capacity of its ordered list is min(products, slab columns that can appear)
__global__ __launch_bounds__(256) void smm_ccs_row_work(int m, int n_slabs, int ws, int ncols, int rowsB, int64_t row_offset, int sym,
                                                        const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                                        const int *__restrict__ seg, int64_t *__restrict__ ub)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        for (int s = 0; s < n_slabs; ++s) {
            int64_t t = 0;
            for (int e = a_ptr[row] + lane; e < a_ptr[row + 1]; e += WAVE) {
                const int *sp = seg + (size_t)a_idx[e] * (n_slabs + 1) + s;
                t += sp[1] - sp[0];
            }
            for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
            if (lane == 0) {
                const int64_t lo = (int64_t)s * ws;
                int64_t cap = (ncols - lo) < ws ? (ncols - lo) : ws;
                cap = cap > 0 ? cap : 0;                       // (exact-walk geometries may end in slabs beyond the last column)
                if (sym) { const int64_t gi = row + row_offset; const int64_t from = gi > lo ? gi : lo; cap = lo + cap > from ? lo + cap - from : 0; }
                ub[(size_t)s * m + row] = t < cap ? t : cap;
            }
        }
    }
}

// Symbolic walk over the chunked stream: the reference's first-touch rule (sparsework.cpp:70-110) for one
// (slab, row) unit per wave -- with one slab that is smm_symbolic's job, output for output (tmp lists, P, row
// counts), at two thirds of its instructions: per chunk one scalar-base load, two test-and-sets and the ordered
// compaction; no mask, no count, no EXEC set-up around the load.
//   unit u (handed out by a global counter, slab-major): slab s = u / nrows, row = rowlist[u % nrows]
//   list_off[s * m + row]   where the unit's list starts in tmp (uint16, slab-local columns)
//   P[s * nnzA + e]         list length of the unit when step e starts;  cnt[s * m + row] final length
//
// DR (round 4, operands with DENSE RUNS of columns: bands, blocks): 128 consecutive columns are 4 bitmap words, i.e. 16 lanes
// of one returning atomic on ONE address, which the LDS serialises at 3.3 cycles per lane (53 instead of 9 cycles per wave
// instruction, scripts/ubench/lds_or_sameword.hip) -- 28 of the 65 ms of a band of half-width 200.  Same-address READS are
// broadcast, and the columns of a chunk are distinct, so here the test reads the words plainly and the set is one
// non-returning OR per group of up to four lanes: a lane's two columns merged when they share a word, then two DPP steps
// of a segmented OR over neighbouring lanes with the same word.  The ~15 extra vector instructions per chunk are what the
// uniform-random walk cannot afford (profiles/r4_symbolic_lean.txt), hence a second instantiation, chosen from the
// share of neighbouring entries of B that share a word (smm_ccs_fill).
//
// BATCH (round 4): units taken per counter round trip.  One atomic per unit on ONE word costs 11.4 ns of L2 time each -- a
// tall-skinny product with 1e6 ten-entry rows spent 10.3 of its 11.45 ms symbolic phase there (1.18 ms with a static split,
// diagnostic build).  The host picks 16 when there are at least 64 units per wave (imbalance then averages out) and 1
// otherwise: as a template parameter, because any more scalar state in the BATCH = 1 loop spills into a walk that sits on the
// issue corner (a guided, run-time batch size cost configs[1] 0.8 ms, profiles/r4_structured.txt).
template <bool SYM, int UNROLL, bool DR = false, int BATCH = 1>
__global__ __launch_bounds__(256) void smm_symbolic_ccs(int m, int n_slabs, const int *__restrict__ rowlist, const int *__restrict__ nrows_p,
                                                        int64_t row_offset, int ws, int bm_words, int rowsB, int64_t nnzA, int guard_chunk,
                                                        const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                                        const int *__restrict__ cptr, const unsigned short *__restrict__ stream,
                                                        const int64_t *__restrict__ list_off, unsigned short *__restrict__ tmp,
                                                        unsigned *__restrict__ P, int *__restrict__ cnt, int *__restrict__ unit_counter)
{
    extern __shared__ unsigned lds_bm[];
    const int lane = lane_id();
    const int lane4 = lane * 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nrows = nrows_p ? *nrows_p : m;
    const int64_t nunits = (int64_t)nrows * n_slabs;
    unsigned *bm = lds_bm + (size_t)wave * (bm_words + WAVE);
    for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0u;
    bm[bm_words + lane] = 0xffffffffu;                          // the guard words: one per lane
    const int guard = ccs_guard_col(bm_words, 2 * lane);        // this lane's guard column (never new)
    for (;;) {
        int u0 = 0;
        if (lane == 0) u0 = atomicAdd(unit_counter, BATCH);
        u0 = rl(u0, 0);
        if (u0 >= nunits) break;
      for (int ui = u0; ui < u0 + BATCH && ui < nunits; ++ui) {
        const int s = ui / nrows;
        const int rr = ui - s * nrows;
        const int row = rowlist ? rowlist[rr] : rr;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const int *__restrict__ cp = cptr + (size_t)s * (rowsB + 1);
        int thresh = 0;                                         // SYM: slab-local columns below the diagonal are dropped
        if (SYM) { const int64_t d = row + row_offset - (int64_t)s * ws; thresh = d < 0 ? 0 : (d > ws ? ws : (int)d); }
        const int64_t lo_off = list_off[(size_t)s * m + row];
        unsigned short *__restrict__ out = tmp + lo_off;
        unsigned *__restrict__ Ps = P + (size_t)s * nnzA;
        int n = 0;
        // (round 4) a unit whose list has no capacity has no product in this slab -- most (slab, row) units of a banded or
        // block operand: its start slots are all 0, and nothing else is read (the walk would still fetch two chunk pointers
        // per entry of A to find every piece empty)
        const bool no_products = __builtin_amdgcn_readfirstlane((int)(list_off[(size_t)s * m + row + 1] != lo_off)) == 0;   // wave-uniform
        if (no_products) {
            for (int e = a0 + lane; e < a1; e += WAVE) Ps[e] = 0u;
        } else if (a1 > a0) {
            auto load_r = [&](int jb) { int e = jb + lane; e = e < a1 ? e : a1 - 1; return a_idx[e]; };
            int r_c = load_r(a0), r_n = load_r(a0 + WAVE);
            int cs = cp[r_c], ce = cp[r_c + 1];
            for (int jb = a0; jb < a1; jb += WAVE) {
                const int rem = a1 - jb;
                const int nb = rem < WAVE ? rem : WAVE;
                const int cs_n = cp[r_n], ce_n = cp[r_n + 1];      // next 64 entries' pieces
                const int r_nn = load_r(jb + 2 * WAVE);            // A indices two batches ahead
                unsigned myP = 0xffffffffu;                     // unset: entry without chunks
                asm volatile("" : "+v"(cs), "+v"(ce));          // the wait for these loads sits here, once (see smm_symbolic)
                const int nch = lane < nb ? ce - cs : 0;
                const int incl = wave_scan_incl(nch);
                const int excl = incl - nch;
                const int T = rl(incl, WAVE - 1);
                for (int tg = 0; tg < T; tg += WAVE) {
                    // lane t - tg: descriptor of chunk t = its index in the stream (entry by binary search over the scan)
                    const int t = tg + lane;
                    int ej = 0;
#pragma unroll
                    for (int sft = WAVE / 2; sft > 0; sft >>= 1)
                        if (__shfl(incl, ej + sft - 1) <= t) ej += sft;
                    // (both shuffles OUTSIDE the select: inside it they would run under t < T only, and a lane whose
                    // entry index ej lies beyond T -- empty pieces before it -- would read an inactive lane: 0)
                    const int cs_e = __shfl(cs, ej), excl_e = __shfl(excl, ej);
                    const int d_chunk = t < T ? cs_e + (t - excl_e) : guard_chunk;      // dead slots: guard columns only
                    const int G = T - tg < WAVE ? T - tg : WAVE;
                    int nrec = 0;                               // lane t - tg: length of the list before chunk t
                    for (int rb = 0; rb < G; rb += UNROLL) {
                        int c[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) {          // all loads first; the chunk index is wave-uniform
                            const unsigned short *cb = stream + (size_t)rl(d_chunk, rb + u) * CCS_CHUNK;
                            asm volatile("global_load_dword %0, %1, %2" : "=v"(c[u]) : "v"(lane4), "s"(cb) : "memory");
                        }
                        unsigned oldl[UNROLL], oldh[UNROLL], bitl[UNROLL], bith[UNROLL];
                        int ch[UNROLL];
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) {
                            wait_vm(c[u], UNROLL - 1 - u);
                            int lo = c[u] & 0xffff, hi = (int)((unsigned)c[u] >> 16);
                            if (SYM) { lo = lo >= thresh ? lo : guard; hi = hi >= thresh ? hi : guard; }
                            c[u] = lo; ch[u] = hi;
                            bitl[u] = 1u << (lo & 31); bith[u] = 1u << (hi & 31);
                            if constexpr (!DR) {
                                oldl[u] = atomicOr(bm + (lo >> 5), bitl[u]);
                                oldh[u] = atomicOr(bm + (hi >> 5), bith[u]);
                            } else {
                                const int wl = lo >> 5, wh = hi >> 5;
                                oldl[u] = bm[wl];                   // (one wave, one bitmap, LDS in issue order: these see every OR of the chunks before)
                                oldh[u] = bm[wh];
                                const bool same = wl == wh;
                                unsigned mm = bitl[u] | (same ? bith[u] : 0u);
                                {   // segmented OR over lanes i-1, then i-2 .. i-3, of the 16-lane row (words ascend with the lanes)
                                    const int pw1 = __builtin_amdgcn_update_dpp(-1, wl, 0x111, 0xf, 0xf, false);
                                    const unsigned pm1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x111, 0xf, 0xf, false);
                                    if (pw1 == wl) mm |= pm1;
                                    const int pw2 = __builtin_amdgcn_update_dpp(-1, wl, 0x112, 0xf, 0xf, false);
                                    const unsigned pm2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mm, 0x112, 0xf, 0xf, false);
                                    if (pw2 == wl) mm |= pm2;
                                }
                                // the last lane of every run of (at most four) lanes with one word issues the OR.  A lane whose
                                // successor has the same word has been merged into it by the steps above iff the run is not cut
                                // by the step pattern: lanes 4k+3 and row ends always issue.
                                const int nw = __builtin_amdgcn_update_dpp(-1, wl, 0x101, 0xf, 0xf, false);     // row_shl:1 = lane + 1
                                if (nw != wl || (lane & 3) == 3) (void)__hip_atomic_fetch_or(bm + wl, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (!same) (void)__hip_atomic_fetch_or(bm + wh, bith[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < UNROLL; ++u) {          // consume strictly in order
                            if (lane == rb + u) nrec = n;
                            const bool nl_ = (bitl[u] & oldl[u]) == 0, nh_ = (bith[u] & oldh[u]) == 0;
                            const unsigned long long ml = __ballot(nl_), mh = __ballot(nh_);
                            const unsigned at = __builtin_amdgcn_mbcnt_hi((unsigned)(mh >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mh,
                                                __builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, (unsigned)n))));
#if SMM_CCS_ASM_STORES
                            // hand-issued list stores: 32-bit byte offsets from the list's scalar base under the two ballot
                            // masks -- no 64-bit address arithmetic, no branch around either store (the compiler's form: two
                            // v_lshl_add_u64, a select, an add, two s_and_saveexec / s_cbranch_execz pairs).  Round 3, interleaved
                            // A/B: smm_symbolic 5.37 -> 5.10 ms at configs[1], 12.0 -> 11.6 at the configs[4] share.
                            {
                                const unsigned off_lo = at << 1;
                                const unsigned off_hi = (at + (nl_ ? 1u : 0u)) << 1;
                                const unsigned packed = (unsigned)c[u] | ((unsigned)ch[u] << 16);
                                asm volatile("s_mov_b64 exec, %4\n\t"
                                             "global_store_short %0, %2, %3\n\t"
                                             "s_mov_b64 exec, %5\n\t"
                                             "global_store_short_d16_hi %1, %2, %3\n\t"
                                             "s_mov_b64 exec, -1"
                                             :: "v"(off_lo), "v"(off_hi), "v"(packed), "s"(out), "s"(ml), "s"(mh) : "memory");
                            }
#else
                            if (nl_) out[at] = (unsigned short)c[u];
                            if (nh_) out[at + (nl_ ? 1u : 0u)] = (unsigned short)ch[u];
#endif
                            n += __popcll(ml) + __popcll(mh);
                        }
                    }
                    // P of the entries whose first chunk was in this group
                    const int tl = excl - tg;
                    const bool has = nch > 0 && tl >= 0 && tl < G;
                    const int pv = __shfl(nrec, has ? tl : 0);
                    if (has) myP = (unsigned)pv;
                }
                {   // an entry whose piece is empty starts where the next one starts
                    unsigned v = myP;
#pragma unroll
                    for (int o = 1; o < WAVE; o <<= 1) {
                        const unsigned y = __shfl_down(v, o);
                        if (lane + o < WAVE && y < v) v = y;
                    }
                    if (v == 0xffffffffu) v = (unsigned)n;
                    if (lane < nb) Ps[jb + lane] = v;
                }
                cs = cs_n; ce = ce_n; r_n = r_nn;
            }
        }
        if (lane == 0) cnt[(size_t)s * m + row] = n;
#if SMM_CCS_ASM_STORES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the hand-issued list stores (read back just below)
#endif
        // reset the marker for the next unit (sparsework.cpp:120-128: memset when the row is long, per entry otherwise)
        if (n >= bm_words) {
            for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0;
        } else {
            for (int s2 = lane; s2 < n; s2 += WAVE) bm[out[s2] >> 5] = 0;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// Sub-run table.  Step e of a row appended the columns list[P[e] .. P[e+1]) in ascending
// order (B's rows are sorted), so the part that falls into coarse tile t is the contiguous
// slot range [runs[e][t], runs[e][t+1]).  One wave per row, one lane per A entry, nct-1
// lower_bounds each.  The list segments of 64 consecutive entries are one contiguous piece of the
// row's list: the wave copies the list into LDS with coalesced loads, RUNS_WIN entries at a time (the
// next window is already on its way while one is searched), and the lanes search there (a first version searched in global memory: ~70 dependent probes per lane into
// lines nobody else used, 19 ms at 200 000 columns / 10 tiles; this one streams the list once).
// The TAIL of a row: from the first step e0 after which every step appends fewer than TAIL_MIN columns, the
// sub-runs are a few entries each (at 99 % fill the last 60 % of the steps hold 13 % of a row) -- one chunk
// load and two partial stores per sub-run.  Those steps get no sub-run table: the numeric epilogue walks the
// list positions [P[e0], end) contiguously, 64 at a time, and every tile's unit keeps the columns that are
// its own (positions in the list ARE positions in the result).  tail[row] = {e0, P[e0]}.
// tail_min is a kernel argument (round 4): 32 for the default walk (interleaved A/B in both orders at configs[1]: 8: 31.63,
// 16: 31.38, 32: 31.13-31.22, 64: 31.26-31.47, 128: 31.37, 256: 31.92 ms per step), 64 for SMM_EXACT (8 waves share the
// tail: 39.1 against 39.5 ms with 32).
constexpr int TAIL_MIN_DEFAULT = 32, TAIL_MIN_EXACT = 64;
constexpr int RUNS_WIN = 2048;
template <typename LT>
__global__ __launch_bounds__(256) void smm_runs(int m, int nct, int wc, const int *__restrict__ rowlist,
                                                const int *__restrict__ a_ptr,
                                                const int64_t *__restrict__ ub_off,
                                                const int *__restrict__ rowcnt,
                                                const unsigned *__restrict__ P,
                                                const LT *__restrict__ tmp_idx,
                                                unsigned *__restrict__ runs, int2 *__restrict__ tail, unsigned *__restrict__ err,
                                                int tail_min)
{
    __shared__ int win_all[4][RUNS_WIN];
    constexpr int NV = RUNS_WIN / WAVE;            // window elements per lane
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;              // 4
    int *win = win_all[threadIdx.x >> 6];
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < m; ri += gridDim.x * wpb) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const LT *__restrict__ list = tmp_idx + ub_off[row];
        const unsigned total = (unsigned)rowcnt[row];
        // where the tail starts: one pass over P (coalesced)
        int e_last = a0 - 1;
        for (int eb = a0; eb < a1; eb += WAVE) {
            const int e = eb + lane;
            if (e < a1) {
                const unsigned q0 = P[e], q1 = e + 1 < a1 ? P[e + 1] : total;
                if (q1 - q0 >= (unsigned)tail_min) e_last = e;
            }
        }
        for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(e_last, o); e_last = y > e_last ? y : e_last; }
        const int e0 = e_last + 1;                  // steps e0 .. a1-1 form the tail
        if (lane == 0) tail[row] = make_int2(e0, (int)(e0 < a1 ? P[e0] : total));
        // The row's list is cut into fixed windows of RUNS_WIN entries.  Window k sits in LDS while
        // window k+1 travels in registers (its loads are issued before the searches in window k).
        int v[NV];
        auto fetch = [&](unsigned wb) {             // window [wb, wb + RUNS_WIN) of the list -> registers
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const unsigned i = wb + u * WAVE + lane;
                v[u] = (int)list[i < total ? i : (total ? total - 1 : 0)];
            }
        };
        unsigned cur = 0xffffffffu;                 // start of the window that is in LDS
        if (total) fetch(0);
        for (int eb = a0; eb < e0; eb += WAVE) {
            const int e = eb + lane;
            const bool valid = e < e0;
            unsigned p0 = valid ? P[e] : total;
            unsigned p1 = (valid && e + 1 < a1) ? P[e + 1] : total;
            // (always on: start slots are another kernel's output -- a value beyond the list or a step that ends before
            // it starts would turn the window loop below into 2^32 / RUNS_WIN iterations)
            if (p0 > total || p1 > total || p1 < p0) {
                plan_err(err, PLAN_ERR_P, row);
                p0 = p0 < total ? p0 : total;
                p1 = p1 < p0 ? p0 : (p1 < total ? p1 : total);
            }
            unsigned *r = runs + (size_t)(valid ? e : e0 - 1) * (nct + 1);
            int tcur = 1;                           // next boundary this lane has to place
            if (valid) {
                r[0] = p0; r[nct] = p1;
                if (p0 == p1) { for (; tcur < nct; ++tcur) r[tcur] = p0; }
            } else {
                tcur = nct;
            }
            const unsigned reg_lo = rl(p0, 0), reg_hi = rl(p1, WAVE - 1);
            for (unsigned wb = reg_lo - reg_lo % RUNS_WIN; wb < reg_hi; wb += RUNS_WIN) {
                const unsigned we = wb + RUNS_WIN < total ? wb + RUNS_WIN : total;
                if (wb != cur) {                    // bring the window in (it is in v), request the next one
                    wave_sync();
#pragma unroll
                    for (int u = 0; u < NV; ++u) win[u * WAVE + lane] = v[u];
                    cur = wb;
                    if (we < total) fetch(we);
                    wave_sync();
                }
                unsigned lo = p0 > wb ? p0 : wb;
                const unsigned hiw = p1 < we ? p1 : we;
                if (lo < hiw) {
                    while (tcur < nct) {
                        const int bound = tcur * wc;
                        unsigned l = lo, h = hiw;
                        while (l < h) {
                            const unsigned mid = l + ((h - l) >> 1);
                            if (win[mid - wb] < bound) l = mid + 1; else h = mid;
                        }
                        if (l < hiw) { r[tcur] = l; lo = l; ++tcur; }            // found inside the window
                        else if (p1 <= we) { r[tcur] = p1; lo = hiw; ++tcur; }   // the segment ends here
                        else break;                                              // it goes on in the next window
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Column slabs (round 3, operands with more than CCS_MAX_WS columns).  smm_symbolic_ccs ran once per (slab, row):
// slab s of row i has its OWN ordered list L_s (slab-local 16-bit columns) and its own P_s[e].  Step e of the row
// appended, slab after slab, the columns L_s[P_s[e] .. P_s[e+1]) -- so in the row of C the sub-run of (step e,
// tile t of slab s) starts at
//     dst = D[e] + sum_{s' < s} c_{s'}[e] + (src - P_s[e]),   D[e] = sum_{e' < e} sum_s c_s[e'],  c_s[e] = P_s[e+1] - P_s[e]
// where src = first position of L_s in [P_s[e], P_s[e+1]) whose column lies in tile t.  This kernel is smm_runs for
// that layout: runs2[t * nnzA + e] = { src | len << 16, dst } (src < 2^16: a slab list is at most CCS_MAX_WS long;
// len <= tile width < 2^15), the table the numeric epilogue walks -- every step is a sub-run step here (no tail:
// positions in a slab's list are not positions in the result).  One wave per row.
__global__ __launch_bounds__(256) void smm_slab_rowcnt(int m, int n_slabs, const int *__restrict__ scnt, int *__restrict__ rowcnt)
{
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < m; row += gridDim.x * blockDim.x) {
        int t = 0;
        for (int s = 0; s < n_slabs; ++s) t += scnt[(size_t)s * m + row];
        rowcnt[row] = t;
    }
}

__global__ __launch_bounds__(256) void smm_runs_slab(int nrows, int m, int n_slabs, int tps, int nct, int wc, int64_t nnzA,
                                                     const int *__restrict__ rowlist, const int *__restrict__ a_ptr,
                                                     const int64_t *__restrict__ list_off, const int *__restrict__ scnt,
                                                     const unsigned *__restrict__ P, const unsigned short *__restrict__ tmp,
                                                     unsigned *__restrict__ dst0, uint2 *__restrict__ runs2,
                                                     unsigned char *__restrict__ tflag, unsigned *__restrict__ err)
{
    __shared__ int win_all[4][RUNS_WIN];
    constexpr int NV = RUNS_WIN / WAVE;
    constexpr int TPS_MAX = 8;                       // tiles per slab (smm_api.hip: tps <= 8)
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    int *win = win_all[threadIdx.x >> 6];
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < nrows; ri += gridDim.x * wpb) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        auto cnt_of = [&](int s, int e) -> unsigned {         // c_s[e] for a valid entry e of this row
            const unsigned *Ps = P + (size_t)s * nnzA;
            const unsigned q1 = e + 1 < a1 ? Ps[e + 1] : (unsigned)scnt[(size_t)s * m + row];
            const unsigned q0 = Ps[e];
            return q1 >= q0 ? q1 - q0 : 0u;     // (a step that ends before it starts is recorded in pass B; a length beyond
                                                //  the list only misplaces dst, which the epilogue checks against the row)
        };
        // A. D[e]: where step e starts in the row of C
        unsigned carry = 0;
        for (int eb = a0; eb < a1; eb += WAVE) {
            const int e = eb + lane;
            int C = 0;
            if (e < a1)
                for (int s = 0; s < n_slabs; ++s) C += (int)cnt_of(s, e);
            const int incl = wave_scan_incl(C);
            if (e < a1) dst0[e] = carry + (unsigned)(incl - C);
            carry += (unsigned)rl(incl, WAVE - 1);
        }
        // B. per slab: tile boundaries inside every step's segment of the slab's list (the list is streamed once
        //    through LDS windows, as in smm_runs), then the table entries of the slab's tiles
        for (int s = 0; s < n_slabs; ++s) {
            const int t0 = s * tps;
            const int tsl = (nct - t0) < tps ? (nct - t0) : tps;             // tiles of this slab
            const unsigned short *__restrict__ list = tmp + list_off[(size_t)s * m + row];
            const unsigned total = (unsigned)scnt[(size_t)s * m + row];
            const unsigned *__restrict__ Ps = P + (size_t)s * nnzA;
            int v[NV];
            auto fetch = [&](unsigned wb) {
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const unsigned i = wb + u * WAVE + lane;
                    v[u] = (int)list[i < total ? i : (total ? total - 1 : 0)];
                }
            };
            unsigned cur = 0xffffffffu;
            if (total) fetch(0);
            for (int eb = a0; eb < a1; eb += WAVE) {
                const int e = eb + lane;
                const bool valid = e < a1;
                unsigned p0 = valid ? Ps[e] : total;
                unsigned p1 = (valid && e + 1 < a1) ? Ps[e + 1] : total;
                if (p0 > total || p1 > total || p1 < p0) {      // always on, see smm_runs
                    plan_err(err, PLAN_ERR_P, row);
                    p0 = p0 < total ? p0 : total;
                    p1 = p1 < p0 ? p0 : (p1 < total ? p1 : total);
                }
                uint2 *r = runs2 + (size_t)t0 * nnzA + (valid ? e : a1 - 1);     // r[tt * nnzA].x = start of tile tt's part
                int tcur = 1;
                if (valid) {
                    r[0].x = p0;
                    if (p0 == p1) { for (; tcur < tsl; ++tcur) r[(size_t)tcur * nnzA].x = p0; }
                } else {
                    tcur = tsl;
                }
                const unsigned reg_lo = rl(p0, 0), reg_hi = rl(p1, WAVE - 1);
                for (unsigned wb = reg_lo - reg_lo % RUNS_WIN; wb < reg_hi; wb += RUNS_WIN) {
                    const unsigned we = wb + RUNS_WIN < total ? wb + RUNS_WIN : total;
                    if (wb != cur) {
                        wave_sync();
#pragma unroll
                        for (int u = 0; u < NV; ++u) win[u * WAVE + lane] = v[u];
                        cur = wb;
                        if (we < total) fetch(we);
                        wave_sync();
                    }
                    unsigned lo = p0 > wb ? p0 : wb;
                    const unsigned hiw = p1 < we ? p1 : we;
                    if (lo < hiw) {
                        while (tcur < tsl) {
                            const int bound = tcur * wc;                 // slab-local column where tile tcur of the slab starts
                            unsigned l = lo, h = hiw;
                            while (l < h) {
                                const unsigned mid = l + ((h - l) >> 1);
                                if (win[mid - wb] < bound) l = mid + 1; else h = mid;
                            }
                            if (l < hiw) { r[(size_t)tcur * nnzA].x = l; lo = l; ++tcur; }
                            else if (p1 <= we) { r[(size_t)tcur * nnzA].x = p1; lo = hiw; ++tcur; }
                            else break;
                        }
                    }
                }
            }
            // the table entries (the starts written above are read back by the lane that wrote them)
            __threadfence_block();
            unsigned any = 0;                                   // bit tt: this lane saw an entry of tile t0 + tt
            for (int eb = a0; eb < a1; eb += WAVE) {
                const int e = eb + lane;
                if (e >= a1) continue;
                unsigned p0 = Ps[e];
                unsigned p1 = e + 1 < a1 ? Ps[e + 1] : total;
                p0 = p0 < total ? p0 : total;                   // (the same clamp as in the pass above)
                p1 = p1 < p0 ? p0 : (p1 < total ? p1 : total);
                unsigned base = dst0[e];
                for (int sp = 0; sp < s; ++sp) base += cnt_of(sp, e);
                uint2 *r = runs2 + (size_t)t0 * nnzA + e;
                unsigned st = r[0].x;
                for (int tt = 0; tt < tsl; ++tt) {
                    const unsigned en = tt + 1 < tsl ? r[(size_t)(tt + 1) * nnzA].x : p1;
                    r[(size_t)tt * nnzA] = make_uint2(st | ((en - st) << 16), base + (st - p0));
                    any |= (en != st ? 1u : 0u) << tt;
                    st = en;
                }
            }
            // which (tile, row) units have anything to emit: the numeric phase skips the others before it touches its tile
            for (int tt = 0; tt < tsl && tt < TPS_MAX; ++tt) {
                const bool some = __ballot((any >> tt) & 1u) != 0ull;
                if (lane == 0) tflag[(size_t)(t0 + tt) * m + row] = some ? 1 : 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Numeric phase.  Workgroup = one (row, coarse tile) unit, tile-major so that concurrent
// units read the same column slab of B; NW waves; LDS = wc f64 accumulators.
//   OUT_DENSE : accumulators start at +0.0 (calloc, sparse_sparse_dense.cpp:97); the tile is
//               written to C[row, lo..hi).
//   OUT_SPARSE: accumulators start at -0.0 (the additive identity: -0.0 + p == p bit for bit,
//               which reproduces `values[index] = p` of sparsework.cpp:108-109).  The epilogue
//               emits the tile's share of the row in the reference's first-touch order: step e
//               of the row put its new columns into slots [runs[e][t], runs[e][t+1]) -- a
//               contiguous sub-run -- so indices and values go out as short contiguous runs
//               straight from LDS (no per-element traffic through L2: a first version that
//               staged the row in an L2-resident scratch row and gathered 8-byte elements from
//               it spent half the kernel on those 2.5e9 L2 requests).
// Two walks fill a tile:
//   EXACT (SMM_EXACT): wave w owns fine tile w and adds the row's products in exactly the
//     reference's order: sums are bit-identical to the CPU loop.  smm_accumulate.
//   default: all waves share the tile and split the work by 64-entry chunks; sums agree to
//     rounding.  smm_accumulate_shared.
// Every load in both walks is unconditional (idle lanes read a dummy word) so that the
// compiler's s_waitcnt distances stay exact -- a predicated load inside the loop degrades
// them to vmcnt(0).
struct NumericArgs {
    int m, ncols, nct, wc, wf, n_ft;
    int64_t row_offset;
    const int *a_ptr, *a_idx; const double *a_val;
    const int *b_idx; const double *b_val;
    const short *b_loc;             // tile-local columns (smm_loc16): the exact walk
    const int2 *tdesc; const double *tpay; int rowsB;   // packed tile-major payload (smm_pack_*): the shared-tile walk
    int piece_epl;                  // 2 / 4: every piece of the payload has <= 128 / 256 entries -> smm_accumulate_pieces; 0: chunk walk
    const int *seg;                 // [rowsB][n_ft+1]
    int kmax;                       // last valid position of b_loc / b_val (exact walk: lanes past a stream's end read it)
    const int *dummy_idx;           // one int  = -1   (read by inactive lanes; its low half is the int16 -1)
    const double *dummy_val;        // one double
    // sparse output
    const int64_t *c_ptr; int *c_idx; double *c_val;
    const int64_t *ub_off; const void *tmp_idx;  // ordered column lists of smm_symbolic (int32, or uint16 when list16)
    int list16;
    const unsigned *runs;           // [nnzA][nct+1] sub-run table (smm_runs), steps before the tail
    const int2 *tail;               // [rows] {first step of the tail, its list position}
    const int *rowlist;             // rows handled by this launch (NULL = all m rows)
    // column slabs (smm_runs_slab): slab-local lists, ub_off indexed [slab * mtot + row]
    const uint2 *runs2; int n_slabs, tps, ws, mtot; int64_t nnzA;
    const unsigned char *tflag;     // [nct][mtot]: 0 = the (tile, row) unit holds no entry of C (smm_runs_slab)
    unsigned long long *stamps;     // diagnostic builds (-DSMM_STAMPS) only: 4 phase totals
    unsigned *err;                  // the context's error word (PLAN_ERR_*)
    unsigned *unit_counter; unsigned n_units;   // persistent workgroups (NULL: one unit per workgroup)
    // dense output
    double *c_dense; int64_t ldc;
};

// Exact walk.  Wave w owns fine tile w of the coarse tile (columns [lo_c + w*wf, +wf)) and is the
// only wave that ever adds into it, and it adds the row's products in exactly the reference's
// order: A's entries in stored order, inside each the entries of B's row in stored order.  Two
// hardware facts make that order cheap to keep:
//   * the LDS executes one wave's ds_add_f64 instructions in issue order;
//   * inside ONE ds_add_f64, lanes that hit the same address are applied in ascending lane order
//     (not promised by the ISA; measured on gfx950 over 6.4e6 random accumulators with 0
//     mismatches -- scripts/ubench/lds_order.hip -- and re-checked by every bit-exact test).
// So the wave does not need one instruction per A entry: the fine-tile segments of 64 A entries
// (about 10 entries of B each) are concatenated into one stream and consumed 64 lanes at a time,
// lane order = stream order.  Which entry j a stream position belongs to is found without a
// search: once per round of 8 chunks every entry drops its index at the stream position where
// its segment starts (a 512-byte LDS scratch), a ballot marks those heads in each chunk, and
// each lane takes the nearest head at or below it
// (positions before the first head belong to the entry that straddles the chunk boundary,
// counted with one more ballot).  Per-entry data (segment base, value of A) is then one LDS read.
constexpr int EX_UNROLL = 8;
struct ExactScratch {          // per wave, in LDS behind the accumulator tile
    int4 tab[WAVE];            // .x = segment start - stream start (k = .x + position), .zw = bits of a
    unsigned char head[WAVE * EX_UNROLL];   // one round of the stream: entry index + 1 where a segment starts
};

template <bool SYM>
__device__ __forceinline__ void smm_accumulate(const NumericArgs &A, double *__restrict__ acc, ExactScratch *__restrict__ sc,
                                               double *__restrict__ sink, const int thresh, const int a0, const int a1, const int ft)
{
    const int lane = lane_id();
    const size_t per = (size_t)A.n_ft + 1;
    const int *__restrict__ segf = A.seg + ft;
    const short *__restrict__ bi = A.b_loc;
    const double *__restrict__ bv = A.b_val;
    const unsigned long long le_mask = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= this one
    const unsigned acc_a = lds_addr(acc), sink_a = lds_addr(sink) + 8u * (unsigned)lane;

    auto load_a = [&](int jb, int &r, double &av) {
        int e = jb + lane;
        e = e < a1 ? e : a1 - 1;
        r = A.a_idx[e];
        av = A.a_val[e];
    };
    auto load_seg = [&](int r, int &s, int &en) {
        const int *sp = segf + (size_t)r * per;
        s = sp[0];
        en = sp[1];
    };

    int r_c, r_n, s_c, e_c;
    double a_c, a_n;
    load_a(a0, r_c, a_c);
    load_a(a0 + WAVE, r_n, a_n);
    load_seg(r_c, s_c, e_c);
    for (int jb = a0; jb < a1; jb += WAVE) {
        const int rem = a1 - jb;
        const int nb = rem < WAVE ? rem : WAVE;
        int s_n, e_n, r_nn;
        double a_nn;
        load_seg(r_n, s_n, e_n);                       // segments of the NEXT 64 entries
        load_a(jb + 2 * WAVE, r_nn, a_nn);             // A entries two batches ahead
        // stream of this batch: entry j owns positions [first_j, first_j + len_j)
        const int len = lane < nb ? e_c - s_c : 0;
        const int incl = wave_scan_incl(len);
        const int first = incl - len;
        const int total = rl(incl, WAVE - 1);
        sc->tab[lane] = make_int4(s_c - first, 0, __double2loint(a_c), __double2hiint(a_c));
        wave_sync();
        for (int g0 = 0; g0 < total; g0 += WAVE * EX_UNROLL) {
            int c[EX_UNROLL], kk[EX_UNROLL];
            double v[EX_UNROLL], a[EX_UNROLL];
            // heads of this round (EX_UNROLL chunks): entries whose segment starts inside it
            wave_sync();
            ((unsigned long long *)sc->head)[lane] = 0ull;
            wave_sync();
            if (len > 0 && first >= g0 && first < g0 + WAVE * EX_UNROLL) sc->head[first - g0] = (unsigned char)(lane + 1);
            wave_sync();
#pragma unroll
            for (int u = 0; u < EX_UNROLL; ++u) {       // map + load EX_UNROLL chunks ...
                const int gbase = g0 + u * WAVE;        // wave-uniform
                const int g = gbase + lane;
                const int hd = sc->head[u * WAVE + lane];
                const unsigned long long heads = __ballot(hd != 0);
                // entry that owns the chunk's first position when no head precedes a lane
                const int carry = (int)__popcll(__ballot(incl <= gbase));
                const unsigned long long mine = heads & le_mask;
                const int hl = 63 - __clzll((long long)(mine | 1ull));          // nearest head at or below
                int j = __shfl(hd, hl) - 1;
                if (mine == 0ull) j = carry;
                j = j < WAVE ? j : WAVE - 1;
                const int4 t = sc->tab[j];
                const int k = t.x + g;                  // positions past the end read the operand's last entry
                kk[u] = k < A.kmax ? k : A.kmax;
                a[u] = __hiloint2double(t.w, t.z);
            }
            // Hand-issued loads and adds (see gload_* above): all 2 x EX_UNROLL loads back to back, then the adds
            // in stream order.  Branch-free: a lane with nothing to add (past the end of the stream, or left of
            // the diagonal) adds -0.0 -- neutral for every accumulator value -- into its slot of the sink.  (With
            // `if (keep) add` the compiler sinks the value load of the round's first chunk into the branch.)
#pragma unroll
            for (int u = 0; u < EX_UNROLL; ++u) {
                gload_sshort(c[u], bi + kk[u]);
                gload_f64(v[u], bv + kk[u]);
            }
#pragma unroll
            for (int u = 0; u < EX_UNROLL; ++u) {
                wait_vm_pair(c[u], v[u], 2 * (EX_UNROLL - 1 - u));
                const bool keep = (g0 + u * WAVE + lane) < total && c[u] >= thresh;
                const double prod = a[u] * v[u];
                lds_add_asm(keep ? acc_a + 8u * (unsigned)c[u] : sink_a, keep ? prod : -0.0);
            }
        }
        wave_sync();
        s_c = s_n; e_c = e_n; a_c = a_n;
        r_n = r_nn; a_n = a_nn;
    }
}

// Shared-tile walk (default mode).  All NW waves of the workgroup add into the SAME coarse
// tile: the segments of 64 A entries are cut into 64-lane chunks, chunk t goes to wave
// t mod NW, and ds_add_f64 (an LDS atomic) makes concurrent adds safe.  Lanes stay ~full
// (a 12k-column tile sees ~125 entries of every row of B: chunks of 64 + 61), 16 waves per CU
// hide both memory and ALU latency, and nothing depends on the segment length.  What is
// given up is the ORDER in which one accumulator receives its products, so values agree with
// the reference to rounding (a few ulp; tests hold them to the north star's 1e-10) instead
// of bit for bit.  SMM_EXACT selects smm_accumulate above instead.
// (Measured dead end: the EXEC-masked, scalar-base chunk loads of smm_symbolic applied here made the
// kernel slower, 29.5 -> 30.6 ms: it is bound by the gather, not by instruction issue, and the scalar
// address arithmetic between the loads spreads their issue out.)
#ifndef SMM_WIDE_CHUNKS
#define SMM_WIDE_CHUNKS 1
#endif
// Chunk loads per wave and round: 8 where the epilogue emits CSR, 4 where the tile goes to a dense row (measured
// with 128-entry chunks: configs[1] 25.75 / 25.9 ms at 8 / 4, configs[2] 20.4 / 20.05, stage 1 of configs[3]
// 19.8 / 18.6).  -DSMM_CH_UNROLL=n forces one value for both.
#ifdef SMM_CH_UNROLL
constexpr int CH_UNROLL_SPARSE = SMM_CH_UNROLL, CH_UNROLL_DENSE = SMM_CH_UNROLL;
#else
constexpr int CH_UNROLL_SPARSE = 8, CH_UNROLL_DENSE = 4;
#endif

template <bool SYM, int NW, int CH_UNROLL>
__device__ __forceinline__ void smm_accumulate_shared(const NumericArgs &A, double *__restrict__ acc, const int lo_c,
                                                      const int thresh, const int a0, const int a1, const int tc,
                                                      const int wave)
{
    const int lane = lane_id();
    const int2 *__restrict__ desc = A.tdesc + (size_t)tc * A.rowsB;
    const double *__restrict__ pay = A.tpay;
    const short *__restrict__ dummy_c = (const short *)A.dummy_idx;

    // Wave w takes the A entries w, w+NW, ... of a round of 64*NW entries (the whole row when it
    // has <= 1024 entries): it alone loads their metadata (no NW-fold redundancy), cuts their
    // segments into 64-lane chunks with one scan, and then issues CH_UNROLL chunk loads at a time.
    for (int rb = a0; rb < a1; rb += NW * WAVE) {
        const int e = rb + wave + NW * lane;
        const bool ev = e < a1;
        const int ec = ev ? e : a1 - 1;
        const int r = A.a_idx[ec];
        const double av = A.a_val[ec];
        const int2 d = desc[r];                     // {first 8-byte unit of the piece, entries}
        const int s_l = d.x;
        const int n_l = ev ? d.y : 0;
#if SMM_WIDE_CHUNKS
        // chunks of 128 entries: one 16-byte load (two values) and one 4-byte load (two columns) per lane
        typedef double dpair __attribute__((ext_vector_type(2), aligned(8)));
        const int nch = (n_l + 2 * WAVE - 1) >> 7;
        const int incl = wave_scan_incl(nch);
        const int total = rl(incl, WAVE - 1);
        for (int t0 = 0; t0 < total; t0 += CH_UNROLL) {
            int c[CH_UNROLL], own[CH_UNROLL];
            dpair v[CH_UNROLL];
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) {       // every load first ...
                const int t = t0 + u;
                int i = (int)__popcll(__ballot(incl <= t));
                i = i < WAVE ? i : WAVE - 1;
                own[u] = i;
                const int first = rl(incl, i) - rl(nch, i);
                const int s = rl(s_l, i), n = rl(n_l, i);
                const int k = ((t - first) << 7) + 2 * lane;
                const bool p = t < total && k < n;
                // a last, single entry reads one element past the values (the piece's own column block) and one
                // past the columns (their padding): both inside the piece, the second half is dropped below
                const dpair *vp = reinterpret_cast<const dpair *>(p ? pay + s + k : A.dummy_val);
                const int *ip = reinterpret_cast<const int *>(p ? (const short *)(pay + s + n) + k : dummy_c);
                int cc = *ip;
                if (!(t < total && k + 1 < n)) cc |= (int)0xffff0000;       // no second entry: column -1
                c[u] = cc;
                v[u] = *vp;
            }
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) {       // ... then the adds
                const int c0 = (int)(short)(c[u] & 0xffff), c1 = c[u] >> 16;
                const double a = rl(av, own[u]);
                if (c0 >= thresh) lds_add(&acc[c0], a * v[u].x);
                if (c1 >= thresh) lds_add(&acc[c1], a * v[u].y);
            }
        }
#else
        const int nch = (n_l + WAVE - 1) >> 6;
        const int incl = wave_scan_incl(nch);
        const int total = rl(incl, WAVE - 1);
        for (int t0 = 0; t0 < total; t0 += CH_UNROLL) {
            int c[CH_UNROLL], own[CH_UNROLL];
            double v[CH_UNROLL];
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) {       // every load first ...
                const int t = t0 + u;
                int i = (int)__popcll(__ballot(incl <= t));
                i = i < WAVE ? i : WAVE - 1;
                own[u] = i;
                const int first = rl(incl, i) - rl(nch, i);
                const int s = rl(s_l, i), n = rl(n_l, i);
                const int k = ((t - first) << 6) + lane;
                const bool p = t < total && k < n;
                const double *vp = p ? pay + s + k : A.dummy_val;
                const short *ip = p ? (const short *)(pay + s + n) + k : dummy_c;
                c[u] = *ip;
                v[u] = *vp;
            }
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u)         // ... then the adds
                if (c[u] >= thresh) lds_add(&acc[c[u]], rl(av, own[u]) * v[u]);
        }
#endif
    }
}

// Piece walk (round 3, default mode).  Where every piece of the packed payload has at most 64 * EPL entries (EPL = 2
// or 4 entries per lane: 128 / 256; the longest piece is known when the payload is built) a wave does not need the
// chunk list at all: ONE piece per iteration, lane l takes its entries EPL*l .. EPL*l + EPL-1 -- one or two 16-byte
// loads of values and one 4- or 8-byte load of columns, all three with a scalar base and a constant per-lane offset
// under an EXEC mask of ceil(n / EPL) lanes -- then EPL multiply-adds into the tile.  No scan, no owner search, no
// per-lane predicate: the columns behind a piece's last entry are the tile's SINK accumulator (smm_pack_fill pads
// the column block with it), so whatever a lane multiplies there lands where nobody reads.  Per piece of ~170
// entries that is ~12 vector and ~12 scalar instructions against ~80 + 40 of the chunk walk; bytes and order of
// issue of the gather are the same.  U pieces are in flight per wave and round.
typedef double dpair_t __attribute__((ext_vector_type(2), aligned(8)));
typedef int ipair_t __attribute__((ext_vector_type(2)));
#ifndef SMM_PIECE_U_SPARSE
#define SMM_PIECE_U_SPARSE 4
#endif
#ifndef SMM_PIECE_U_DENSE
#define SMM_PIECE_U_DENSE 2
#endif
__device__ __forceinline__ void wait_vm_n(int n) {
#define SMM_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        SMM_W(0) SMM_W(1) SMM_W(2) SMM_W(3) SMM_W(4) SMM_W(5) SMM_W(6) SMM_W(7)
        SMM_W(8) SMM_W(9) SMM_W(10) SMM_W(11) SMM_W(12) SMM_W(13) SMM_W(14) SMM_W(15)
        SMM_W(16) SMM_W(17) SMM_W(18) SMM_W(19) SMM_W(20) SMM_W(21) SMM_W(22) SMM_W(23)
        SMM_W(24) SMM_W(25) SMM_W(26) SMM_W(27) SMM_W(28) SMM_W(29) SMM_W(30) SMM_W(31)
    }
#undef SMM_W
}
template <bool SYM, int NW, int EPL, int U>
__device__ __forceinline__ void smm_accumulate_pieces(const NumericArgs &A, double *__restrict__ acc, const int thresh,
                                                      const int a0, const int a1, const int tc, const int wave)
{
    static_assert(EPL == 2 || EPL == 4, "entries per lane");
    constexpr int LPP = EPL == 4 ? 3 : 2;                   // loads per piece
    const int lane = lane_id();
    const int2 *__restrict__ desc = A.tdesc + (size_t)tc * A.rowsB;
    const double *__restrict__ pay = A.tpay;
    const unsigned acc_a = lds_addr(acc);
    const int sink = (A.wc + 1) & ~1;
    const int voff = lane * (EPL * 8), coff = lane * (EPL * 2);
    for (int rb = a0; rb < a1; rb += NW * WAVE) {
        const int e = rb + wave + NW * lane;
        const bool ev = e < a1;
        const int ec = ev ? e : a1 - 1;
        const int r = A.a_idx[ec];
        const double av = A.a_val[ec];
        const int2 d = desc[r];                             // {first 8-byte unit of the piece, entries}
        const int s_l = d.x, n_l = ev ? d.y : 0;
        int cnt = (a1 - rb - wave + NW - 1) / NW;           // this wave's entries of the round (wave-uniform)
        cnt = cnt < 0 ? 0 : (cnt > WAVE ? WAVE : cnt);
        for (int i0 = 0; i0 < cnt; i0 += U) {
            dpair_t v0[U], v1[U];
            ipair_t cc[U];
            unsigned long long mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                   // every load of the round first
                const int i = i0 + u < cnt ? i0 + u : cnt - 1;
                const int sp = rl(s_l, i);
                const int n = i0 + u < cnt ? rl(n_l, i) : 0;
                const int nl = (n + EPL - 1) / EPL;         // lanes with entries
                mk[u] = nl >= WAVE ? ~0ull : (1ull << nl) - 1ull;
                // (an empty piece or a dead slot still loads with lane 0 -- from inside the payload, never added -- so that
                // every slot counts exactly LPP loads and the counted waits below hold)
                const unsigned long long lm = mk[u] | 1ull;
                const double *vb = pay + sp;
                const void *cb = (const void *)(pay + sp + n);
                if constexpr (EPL == 4)
                    asm volatile("s_mov_b64 exec, %7\n\t"
                                 "global_load_dwordx4 %0, %3, %5\n\t"
                                 "global_load_dwordx4 %1, %3, %5 offset:16\n\t"
                                 "global_load_dwordx2 %2, %4, %6\n\t"
                                 "s_mov_b64 exec, -1"
                                 : "=&v"(v0[u]), "=&v"(v1[u]), "=&v"(cc[u])      // early-clobber: no output may share a register with voff / coff,
                                 : "v"(voff), "v"(coff), "s"(vb), "s"(cb), "s"(lm) : "memory");   // which the later loads of the block still read
                else
                    asm volatile("s_mov_b64 exec, %6\n\t"
                                 "global_load_dwordx4 %0, %2, %4\n\t"
                                 "global_load_dword %1, %3, %5\n\t"
                                 "s_mov_b64 exec, -1"
                                 : "=&v"(v0[u]), "=&v"(cc[u].x)
                                 : "v"(voff), "v"(coff), "s"(vb), "s"(cb), "s"(lm) : "memory");
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                wait_vm_n(LPP * (U - 1 - u));
                if constexpr (EPL == 4) asm volatile("" : "+v"(v0[u]), "+v"(v1[u]), "+v"(cc[u]));
                else asm volatile("" : "+v"(v0[u]), "+v"(cc[u].x));
                const int i = i0 + u < cnt ? i0 + u : cnt - 1;
                const double a = rl(av, i);
                int c0 = (int)(short)(cc[u].x & 0xffff), c1 = cc[u].x >> 16;
                if (SYM) { c0 = c0 >= thresh ? c0 : sink; c1 = c1 >= thresh ? c1 : sink; }
                const unsigned d0 = acc_a + 8u * (unsigned)c0, d1 = acc_a + 8u * (unsigned)c1;
                const double p0 = a * v0[u].x, p1 = a * v0[u].y;
                if constexpr (EPL == 4) {
                    int c2 = (int)(short)(cc[u].y & 0xffff), c3 = cc[u].y >> 16;
                    if (SYM) { c2 = c2 >= thresh ? c2 : sink; c3 = c3 >= thresh ? c3 : sink; }
                    const unsigned d2 = acc_a + 8u * (unsigned)c2, d3 = acc_a + 8u * (unsigned)c3;
                    const double p2 = a * v1[u].x, p3 = a * v1[u].y;
                    asm volatile("s_mov_b64 exec, %8\n\t"
                                 "ds_add_f64 %0, %1\n\tds_add_f64 %2, %3\n\tds_add_f64 %4, %5\n\tds_add_f64 %6, %7\n\t"
                                 "s_mov_b64 exec, -1"
                                 :: "v"(d0), "v"(p0), "v"(d1), "v"(p1), "v"(d2), "v"(p2), "v"(d3), "v"(p3), "s"(mk[u]) : "memory");
                } else {
                    asm volatile("s_mov_b64 exec, %4\n\t"
                                 "ds_add_f64 %0, %1\n\tds_add_f64 %2, %3\n\t"
                                 "s_mov_b64 exec, -1"
                                 :: "v"(d0), "v"(p0), "v"(d1), "v"(p1), "s"(mk[u]) : "memory");
                }
            }
        }
    }
    wait_lgkm0();                                           // the hand-issued ds_add's (the compiler does not count them)
}

#ifndef SMM_EPI_UNROLL
#define SMM_EPI_UNROLL 8
#endif
// Streams that are touched once (the result, the ordered lists) carry the non-temporal hint so that they
// do not push B -- re-read by every row -- out of the Infinity Cache.  -DSMM_NT=0 builds without it.
#ifndef SMM_NT
#define SMM_NT 1
#endif
template <typename T> __device__ __forceinline__ void st_stream(T *p, T v) {
#if SMM_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) {
#if SMM_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
constexpr int EPI_UNROLL = SMM_EPI_UNROLL;

// SCR: the tile is not accumulated here but loaded from the dense scratch rows smm_dense_slab left
// in column order (A.c_dense / A.ldc, one scratch row per row of the launch); the kernel is then only
// the first-touch emission.
// L16: the ordered lists are uint16 (a template parameter, not a run-time switch: a branch inside the epilogue's
// "all loads first" loop made every load wait for the one before it -- 30 -> 37 ms).
template <int OUT, bool SYM, int NW, bool EXACT, bool SCR = false, bool L16 = false, bool SLAB = false>
__device__ __forceinline__ void smm_numeric_unit(const NumericArgs &A, double *__restrict__ acc, const unsigned unit_index)
{
    static_assert(!SCR || OUT == OUT_SPARSE, "the scratch source feeds the CSR emission only");
    static_assert(!SLAB || (OUT == OUT_SPARSE && L16 && !SCR), "slab-local lists: CSR output, 16-bit lists");
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int NT = NW * 64;
    const double zero = OUT == OUT_SPARSE ? -0.0 : 0.0;
#ifdef SMM_STAMPS
    // Diagnostic build only (never the shipped library): thread 0 sums the cycles of each phase
    // and adds them to a buffer nothing else reads.
    unsigned long long t_init = 0, t_acc = 0, t_epi = 0, t_mark;
#define SMM_MARK() (t_mark = __builtin_readcyclecounter())
#define SMM_LAP(var) { const unsigned long long now_ = __builtin_readcyclecounter(); var += now_ - t_mark; t_mark = now_; }
#else
#define SMM_MARK()
#define SMM_LAP(var)
#endif
    const int tc = (int)(unit_index / (unsigned)A.m);     // tile-major: concurrent units share B's slab
    const int ridx = (int)(unit_index - (unsigned)tc * (unsigned)A.m);
    const int row = A.rowlist ? A.rowlist[ridx] : ridx;
    // (slab-local lists: operands with structure -- a band, blocks -- leave most tiles of a wide row empty)
    if constexpr (SLAB) { if (!A.tflag[(size_t)tc * A.mtot + row]) return; }
    const int a0 = A.a_ptr[row], a1 = A.a_ptr[row + 1];
    const int64_t gi = row + A.row_offset;
    const int lo_c = tc * A.wc;
    const int w = (A.ncols - lo_c) < A.wc ? (A.ncols - lo_c) : A.wc;
    if (w <= 0) return;
    int64_t rs = 0;
    if (OUT == OUT_SPARSE) {
        rs = A.c_ptr[row];
        if (A.c_ptr[row + 1] == rs) return;             // empty row of C (workgroup-uniform)
    }
    // a tile entirely left of the diagonal holds nothing under SMM_SYMMETRIC
    const bool below = SYM && ((int64_t)lo_c + A.wc <= gi);
    if (OUT == OUT_SPARSE && below) return;
    // the walks see tile-local columns: keep c >= gi - lo_c (also drops the idle lanes' -1)
    int thresh = 0;
    if (SYM) { const int64_t d = gi - lo_c; thresh = d < 0 ? 0 : (d > 32767 ? 32767 : (int)d); }

    SMM_MARK();
    if (SCR) {
        const double *__restrict__ src = A.c_dense + (int64_t)ridx * A.ldc + lo_c;
        for (int x = threadIdx.x; x < w; x += NT) acc[x] = src[x];
        SMM_LAP(t_init);
    } else {
        for (int x = threadIdx.x; x < A.wc; x += NT) acc[x] = zero;
        if (NW > 1) __syncthreads();
        SMM_LAP(t_init);
        if (!below && a1 > a0) {
            if (EXACT) {
                ExactScratch *scr = (ExactScratch *)(acc + ((A.wc + 1) & ~1) + 2);      // (+ the sink accumulators of the piece walk)
                smm_accumulate<SYM>(A, acc, scr + wave, (double *)(scr + NW), thresh, a0, a1, tc * NW + wave);
                wait_lgkm0();                   // its hand-issued ds_add's (the compiler does not count them)
            }
            else if (A.piece_epl == 4) smm_accumulate_pieces<SYM, NW, 4, (OUT == OUT_DENSE ? SMM_PIECE_U_DENSE : SMM_PIECE_U_SPARSE)>(A, acc, thresh, a0, a1, tc, wave);
            else if (A.piece_epl == 2) smm_accumulate_pieces<SYM, NW, 2, (OUT == OUT_DENSE ? SMM_PIECE_U_DENSE : SMM_PIECE_U_SPARSE)>(A, acc, thresh, a0, a1, tc, wave);
            else       smm_accumulate_shared<SYM, NW, (OUT == OUT_DENSE ? CH_UNROLL_DENSE : CH_UNROLL_SPARSE)>(A, acc, lo_c, thresh, a0, a1, tc, wave);
        }
    }
    if (NW > 1) __syncthreads();
    SMM_LAP(t_acc);

    if (OUT == OUT_DENSE) {
        double *__restrict__ dst = A.c_dense + (int64_t)row * A.ldc + lo_c;
        for (int x = threadIdx.x; x < w; x += NT) st_stream(&dst[x], acc[x]);
    } else if constexpr (SLAB) {
        // Epilogue over slab-local lists (smm_runs_slab): step e's columns of this tile are the entries
        // [src, src + len) of the slab's list and go to [dst, dst + len) of the row of C.  Same chunking as below.
        const int sl_ = tc / A.tps;
        const unsigned short *__restrict__ list = (const unsigned short *)A.tmp_idx + A.ub_off[(size_t)sl_ * A.mtot + row];
        const uint2 *__restrict__ rt = A.runs2 + (size_t)tc * A.nnzA;
        const int cshift = sl_ * A.ws;                       // slab-local column -> global column
        const int ashift = cshift - lo_c;                    // slab-local column -> accumulator of this tile
        int *__restrict__ oi = A.c_idx + rs;
        double *__restrict__ ov = A.c_val + rs;
        const unsigned rowlen = (unsigned)(A.c_ptr[row + 1] - rs);
        for (int rb = a0; rb < a1; rb += NW * WAVE) {
            const int e = rb + wave + NW * lane;
            const bool ev = e < a1;
            const uint2 d = rt[ev ? e : a1 - 1];
            const unsigned src0 = d.x & 0xffffu, dst0 = d.y;
            unsigned len = ev ? d.x >> 16 : 0u;
            // (always on: a sub-run must lie inside the row of C and inside one tile; a table that says otherwise costs
            // this row its entries, never a store outside it -- the list side is covered by the slack behind the lists)
#if SMM_CLAMPS
            if (len > (unsigned)A.wc || dst0 > rowlen || len > rowlen - dst0) { plan_err(A.err, PLAN_ERR_RUNS2, row); len = 0u; }
#endif
            const int nch = (int)((len + WAVE - 1) >> 6);
            const int incl = wave_scan_incl(nch);
            const int total = rl(incl, WAVE - 1);
            for (int t0 = 0; t0 < total; t0 += EPI_UNROLL) {
                int c[EPI_UNROLL];
                unsigned dl[EPI_UNROLL];
#pragma unroll
                for (int u = 0; u < EPI_UNROLL; ++u) {          // all chunk loads of the round first
                    const int t = t0 + u;
                    int i = (int)__popcll(__ballot(incl <= t));
                    i = i < WAVE ? i : WAVE - 1;
                    const int first = rl(incl, i) - rl(nch, i);
                    const unsigned off = ((unsigned)(t - first) << 6) + (unsigned)lane;
                    const bool p = t < total && off < rl(len, i);
                    dl[u] = rl(dst0, i) + off;
                    const unsigned short *ip = p ? list + rl(src0, i) + off : (const unsigned short *)A.dummy_idx;
                    c[u] = (int)ld_stream(ip);
                }
#pragma unroll
                for (int u = 0; u < EPI_UNROLL; ++u)        // idle lanes read the dummy word: 0xffff, no slab-local column (ws <= 63 456)
                    if (c[u] != 0xffff) { st_stream(&oi[dl[u]], c[u] + cshift); st_stream(&ov[dl[u]], acc[c[u] + ashift]); }
            }
        }
    } else {
        // Epilogue.  Step e of the row put its new columns of this tile into the contiguous slots
        // [runs[e][tc], runs[e][tc+1]).  The sub-runs are cut into 64-lane chunks; wave w takes the
        // entries w, w+NW, ... of a round of 64*NW entries (long sub-runs belong to the first steps,
        // so interleaving balances them), and for its <= 64 entries it finds the chunk list with one
        // scan, issues EPI_UNROLL chunk loads at a time and then stores.  The whole epilogue of a
        // unit costs a handful of memory round trips (an earlier version walked the entries 64 at a
        // time, one dependent round trip per batch and per long sub-run, and took 37 % of the kernel).
        using LT = std::conditional_t<L16, unsigned short, int>;
        const LT *__restrict__ list = (const LT *)A.tmp_idx + A.ub_off[row];
        int *__restrict__ oi = A.c_idx + rs;
        double *__restrict__ ov = A.c_val + rs;
        const size_t per = (size_t)A.nct + 1;
        const unsigned rowlen = (unsigned)(A.c_ptr[row + 1] - rs);
        const int2 tl = A.tail[row];
        int e0 = tl.x;
        unsigned tail0 = (unsigned)tl.y;
        // (always on: the tail descriptor and the sub-run bounds are smm_runs' output.  Out of range they would make
        // r1 - r0 wrap to 2^26 chunks -- a hang -- or place stores past the row: clamped, and recorded)
#if SMM_CLAMPS
        if (e0 < a0 || e0 > a1 || tail0 > rowlen) {
            if (threadIdx.x == 0) plan_err(A.err, PLAN_ERR_TAIL, row);
            e0 = e0 < a0 ? a0 : (e0 > a1 ? a1 : e0);
            tail0 = rowlen;
        }
#endif
        for (int rb = a0; rb < e0; rb += NW * WAVE) {
            const int e = rb + wave + NW * lane;
            const bool ev = e < e0;
            const unsigned *rp = A.runs + (size_t)(ev ? e : e0 - 1) * per + tc;
            const unsigned r0 = rp[0];
            unsigned r1 = ev ? rp[1] : r0;
#if SMM_CLAMPS
            if (r1 < r0 || r1 > rowlen) { plan_err(A.err, PLAN_ERR_RUNS, row); r1 = r0; }
#endif
            const int nch = (int)((r1 - r0 + WAVE - 1) >> 6);
            const int incl = wave_scan_incl(nch);
            const int total = rl(incl, WAVE - 1);
            for (int t0 = 0; t0 < total; t0 += EPI_UNROLL) {
                int c[EPI_UNROLL];
                unsigned sl[EPI_UNROLL];
#pragma unroll
                for (int u = 0; u < EPI_UNROLL; ++u) {          // all chunk loads of the round first
                    const int t = t0 + u;
                    int i = (int)__popcll(__ballot(incl <= t));
                    i = i < WAVE ? i : WAVE - 1;
                    const int first = rl(incl, i) - rl(nch, i);
                    const unsigned s0 = rl(r0, i), s1 = rl(r1, i);
                    sl[u] = s0 + ((unsigned)(t - first) << 6) + (unsigned)lane;
                    const bool p = t < total && sl[u] < s1;
                    const LT *ip = p ? list + sl[u] : (const LT *)A.dummy_idx;
                    c[u] = (int)ld_stream(ip);
                }
#pragma unroll
                for (int u = 0; u < EPI_UNROLL; ++u)        // idle lanes read the dummy word: -1, or 0xffff as uint16 (no column: B has < 65 535)
                    if (L16 ? c[u] != 0xffff : c[u] >= 0) { st_stream(&oi[sl[u]], c[u]); st_stream(&ov[sl[u]], acc[c[u] - lo_c]); }
            }
        }
        // the tail (see TAIL_MIN): list positions [tl.y, row length) 64 at a time, this tile's columns kept.
        // (the dummy word of idle lanes, -1 or 0xffff, lies in no tile: B has < 65 535 columns when lists are 16-bit)
        const unsigned ntc = rowlen > tail0 ? (rowlen - tail0 + WAVE - 1) >> 6 : 0u;
        for (unsigned q0 = (unsigned)wave; q0 < ntc; q0 += NW * EPI_UNROLL) {
            int c[EPI_UNROLL];
            unsigned ps[EPI_UNROLL];
#pragma unroll
            for (int u = 0; u < EPI_UNROLL; ++u) {
                const unsigned q = q0 + (unsigned)u * NW;
                ps[u] = tail0 + (q << 6) + (unsigned)lane;
                const bool p = q < ntc && ps[u] < rowlen;
                const LT *ip = p ? list + ps[u] : (const LT *)A.dummy_idx;
                c[u] = (int)ld_stream(ip);
            }
#pragma unroll
            for (int u = 0; u < EPI_UNROLL; ++u) {
                const unsigned cc = (unsigned)(c[u] - lo_c);
                if (cc < (unsigned)w) { st_stream(&oi[ps[u]], c[u]); st_stream(&ov[ps[u]], acc[cc]); }
            }
        }
    }
    SMM_LAP(t_epi);
#ifdef SMM_STAMPS
    if (threadIdx.x == 0 && A.stamps) {
        atomicAdd(&A.stamps[0], t_init); atomicAdd(&A.stamps[1], t_acc); atomicAdd(&A.stamps[2], t_epi);
    }
#endif
#undef SMM_MARK
#undef SMM_LAP
}

// The kernel: one (row, tile) unit per workgroup (A.unit_counter == NULL: unit = blockIdx.x), or -- round 4 -- PERSISTENT
// workgroups that take units from a global counter, in order (tile-major order is kept: the counter only ever grows), until
// it passes the last unit: a resident workgroup never gives its CU and its 133 KB of LDS back between two units.
template <int OUT, bool SYM, int NW, bool EXACT, bool SCR = false, bool L16 = false, bool SLAB = false>
__global__ __launch_bounds__(NW * 64) void smm_numeric(const NumericArgs A)
{
    extern __shared__ double acc[];
    if (!A.unit_counter) {
        smm_numeric_unit<OUT, SYM, NW, EXACT, SCR, L16, SLAB>(A, acc, blockIdx.x);
        return;
    }
    // Units are taken in batches of up to 8 (one counter round trip per batch: a unit that turns out to be empty -- a tile
    // left of the diagonal, an empty tile of a banded row -- costs a test, not a round trip), shrinking to 1 towards the
    // end of the launch so that the last workgroups finish together.
    __shared__ unsigned s_unit[2];
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned seen = __hip_atomic_load(A.unit_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned b = (seen < A.n_units ? A.n_units - seen : 0u) / (4u * gridDim.x);
            b = b < 1u ? 1u : (b > 8u ? 8u : b);
            s_unit[0] = atomicAdd(A.unit_counter, b);
            s_unit[1] = b;
        }
        __syncthreads();
        const unsigned u0 = s_unit[0];
        unsigned u1 = u0 + s_unit[1];
        if (u0 >= A.n_units) break;                     // (every workgroup leaves: the counter passes the last unit for each of them)
        u1 = u1 < A.n_units ? u1 : A.n_units;
        for (unsigned unit = u0; unit < u1; ++unit) {
            smm_numeric_unit<OUT, SYM, NW, EXACT, SCR, L16, SLAB>(A, acc, unit);
            __syncthreads();                            // nobody reads the tile or s_unit any more
        }
    }
}

// ---------------------------------------------------------------------------------------
// Row binning after the symbolic phase: rows of C with few nonzeros go to the hash kernels
// below, the rest to the dense-tile kernel above.  lists[b] receives the rows of bin b (order
// irrelevant), counts[b] their number.  bin 0: 1..small_max, bin 1: ..med_max, bin 2: larger.
struct BinSpec { int nb; int thr[6]; int tiny_bin, tiny_bin2; };   // bins 0 .. nb-1: 0 < n <= thr[b] (ascending); bin nb: the rest; tiny bins: see below
template <typename T>
__global__ __launch_bounds__(1024) void smm_bin_rows(int m, const BinSpec spec, const T *__restrict__ rowcnt,
                                                     int *__restrict__ lists, int *__restrict__ counts,
                                                     int tiny_max = 0, const int64_t *__restrict__ tiny_ub = nullptr,
                                                     const int *__restrict__ a_ptr = nullptr)
{
    // bins spec.tiny_bin / tiny_bin2 (round 4): TINY rows -- at most tiny_max (2 x tiny_max) products (tiny_ub) from at most
    // as many entries of A -- which the smm_*_tiny kernels handle four (two) to a wave; the predicate does not depend on
    // rowcnt, so the symbolic and the numeric binning put the same rows there.
    constexpr int NBIN = 8;
    __shared__ int wcnt[16][NBIN], wbase[16][NBIN];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int base = blockIdx.x * 1024; base < m; base += gridDim.x * 1024) {        // workgroup-uniform trip count
        const int row = base + threadIdx.x;
        const T n = row < m ? rowcnt[row] : 0;
        int b = -1;
        if (n > 0) {
            b = spec.nb;
#pragma unroll
            for (int q = 5; q >= 0; --q) if (q < spec.nb && n <= (T)spec.thr[q]) b = q;
        }
        if (tiny_max > 0 && row < m) {
            const int64_t u = tiny_ub[row];
            const int na = a_ptr[row + 1] - a_ptr[row];
            if (u > 0 && u <= tiny_max && na <= tiny_max) b = n > 0 ? spec.tiny_bin : -1;
            else if (u > 0 && u <= 2 * tiny_max && na <= 2 * tiny_max) b = n > 0 ? spec.tiny_bin2 : -1;
        }
        unsigned long long mine = 0ull;
#pragma unroll
        for (int bin = 0; bin < NBIN; ++bin) {
            const unsigned long long mask = __ballot(b == bin);
            if (b == bin) mine = mask;
            if (lane == 0) wcnt[wave][bin] = (int)__popcll(mask);
        }
        __syncthreads();
        if (threadIdx.x < NBIN) {                           // one atomic per workgroup and bin, not per row
            int total = 0;
            for (int w = 0; w < 16; ++w) total += wcnt[w][threadIdx.x];
            int at = total ? atomicAdd(&counts[threadIdx.x], total) : 0;
            for (int w = 0; w < 16; ++w) { wbase[w][threadIdx.x] = at; at += wcnt[w][threadIdx.x]; }
        }
        __syncthreads();
        if (b >= 0) lists[(size_t)b * m + wbase[wave][b] + mbcnt(mine)] = row;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// TINY rows (round 4): at most G products from at most G entries of A; G = 16 lanes per row (four rows per wave) for up to
// 16 products, G = 32 (two rows per wave) for 17 ... 32.
// The hash kernels give every row a whole wave and a chain of ~5 dependent loads with ONE row in flight per wave: at
// 4 entries per row 3 % of the lanes work (30 M rows x 2 entries: 35 + 33 ms).  Here lane p of a row's group IS
// product p (entries of A in order, inside each the row of B in order -- the reference's loop order,
// sparsework.cpp:59-76), so "first touch" needs no marker at all: a column is new iff no EARLIER lane of the group
// carries it, found with G-1 DPP shifts inside the 16-lane row; its slot is the number of new lanes before it, and
// the numeric kernel adds the later lanes of the same column to the first one in ascending order -- the reference's
// order of additions, bit for bit.  B may be unsorted and may repeat columns (the earliest product wins, as in the
// reference).  SYM: columns left of the diagonal are dropped (sparsework.cpp:160-167).
constexpr int TINY_G = 16;         // lanes per row of the first tiny class (one DPP row); the second class takes 32 (two DPP rows)
constexpr int TINY_G2 = 32;
struct TinyMap { int col; int k; int e; bool valid; int nprod; int excl; };
template <bool SYM, int G>
__device__ __forceinline__ TinyMap tiny_map(const int row, const bool have, const int64_t row_offset, const int pl,
                                            const int gbase, const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                            const int *__restrict__ b_ptr, const int *__restrict__ b_idx)
{
    static_assert(G == 16 || G == 32, "one or two DPP rows per group");
    // entry lanes: lane i of the group holds entry a0 + i (bs, len); product lanes: lane p holds product p
    const int a0 = have ? a_ptr[row] : 0;
    const int na = have ? a_ptr[row + 1] - a0 : 0;
    const bool ev = pl < na;
    const int r = ev ? a_idx[a0 + pl] : 0;
    const int bs = ev ? b_ptr[r] : 0;
    int len = ev ? b_ptr[r + 1] - bs : 0;
    len = len < 0 ? 0 : (len > G ? G + 1 : len);               // (a row beyond the class never gets here; clamped all the same)
    // inclusive scan inside the group: the 16-lane row, then (G = 32) lane 15 of the even row into the odd row
    int incl = len;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);
    if constexpr (G == 32) incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    const int excl = incl - len;
    int nprod = __shfl(incl, gbase + G - 1);
    nprod = (nprod > G || na > G) ? -1 : nprod;                 // (never: the class holds rows with <= G products from <= G entries; reported)
    // product p = pl: its entry is the number of entries that end at or before p
    int ej = 0;
#pragma unroll
    for (int sft = G / 2; sft > 0; sft >>= 1)
        if (__shfl(incl, gbase + ej + sft - 1) <= pl) ej += sft;
    const int e_bs = __shfl(bs, gbase + ej), e_ex = __shfl(excl, gbase + ej);
    TinyMap t;
    t.valid = pl < nprod;
    t.k = t.valid ? e_bs + (pl - e_ex) : 0;
    t.e = a0 + ej;
    t.col = t.valid ? b_idx[t.k] : -1;
    if (SYM) { const int64_t gi = row + row_offset; if (t.valid && (int64_t)t.col < gi) { t.valid = false; t.col = -1; } }
    t.nprod = nprod;
    t.excl = excl;
    return t;
}
// dup = an EARLIER lane of the group carries this lane's column: lanes before it in its own 16-lane row (DPP row_shr), and
// (G = 32) for the lanes of the odd row every lane of the even row (ds_bpermute: no DPP form crosses rows by a distance)
template <int D>
__device__ __forceinline__ void tiny_first_row(const int col, const int pl16, bool &dup)
{
    if constexpr (D < 16) {
        const int prev = __builtin_amdgcn_update_dpp(-2, col, 0x110 + D, 0xf, 0xf, false);      // row_shr:D (lanes shifted in keep -2)
        if (pl16 >= D && prev == col) dup = true;
        tiny_first_row<D + 1>(col, pl16, dup);
    }
}
template <int G>
__device__ __forceinline__ bool tiny_dup(const int col, const int pl, const int gbase)
{
    bool dup = false;
    tiny_first_row<1>(col, pl & 15, dup);
    if constexpr (G == 32) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int other = __shfl(col, gbase + j);
            if (pl >= 16 && other == col) dup = true;
        }
    }
    return dup;
}
template <bool SYM, typename IT, int G>
__global__ __launch_bounds__(256) void smm_symbolic_tiny(int nrows, const int *__restrict__ rowlist, int64_t row_offset,
                                                         const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                                         const int *__restrict__ b_ptr, const int *__restrict__ b_idx,
                                                         const int64_t *__restrict__ ub_off, IT *__restrict__ tmp_idx,
                                                         unsigned *__restrict__ P, int *__restrict__ rowcnt, unsigned *__restrict__ err)
{
    constexpr int RW = WAVE / G;                                // rows per wave
    constexpr unsigned GMASK = G == 32 ? 0xffffffffu : 0xffffu;
    const int lane = lane_id(), pl = lane & (G - 1), g = lane / G, gbase = g * G;
    const int wave = (int)(threadIdx.x >> 6), wpb = blockDim.x / WAVE;
    for (int rb = (blockIdx.x * wpb + wave) * RW; rb < nrows; rb += gridDim.x * wpb * RW) {       // wave-uniform
        const int ri = rb + g;
        const bool have = ri < nrows;
        const int row = have ? rowlist[ri] : 0;
        const TinyMap t = tiny_map<SYM, G>(row, have, row_offset, pl, gbase, a_ptr, a_idx, b_ptr, b_idx);
        const bool isnew = t.valid && !tiny_dup<G>(t.col, pl, gbase);
        const unsigned gm = (unsigned)(__ballot(isnew) >> gbase) & GMASK;
        const unsigned below = pl ? (0xffffffffu >> (32 - pl)) : 0u;                    // lanes of the group before this one
        if (have && t.nprod < 0) { if (pl == 0) { plan_err(err, PLAN_ERR_COUNT, row); rowcnt[row] = 0; } }
        else if (have) {
            if (isnew) tmp_idx[ub_off[row] + __popc(gm & below)] = (IT)t.col;
            if (pl == 0) rowcnt[row] = __popc(gm);
            // start slots: the list length when step e starts = new products before the entry's first product
            const int na = a_ptr[row + 1] - a_ptr[row];
            const int ex = t.excl < G ? t.excl : G;
            if (pl < na) P[a_ptr[row] + pl] = (unsigned)__popc(gm & (ex ? (0xffffffffu >> (32 - ex)) : 0u));
        }
    }
}
// the first lane of a column adds the later lanes of that column in ascending lane order = the reference's order of additions
template <int D>
__device__ __forceinline__ void tiny_gather_row(const int col, const int pl16, const bool lead, const double v, double &sum)
{
    if constexpr (D < 16) {
        const int nc = __builtin_amdgcn_update_dpp(-2, col, 0x100 + D, 0xf, 0xf, false);         // row_shl:D: lane pl + D of the row
        const unsigned long long vb = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)vb, 0x100 + D, 0xf, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(vb >> 32), 0x100 + D, 0xf, 0xf, false);
        const double nv = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        if (lead && pl16 + D < 16 && nc == col) sum = sum + nv;
        tiny_gather_row<D + 1>(col, pl16, lead, v, sum);
    }
}
template <bool SYM, int G>
__global__ __launch_bounds__(256) void smm_numeric_tiny(int nrows, const int *__restrict__ rowlist, int64_t row_offset,
                                                        const int *__restrict__ a_ptr, const int *__restrict__ a_idx,
                                                        const double *__restrict__ a_val, const int *__restrict__ b_ptr,
                                                        const int *__restrict__ b_idx, const double *__restrict__ b_val,
                                                        const int64_t *__restrict__ c_ptr, int *__restrict__ c_idx,
                                                        double *__restrict__ c_val, unsigned *__restrict__ err)
{
    constexpr int RW = WAVE / G;
    constexpr unsigned GMASK = G == 32 ? 0xffffffffu : 0xffffu;
    const int lane = lane_id(), pl = lane & (G - 1), g = lane / G, gbase = g * G;
    const int wave = (int)(threadIdx.x >> 6), wpb = blockDim.x / WAVE;
    for (int rb = (blockIdx.x * wpb + wave) * RW; rb < nrows; rb += gridDim.x * wpb * RW) {
        const int ri = rb + g;
        const bool have = ri < nrows;
        const int row = have ? rowlist[ri] : 0;
        const TinyMap t = tiny_map<SYM, G>(row, have, row_offset, pl, gbase, a_ptr, a_idx, b_ptr, b_idx);
        const double v = t.valid ? a_val[t.e] * b_val[t.k] : 0.0;
        const bool isnew = t.valid && !tiny_dup<G>(t.col, pl, gbase);
        double sum = v;                                         // `values[index] = p` (sparsework.cpp:108), then += in product order
        tiny_gather_row<1>(t.col, pl & 15, isnew, v, sum);      // the later lanes of its own 16-lane row ...
        if constexpr (G == 32) {                                // ... then, for a first lane in the even row, the odd row in lane order
            const unsigned long long vb = __builtin_bit_cast(unsigned long long, v);
#pragma unroll
            for (int j = 16; j < 32; ++j) {
                const int oc = __shfl(t.col, gbase + j);
                const unsigned lo = (unsigned)__shfl((int)(unsigned)vb, gbase + j), hi = (unsigned)__shfl((int)(unsigned)(vb >> 32), gbase + j);
                const double ov = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                if (isnew && pl < 16 && oc == t.col) sum = sum + ov;
            }
        }
        const unsigned gm = (unsigned)(__ballot(isnew) >> gbase) & GMASK;
        const unsigned below = pl ? (0xffffffffu >> (32 - pl)) : 0u;
        if (have) {
            const int64_t rs = c_ptr[row];
            // (always on: the symbolic phase counted this row with the same map -- a different count would put stores
            // into the neighbouring rows)
            if ((int64_t)__popc(gm) != c_ptr[row + 1] - rs) { if (pl == 0) plan_err(err, PLAN_ERR_COUNT, row); }
            else if (isnew) {
                const int64_t at = rs + __popc(gm & below);
                c_idx[at] = t.col;
                c_val[at] = sum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Numeric phase for rows with few nonzeros: an LDS hash table column -> slot, where slot is the
// column's position in the row's first-touch list (smm_symbolic), and one f64 accumulator per
// slot.  The dense tile above costs O(columns of B) per row whatever the row holds; this costs
// O(products + nonzeros), and its output is in the reference's order by construction
// (values[slot]), fully coalesced.  It walks whole rows of B, so B need not be sorted.
//   team = NW waves per row (TPB teams per workgroup); HSIZE >= 2 x nonzeros of the row.
//   NW == 1: one wave adds the row's products in the reference's order -> bit-exact values
//   (used for SMM_EXACT and for the small bin); NW > 1: the waves split A's entries (atomics).
struct HashArgs {
    int nrows;                      // rows in rowlist
    int64_t row_offset;
    const int *rowlist;
    const int *a_ptr, *a_idx; const double *a_val;
    const int *b_ptr, *b_idx; const double *b_val;
    const int64_t *c_ptr; int *c_idx; double *c_val;
    const int64_t *ub_off; const void *tmp_idx; int list16;
    const int *dummy_idx; const double *dummy_val;
    unsigned *err;                  // the context's error word (PLAN_ERR_*)
};

constexpr int HASH_UNROLL = 8;

template <bool SYM, int HSIZE, int NW, int TPB>
__global__ __launch_bounds__(NW * TPB * 64) void smm_numeric_hash(const HashArgs A)
{
    constexpr int HBITS = __builtin_ctz(HSIZE);
    constexpr int NVAL = HSIZE / 2;
    __shared__ int keys_s[TPB][HSIZE];
    __shared__ unsigned short slots_s[TPB][HSIZE];
    __shared__ double vals_s[TPB][NVAL];
    const int lane = lane_id();
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int team = wid / NW, wave = wid % NW;
    const int tlane = wave * WAVE + lane;              // lane index inside the team
    constexpr int TN = NW * WAVE;
    int *__restrict__ keys = keys_s[team];
    unsigned short *__restrict__ slots = slots_s[team];
    double *__restrict__ vals = vals_s[team];
    auto team_sync = [&]() { if (NW * TPB > 1) __syncthreads(); else wave_sync(); };
    auto hash = [&](int c) { return (int)(((unsigned)c * 0x9E3779B1u) >> (32 - HBITS)); };

    for (int ri = blockIdx.x * TPB + team; ri - team < A.nrows; ri += gridDim.x * TPB) {
        // (every team of a workgroup runs the same number of iterations: the barriers are workgroup-wide)
        const bool have = ri < A.nrows;
        const int row = have ? A.rowlist[ri] : 0;
        const int64_t rs = have ? A.c_ptr[row] : 0;
        const int cnt = have ? (int)(A.c_ptr[row + 1] - rs) : 0;
        const bool l16 = A.list16 != 0;
        const int64_t lbase = have ? A.ub_off[row] : 0;
        // 1. empty table, accumulators at -0.0 (additive identity: reproduces `values[i] = p`)
        for (int h = tlane; h < HSIZE; h += TN) keys[h] = -1;
        for (int s = tlane; s < cnt && s < NVAL; s += TN) vals[s] = -0.0;
        team_sync();
        // 2. insert the row's columns: slot = position in the first-touch list
        // (always on: every probe sequence is bounded by the table size -- a row count beyond the bin's limit, or a
        // product whose column the symbolic phase did not list, is recorded instead of spinning for ever)
        for (int s = tlane; s < cnt && s < NVAL; s += TN) {
            const int c = list_at(A.tmp_idx, lbase + s, l16);
            int h = hash(c), tries = 0;
            while (atomicCAS(&keys[h], -1, c) != -1 && ++tries < HSIZE) h = (h + 1) & (HSIZE - 1);
            if (tries < HSIZE) slots[h] = (unsigned short)s;
        }
        if (cnt > NVAL && tlane == 0) plan_err(A.err, PLAN_ERR_COUNT, row);
        team_sync();
        // 3. products: wave w of the team takes A's entries w, w+NW, ... (whole rows of B)
        if (have && cnt > 0) {
            const int a0 = A.a_ptr[row], a1 = A.a_ptr[row + 1];
            int thresh = 0;
            if (SYM) { const int64_t gi = row + A.row_offset; thresh = gi > 0x7fffffff ? 0x7fffffff : (int)gi; }
            auto load_a = [&](int jb, int &r, double &av) {
                int e = jb + lane;
                e = e < a1 ? e : a1 - 1;
                r = A.a_idx[e];
                av = A.a_val[e];
            };
            int r_c, r_n;
            double a_c, a_n;
            load_a(a0, r_c, a_c);
            load_a(a0 + WAVE, r_n, a_n);
            int bs = A.b_ptr[r_c], be = A.b_ptr[r_c + 1];
            for (int jb = a0; jb < a1; jb += WAVE) {
                const int rem = a1 - jb;
                const int nb = rem < WAVE ? rem : WAVE;
                const int bs_n = A.b_ptr[r_n], be_n = A.b_ptr[r_n + 1];
                int r_nn;
                double a_nn;
                load_a(jb + 2 * WAVE, r_nn, a_nn);
                int j = wave - NW, kb = 0, en = 0;
                double aj = 0.0;
                auto advance = [&]() {                                  // next own entry with a non-empty row of B
                    do {
                        j += NW;
                        if (j >= nb) break;
                        kb = rl(bs, j);
                        en = rl(be, j);
                        aj = rl(a_c, j);
                    } while (kb >= en);
                };
                advance();
                while (j < nb) {
                    int c[HASH_UNROLL];
                    double v[HASH_UNROLL], a[HASH_UNROLL];
#pragma unroll
                    for (int u = 0; u < HASH_UNROLL; ++u) {             // all loads first
                        const bool live = j < nb;
                        const int k = kb + lane;
                        const bool ok = live && k < en;
                        const int *ip = ok ? A.b_idx + k : A.dummy_idx;
                        const double *vp = ok ? A.b_val + k : A.dummy_val;
                        c[u] = *ip;
                        v[u] = *vp;
                        a[u] = aj;
                        kb += WAVE;
                        if (live && kb >= en) advance();
                    }
#pragma unroll
                    for (int u = 0; u < HASH_UNROLL; ++u) {
                        if (c[u] >= thresh) {                           // also drops the dummy -1
                            int h = hash(c[u]), tries = 0;
                            while (keys[h] != c[u] && ++tries < HSIZE) h = (h + 1) & (HSIZE - 1);
                            if (tries < HSIZE) lds_add(&vals[slots[h]], a[u] * v[u]);
                            else plan_err(A.err, PLAN_ERR_HASH, row);
                        }
                    }
                }
                bs = bs_n; be = be_n; a_c = a_n;
                r_n = r_nn; a_n = a_nn;
            }
        }
        team_sync();
        // 4. the row, in first-touch order
        for (int s = tlane; s < cnt && s < NVAL; s += TN) {
            A.c_idx[rs + s] = list_at(A.tmp_idx, lbase + s, l16);
            A.c_val[rs + s] = vals[s];
        }
        team_sync();
    }
}

// ---------------------------------------------------------------------------------------
// Copy the ordered lists from their capacity-strided slots to the CSR index array (only the
// general path below needs it; smm_numeric emits indices itself).
__global__ __launch_bounds__(256) void smm_copy_lists(int m, const int *__restrict__ rowlist,
                                                      const int64_t *__restrict__ ub_off,
                                                      const int64_t *__restrict__ c_ptr,
                                                      const void *__restrict__ tmp_idx, int list16,
                                                      int *__restrict__ c_idx)
{
    for (int ri = blockIdx.x; ri < m; ri += gridDim.x) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int64_t src = ub_off[row], dst = c_ptr[row];
        const int n = (int)(c_ptr[row + 1] - dst);
        for (int s = threadIdx.x; s < n; s += blockDim.x) c_idx[dst + s] = list_at(tmp_idx, src + s, list16 != 0);
    }
}

// ---------------------------------------------------------------------------------------
// General numeric fallbacks for an operand B whose rows are NOT sorted (tile segments are
// then not contiguous).  One wave per row, the reference's own structure: a column -> slot
// map in HBM (workArray, sparsework.cpp:45) and one global f64 atomic per product.  Values
// agree to rounding (atomics), indices are untouched.  Correctness path, not a fast path.
// ORDERED (SMM_EXACT): the wave that owns the row applies its products strictly in the reference's order
// (sparsework.cpp:59-76) with plain read-modify-write instead of atomics: one 64-entry piece of B's row per step,
// every step's stores acknowledged by L2 before the next step's loads (both bypass the L1), and lanes of ONE step
// that meet on the same accumulator -- a row of B that repeats a column -- go one after the other in ascending lane
// order.  Bit-identical values for any legal CSR operand; a correctness path, not a fast path.
__device__ __forceinline__ double ld_l2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_l2(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ordered_add(double *acc, int *owner, int slot, bool keep, double prod, int lane)
{
    // who else of this step adds to my accumulator?  (last writer of owner[slot] wins; everybody else is a loser)
    if (keep) __hip_atomic_store(&owner[slot], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const bool lone = !keep || __hip_atomic_load(&owner[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lane;
    unsigned long long clash = __ballot(!lone);
    if (clash == 0ull) {                                      // the usual step: all accumulators distinct
        if (keep) st_l2(&acc[slot], ld_l2(&acc[slot]) + prod);
    } else {
        // some accumulator is shared: every lane whose accumulator is shared with ANY other lane of the step takes
        // its turn in ascending lane order (the winners of those accumulators included); the others go together first
        unsigned long long shared = 0ull;
        while (clash) {
            const int x = __ffsll((long long)clash) - 1;
            const int sx = __shfl(slot, x);
            const unsigned long long grp = __ballot(keep && slot == sx);
            shared |= grp; clash &= ~grp;
        }
        if (keep && !((shared >> lane) & 1ull)) st_l2(&acc[slot], ld_l2(&acc[slot]) + prod);
        __threadfence();
        while (shared) {
            const int x = __ffsll((long long)shared) - 1;
            if (lane == x) st_l2(&acc[slot], ld_l2(&acc[slot]) + prod);
            __threadfence();
            shared &= shared - 1ull;
        }
    }
    __threadfence();                                          // the stores have reached L2 before the next step loads
}

template <bool SYM, bool ORDERED = false>
__global__ __launch_bounds__(256) void smm_numeric_general(int m, int ncols, int64_t row_offset,
                                                           const int *__restrict__ rowlist,
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
    // ORDERED: two maps per wave -- column -> slot, and slot -> last lane of the current step (ordered_add)
    int *map = slotmap + ((size_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * (size_t)ncols * (ORDERED ? 2 : 1);
    int *owner = map + ncols;
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < m; ri += gridDim.x * wpb) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int64_t rs = c_ptr[row];
        const int n = (int)(c_ptr[row + 1] - rs);
        const int64_t gi = row + row_offset;
        for (int s = lane; s < n; s += WAVE) { map[c_idx[rs + s]] = s; c_val[rs + s] = -0.0; }
        __threadfence();
        for (int j = a_ptr[row]; j < a_ptr[row + 1]; ++j) {
            const int r = a_idx[j];
            const double a = a_val[j];
            if constexpr (ORDERED) {
                for (int k0 = b_ptr[r]; k0 < b_ptr[r + 1]; k0 += WAVE) {         // wave-uniform trip count
                    const int k = k0 + lane;
                    const bool in = k < b_ptr[r + 1];
                    const int c = in ? b_idx[k] : 0;
                    const bool keep = in && !(SYM && (int64_t)c < gi);
                    ordered_add(c_val + rs, owner, keep ? map[c] : 0, keep, keep ? a * b_val[k] : 0.0, lane);
                }
            } else {
                for (int k = b_ptr[r] + lane; k < b_ptr[r + 1]; k += WAVE) {
                    const int c = b_idx[k];
                    if (SYM && (int64_t)c < gi) continue;
                    glb_add(&c_val[rs + map[c]], a * b_val[k]);
                }
            }
        }
        __threadfence();
    }
}

template <bool SYM, bool ORDERED = false>
__global__ __launch_bounds__(256) void smm_dense_general(int m, int ncols, int64_t row_offset,
                                                         const int *__restrict__ a_ptr,
                                                         const int *__restrict__ a_idx,
                                                         const double *__restrict__ a_val,
                                                         const int *__restrict__ b_ptr,
                                                         const int *__restrict__ b_idx,
                                                         const double *__restrict__ b_val,
                                                         double *__restrict__ c, int64_t ldc, int *__restrict__ owner_all)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    int *owner = ORDERED ? owner_all + ((size_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * (size_t)ncols : nullptr;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        double *dst = c + (int64_t)row * ldc;
        const int64_t gi = row + row_offset;
        for (int x = lane; x < ncols; x += WAVE) dst[x] = 0.0;
        __threadfence();
        for (int j = a_ptr[row]; j < a_ptr[row + 1]; ++j) {
            const int r = a_idx[j];
            const double a = a_val[j];
            if constexpr (ORDERED) {
                for (int k0 = b_ptr[r]; k0 < b_ptr[r + 1]; k0 += WAVE) {
                    const int k = k0 + lane;
                    const bool in = k < b_ptr[r + 1];
                    const int cb = in ? b_idx[k] : 0;
                    const bool keep = in && !(SYM && (int64_t)cb < gi);
                    ordered_add(dst, owner, keep ? cb : 0, keep, keep ? a * b_val[k] : 0.0, lane);
                }
            } else {
                for (int k = b_ptr[r] + lane; k < b_ptr[r + 1]; k += WAVE) {
                    const int cb = b_idx[k];
                    if (SYM && (int64_t)cb < gi) continue;
                    glb_add(&dst[cb], a * b_val[k]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Triple product, stage 2 (sparse_sparse_dense.cpp:201-216): C[i,k] = sum over row k of H of
// T[i, col] * val, for k >= i (or all k), where T = H*Q is the dense n x K matrix stage 1
// (smm_numeric<OUT_DENSE>) left in HBM.  With SMM_EXACT the sum runs in H's stored order starting
// from 0.0, product rounded before the addition, exactly as the reference's scalar loop: stage 2 is
// bit-exact given T (lanes only sit steps out to keep their LDS reads apart, smm_ell_fill<2>).  By
// default the entries of a row inside one chunk are taken in whatever order keeps the LDS gather free
// of bank conflicts (smm_ell_fill<1>) and multiply-add is fused: the same sum, within the 1e-10 of
// the default mode.
//
// Layout.  A workgroup owns a block of R rows of T and a group of NW*64 rows k of H (one k per
// lane, one 64-row slice per wave); its R running sums per lane stay in registers from the
// first column of H to the last.  K is cut into chunks of <= 1024 columns; the R x chunk piece
// of T sits in LDS column-major with a 2-double pad ([chunk][R+2]: the R values of one column
// are one contiguous run, read with ds_read_b128; the pad spreads a wave's random columns over
// all banks -- 1.46x the gather rate of the row-major tile it replaced,
// scripts/ubench/lds_gather.hip).  Chunks are visited in ascending column order = stored order
// of a sorted H, so with the ELL steps in stored order the order of additions is the reference's.
// A lane walking its own CSR row would make every load 64 separate 12-byte requests, so H is
// re-laid once (cached on the handle) as sliced ELL per chunk: for 64 consecutive rows k and
// chunk q, step s of all 64 rows is stored contiguously -- int16 chunk-local column + f64
// value -- padded to the longest of the 64 segments (column -1 = no entry).
// Traffic: the ELL copy of H is streamed once per row block (n/R times; consecutive workgroups
// share one k-group, so it is served by the Infinity Cache), T once per k-group.
struct EllArgs {
    int n, nchunks, chunk, nslices;
    const int *h_ptr, *h_idx; const double *h_val;
    const int *hseg;                  // [n][nchunks+1]
    int64_t *cnt;                     // [nchunks][nslices]  64 * longest segment of the slice
    const int64_t *off;               // exclusive scan of cnt (+ total)
    short *col; double *val;          // ELL payload; col < 0 = no entry in this step
};

// pass 1: per-slice maxima (one wave per (chunk, slice))
__global__ __launch_bounds__(256) void smm_ell_count(const EllArgs A)
{
    const int lane = lane_id();
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    if (item >= (int64_t)A.nchunks * A.nslices) return;
    const int q = (int)(item / A.nslices), sl = (int)(item % A.nslices);
    const int k = sl * WAVE + lane;
    int len = 0;
    if (k < A.n) {
        const int *sp = A.hseg + (size_t)k * (A.nchunks + 1) + q;
        len = sp[1] - sp[0];
    }
    int mx = len;
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    if (lane == 0) A.cnt[item] = (int64_t)mx * WAVE;
}

// pass 2: payload (one wave per (chunk, slice)).  Which entry of its segment a lane meets in which step decides
// how often the wave's LDS reads collide in stage 2: the tile is [column][R + 2] doubles with R = 16, so the
// 16-byte slot of a lane's read j is (9 c + j) mod 16, and the 16 lanes of one ds_read_b128 conflict group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32: MI355X_MICROARCH.md, LDS table) read without a bank
// conflict iff their columns are distinct mod 16.  Entry s in step s costs 2.3 LDS passes per step on random
// columns (scripts/ubench/lds_conflict.hip: 1.65x the gather rate without them).  The block is as long as the
// longest of the 64 segments, so most lanes have steps to spare, and a scheduler uses them: step by step, the
// lanes of a group choose in the order of their spare steps (fewest first); a lane whose candidate's residue
// class is taken sits the step out (column -1) unless it has no step to spare, then it takes the conflict.
//   ORDER 0  entry s in step s (blocks longer than 64 steps, whatever the mode)
//   ORDER 1  default mode: the candidate is ANY entry the lane still holds (lowest free residue class; stored
//            order inside a class) -- the order of additions changes, 1.06 passes per step
//   ORDER 2  SMM_EXACT: the candidate is the NEXT entry in stored order, only idle steps are inserted -- the
//            reference's order of additions, 1.3 passes per step
template <int ORDER>
__global__ __launch_bounds__(256) void smm_ell_fill(const EllArgs A)
{
    __shared__ unsigned char s_ord[4][WAVE][64], s_ptr[4][WAVE][16], s_end[4][WAVE][16];
    const int lane = lane_id();
    const int64_t item = (int64_t)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6);
    if (item >= (int64_t)A.nchunks * A.nslices) return;
    const int q = (int)(item / A.nslices), sl = (int)(item % A.nslices);
    const int k = sl * WAVE + lane;
    int s = 0, len = 0;
    if (k < A.n) {
        const int *sp = A.hseg + (size_t)k * (A.nchunks + 1) + q;
        s = sp[0]; len = sp[1] - sp[0];
    }
    const int64_t base = A.off[item];
    const int steps = (int)((A.off[item + 1] - base) / WAVE);
    const int lo = q * A.chunk;
    if (ORDER == 0 || steps > 64) {
        for (int st = 0; st < steps; ++st) {
            short c = -1; double v = 0.0;
            if (st < len) { c = (short)(A.h_idx[s + st] - lo); v = A.h_val[s + st]; }
            A.col[base + (int64_t)st * WAVE + lane] = c;
            A.val[base + (int64_t)st * WAVE + lane] = v;
        }
        return;
    }
    const int w = threadIdx.x >> 6;
    unsigned char *ord = s_ord[w][lane], *ptr = s_ptr[w][lane], *end = s_end[w][lane];
    unsigned avail = 0;
    if (ORDER == 1) {
        // positions of the segment sorted by column mod 16 (stable): ord[ptr[r] .. end[r])
        for (int r = 0; r < 16; ++r) ptr[r] = 0;
        for (int p = 0; p < len; ++p) ++ptr[(A.h_idx[s + p] - lo) & 15];
        int acc = 0;
        for (int r = 0; r < 16; ++r) { const int c = ptr[r]; ptr[r] = (unsigned char)acc; acc += c; }
        for (int p = 0; p < len; ++p) ord[ptr[(A.h_idx[s + p] - lo) & 15]++] = (unsigned char)p;
        for (int r = 15; r >= 0; --r) {
            end[r] = ptr[r];
            ptr[r] = r ? ptr[r - 1] : 0;
        }
        for (int r = 0; r < 16; ++r) avail |= (ptr[r] < end[r] ? 1u : 0u) << r;
    } else {
        for (int p = 0; p < len; ++p) ord[p] = (unsigned char)((A.h_idx[s + p] - lo) & 15);   // residue class of entry p
    }
    // conflict group and position in it; lane at position i of this lane's group
    const int l32 = lane & 31;
    const int odd = (l32 >= 4 && l32 < 12) || (l32 >= 16 && l32 < 20) || l32 >= 28;
    const int gp = l32 < 4 ? l32 : l32 < 12 ? l32 - 4 : l32 < 20 ? l32 - 8 : l32 < 28 ? l32 - 12 : l32 - 16;
    auto at = [&](int i) { return (lane & 32) + (odd ? (i < 8 ? i + 4 : i < 12 ? i + 8 : i + 16) : (i < 4 ? i : i < 8 ? i + 8 : i + 12)); };
    int rem = len, next = 0;
    for (int st = 0; st < steps; ++st) {
        const int spare = rem > 0 ? steps - st - rem : 255;          // 0: this lane must take an entry in every step left
        const int key = (spare << 4) | gp;
        int rank = 0;
        for (int i = 0; i < 16; ++i) rank += __shfl(key, at(i)) < key ? 1 : 0;
        // the lane at position j of the group learns which lane chooses j-th
        const int chooser = __builtin_amdgcn_ds_permute(at(rank) << 2, lane);
        unsigned used = 0; int pick = -1;
        for (int j = 0; j < 16; ++j) {
            const int who = __shfl(chooser, at(j));
            unsigned bit = 0;
            if (lane == who && rem > 0) {
                if (ORDER == 1) {
                    const unsigned m = avail & ~used;
                    if (m) pick = __ffs(m) - 1;
                    else if (spare == 0) pick = __ffs(avail) - 1;    // no step to spare: take the conflict
                } else {
                    const int r = ord[next];
                    if (!((used >> r) & 1u) || spare == 0) pick = r;
                }
                if (pick >= 0) bit = 1u << pick;
            }
            used |= (unsigned)__shfl((int)bit, who);
        }
        short c = -1; double v = 0.0;
        if (pick >= 0) {
            int p;
            if (ORDER == 1) {
                p = ord[ptr[pick]];
                if (++ptr[pick] == end[pick]) avail &= ~(1u << pick);
            } else {
                p = next++;
            }
            --rem;
            c = (short)(A.h_idx[s + p] - lo); v = A.h_val[s + p];
        }
        A.col[base + (int64_t)st * WAVE + lane] = c;
        A.val[base + (int64_t)st * WAVE + lane] = v;
    }
}

struct TripleArgs {
    int n, K, nchunks, chunk, nslices;
    int nib, nkg, gk;                 // row blocks, k-groups; block order: see smm_triple_stage2
    int64_t row_begin, row_end;
    int full;
    const int64_t *off;               // [nchunks*nslices + 1]
    const short *col; const double *val;
    const double *T;                  // (row_end-row_begin) x K
    double *C; int64_t ldc;           // row row_begin at C
    unsigned long long *stamps;       // diagnostic builds (-DSMM_S2_STAMPS) only: cycles per phase, summed over waves
};

// diagnostic builds (make variant EXTRA=-DSMM_S2_DIAG=n): 1 = no loads of T, 4 = no barriers, 16 = every block
// loads the tiles of row block 0 (L2 hits).  Wrong results, timing only.
#ifndef SMM_S2_DIAG
#define SMM_S2_DIAG 0
#endif
#ifndef SMM_S2_STAGGER
#define SMM_S2_STAGGER 8
#endif
#ifndef SMM_S2_DEPTH
#define SMM_S2_DEPTH 3
#endif
#if SMM_S2_DIAG & 4
#define S2_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define S2_SYNC() __syncthreads()
#endif
#ifdef SMM_S2_STAMPS
#define S2_MARK() (t_mark = __builtin_readcyclecounter())
#define S2_LAP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); t_ph[i] += now_ - t_mark; t_mark = now_; }
#else
#define S2_MARK()
#define S2_LAP(i)
#endif
template <int R, int NW, int CW, bool FMA>
__global__ __launch_bounds__(NW * 64) void smm_triple_stage2(const TripleArgs A)
{
#ifdef SMM_S2_STAMPS
    unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, t_mark;
#endif
    constexpr int RP = NW * 64 / CW;                   // tile rows filled per pass (chunk <= CW columns)
    constexpr int NV = R / RP;                         // tile elements per thread
    static_assert(RP >= 1 && RP * CW == NW * 64 && R % (2 * RP) == 0 && NV == 16, "tile fill: CW threads per row");
    extern __shared__ double tl[];                     // [chunk][R + 2]
    constexpr int LD = R + 2;
    constexpr int D = SMM_S2_DEPTH;                    // steps of H entries in flight
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Block order.  The gk k-groups of one super-group that meet the same row block follow each other on the
    // same XCD (workgroup b runs on XCD b % 8), so the tile of T that the first of them pulls from HBM is in
    // that XCD's L2 for the others; the row blocks of a super-group, 8 at a time over the XCDs, all stream
    // the same gk ELL copies.
    int ib, kg;
    {
        const unsigned x = blockIdx.x & 7u; unsigned t = blockIdx.x >> 3;
        const unsigned nib8 = ((unsigned)A.nib + 7u) >> 3;
        const unsigned g = t % (unsigned)A.gk; t /= (unsigned)A.gk;
        ib = (int)((t % nib8) * 8u + x); kg = (int)((t / nib8) * (unsigned)A.gk + g);
        if (ib >= A.nib || kg >= A.nkg) return;
    }
    const int64_t i0 = A.row_begin + (int64_t)ib * R;
    const int nr = (A.row_end - i0) < R ? (int)(A.row_end - i0) : R;
    const int sl = kg * NW + wave;                     // this wave's 64 rows of H
    const int64_t k = (int64_t)sl * WAVE + lane;
    const bool kin = k < A.n;
    double *crow = A.C + (int64_t)(i0 - A.row_begin) * A.ldc + k;
    // the whole k-group lies left of the diagonal: the reference's calloc'd zeros
    if (!A.full && (int64_t)(kg + 1) * NW * WAVE <= i0) {
        if (kin)
            for (int r = 0; r < nr; ++r) crow[(int64_t)r * A.ldc] = 0.0;
        return;
    }
    const bool work = sl < A.nslices && (A.full || (int64_t)(sl + 1) * WAVE > i0);
    double sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = 0.0;
    // SMM_EXACT: product rounded, then added (the reference's x86-64 build has no fused multiply-add);
    // default: one v_fma_f64 -- half the VALU work of the gather loop, and one rounding less
    auto mad = [](double x, double h, double acc) { return FMA ? __builtin_fma(x, h, acc) : acc + x * h; };
    // Tile fill: thread (fr, fx) owns column fx of tile rows fr, fr + RP, ...  The NV elements of the
    // NEXT chunk travel in registers while the current chunk is consumed (a fill between the two
    // barriers cost 15 % of the kernel).  Rows past the last row of the block and columns past the
    // chunk are clamped (loaded twice, stored never / never read).
    const int fx = threadIdx.x % CW, fr = threadIdx.x / CW;
    const double *trow[1];
#if SMM_S2_DIAG & 16
    trow[0] = A.T;                                     // every block reads the tiles of row block 0: L2 hits
#else
    trow[0] = A.T + (i0 - A.row_begin) * A.K;
#endif
    double v[NV];
#if SMM_S2_DIAG & 1
#define TILE_LD(t, lo_, w_) v[t] = 1.0 + (t) + (lo_) + (w_)
#else
#define TILE_LD(t, lo_, w_)                                                                     \
    v[t] = trow[0][(int64_t)((fr + RP * (t)) < nr ? (fr + RP * (t)) : nr - 1) * A.K + (lo_) +   \
                   (fx < (w_) ? fx : (w_) - 1)]
#endif
    {
        const int w0 = A.K < A.chunk ? A.K : A.chunk;
#pragma unroll
        for (int t = 0; t < NV; ++t) TILE_LD(t, 0, w0);
    }
    // Per-chunk metadata (the slice's block in the ELL arrays) is loaded one chunk ahead, and the first
    // D steps of a chunk are requested before the barrier, so no wave starts a chunk with a chain of
    // dependent round trips.  Waves without work (slice beyond n or left of the diagonal) run the same
    // loads on clamped indices and 0 steps.  A lane has an entry in a step iff its column is >= 0.
    const int slc = sl < A.nslices ? sl : A.nslices - 1;
    int64_t base_n = A.off[slc], end_n = A.off[slc + 1];

    for (int q = 0; q < A.nchunks; ++q) {
        const int lo = q * A.chunk;
        const int w = (A.K - lo) < A.chunk ? (A.K - lo) : A.chunk;
        const int steps = work ? (int)((end_n - base_n) / WAVE) : 0;
        const short *cp = A.col + base_n + lane;
        const double *vp = A.val + base_n + lane;
        // The entries of the next D steps travel in registers (indices are clamped into the slice's block; a
        // clamped step has st >= steps and adds nothing).  vmcnt retires in order, so a wait for an H entry
        // also waits for every tile load issued before it; the NV loads of the NEXT chunk's tile therefore go
        // out in one burst per wave (see below for when).  D = 3 with the waves' bursts spread over 8 step
        // groups measured best (50.7 ms at BASELINE configs[3]; D = 8 / 4 groups 53.6, D = 2 54-55: sweep in
        // profiles/r2_s2_sweeps.txt) -- a short ring leaves the registers to the LDS reads in flight.
        const int last = steps > 0 ? steps - 1 : 0;
        int cc[D]; double hh[D];
        S2_MARK();
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int su = u < last ? u : last;
            cc[u] = cp[(int64_t)su * WAVE]; hh[u] = vp[(int64_t)su * WAVE];
        }
        S2_LAP(0);
        S2_SYNC();                               // nobody reads the previous tile any more
        S2_LAP(1);
        if (fx < w) {
            double *dst = tl + fx * LD + fr;
#pragma unroll
            for (int t = 0; t < NV; ++t) dst[RP * t] = v[t];
        }
#ifdef SMM_S2_STAMPS
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the tile writes have landed
#endif
        S2_LAP(2);
        const int qn = q + 1 < A.nchunks ? q + 1 : q;
        {
            const int mi = __builtin_amdgcn_readfirstlane(qn * A.nslices + slc);    // wave-uniform: scalar loads
            base_n = A.off[mi]; end_n = A.off[mi + 1];
        }
        const int lon = qn * A.chunk;
        const int wn = (A.K - lon) < A.chunk ? (A.K - lon) : A.chunk;
        S2_SYNC();
        S2_LAP(3);
        // The 16 waves' tile loads are 128 KB through a 64 B/clk vector-memory pipe: issued by all waves at once
        // behind the barrier they block every wave at the issue for ~2000 cycles (13 % of the kernel in a stamped
        // build) while the LDS pipe idles.  So the waves take turns: wave w issues its NV loads in front of step
        // group w mod S2_STAGGER, the others gather meanwhile.  (One unconditional site for the loads between two
        // copies of the step loop: with the loads under a condition inside the loop the compiler spilled 99 VGPRs.)
        S2_LAP(4);
#define STEP_GROUP(st_)                                                                         \
        _Pragma("unroll") for (int u = 0; u < D; ++u) {                                         \
            if ((st_) + u < steps && cc[u] >= 0) {                                              \
                const double2 *p = reinterpret_cast<const double2 *>(tl + cc[u] * LD);          \
                _Pragma("unroll") for (int r = 0; r < R; r += 2) {                              \
                    const double2 x = p[r >> 1];                                                \
                    sum[r] = mad(x.x, hh[u], sum[r]);                                           \
                    sum[r + 1] = mad(x.y, hh[u], sum[r + 1]);                                   \
                }                                                                               \
            }                                                                                   \
            const int sn = (st_) + u + D < last ? (st_) + u + D : last;                         \
            cc[u] = cp[(int64_t)sn * WAVE]; hh[u] = vp[(int64_t)sn * WAVE];                     \
        }
        {
            const int ng = steps > D ? (steps + D - 1) / D : 1;        // at least one group: the tile loads must go out
            int tg = wave % SMM_S2_STAGGER;
            tg = tg < ng ? tg : ng - 1;
            int g = 0;
            for (; g < tg; ++g) { STEP_GROUP(g * D) }
#pragma unroll
            for (int t = 0; t < NV; ++t) TILE_LD(t, lon, wn);
            for (; g < ng; ++g) { STEP_GROUP(g * D) }
        }
#undef STEP_GROUP
        S2_LAP(5);
    }
#ifdef SMM_S2_STAMPS
    if (lane == 0 && work && A.stamps)
        for (int i = 0; i < 6; ++i) atomicAdd(&A.stamps[i], t_ph[i]);
#endif
#undef TILE_LD
    if (kin) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < nr) crow[(int64_t)r * A.ldc] = (A.full || k >= i0 + r) ? sum[r] : 0.0;
    }
}

// Stage 2 for an H whose rows are not sorted: one thread per (i, k), the reference's scalar loop
// (sparse_sparse_dense.cpp:203-211) in H's stored order.  Correctness path, not a fast path.
__global__ __launch_bounds__(256) void smm_triple_stage2_general(int n, int K, int64_t row_begin, int64_t row_end, int full,
                                                                 const int *__restrict__ h_ptr, const int *__restrict__ h_idx,
                                                                 const double *__restrict__ h_val, const double *__restrict__ T,
                                                                 double *__restrict__ C, int64_t ldc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int s = h_ptr[k], e = h_ptr[k + 1];
    for (int64_t i = row_begin + blockIdx.y; i < row_end; i += gridDim.y) {
        const double *t = T + (i - row_begin) * K;
        double sum = 0.0;
        if (full || k >= i)
            for (int jp = s; jp < e; ++jp) sum += t[h_idx[jp]] * h_val[jp];
        C[(i - row_begin) * ldc + k] = sum;
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

// Mirror epilogue: the lower triangle of an n x n upper-triangle result becomes the mirror
// image of the upper one (what a caller of symmetric=True / compute_full_matrix=0 needs to use the result
// as a full matrix; the reference's own compute_full_matrix=1 doubles the off-diagonal instead).
// Tiles of 64 x 64 through LDS so that both the read and the write are row-contiguous.
__global__ __launch_bounds__(256) void smm_mirror_upper(int n, double *__restrict__ C, int64_t ldc)
{
    __shared__ double t[64][65];
    const int tiles = (n + 63) / 64;
    // block b -> tile pair (ti <= tj) of the upper triangle, row-major over pairs
    int64_t b = blockIdx.x;
    int ti = 0;
    while (b >= tiles - ti) { b -= tiles - ti; ++ti; }
    const int tj = ti + (int)b;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 4 rows per pass
    for (int r = ty; r < 64; r += 4) {
        const int a = ti * 64 + r, c = tj * 64 + tx;
        t[r][tx] = (a < n && c < n) ? C[(int64_t)a * ldc + c] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int a = tj * 64 + r, c = ti * 64 + tx;            // target (a, c) = source (c, a)
        if (a < n && c < n && a > c) C[(int64_t)a * ldc + c] = t[tx][r];
    }
}

// ---------------------------------------------------------------------------------------
// CSR mirror epilogue (not in the reference, opt-in): an n x n result that holds only i <= col
// (sparsework.cpp:217) becomes the full symmetric matrix, on the device.  Row i of the full matrix is
//     [ the mirrored entries (i, j), j < i, in ASCENDING column order ]  followed by
//     [ the row's own entries (i, col >= i) exactly as the upper result holds them: first-touch order ].
// Three steps: count (one atomic per strictly-upper entry on its target row), fill (own entries copied behind
// the space of the mirrored ones, mirrored ones dropped into it through a per-row cursor: arrival order), and a
// sort of each row's mirrored segment by column, which makes the result deterministic.  The sort runs in LDS
// (one wave per row up to 64 mirrored entries, one workgroup up to MIRROR_MAX_SEG); longer segments -- results as full
// as the BASELINE configs' -- are staged and placed by RANK instead (smm_mirror_rank below): no limit on a row.
constexpr int MIRROR_MAX_SEG = 8192;

__global__ __launch_bounds__(256) void smm_mirror_count(int n, const int64_t *__restrict__ uptr, const int *__restrict__ uidx,
                                                        int *__restrict__ mcnt, unsigned *__restrict__ flags)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    unsigned bad = 0;
    for (int i = blockIdx.x * wpb + (threadIdx.x >> 6); i < n; i += gridDim.x * wpb) {
        for (int64_t k = uptr[i] + lane; k < uptr[i + 1]; k += WAVE) {
            const int j = uidx[k];
            if (j > i && j < n) atomicAdd(&mcnt[j], 1);
            else if (j != i) bad = 1;                       // an entry left of the diagonal, or outside the square
        }
    }
    if (bad) atomicOr(flags, 1u);
}
// full row lengths (for the scan) and the longest mirrored segment
__global__ __launch_bounds__(256) void smm_mirror_rowlen(int n, const int64_t *__restrict__ uptr, const int *__restrict__ mcnt,
                                                         int64_t *__restrict__ flen, int *__restrict__ maxseg)
{
    int mx = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        flen[i] = (int64_t)mcnt[i] + (uptr[i + 1] - uptr[i]);
        mx = mcnt[i] > mx ? mcnt[i] : mx;
    }
    for (int o = 32; o > 0; o >>= 1) { const int y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    if (lane_id() == 0 && mx > 0) atomicMax(maxseg, mx);
}
// big[i] = mirrored entries of row i when they are too many for the LDS sort (they travel through a staging array and
// are RANKED instead: smm_mirror_rank), else 0
__global__ __launch_bounds__(256) void smm_mirror_big(int n, const int *__restrict__ mcnt, int64_t *__restrict__ big)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        big[i] = mcnt[i] > MIRROR_MAX_SEG ? (int64_t)mcnt[i] : 0;
}
// toff / tidx / tval: staging array of the long segments (toff = exclusive scan of big[]; NULL when there is none)
__global__ __launch_bounds__(256) void smm_mirror_fill(int n, const int64_t *__restrict__ uptr, const int *__restrict__ uidx,
                                                       const double *__restrict__ uval, const int64_t *__restrict__ fptr,
                                                       const int *__restrict__ mcnt, int *__restrict__ cursor,
                                                       int *__restrict__ fidx, double *__restrict__ fval,
                                                       const int64_t *__restrict__ toff, int *__restrict__ tidx, double *__restrict__ tval)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int i = blockIdx.x * wpb + (threadIdx.x >> 6); i < n; i += gridDim.x * wpb) {
        const int64_t s = uptr[i], own = fptr[i] + mcnt[i];
        for (int64_t k = s + lane; k < uptr[i + 1]; k += WAVE) {
            const int j = uidx[k];
            const double v = uval[k];
            fidx[own + (k - s)] = j;
            fval[own + (k - s)] = v;
            if (j > i) {
                const int slot = atomicAdd(&cursor[j], 1);
                if (toff && mcnt[j] > MIRROR_MAX_SEG) {
                    const int64_t at = toff[j] + slot;
                    tidx[at] = i; tval[at] = v;
                } else {
                    const int64_t at = fptr[j] + slot;
                    fidx[at] = i; fval[at] = v;
                }
            }
        }
    }
}
// Mirrored segments longer than MIRROR_MAX_SEG (results as full as the BASELINE configs': a row receives up to n
// entries): no comparison sort.  The columns of a segment are DISTINCT integers below i, so the sorted position of an
// entry is its column's rank in the segment = the number of set bits below it in a bitmap of the segment's columns.
// One workgroup per row: bitmap of a range of RANK_WORDS * 32 columns in LDS (atomic OR), per-word exclusive prefix of
// the popcounts (block scan), then every staged entry goes to fptr[i] + (entries below the range) + rank.  Rows wider
// than one range take one pass per range.  Linear in the segment, deterministic, any length.
constexpr int RANK_WORDS = 16384;                 // 64 KB of bitmap + 64 KB of prefixes: 524 288 columns per pass
__global__ __launch_bounds__(1024) void smm_mirror_rank(int n, const int64_t *__restrict__ fptr, const int *__restrict__ mcnt,
                                                        const int64_t *__restrict__ toff, const int *__restrict__ tidx,
                                                        const double *__restrict__ tval, int *__restrict__ fidx, double *__restrict__ fval)
{
    extern __shared__ unsigned rk_lds[];
    unsigned *bm = rk_lds, *pre = rk_lds + RANK_WORDS;
    __shared__ unsigned wsum[16];
    __shared__ unsigned below_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int len = mcnt[i];                                       // workgroup-uniform
        if (len <= MIRROR_MAX_SEG) continue;
        const int *__restrict__ tk = tidx + toff[i];
        const double *__restrict__ tv = tval + toff[i];
        const int64_t dst = fptr[i];
        for (int lo = 0; lo < i; lo += RANK_WORDS * 32) {
            const int span = (i - lo) < RANK_WORDS * 32 ? (i - lo) : RANK_WORDS * 32;     // columns lo .. lo + span - 1
            const int words = (span + 31) >> 5;
            for (int w = threadIdx.x; w < words; w += blockDim.x) bm[w] = 0u;
            if (threadIdx.x == 0) below_s = 0u;
            __syncthreads();
            unsigned below = 0;
            for (int x = threadIdx.x; x < len; x += blockDim.x) {
                const int c = tk[x] - lo;
                if (c < 0) ++below;
                else if (c < span) atomicOr(&bm[c >> 5], 1u << (c & 31));
            }
            for (int o = 32; o > 0; o >>= 1) below += __shfl_down(below, o);
            if (lane == 0 && below) atomicAdd(&below_s, below);
            __syncthreads();
            // exclusive prefix of the words' popcounts: thread t owns the words [t * per, (t + 1) * per)
            const int per = (words + (int)blockDim.x - 1) / (int)blockDim.x;
            const int w0 = threadIdx.x * per, w1 = (w0 + per) < words ? (w0 + per) : words;
            unsigned mine = 0;
            for (int w = w0; w < w1; ++w) mine += (unsigned)__popc(bm[w]);
            unsigned incl = mine;
            for (int o = 1; o < WAVE; o <<= 1) { const unsigned y = __shfl_up(incl, o); if (lane >= o) incl += y; }
            if (lane == WAVE - 1) wsum[wave] = incl;
            __syncthreads();
            unsigned run = incl - mine + below_s;
            for (int w = 0; w < wave; ++w) run += wsum[w];
            for (int w = w0; w < w1; ++w) { pre[w] = run; run += (unsigned)__popc(bm[w]); }
            __syncthreads();
            for (int x = threadIdx.x; x < len; x += blockDim.x) {
                const int col = tk[x], c = col - lo;
                if (c >= 0 && c < span) {
                    const unsigned r = pre[c >> 5] + (unsigned)__popc(bm[c >> 5] & ((1u << (c & 31)) - 1u));
                    fidx[dst + r] = col;
                    fval[dst + r] = tv[x];
                }
            }
            __syncthreads();
        }
    }
}
// ascending-column sort of the mirrored segment of every row: bitonic network on (column, value) pairs.
// LARGE = false: one wave per row, segments of 2..64 entries, exchanged by shuffles.
// LARGE = true : one workgroup per row, segments of 65..MIRROR_MAX_SEG entries, in LDS.
template <bool LARGE>
__global__ __launch_bounds__(256) void smm_mirror_sort(int n, const int64_t *__restrict__ fptr, const int *__restrict__ mcnt,
                                                       int *__restrict__ fidx, double *__restrict__ fval)
{
    extern __shared__ double srt_val[];
    if constexpr (!LARGE) {
        const int lane = lane_id();
        const int wpb = blockDim.x / WAVE;
        for (int i = blockIdx.x * wpb + (threadIdx.x >> 6); i < n; i += gridDim.x * wpb) {
            const int len = mcnt[i];
            if (len < 2 || len > WAVE) continue;
            const int64_t b = fptr[i];
            int key = lane < len ? fidx[b + lane] : 0x7fffffff;
            double val = lane < len ? fval[b + lane] : 0.0;
            for (int k = 2; k <= WAVE; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const int pk = __shfl_xor(key, j);
                    const double pv = __shfl_xor(val, j);
                    const bool up = (lane & k) == 0, low = (lane & j) == 0;
                    const bool take = (pk < key) == (up == low);          // lower lane of an ascending pair keeps the smaller key
                    if (pk != key && take) { key = pk; val = pv; }
                }
            if (lane < len) { fidx[b + lane] = key; fval[b + lane] = val; }
        }
    } else {
        int *srt_key = (int *)(srt_val + MIRROR_MAX_SEG);
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            const int len = mcnt[i];                                       // workgroup-uniform
            if (len <= WAVE || len > MIRROR_MAX_SEG) continue;
            const int64_t b = fptr[i];
            int P = 128;
            while (P < len) P <<= 1;
            for (int x = threadIdx.x; x < P; x += blockDim.x) {
                srt_key[x] = x < len ? fidx[b + x] : 0x7fffffff;
                srt_val[x] = x < len ? fval[b + x] : 0.0;
            }
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int x = threadIdx.x; x < P; x += blockDim.x) {
                        const int y = x ^ j;
                        if (y > x) {
                            const bool up = (x & k) == 0;
                            const int kx = srt_key[x], ky = srt_key[y];
                            if ((kx > ky) == up) {
                                const double vx = srt_val[x];
                                srt_key[x] = ky; srt_key[y] = kx;
                                srt_val[x] = srt_val[y]; srt_val[y] = vx;
                            }
                        }
                    }
                    __syncthreads();
                }
            for (int x = threadIdx.x; x < len; x += blockDim.x) { fidx[b + x] = srt_key[x]; fval[b + x] = srt_val[x]; }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// Plan checker (SMM_CHECK=1 / smm_ctx_set_check / smm_plan_check; every test runs with it).  It verifies, row by
// row, every invariant the numeric phase relies on, and records violations as PLAN_ERR_* bits:
//   capacities   ub_off non-decreasing (a negative capacity makes two lists overlap), list length <= capacity
//   counts       c_ptr[row+1] - c_ptr[row] == list length (slabs: the sum over the row's slab lists)
//   start slots  P[a0] == 0, P[e] <= P[e+1] <= list length
//   lists        every entry is a column of the result (of the slab)
//   sub-runs     runs[e][0] == P[e], runs[e][nct] == P[e+1], non-decreasing in between, and every list entry of
//                sub-run (e, t) is a column of tile t; tail = {e0 in [a0, a1], P[e0]}
//   slab table   runs2[t][e]: source inside [P_s[e], P_s[e+1]), destination + length inside the row, columns of
//                tile t only, and the lengths of a row add up to the row
// One wave per row; `units` = 1 list per row, or n_slabs (list s of row r at index s * m + r).
template <typename LT>
__global__ __launch_bounds__(256) void smm_plan_check_rows(int m, int n_slabs, int ws, int ncols, int64_t nnzA, int64_t total_cap,
                                                           const int *__restrict__ a_ptr, const int64_t *__restrict__ ub_off,
                                                           const int *__restrict__ cnt, const int *__restrict__ rowcnt,
                                                           const int64_t *__restrict__ c_ptr, const unsigned *__restrict__ P,
                                                           const LT *__restrict__ tmp, unsigned *__restrict__ err)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (c_ptr[0] != 0 || ub_off[0] != 0 || ub_off[(size_t)n_slabs * m] != total_cap) plan_err(err, PLAN_ERR_CAP, 0);
    }
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < m; row += gridDim.x * wpb) {
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        int64_t rowtot = 0;
        unsigned bad = 0;
        for (int s = 0; s < n_slabs; ++s) {
            const size_t u = (size_t)s * m + row;
            const int64_t lo = ub_off[u], cap = ub_off[u + 1] - lo;
            const int n = cnt[u];
            if (cap < 0 || lo < 0 || lo > total_cap) { bad |= PLAN_ERR_CAP; continue; }
            if (n < 0 || n > cap) { bad |= PLAN_ERR_CAP; continue; }
            rowtot += n;
            const int64_t slab_lo = (int64_t)s * ws;
            const int64_t width = n_slabs > 1 ? ((ncols - slab_lo) < ws ? (ncols - slab_lo) : ws) : ncols;
            for (int i = lane; i < n; i += WAVE)
                if ((int64_t)tmp[lo + i] < 0 || (int64_t)tmp[lo + i] >= width) bad |= PLAN_ERR_LIST;
            if (n > 0 || n_slabs > 1) {                // (rows no symbolic kernel visited -- no products -- have no start slots)
                const unsigned *Ps = P + (size_t)s * nnzA;
                for (int e = a0 + lane; e < a1; e += WAVE) {
                    const unsigned q0 = Ps[e], q1 = e + 1 < a1 ? Ps[e + 1] : (unsigned)n;
                    if (q0 > q1 || q1 > (unsigned)n || (e == a0 && q0 != 0u)) bad |= PLAN_ERR_P;
                }
            }
        }
        if (rowcnt[row] != rowtot || c_ptr[row + 1] - c_ptr[row] != rowtot) bad |= PLAN_ERR_COUNT;
        if (bad) plan_err(err, bad, row);
    }
}

template <typename LT>
__global__ __launch_bounds__(256) void smm_plan_check_runs(int nrows, int nct, int wc, const int *__restrict__ rowlist,
                                                           const int *__restrict__ a_ptr, const int64_t *__restrict__ ub_off,
                                                           const int *__restrict__ rowcnt, const unsigned *__restrict__ P,
                                                           const LT *__restrict__ tmp, const unsigned *__restrict__ runs,
                                                           const int2 *__restrict__ tail, unsigned *__restrict__ err)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < nrows; ri += gridDim.x * wpb) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const unsigned total = (unsigned)rowcnt[row];
        const LT *__restrict__ list = tmp + ub_off[row];
        const int2 tl = tail[row];
        unsigned bad = 0;
        if (tl.x < a0 || tl.x > a1 || (unsigned)tl.y != (tl.x < a1 ? P[tl.x < a0 ? a0 : tl.x] : total)) bad |= PLAN_ERR_TAIL;
        const int e0 = tl.x < a0 ? a0 : (tl.x > a1 ? a1 : tl.x);
        for (int e = a0; e < e0; ++e) {                         // one step at a time, its sub-runs spread over the lanes
            const unsigned *r = runs + (size_t)e * (nct + 1);
            const unsigned q0 = P[e], q1 = e + 1 < a1 ? P[e + 1] : total;
            if (r[0] != q0 || r[nct] != q1) bad |= PLAN_ERR_RUNS;
            if (q0 > q1 || q1 > total) { bad |= PLAN_ERR_P; continue; }
            for (int t = 0; t < nct; ++t) {
                const unsigned s0 = r[t], s1 = r[t + 1];
                if (s0 > s1 || s0 < q0 || s1 > q1) { bad |= PLAN_ERR_RUNS; continue; }
                for (unsigned i = s0 + lane; i < s1; i += WAVE)
                    if ((int)list[i] / wc != t) bad |= PLAN_ERR_LIST;
            }
        }
        if (bad) plan_err(err, bad, row);
    }
}

__global__ __launch_bounds__(256) void smm_plan_check_runs2(int nrows, int m, int n_slabs, int tps, int nct, int wc, int64_t nnzA,
                                                            const int *__restrict__ rowlist, const int *__restrict__ a_ptr,
                                                            const int64_t *__restrict__ ub_off, const int *__restrict__ scnt,
                                                            const int64_t *__restrict__ c_ptr, const unsigned *__restrict__ P,
                                                            const unsigned short *__restrict__ tmp, const uint2 *__restrict__ runs2,
                                                            unsigned *__restrict__ err)
{
    const int lane = lane_id();
    const int wpb = blockDim.x / WAVE;
    for (int ri = blockIdx.x * wpb + (threadIdx.x >> 6); ri < nrows; ri += gridDim.x * wpb) {
        const int row = rowlist ? rowlist[ri] : ri;
        const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
        const unsigned rowlen = (unsigned)(c_ptr[row + 1] - c_ptr[row]);
        unsigned bad = 0;
        unsigned long long covered = 0;
        for (int t = 0; t < nct; ++t) {
            const int s = t / tps, tl = t - s * tps;
            const unsigned total = (unsigned)scnt[(size_t)s * m + row];
            const unsigned short *__restrict__ list = tmp + ub_off[(size_t)s * m + row];
            const unsigned *Ps = P + (size_t)s * nnzA;
            for (int e = a0; e < a1; ++e) {
                const uint2 d = runs2[(size_t)t * nnzA + e];
                const unsigned src = d.x & 0xffffu, len = d.x >> 16, dst = d.y;
                const unsigned q0 = Ps[e], q1 = e + 1 < a1 ? Ps[e + 1] : total;
                if (q0 > q1 || q1 > total) { bad |= PLAN_ERR_P; continue; }
                if (src < q0 || src + len > q1 || dst > rowlen || len > rowlen - dst) { bad |= PLAN_ERR_RUNS2; continue; }
                covered += len;
                for (unsigned i = src + lane; i < src + len; i += WAVE)
                    if ((int)list[i] / wc != tl) bad |= PLAN_ERR_LIST;
            }
        }
        if (covered != rowlen) bad |= PLAN_ERR_RUNS2;
        if (bad) plan_err(err, bad, row);
    }
}

// Test hook (smm_plan_inject_fault): damage ONE piece of a plan's metadata the way a defect in the kernel that
// produces it would -- so that the tests can assert the consumer reports it instead of hanging or faulting.
//   1 a sub-run that ends before it starts     2 a sub-run bound beyond the row      3 a tail descriptor outside the row
//   4 a row count that disagrees with c_ptr    5 a list entry that is no column      6 a slab sub-run with absurd length / target
//   7 a start slot beyond the list             8 a negative list capacity
__global__ void smm_plan_corrupt(int kind, int row, int nct, int64_t nnzA, const int *__restrict__ a_ptr, int64_t *__restrict__ ub_off,
                                 int *__restrict__ rowcnt, unsigned *__restrict__ P, void *__restrict__ tmp, int list16,
                                 unsigned *__restrict__ runs, int2 *__restrict__ tail, uint2 *__restrict__ runs2)
{
    if (blockIdx.x || threadIdx.x) return;
    const int a0 = a_ptr[row], a1 = a_ptr[row + 1];
    switch (kind) {
    case 1: if (runs) { unsigned *r = runs + (size_t)a0 * (nct + 1); const unsigned x = r[0]; r[0] = r[nct] + 7u; r[nct] = x; } break;
    case 2: if (runs) runs[(size_t)a0 * (nct + 1) + nct] = 0x7ffffff0u; break;
    case 3: if (tail) tail[row] = make_int2(a1 + 5, 0x0ffffff0); break;
    case 4: rowcnt[row] += 1; break;
    case 5: if (list16) ((unsigned short *)tmp)[ub_off[row]] = 0xfffeu; else ((int *)tmp)[ub_off[row]] = 0x7ffffff0; break;
    case 6: if (runs2) runs2[a0] = make_uint2(0xffff0000u, 0xfffffff0u); break;
    case 7: P[a0 + (a1 - a0 > 1 ? 1 : 0)] = 0xfffffff0u; break;
    case 8: ub_off[row + 1] = ub_off[row] - 1; break;
    }
}

}  // namespace smm
