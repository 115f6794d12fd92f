// smm_ring.hpp -- triple product, stage 2, round 4: the tile of T is a RING of column pieces and the waves of a
// workgroup synchronise through progress words in LDS instead of barriers.
//
// Reference loop: src/sparse_sparse_dense.cpp:201-216 -- C[i,k] = sum over row k of H of T[i, col] * H[k, col], k >= i.
//
// What round 2/3's kernel (smm_triple_stage2, smm_kernels.hpp) loses.  A workgroup owns 16 rows of T x 1024 rows k
// (one k per lane, sums in registers); T is staged in LDS one 1024-column chunk at a time and every lane walks its row
// of H chunk by chunk.  Inside a chunk a wave takes max-over-its-64-lanes steps (Poisson(20) segments: 31 steps for 20.5
// entries), and at the chunk's end ALL 16 waves wait at a barrier for the slowest (35.5 steps), then for the refill.
// PMC (profiles/r2_zz_c3_pmc.txt): LDS pipe 64 % busy, VALU 39 %, vector memory 27 % -- nothing is saturated; the
// kernel waits at barriers.  Two workgroups per CU would hide that but halve the chunk (LDS) and with it the lanes'
// efficiency (measured in round 2: 71 ms against 54).
//
// The ring.  Same LDS (147 KB), cut into NB = 2 buffers of PW = 512 columns.  Piece p of the columns lives in buffer
// p mod NB.  A lane that has finished its entries of piece p goes straight on with piece p+1 while other lanes (and
// other waves) are still in p -- the window a lane can work in is the two resident pieces, 1024 columns as before, but
// it SLIDES piece by piece instead of jumping chunk by chunk.  Nobody waits at a barrier:
//   prog[w]   = number of pieces wave w has finished        (written by wave w after its last read of the piece)
//   fprog[w]  = number of pieces whose share wave w has written into the ring
// piece p may be written into its buffer once min_w prog[w] >= p - NB + 1 (every wave is done with piece p - NB), and
// may be read once min_w fprog[w] >= p + 1.  Every wave fills 1/NW of every piece (its columns w*PW/NW ..), loads for the
// next share travel in registers, the fill is attempted -- never waited for -- once per step, and a wave only ever waits
// when its next step needs a piece that is not complete.  One wave's LDS operations execute in issue order, so "data
// writes, then the progress word" needs no fence, and neither does "progress word read, then the data".
//
// The schedule (smm_ring_build).  Which entry a lane executes in which step is decided once per H (cached on the
// handle), by simulating a whole k-group -- 16 waves x 64 lanes -- in lock step: an entry of piece p may be taken at time t
// only if the piece is resident by then (all lanes of all waves finished p - NB, plus a refill allowance), lanes of a
// ds_read_b128 conflict group take entries of distinct bank classes (column mod 16: round 2's conflict-free order),
// default mode picks among the next LOOK = 8 entries of the lane, SMM_EXACT only the next one (stored order: bit-exact).
// Per wave the steps in which at least one lane moves are stored, [step][64 lanes] (int16 ring position | f64 value), with a
// header per step: the highest piece the step needs and the number of pieces the wave has finished after it.  At run time
// the progress words enforce what the simulation predicted; a wave that is ahead of it simply waits where it must.
// Every wait is bounded (SPIN_MAX polls): a protocol defect would end as PLAN_ERR_RING in the context's error word,
// not as a hung GPU.
#pragma once
#include "smm_kernels.hpp"

namespace smm {

constexpr int RING_PW = 512;        // columns per piece
constexpr int RING_NB = 2;          // pieces resident
constexpr int RING_FILLD = 3;       // steps the simulation allows for a refill
constexpr unsigned PLAN_ERR_RING = 256;

struct RingBuildArgs {
    int n, K, npieces, nkg;
    const int *h_ptr, *h_idx; const double *h_val;
    int64_t *cnt;                 // [nkg * 16] steps per (k-group, wave)            (count pass)
    const int64_t *off;           // exclusive scan of cnt (+ total)                  (fill pass)
    short *col; double *val;      // [total steps][64]
    unsigned *hdr;                // [total steps + nkg * 16]: stream s starts at off[s] + s, holds T + 1 headers
    unsigned *err;
};

// lane at position i of this lane's ds_read_b128 conflict group, and this lane's position (see smm_ell_fill)
__device__ __forceinline__ int ring_group_at(int lane, int odd, int i) {
    return (lane & 32) + (odd ? (i < 8 ? i + 4 : i < 12 ? i + 8 : i + 16) : (i < 4 ? i : i < 8 ? i + 8 : i + 12));
}

template <int LOOK, bool WRITE>
__global__ __launch_bounds__(1024) void smm_ring_build(const RingBuildArgs A)
{
    constexpr int NW = 16, PW = RING_PW, NB = RING_NB;
    __shared__ int w_oldest[NW];
    __shared__ int s_released;
    __shared__ int s_avail[NB];              // time from which the piece in this buffer may be read
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int kg = blockIdx.x;
    const int64_t k = (int64_t)kg * (NW * WAVE) + wave * WAVE + lane;
    const int s0 = k < A.n ? A.h_ptr[k] : 0, e0 = k < A.n ? A.h_ptr[k + 1] : 0;
    const int sidx = kg * NW + wave;
    const int64_t sbase = WRITE ? A.off[sidx] : 0;
    unsigned *hdr = WRITE ? A.hdr + sbase + sidx : nullptr;
    short *ocol = WRITE ? A.col + sbase * WAVE + lane : nullptr;
    double *oval = WRITE ? A.val + sbase * WAVE + lane : nullptr;
    // conflict group geometry of ds_read_b128 (MI355X_MICROARCH.md, LDS table)
    const int l32 = lane & 31;
    const int odd = (l32 >= 4 && l32 < 12) || (l32 >= 16 && l32 < 20) || l32 >= 28;
    const int gp = l32 < 4 ? l32 : l32 < 12 ? l32 - 4 : l32 < 20 ? l32 - 8 : l32 < 28 ? l32 - 12 : l32 - 16;

    if (threadIdx.x == 0) s_released = 0;
    if (threadIdx.x < NB) s_avail[threadIdx.x] = 0;
    __syncthreads();

    int base = s0;                  // first entry of the row not executed yet
    unsigned cons = 0;              // bit i: entry base + i has been executed (out of order, default mode)
    int my_step = 0;                // steps of this wave so far (wave-uniform)
    int need_upto = -1;             // highest piece any step of this wave has used
    int done_before = 0;            // pieces this wave has finished so far (wave-uniform)
    const int64_t t_max = (int64_t)A.K + 4ll * A.npieces + 64;       // more steps than any schedule can take
    for (int64_t t = 0; t < t_max; ++t) {
        const int released = s_released;                             // pieces < released are finished by everybody
        if (released >= A.npieces) break;
        // candidates: the next LOOK entries of the row that lie in a resident piece
        int cand_col[LOOK];
        bool cand_ok[LOOK];
        int n_left = 0;                                              // unexecuted entries of piece `released` (priority)
#pragma unroll
        for (int i = 0; i < LOOK; ++i) {
            const int pos = base + i;
            const int c = pos < e0 ? A.h_idx[pos] : 0x7fffffff;
            const int p = c / PW;
            cand_col[i] = c;
            const bool in_window = pos < e0 && p < released + NB;
            cand_ok[i] = in_window && !((cons >> i) & 1u) && s_avail[p % NB] <= (int)t;
            n_left += (pos < e0 && p == released && !((cons >> i) & 1u)) ? 1 : 0;
        }
        for (int i = LOOK; i < 24; ++i) {                            // (priority only: what lies behind the look-ahead, roughly)
            const int pos = base + i;
            n_left += (pos < e0 && A.h_idx[pos] / PW == released) ? 1 : 0;
        }
        // lanes of a conflict group choose one after the other, those with most left in the oldest piece first
        const int key = ((255 - (n_left > 255 ? 255 : n_left)) << 4) | gp;
        int rank = 0;
        for (int i = 0; i < 16; ++i) rank += __shfl(key, ring_group_at(lane, odd, i)) < key ? 1 : 0;
        const int chooser = __builtin_amdgcn_ds_permute(ring_group_at(lane, odd, rank) << 2, lane);
        unsigned used = 0;
        int pick = -1;
        for (int j = 0; j < 16; ++j) {
            const int who = __shfl(chooser, ring_group_at(lane, odd, j));
            unsigned bit = 0;
            if (lane == who) {
#pragma unroll
                for (int i = 0; i < LOOK; ++i)
                    if (pick < 0 && cand_ok[i] && !((used >> (cand_col[i] & 15)) & 1u)) pick = i;
                // a lane that holds the piece everybody waits for never sits a step out: it takes the conflict
                if (pick < 0 && cand_ok[0] && cand_col[0] / PW == released) pick = 0;
                if (pick >= 0) bit = 1u << (cand_col[pick] & 15);
            }
            used |= (unsigned)__shfl((int)bit, who);
        }
        const bool moved = pick >= 0;
        const unsigned long long any = __ballot(moved);
        int piece_used = -1;
        if (moved) {
            const int c = cand_col[pick];
            piece_used = c / PW;
            if (WRITE) {
                ocol[(int64_t)my_step * WAVE] = (short)((piece_used % NB) * PW + (c - piece_used * PW));
                oval[(int64_t)my_step * WAVE] = A.h_val[base + pick];
            }
            cons |= 1u << pick;
            while (cons & 1u) { cons >>= 1; ++base; }
        } else if (WRITE && any) {
            ocol[(int64_t)my_step * WAVE] = (short)-1;
            oval[(int64_t)my_step * WAVE] = 0.0;
        }
        // this lane's oldest unfinished piece; the wave's; the workgroup's
        int mine = base < e0 ? A.h_idx[base] / PW : A.npieces;
        int pu = piece_used;
        for (int o = 32; o > 0; o >>= 1) {
            const int y = __shfl_xor(mine, o); mine = y < mine ? y : mine;
            const int z = __shfl_xor(pu, o); pu = z > pu ? z : pu;
        }
        if (any) {
            need_upto = pu > need_upto ? pu : need_upto;
            // header of a step: the highest piece it (or an earlier step) reads | the pieces this wave had finished BEFORE it
            if (WRITE && lane == 0) hdr[my_step] = (unsigned)(need_upto & 0xffff) | ((unsigned)done_before << 16);
            ++my_step;
        }
        done_before = mine;                                          // pieces < mine are finished after this time step
        if (lane == 0) w_oldest[wave] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            int m = w_oldest[0];
            for (int w = 1; w < NW; ++w) m = w_oldest[w] < m ? w_oldest[w] : m;
            int rel = released;
            while (rel < m && rel < A.npieces) {                     // piece rel is finished: its buffer takes piece rel + NB
                s_avail[rel % NB] = (int)t + 1 + RING_FILLD;
                ++rel;
            }
            s_released = rel;
        }
        __syncthreads();
    }
    if (s_released < A.npieces && threadIdx.x == 0) plan_err(A.err, PLAN_ERR_RING, kg);      // (cannot happen: t_max bounds every schedule)
    if (lane == 0) {
        if (WRITE) hdr[my_step] = (unsigned)(need_upto & 0xffff) | ((unsigned)A.npieces << 16);
        else A.cnt[sidx] = my_step;
    }
}

struct RingArgs {
    int n, K, npieces, nslices;
    int nib, nkg, gk;
    int64_t row_begin, row_end;
    int full;
    const int64_t *off;           // [nkg * 16 + 1]
    const short *col; const double *val; const unsigned *hdr;
    const double *T;              // (row_end - row_begin) x K
    double *C; int64_t ldc;
    unsigned *err;
};

#ifndef SMM_RING_POLL
#define SMM_RING_POLL 2             // (a power of two) steps between two looks at the progress words while a share is owed
#endif
#ifndef SMM_RING_SPIN_MAX
#define SMM_RING_SPIN_MAX (1 << 22)
#endif

template <int R, int NW, bool FMA>
__global__ __launch_bounds__(NW * 64) void smm_triple_stage2_ring(const RingArgs A)
{
    constexpr int PW = RING_PW, NB = RING_NB, LD = R + 2, D = 3;
    constexpr int CS = PW / NW;                     // columns of a piece this wave fills
    constexpr int RPI = WAVE / CS;                  // rows covered by one load instruction of the share
    constexpr int NVS = R / RPI;                    // loads per lane and share
    static_assert(NW == 16 && CS * RPI == WAVE && NVS * RPI == R, "fill share: PW / 16 columns x 16 rows per wave");
    extern __shared__ double tl[];                  // [NB * PW][LD]
    // progress words: [0, NW) prog, [NW, 2 NW) fprog.  Accessed with hand-issued ds_write_b32 / ds_read_b32 on their LDS
    // addresses: through C++ (a volatile pointer captured by the lambdas below) hipcc made them FLAT instructions with a
    // vmcnt(0) behind each.
    __shared__ int s_words[2 * NW];

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned words_a = lds_addr(s_words);
    auto set_prog = [&](int v_) { asm volatile("ds_write_b32 %0, %1" :: "v"(words_a + 4u * (unsigned)wave), "v"(v_) : "memory"); };
    auto set_fprog = [&](int v_) { asm volatile("ds_write_b32 %0, %1" :: "v"(words_a + 4u * (unsigned)(NW + wave)), "v"(v_) : "memory"); };
    // block order: as smm_triple_stage2 (k-groups of one super-group that meet the same row block follow each other on one XCD)
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
    const int sl = kg * NW + wave;                   // this wave's 64 rows of H = its stream
    const int64_t k = (int64_t)sl * WAVE + lane;
    const bool kin = k < A.n;
    double *crow = A.C + (int64_t)(i0 - A.row_begin) * A.ldc + k;
    if (!A.full && (int64_t)(kg + 1) * NW * WAVE <= i0) {           // the whole k-group lies left of the diagonal
        if (kin)
            for (int r = 0; r < nr; ++r) crow[(int64_t)r * A.ldc] = 0.0;
        return;
    }
    const bool work = sl < A.nslices && (A.full || (int64_t)(sl + 1) * WAVE > i0);
    const int npieces = A.npieces;
    if (lane == 0) { set_prog(work ? 0 : npieces); set_fprog(0); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                 // the only barrier of the kernel

    double sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] = 0.0;
    auto mad = [](double x, double h, double acc) { return FMA ? __builtin_fma(x, h, acc) : acc + x * h; };

    // this wave's share of a piece: columns wave*CS + (lane % CS), rows (lane / CS) + RPI * t
    const int cl = wave * CS + (lane & (CS - 1)), rr = lane / CS;
    const double *trow = A.T + (i0 - A.row_begin) * A.K;
    // The loads of a share are issued by hand (gload_f64) and waited for by hand: they are issued on the rare path of
    // a step, and a compiler-visible load there made hipcc wait with vmcnt(0) where that path joins the common one --
    // at every step, draining the ring of H entries.  `young` counts the loads issued since: the wait before the
    // share is written needs vmcnt(2 D) only when at least that many younger loads (ring entries) are behind it.
    double v[NVS];
    int young = 0, foreign = 0;
    const double *tsrc[NVS];
#pragma unroll
    for (int t = 0; t < NVS; ++t) {
        const int row = rr + RPI * t;
        tsrc[t] = trow + (int64_t)(row < nr ? row : nr - 1) * A.K;
    }
    auto load_share = [&](int p) {
        const int lo = p * PW;
        const int w = (A.K - lo) < PW ? (A.K - lo) : PW;
        const int c = lo + (cl < w ? cl : w - 1);
#pragma unroll
        for (int t = 0; t < NVS; ++t) gload_f64(v[t], tsrc[t] + c);
        young = 0;
        foreign = __builtin_amdgcn_readfirstlane(foreign + NVS);      // (the ring's waits count these)
    };
    auto write_share = [&](int p) {
#if defined(SMM_RING_SAFE_WAITS) || defined(SMM_RING_SAFE_SHARE)
        if (false) {}
#else
        if (young >= 2 * D) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); static_assert(D == 3, "vmcnt(2 D)"); }
#endif
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NVS; ++t) asm volatile("" : "+v"(v[t]));
        double *dst = tl + (size_t)((p % NB) * PW + cl) * LD + rr;
#pragma unroll
        for (int t = 0; t < NVS; ++t) dst[RPI * t] = v[t];
    };
    int fp = 0;                                      // next piece this wave has to contribute to
    for (; fp < NB && fp < npieces; ++fp) { load_share(fp); write_share(fp); }
    asm volatile("" ::: "memory");
    if (lane == 0) set_fprog(fp);                    // (LDS executes a wave's operations in order: the data is there before the word)
    if (fp < npieces) load_share(fp);

    const int64_t sbase = A.off[sl];                 // (a stream exists for every wave of every k-group, rows beyond n included: empty)
    const int T_w = work ? (int)(A.off[sl + 1] - sbase) : 0;
    const short *cp = A.col + sbase * WAVE + lane;
    const double *vp = A.val + sbase * WAVE + lane;
    const unsigned *hp = A.hdr + sbase + sl;
    const int last = T_w > 0 ? T_w - 1 : 0;
    // Step headers travel 63 at a time, one per lane (lane l: step s0 + l), loaded by hand one block ahead: a compiler-
    // visible load whose use is a whole block of steps away made hipcc wait with vmcnt(0) at every step -- which also
    // drains the ring of H entries (measured: 1770 cycles per step instead of 850).
    constexpr int HB = 63;                           // a multiple of D
    int hnext;
    {
        const unsigned *q = hp + (lane <= T_w ? lane : T_w);
        asm volatile("global_load_dword %0, %1, off" : "=v"(hnext) : "v"(q) : "memory");
    }
    // The ring of H entries is hand-issued as well (gload_sshort / gload_f64 + counted waits): with the share and header
    // loads outside the compiler's view its own counts for the ring came out as vmcnt(0..1) at the head of every group of
    // steps.  ex[u] = loads issued since slot u's pair that are not ring entries (a share: 8, a header block: 1): the
    // wait for slot u is vmcnt(2 (D - 1) + ex[u]), rounded down to one of four encodable cases.
    int cc[D]; double hh[D];
    int ex[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int su = u < last ? u : last;          // (T_w == 0: position 0 of the stream block -- allocated, never used)
        gload_sshort(cc[u], cp + (int64_t)su * WAVE);
        gload_f64(hh[u], vp + (int64_t)su * WAVE);
        ex[u] = foreign;
    }
    // (the wait itself carries no operands: with the values tied to four alternative asm statements the compiler copied
    // them into the common result register BEFORE three of the four waits -- i.e. before the loads had landed.  One
    // empty statement behind the branches ties the values instead.)
    auto ring_wait = [&](int &c, double &v_, int e) {
#if defined(SMM_RING_SAFE_WAITS) || defined(SMM_RING_SAFE_RING)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (e == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (e < NVS) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (e == NVS) { if (NVS == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else { if (NVS == 8) asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
        static_assert(D == 3 && (NVS == 4 || NVS == 8), "vmcnt(2 (D - 1) + foreign loads): a share is NVS loads");
#endif
        asm volatile("" : "+v"(c), "+v"(v_) : : "memory");
    };
    bool broken = false;                             // a wait ran out: stop waiting, report, finish
    int dbg0 = 0, dbg1 = 0, dbg2 = -1;               // (what the wave saw when it gave up: reported with the error)

    auto try_fill = [&](int pmin) {
        if (fp < npieces && pmin >= fp - NB + 1) {    // everybody is done with the piece this one replaces
            write_share(fp);
            asm volatile("" ::: "memory");
            fp = __builtin_amdgcn_readfirstlane(fp + 1);      // (wave-uniform state is kept in SGPRs: readfirstlane tells the compiler so)
            if (lane == 0) set_fprog(fp);
            if (fp < npieces) load_share(fp);
        }
    };

    // Per step the common path costs three scalar instructions: the step's header is compared with the last one.  Only
    // when it changes (a new piece is needed, or the wave has finished one: ~twice per piece) the wave looks at the
    // progress words; and while a share of its own is owed and the wave itself is done with the piece it replaces, it
    // polls every other step.
#ifdef SMM_RING_STAMPS
    unsigned long long t_wait = 0, t_tail = 0, n_slow = 0, n_spin = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
    unsigned hcur = 0xffffffffu;                     // (no header looks like this: the first step takes the slow path)
    int my_done = 0;
    // Two header registers take turns (hA: even blocks, hB: odd blocks): each is loaded by hand one block ahead, waited
    // for once, and only READ until its next load -- a single register re-loaded while its old value was still in use made
    // the compiler copy it BEFORE the hand-issued wait, i.e. before the load had landed.
    auto hdr_load = [&](int &hx, int block_s0) {
        const int qn = block_s0 + lane;
        const unsigned *q = hp + (qn <= T_w ? qn : T_w);
        asm volatile("global_load_dword %0, %1, off" : "=v"(hx) : "v"(q) : "memory");
        foreign = __builtin_amdgcn_readfirstlane(foreign + 1); young = __builtin_amdgcn_readfirstlane(young + 1);
    };
    // Looks at the progress words cost no round trip of their own: ONE ds_read_b32 of the words is issued at the end
    // of every step (2 LDS cycles beside the step's 32), behind that step's tile reads, and has landed when the next
    // step starts; the step reduces it (DPP minima) only when it has a reason to -- its header differs from the last
    // one (a piece is needed, or was finished), or a share of its own is owed and it is time to look again.  One issue
    // site, one wait site: the value is never re-loaded while in flight and never merged with another in-flight value
    // (either made the compiler copy or recycle the register before the data had landed).
    int px;
    asm volatile("ds_read_b32 %0, %1" : "=v"(px) : "v"(words_a + 4u * (unsigned)(lane & (2 * NW - 1))) : "memory");
    int known_p = 0, known_f = 0;
    auto reduce = [&](int x) {
#define SMM_ROWMIN(ctrl) { const int y = __builtin_amdgcn_update_dpp(0x7fffffff, x, ctrl, 0xf, 0xf, false); x = y < x ? y : x; }
        SMM_ROWMIN(0x111) SMM_ROWMIN(0x112) SMM_ROWMIN(0x114) SMM_ROWMIN(0x118)
#undef SMM_ROWMIN
        known_p = __builtin_amdgcn_readlane(x, 15); known_f = __builtin_amdgcn_readlane(x, 31);
    };
    auto poll_now = [&]() {                           // a synchronous look (only while a wave really waits)
        int x;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(words_a + 4u * (unsigned)(lane & (2 * NW - 1))) : "memory");
        reduce(x);
    };
    auto run_block = [&](const int s0, const int hdrv) {
        const int nst = (T_w - s0) < HB ? (T_w - s0) : HB;
        for (int g0 = 0; g0 < nst; g0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int sl_ = g0 + u;
                const int s = s0 + sl_;
                const bool live = sl_ < nst;               // (the last group of a stream may be partial: its dead slots only reload)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px) : : "memory");       // the look issued at the end of the step before
                if (live) {
                    const unsigned h = (unsigned)__builtin_amdgcn_readlane(hdrv, sl_);
                    const bool owe = fp < npieces && my_done >= fp - NB + 1 && (s & (SMM_RING_POLL - 1)) == 0;
                    if (h != hcur || owe) {
                        reduce(px);
                        try_fill(known_p);
                    }
                    if (h != hcur) {
                        hcur = h;
                        const int need = (int)(h & 0xffffu), done = (int)(h >> 16);
                        asm volatile("" ::: "memory");     // (the tile reads of the steps before are issued before the word moves)
                        if (done > my_done) { my_done = __builtin_amdgcn_readfirstlane(done); if (lane == 0) set_prog(done); }
                        if (known_f < need + 1) {          // the step reads a piece that (as far as this wave knows) is not complete
                            int spin = 0;
#ifdef SMM_RING_STAMPS
                            const unsigned long long tw0 = __builtin_readcyclecounter();
                            ++n_slow;
#endif
                            do {
                                poll_now();
                                try_fill(known_p);
                                if (++spin > SMM_RING_SPIN_MAX) { broken = true; dbg0 = (need << 16) | (known_f & 0xffff); dbg1 = (known_p << 16) | (fp & 0xffff); dbg2 = s; }
                            } while (known_f < need + 1 && !broken);
#ifdef SMM_RING_STAMPS
                            t_wait += __builtin_readcyclecounter() - tw0; n_spin += spin;
#endif
                        }
                        asm volatile("" ::: "memory");     // (no tile read of this step moves above the wait)
                    }
                }
                // (the ring loads stand outside every condition, as in smm_triple_stage2: one fixed sequence of loads per
                // group; they are hand-issued, their waits counted: ring_wait)
                ring_wait(cc[u], hh[u], foreign - ex[u]);
                if (live && cc[u] >= 0) {
                    const double2 *p = reinterpret_cast<const double2 *>(tl + cc[u] * LD);
#pragma unroll
                    for (int r = 0; r < R; r += 2) {
                        const double2 x = p[r >> 1];
                        sum[r] = mad(x.x, hh[u], sum[r]);
                        sum[r + 1] = mad(x.y, hh[u], sum[r + 1]);
                    }
                }
                asm volatile("ds_read_b32 %0, %1" : "=v"(px) : "v"(words_a + 4u * (unsigned)(lane & (2 * NW - 1))) : "memory");
                const int sn = s + D < last ? s + D : last;
                gload_sshort(cc[u], cp + (int64_t)sn * WAVE);
                gload_f64(hh[u], vp + (int64_t)sn * WAVE);
                ex[u] = foreign;
                young = __builtin_amdgcn_readfirstlane(young + 2);
            }
        }
    };
    int hB;
    for (int s0 = 0; s0 < T_w; s0 += 2 * HB) {
        wait_vm(hnext, 2 * D);                       // (at least 2 D younger loads are behind it: the ring)
        hdr_load(hB, s0 + HB);
        run_block(s0, hnext);
        wait_vm(hB, 2 * D);
        hdr_load(hnext, s0 + 2 * HB);
        run_block(s0 + HB, hB);                      // (past the end of the stream: zero steps)
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hnext));     // (the last hand-issued loads: none may outlive the wave's use of its registers)
#pragma unroll
    for (int u = 0; u < D; ++u) asm volatile("" : "+v"(cc[u]), "+v"(hh[u]));
    asm volatile("" ::: "memory");
    // the stream is done; the shares of the remaining pieces are still owed to the other waves
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px) : : "memory");
    if (lane == 0) set_prog(npieces);
#ifdef SMM_RING_STAMPS
    const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
    {
        int spin = 0;
        while (fp < npieces && !broken) {
            poll_now();
            try_fill(known_p);
            if (++spin > SMM_RING_SPIN_MAX) broken = true;      // (reported once, at the end: no VMEM operation inside a wait loop)
        }
    }
#ifdef SMM_RING_STAMPS
    t_tail = __builtin_readcyclecounter() - tt0;
    if (lane == 0 && work) {
        unsigned long long *st = (unsigned long long *)(A.err + 4);
        atomicAdd(st + 0, __builtin_readcyclecounter() - t_begin); atomicAdd(st + 1, t_wait); atomicAdd(st + 2, t_tail);
        atomicAdd(st + 3, n_slow); atomicAdd(st + 4, n_spin); atomicAdd(st + 5, (unsigned long long)T_w);
    }
#endif
    if (broken && lane == 0) {
        plan_err(A.err, PLAN_ERR_RING, kg);
        A.err[8] = (unsigned)dbg0; A.err[9] = (unsigned)dbg1; A.err[10] = (unsigned)dbg2; A.err[11] = (unsigned)(wave | (ib << 8)); A.err[12] = (unsigned)T_w;
    }
    if (kin) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < nr) crow[(int64_t)r * A.ldc] = (A.full || k >= i0 + r) ? sum[r] : 0.0;
    }
}

}  // namespace smm
