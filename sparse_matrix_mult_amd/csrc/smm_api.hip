// smm_api.hip -- host side of libsmm_hip.so: the v2 C ABI declared in include/smm_hip.h.
// Owns the device context (stream, pooled workspace, per-kernel timing) and sequences the
// kernels of smm_kernels.hpp.  There is deliberately no CPU compute path in this file: with
// no device every entry point fails with SMM_ERR_NO_DEVICE.
#include "smm_kernels.hpp"
#include "smm_slab.hpp"
#include "smm_ring.hpp"
#include "../../include/smm_hip.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace smm;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(SMM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                \
    } while (0)
#define CHK(expr)                      \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != SMM_OK) return rc_; \
    } while (0)

extern "C" const char *smm_last_error(void) { return g_err; }

extern "C" int smm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ------------------------------------------------------------------------------ context
struct PoolBlock { void *p; size_t bytes; };
struct TimedLaunch { std::string name; hipEvent_t t0, t1; };

struct smm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    int sym_wide = 1;        // symbolic phase on 16-bit columns: chunks of 128 entries (env SMM_SYM_WIDE=0: 64)
    int sym_ccs = 1;         // symbolic phase over the chunk-padded column stream (smm_symbolic_ccs; env SMM_SYM_CCS=0: smm_symbolic)
    int piece_walk = 1;      // default mode: one piece of B per wave iteration where every piece has <= 256 entries (env SMM_PIECE_WALK=0: chunk walk)
    int sym_max_ws = 0;      // widest column slab of that walk (0 = CCS_MAX_WS); B with more columns is walked slab by slab
                             // (smm_ctx_tune_symbolic; tests set it small to reach the slab path with small matrices)
    int sym_dense = 1;       // symbolic walk over operands with dense runs of columns: 0 never, 1 chosen from B (>= 80 % of the neighbouring
                             // entries share a bitmap word), 2 always (tests) -- env SMM_SYM_DENSE, smm_ctx_tune_dense_runs
    int numeric_persist = 1; // smm_numeric: persistent workgroups fed by a unit counter (env SMM_NUMERIC_PERSIST: 0 never, 1 CSR output without triangle, 2 always)
    int s2_ring = 0;         // triple stage 2: 1 = the ring kernel of round 4 (smm_ring.hpp: correct, measured 60-63 ms against 51 at
                             // BASELINE configs[3] -- kept as an alternative, env SMM_S2_RING=1 / smm_ctx_tune_stage2); 0 = the chunk kernel
    int s2_group = 5;        // triple stage 2: k-groups whose blocks follow each other on one XCD and share a tile of T
                             // through its L2 (env SMM_S2_GROUP; at BASELINE configs[3] 1: 58.2, 2: 56.3, 4: 60.1, 5: 54.8,
                             // 7: 54.9, 8: 58.9, 10: 54.8 ms -- powers of two lose, profiles/r2_s2_sweeps.txt)
    int lds_cols = 18000;    // SMM_EXACT walk: accumulator columns per workgroup (x8 B of LDS; 1.5 KB of
    int waves = 8;           // scratch per wave sit behind them) and waves per workgroup (each owns 1/waves)
    int lds_cols_shared = 20000;   // default (shared-tile) walk: tile columns and waves per workgroup
    int waves_shared = 16;
    int hash_small = 256;    // rows of C with <= hash_small nonzeros: one wave per row, LDS hash (0 = off)
    int hash_medium = 2048;  // ... <= hash_medium: one workgroup per row, LDS hash; above: dense LDS tiles
    int tiny_max = TINY_G;   // rows with <= 16 products from <= 16 entries of A: smm_*_tiny, four rows per wave (env SMM_TINY=0: off)
    // row block x column slab kernels (smm_slab.hpp): mode 0 = where they pay, 1 = never, 2 = wherever
    // they can run; ws = slab width (0 = sized so that one slab of B is ~3 MB, L2-resident);
    // rows per wave 2 or 4 (8 waves per workgroup: 16 or 32 rows per block)
    int slab_mode = 0, slab_ws = 0, slab_rw = 4;
    int narrow_idx = 1;      // 1: operands with < 65535 columns go through the symbolic phase as uint16 (column stream and lists)
    int n_cu = 256;
    std::vector<PoolBlock> pool;          // free blocks
    std::map<void *, size_t> live;        // blocks handed out
    std::vector<TimedLaunch> launches;
    std::map<std::string, std::pair<double, int64_t>> totals;
    // result download (download() below): ring of pinned bounce buffers, allocated on first use
    static constexpr int PIN_SLOTS = 8;
    static constexpr size_t PIN_BYTES = (size_t)32 << 20;
    void *pin[PIN_SLOTS] = {nullptr};
    hipEvent_t pin_ev[PIN_SLOTS] = {nullptr};
    int exact_checked = 0;           // SMM_EXACT guard (smm_ctx_exact_selftest): 0 not run yet, 1 passed, -1 failed
    int check = 0;                   // 1: every symbolic phase ends with the plan checker (env SMM_CHECK, smm_ctx_set_check)
    int inject_alloc_nth = 0;        // test hook: the n-th dev_malloc from now fails (its first attempt, or both: _hard)
    bool inject_alloc_hard = false;
    int64_t alloc_retries = 0;       // allocations that needed the pool flushed
    unsigned *d_flags = nullptr;     // [0] validation flags; +64: int -1 and +128: double 0 read by idle lanes
    unsigned *d_err = nullptr;       // plan error word: [0] PLAN_ERR_* bits, [1] lowest row that tripped one (kernels' always-on clamps, plan checker)
    std::recursive_mutex mu;         // every entry point that touches the context takes it: calls from
                                     // several host threads on one context serialise (one stream anyway)
};
#define CTX_LOCK(c) std::lock_guard<std::recursive_mutex> ctx_lock_((c)->mu)

// Return every free block of the context's pool to the device (blocks handed out stay).  hipFree waits for the
// device, so work still queued on blocks that went back to the pool has finished by then.
static void pool_flush(smm_ctx *c)
{
    for (auto &b : c->pool) (void)hipFree(b.p);
    c->pool.clear();
}
// hipMalloc for everything that is not pooled (operands and their cached copies).  An allocation that fails is
// retried once after the pool's free blocks -- the multi-GB lists of closed plans live there -- went back to the
// device; a failure never leaves HIP's sticky last error behind (the next LAUNCH_CHECK would report it).
// Test hook (smm_ctx_inject_alloc_failure): the n-th call from now fails its first attempt -- or both.
static hipError_t dev_malloc(smm_ctx *c, void **p, size_t bytes)
{
    bool fail_first = false, fail_both = false;
    if (c->inject_alloc_nth > 0 && --c->inject_alloc_nth == 0) { fail_first = true; fail_both = c->inject_alloc_hard; }
    hipError_t e = fail_first ? hipErrorOutOfMemory : hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    pool_flush(c);
    ++c->alloc_retries;
    e = fail_both ? hipErrorOutOfMemory : hipMalloc(p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; }
    return e;
}

static int pool_alloc(smm_ctx *c, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < (int)c->pool.size(); ++i)
        if (c->pool[i].bytes >= bytes && c->pool[i].bytes <= bytes + bytes / 4 + (1 << 20) &&
            (best < 0 || c->pool[i].bytes < c->pool[best].bytes))
            best = i;
    if (best >= 0) {
        *out = c->pool[best].p;
        c->live[*out] = c->pool[best].bytes;
        c->pool.erase(c->pool.begin() + best);
        return SMM_OK;
    }
    void *p = nullptr;
    hipError_t e = dev_malloc(c, &p, bytes);           // (drops the pool's free blocks and retries once)
    if (e != hipSuccess) return fail(SMM_ERR_ALLOC, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    c->live[p] = bytes;
    *out = p;
    return SMM_OK;
}
static void pool_free(smm_ctx *c, void *p)
{
    if (!p) return;
    auto it = c->live.find(p);
    if (it == c->live.end()) return;
    c->pool.push_back({p, it->second});
    c->live.erase(it);
}
template <typename T> static int pool_get(smm_ctx *c, size_t count, T **out)
{
    void *p = nullptr;
    int rc = pool_alloc(c, count * sizeof(T), &p);
    *out = (T *)p;
    return rc;
}

struct LaunchTimer {
    smm_ctx *c; const char *name; hipEvent_t t0 = nullptr, t1 = nullptr;
    LaunchTimer(smm_ctx *ctx, const char *n) : c(ctx), name(n)
    {
        if (c->timing) {
            (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
            (void)hipEventRecord(t0, c->stream);
        }
    }
    ~LaunchTimer()
    {
        if (c->timing) {
            (void)hipEventRecord(t1, c->stream);
            c->launches.push_back({name, t0, t1});
        }
    }
};
#define LAUNCH(ctx, name, kern, grid, block, lds, ...)                                         \
    do {                                                                                       \
        LaunchTimer lt_(ctx, name);                                                            \
        hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (size_t)(lds), \
                           (ctx)->stream, __VA_ARGS__);                                        \
    } while (0)
#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

extern "C" int smm_ctx_create(int device, void *hip_stream, smm_ctx **out)
{
    if (!out) return fail(SMM_ERR_INVALID, "smm_ctx_create: out is NULL");
    *out = nullptr;
    int n = smm_device_count();
    if (n <= 0) return fail(SMM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(SMM_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SMM_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    smm_ctx *c = new smm_ctx();
    c->device = device;
    if (const char *e = getenv("SMM_NARROW_IDX")) c->narrow_idx = atoi(e) != 0;     // A/B switch (scripts/ab_env.sh)
    if (const char *e = getenv("SMM_S2_GROUP")) c->s2_group = std::max(1, atoi(e));
    if (const char *e = getenv("SMM_S2_RING")) c->s2_ring = atoi(e) != 0;
    if (const char *e = getenv("SMM_NUMERIC_PERSIST")) c->numeric_persist = atoi(e);
    if (const char *e = getenv("SMM_TINY")) c->tiny_max = atoi(e) != 0 ? TINY_G : 0;
    if (const char *e = getenv("SMM_SYM_DENSE")) c->sym_dense = std::max(0, std::min(2, atoi(e)));
    if (const char *e = getenv("SMM_SYM_WIDE")) c->sym_wide = atoi(e) != 0;
    if (const char *e = getenv("SMM_SYM_CCS")) c->sym_ccs = atoi(e) != 0;
    if (const char *e = getenv("SMM_PIECE_WALK")) c->piece_walk = atoi(e);
    if (const char *e = getenv("SMM_SYM_MAX_WS")) c->sym_max_ws = std::max(0, std::min(atoi(e), (int)CCS_MAX_WS));
    if (const char *e = getenv("SMM_CHECK")) c->check = atoi(e) != 0;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hip_stream == SMM_STREAM_DEFAULT) { c->stream = nullptr; c->own_stream = false; }   // the device's null stream
    else if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(SMM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    if (hipMalloc((void **)&c->d_flags, 512) != hipSuccess) { delete c; return fail(SMM_ERR_ALLOC, "hipMalloc flags"); }    // (+256: bin counts, +288 / +320: row counters)
    {
        unsigned char init[256];
        memset(init, 0, sizeof(init));
        const int neg1 = -1;
        memcpy(init + 64, &neg1, sizeof(neg1));
        if (hipMemcpy(c->d_flags, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(c->d_flags); delete c;
            return fail(SMM_ERR_HIP, "hipMemcpy of the context constants failed");
        }
    }
    {
        const unsigned clean[2] = {0u, 0xffffffffu};
        if (hipMalloc((void **)&c->d_err, 64) != hipSuccess || hipMemcpy(c->d_err, clean, sizeof(clean), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(c->d_err); (void)hipFree(c->d_flags); delete c;
            return fail(SMM_ERR_ALLOC, "hipMalloc of the context's error word failed");
        }
    }
    *out = c;
    return SMM_OK;
}

extern "C" void smm_ctx_destroy(smm_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &l : c->launches) { (void)hipEventDestroy(l.t0); (void)hipEventDestroy(l.t1); }
    for (auto &b : c->pool) (void)hipFree(b.p);
    for (auto &kv : c->live) (void)hipFree(kv.first);
    (void)hipFree(c->d_flags);
    (void)hipFree(c->d_err);
    for (int i = 0; i < smm_ctx::PIN_SLOTS; ++i) {
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Fetch-and-clear the context's plan error word (what the kernels' always-on clamps and the plan checker record);
// synchronises the stream.  A non-zero word means plan metadata was inconsistent: the result of the product that
// tripped it is not to be trusted, and the caller gets SMM_ERR_INTERNAL -- never a hang or a fault.
static int take_plan_error(smm_ctx *c, const char *where)
{
    unsigned h[16] = {0u};
    HIPCHK(hipMemcpyAsync(h, c->d_err, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!h[0]) return SMM_OK;
    if ((h[0] & PLAN_ERR_RING) && getenv("SMM_DEBUG"))
        fprintf(stderr, "[smm] ring wait ran out: need %u fmin %d pmin %d fp %u step %d wave %u row block %u T_w %u\n", h[8] >> 16, (int)(short)(h[8] & 0xffff),
                (int)h[9] >> 16, h[9] & 0xffff, (int)h[10], h[11] & 0xff, h[11] >> 8, h[12]);
    const unsigned clean[2] = {0u, 0xffffffffu};
    HIPCHK(hipMemcpyAsync(c->d_err, clean, sizeof(clean), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    static const char *names[] = {"sub-run bounds", "tail descriptor", "slab sub-run table", "start slots", "hash look-up",
                                  "list capacity", "row counts", "list entries", "ring schedule / progress words of triple-product stage 2"};
    std::string what;
    for (int b = 0; b < 9; ++b)
        if (h[0] & (1u << b)) { if (!what.empty()) what += ", "; what += names[b]; }
    return fail(SMM_ERR_INTERNAL, "%s: inconsistent plan metadata (%s; first at row %u of A) -- the result of this product is not valid; "
                                  "please report this with the operands (SMM_CHECK=1 verifies every plan)", where, what.c_str(), h[1]);
}

extern "C" int smm_ctx_synchronize(smm_ctx *c)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    return take_plan_error(c, "smm_ctx_synchronize");       // (synchronises the stream)
}

extern "C" int smm_ctx_release_pool(smm_ctx *c)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    pool_flush(c);
    return SMM_OK;
}
extern "C" int64_t smm_ctx_pool_bytes(smm_ctx *c)
{
    if (!c) return -1;
    CTX_LOCK(c);
    int64_t t = 0;
    for (auto &b : c->pool) t += (int64_t)b.bytes;
    return t;
}
extern "C" int smm_ctx_inject_alloc_failure(smm_ctx *c, int nth, int hard)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    c->inject_alloc_nth = nth > 0 ? nth : 0;
    c->inject_alloc_hard = hard != 0;
    return SMM_OK;
}
extern "C" int64_t smm_ctx_alloc_retries(smm_ctx *c)
{
    if (!c) return -1;
    CTX_LOCK(c);
    return c->alloc_retries;
}

extern "C" int smm_ctx_set_check(smm_ctx *c, int enable)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    c->check = enable != 0;
    return SMM_OK;
}

static int drain_timers(smm_ctx *c)
{
    if (c->launches.empty()) return SMM_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &l : c->launches) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, l.t0, l.t1);
        auto &t = c->totals[l.name];
        t.first += ms; t.second += 1;
        (void)hipEventDestroy(l.t0); (void)hipEventDestroy(l.t1);
    }
    c->launches.clear();
    return SMM_OK;
}
extern "C" int smm_ctx_timing(smm_ctx *c, int enable)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(drain_timers(c));
    c->timing = enable != 0;
    return SMM_OK;
}
extern "C" int smm_ctx_timing_reset(smm_ctx *c)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(drain_timers(c));
    c->totals.clear();
    return SMM_OK;
}
extern "C" int smm_ctx_kernel_time(smm_ctx *c, const char *kernel, double *ms_total, int64_t *launches)
{
    if (!c || !kernel) return fail(SMM_ERR_INVALID, "smm_ctx_kernel_time: NULL argument");
    CTX_LOCK(c);
    CHK(drain_timers(c));
    auto it = c->totals.find(kernel);
    if (ms_total) *ms_total = it == c->totals.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == c->totals.end() ? 0 : it->second.second;
    return SMM_OK;
}
extern "C" int smm_ctx_tune(smm_ctx *c, int lds_cols, int waves)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (lds_cols) {
        if (lds_cols < 64 || lds_cols > 20000) return fail(SMM_ERR_INVALID, "lds_cols must be in [64,20000]");
        c->lds_cols = lds_cols;
    }
    if (waves) {
        if (waves != 1 && waves != 2 && waves != 4 && waves != 8 && waves != 16)
            return fail(SMM_ERR_INVALID, "waves must be 1, 2, 4, 8 or 16");
        c->waves = waves;
    }
    return SMM_OK;
}
extern "C" int smm_ctx_tune_hash(smm_ctx *c, int small_max, int medium_max)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (small_max < 0 || small_max > 256 || medium_max < 0 || medium_max > 2048)
        return fail(SMM_ERR_INVALID, "hash thresholds must be in [0,256] and [0,2048]");
    c->hash_small = small_max;
    c->hash_medium = std::max(medium_max, small_max);
    return SMM_OK;
}
extern "C" int smm_ctx_tune_slab(smm_ctx *c, int mode, int ws, int rows_per_wave)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (mode < 0 || mode > 2) return fail(SMM_ERR_INVALID, "slab mode must be 0 (auto), 1 (off) or 2 (force)");
    if (ws < 0 || ws > 32766) return fail(SMM_ERR_INVALID, "slab width must be in [0,32766]");
    if (rows_per_wave != 0 && rows_per_wave != 2 && rows_per_wave != 4)
        return fail(SMM_ERR_INVALID, "rows per wave must be 0 (keep), 2 or 4");
    c->slab_mode = mode;
    c->slab_ws = ws;
    if (rows_per_wave) c->slab_rw = rows_per_wave;
    return SMM_OK;
}
extern "C" int smm_ctx_tune_narrow(smm_ctx *c, int enable)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    c->narrow_idx = enable != 0;
    return SMM_OK;
}
extern "C" int smm_ctx_tune_symbolic(smm_ctx *c, int max_slab_cols)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (max_slab_cols < 0 || max_slab_cols > CCS_MAX_WS) return fail(SMM_ERR_INVALID, "slab width must be in [0,%d]", CCS_MAX_WS);
    c->sym_max_ws = max_slab_cols;
    return SMM_OK;
}
extern "C" int smm_ctx_tune_dense_runs(smm_ctx *c, int mode)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (mode < 0 || mode > 2) return fail(SMM_ERR_INVALID, "mode must be 0 (never), 1 (chosen from the operand) or 2 (always)");
    c->sym_dense = mode;
    return SMM_OK;
}
extern "C" int smm_ctx_tune_stage2(smm_ctx *c, int ring)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    c->s2_ring = ring != 0;
    return SMM_OK;
}
extern "C" int smm_ctx_tune_shared(smm_ctx *c, int lds_cols, int waves)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    if (lds_cols) {
        if (lds_cols < 64 || lds_cols > 20000) return fail(SMM_ERR_INVALID, "lds_cols must be in [64,20000]");
        c->lds_cols_shared = lds_cols;
    }
    if (waves) {
        if (waves != 4 && waves != 8 && waves != 16) return fail(SMM_ERR_INVALID, "waves must be 4, 8 or 16");
        c->waves_shared = waves;
    }
    return SMM_OK;
}

// ------------------------------------------------------------------------------ SMM_EXACT guard
extern "C" int smm_ctx_exact_selftest(smm_ctx *c, int inject_fault)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    unsigned *d_bad = (unsigned *)((char *)c->d_flags + 240);
    HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream));
    constexpr int NTRIAL = 16384;
    LAUNCH(c, "smm_lds_order_selftest", smm_lds_order_selftest, 512, 64, 0, NTRIAL, inject_fault ? 1 : 0, d_bad);
    LAUNCH_CHECK();
    unsigned bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad)
        return fail(SMM_ERR_UNSUPPORTED,
                    "SMM_EXACT self-test failed: %u of %d accumulators differ from the lane-ordered sum%s -- this device does not "
                    "apply the lanes of one ds_add_f64 in ascending lane order, so reference-order (bit-exact) accumulation is "
                    "not available; run without SMM_EXACT (values agree to rounding)",
                    bad, NTRIAL * 32, inject_fault ? " (fault injected)" : "");
    return SMM_OK;
}
// every SMM_EXACT product goes through this first; the kernel runs once per context
static int exact_guard(smm_ctx *c, int flags)
{
    if (!(flags & SMM_EXACT) || c->exact_checked > 0) return SMM_OK;
    if (c->exact_checked == 0) {
        const char *e = getenv("SMM_EXACT_INJECT_FAULT");
        const int rc = smm_ctx_exact_selftest(c, e && *e && strcmp(e, "0") != 0);
        if (rc != SMM_OK && rc != SMM_ERR_UNSUPPORTED) return rc;      // could not run: try again next time
        c->exact_checked = rc == SMM_OK ? 1 : -1;
        if (rc != SMM_OK) return rc;
        return SMM_OK;
    }
    return fail(SMM_ERR_UNSUPPORTED, "SMM_EXACT is not available on this device (its self-test failed earlier on this context)");
}

// ------------------------------------------------------------------------------ host-side content hash
// h' = rotl(h ^ w, 27) * P is a bijection of h for a fixed word and of the word for a fixed h: a buffer that
// differs from another in one 8-byte word gives a different lane state from there on, whatever follows.
// Blocks of 4 MB are hashed independently (four interleaved lanes each; by several threads when the buffer
// is large) and their hashes are chained in order by the same step.
static inline uint64_t hash_step(uint64_t h, uint64_t w)
{
    h ^= w;
    h = (h << 27) | (h >> 37);
    return h * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
}
static inline uint64_t hash_final(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
static uint64_t hash_block(const unsigned char *p, size_t bytes)
{
    uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull, h2 = 0xA4093822299F31D0ull, h3 = 0x082EFA98EC4E6C89ull;
    size_t i = 0;
    for (; i + 32 <= bytes; i += 32) {
        uint64_t w0, w1, w2, w3;
        memcpy(&w0, p + i, 8); memcpy(&w1, p + i + 8, 8); memcpy(&w2, p + i + 16, 8); memcpy(&w3, p + i + 24, 8);
        h0 = hash_step(h0, w0); h1 = hash_step(h1, w1); h2 = hash_step(h2, w2); h3 = hash_step(h3, w3);
    }
    for (; i < bytes; i += 8) {                         // tail: whole words, the last one zero-padded
        uint64_t w = 0;
        memcpy(&w, p + i, std::min<size_t>(8, bytes - i));
        h0 = hash_step(h0, w);
    }
    uint64_t h = hash_step(hash_step(hash_step(hash_step(0x452821E638D01377ull, h0), h1), h2), h3);
    return hash_step(h, (uint64_t)bytes);
}
extern "C" uint64_t smm_host_hash64(const void *ptr, int64_t nbytes)
{
    if (!ptr || nbytes <= 0) return hash_final(0x9E3779B97F4A7C15ull);
    const unsigned char *p = (const unsigned char *)ptr;
    const size_t bytes = (size_t)nbytes;
    constexpr size_t BLK = (size_t)4 << 20;
    const size_t nblk = (bytes + BLK - 1) / BLK;
    std::vector<uint64_t> hb(nblk);
    auto run = [&](size_t b0, size_t stride) {
        for (size_t b = b0; b < nblk; b += stride) hb[b] = hash_block(p + b * BLK, std::min(BLK, bytes - b * BLK));
    };
    unsigned nt = 1;
    if (nblk >= 4) {
        unsigned hw = std::thread::hardware_concurrency();
        if (const char *e = getenv("SMM_HASH_THREADS")) hw = (unsigned)std::max(1, atoi(e));
        nt = (unsigned)std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 8), nblk);
    }
    if (nt <= 1) run(0, 1);
    else {
        // (a thread that cannot be started -- resource limits -- must not throw across the C ABI: its blocks are
        // hashed by this thread instead; the result does not depend on who hashes which block)
        std::vector<std::thread> th;
        std::vector<unsigned> inline_lanes;
        for (unsigned t = 1; t < nt; ++t) {
            try { th.emplace_back(run, (size_t)t, (size_t)nt); }
            catch (...) { inline_lanes.push_back(t); }
        }
        run(0, nt);
        for (unsigned t : inline_lanes) run(t, nt);
        for (auto &t : th) t.join();
    }
    uint64_t h = 0x3F84D5B5B5470917ull;
    for (size_t b = 0; b < nblk; ++b) h = hash_step(h, hb[b]);
    return hash_final(hash_step(h, (uint64_t)bytes));
}

// ------------------------------------------------------------------------------ result download
// Device -> host copy of a result into the caller's (pageable, usually freshly allocated) array
// (reference: sparsemat_to_csr / darray_to_numpy copy the result out, matrix_ops.py:205-240).  One big
// hipMemcpy into such memory runs at the speed of ONE thread taking page faults 4 KB at a time.  Here:
//  * the destination is first advised to use transparent huge pages (512x fewer faults where the kernel
//    allows it; harmless where it does not);
//  * chunks of 32 MB travel by DMA into a ring of pinned buffers on the context's stream, and a few host
//    threads copy finished chunks into the destination in parallel -- so the faults are taken by several
//    threads and overlap with the DMA of the following chunks.
// Small results take the plain copy (measured: at 118 MB the pipeline's start-up still cost 5 ms more than it
// saved, at 265 MB it was ahead).
// widen: the source holds int32, the destination receives int64 (CSR column indices of a result whose nnz
// does not fit int32: scipy wants indptr and indices of one dtype); `bytes` counts SOURCE bytes.
static int download(smm_ctx *c, void *dst, const void *src_dev, size_t bytes, bool widen = false)
{
    if (bytes == 0) return SMM_OK;
    static const bool plain = getenv("SMM_DOWNLOAD_PLAIN") != nullptr;      // A/B switch for scripts/api_e2e.py
    if ((bytes < ((size_t)256 << 20) || plain) && !widen) {
        HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return SMM_OK;
    }
    constexpr int R = smm_ctx::PIN_SLOTS;
    constexpr size_t CH = smm_ctx::PIN_BYTES;
    for (int i = 0; i < R; ++i) {
        if (!c->pin[i]) {
            if (hipHostMalloc(&c->pin[i], CH, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                c->pin[i] = nullptr;
                if (widen) return fail(SMM_ERR_ALLOC, "hipHostMalloc of the download ring failed");
                HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));   // no pinned memory: plain copy
                HIPCHK(hipStreamSynchronize(c->stream));
                return SMM_OK;
            }
            HIPCHK(hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
    }
    {   // huge pages for the page-aligned interior of the destination
        const size_t pg = (size_t)2 << 20;
        const size_t dbytes = widen ? 2 * bytes : bytes;
        const uintptr_t lo = ((uintptr_t)dst + pg - 1) & ~(uintptr_t)(pg - 1), hi = ((uintptr_t)dst + dbytes) & ~(uintptr_t)(pg - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    const size_t nchunks = (bytes + CH - 1) / CH;
    const int W = (int)std::min<size_t>(4, nchunks);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> issued(nchunks, 0), copied(nchunks, 0);
    hipError_t worker_err = hipSuccess;
    auto worker = [&](int w) {
        (void)hipSetDevice(c->device);
        for (size_t i = (size_t)w; i < nchunks; i += (size_t)W) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return issued[i] != 0; });
                if (issued[i] == 2) return;                         // the issuing side failed: stop
            }
            const hipError_t e = hipEventSynchronize(c->pin_ev[i % R]);
            const size_t off = i * CH, len = std::min(CH, bytes - off);
            if (e == hipSuccess) {
                if (!widen) memcpy((char *)dst + off, c->pin[i % R], len);
                else {
                    const int32_t *sp = (const int32_t *)c->pin[i % R];
                    int64_t *dp = (int64_t *)dst + off / sizeof(int32_t);
                    for (size_t k = 0; k < len / sizeof(int32_t); ++k) dp[k] = sp[k];
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (e != hipSuccess) worker_err = e;
                copied[i] = 1;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threads;
    bool spawned = true;
    try {
        for (int w = 0; w < W; ++w) threads.emplace_back(worker, w);
    } catch (...) {                                     // (resource limits: nothing may throw across the C ABI)
        spawned = false;
    }
    if (!spawned) {
        // the workers that did start are told to stop; this thread moves the chunks alone, one at a time
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t k = 0; k < nchunks; ++k) issued[k] = 2;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
        for (size_t i = 0; i < nchunks; ++i) {
            const size_t off = i * CH, len = std::min(CH, bytes - off);
            HIPCHK(hipMemcpyAsync(c->pin[0], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!widen) memcpy((char *)dst + off, c->pin[0], len);
            else {
                const int32_t *sp = (const int32_t *)c->pin[0];
                int64_t *dp = (int64_t *)dst + off / sizeof(int32_t);
                for (size_t k = 0; k < len / sizeof(int32_t); ++k) dp[k] = sp[k];
            }
        }
        return SMM_OK;
    }
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < nchunks; ++i) {
        if (i >= (size_t)R) {                                       // the slot's previous chunk must have left it
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return copied[i - R] != 0; });
        }
        const size_t off = i * CH, len = std::min(CH, bytes - off);
        if (err == hipSuccess) err = hipMemcpyAsync(c->pin[i % R], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, c->stream);
        if (err == hipSuccess) err = hipEventRecord(c->pin_ev[i % R], c->stream);
        {
            std::lock_guard<std::mutex> lk(mu);
            issued[i] = err == hipSuccess ? 1 : 2;
            if (err != hipSuccess)
                for (size_t k = i; k < nchunks; ++k) issued[k] = 2;
        }
        cv.notify_all();
        if (err != hipSuccess) break;
    }
    for (auto &t : threads) t.join();
    if (err == hipSuccess) err = worker_err;
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(SMM_ERR_HIP, "result download: %s", hipGetErrorString(err)); }
    return SMM_OK;
}

// ------------------------------------------------------------------------------ operands
struct smm_csr {
    smm_ctx *ctx = nullptr;
    int64_t rows = 0, cols = 0, nnz = 0;
    const int *ptr = nullptr, *idx = nullptr;
    const double *val = nullptr;
    bool owned = false;
    bool validated = false;
    unsigned vflags = 0;
    // cached tile indices (sorted operands only), one per tile geometry that has been asked for:
    // plans keep the pointer of theirs, so a later product with another geometry (SMM_EXACT vs
    // default, another tuning, the ELL chunks of the triple product) never invalidates it
    struct SegCache { int wf, n_ft; int *seg; };
    struct LocCache { int wc; short *loc; };          // tile-local columns for coarse width wc
    struct SlabCache { int ws, n_slabs; int *soff; short *scol; double *sval; };   // slab-major copy (smm_slab.hpp)
    std::vector<SegCache> segs;
    std::vector<LocCache> locs;
    struct PackCache { int wc, nct; int2 *desc; double *pay; int maxlen; };   // packed tile-major payload (smm_pack_*); longest piece
    std::vector<SlabCache> slabs;
    std::vector<PackCache> packs;
    // chunk-padded 16-bit column stream per column slab (smm_ccs_*): the symbolic phase's gather stream
    struct CcsCache { int ws, n_slabs, bm_words, guard_chunk; int *cptr; unsigned short *stream; double same_word; };   // same_word: share of neighbouring entries in one bitmap word
    std::vector<CcsCache> ccs;
    unsigned short *idx16 = nullptr;             // 16-bit copy of idx (cols < 65535): the symbolic phase's gather stream
    int *idx_pad = nullptr;                      // borrowed operands with >= 65535 columns: a copy of idx with two ints of slack (wide symbolic walk)
    // sliced-ELL copy for triple-product stage 2 (chunk width ell_chunk)
    int ell_chunk = 0, ell_nchunks = 0; bool ell_spread = false; int64_t *ell_off = nullptr;
    short *ell_col = nullptr; double *ell_val = nullptr;
    int64_t ell_bytes = 0;                       // HBM of the ELL copy
    // scheduled streams for the ring kernel of triple-product stage 2 (smm_ring.hpp), one per (k-group, wave)
    int64_t *ring_off = nullptr; short *ring_col = nullptr; double *ring_val = nullptr; unsigned *ring_hdr = nullptr;
    int ring_npieces = 0; bool ring_spread = false; int64_t ring_bytes = 0;
    int64_t derived_bytes = 0;                   // HBM of every other cached copy (tile indices, payloads, ...)
};

static int validate(smm_ctx *c, smm_csr *m)
{
    if (m->validated) return (m->vflags & CSR_BAD) ? fail(SMM_ERR_INVALID, "malformed CSR operand") : SMM_OK;
    HIPCHK(hipMemsetAsync(c->d_flags, 0, sizeof(unsigned), c->stream));
    const int grid = (int)std::min<int64_t>(std::max<int64_t>((m->rows + 3) / 4, 1), 8192);
    LAUNCH(c, "smm_validate", smm_validate, grid, 256, 0, (int)m->rows, (int)m->cols, (int)m->nnz, m->ptr, m->idx,
           c->d_flags);
    LAUNCH_CHECK();
    unsigned f = 0;
    HIPCHK(hipMemcpyAsync(&f, c->d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    m->vflags = f;
    m->validated = true;
    if (f & CSR_BAD)
        return fail(SMM_ERR_INVALID, "malformed CSR operand (non-monotone indptr or column index out of range)");
    return SMM_OK;
}

static int csr_common(smm_ctx *c, int64_t rows, int64_t cols, int64_t nnz)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    if (rows < 0 || cols < 0 || nnz < 0) return fail(SMM_ERR_INVALID, "negative dimension");
    if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX)
        return fail(SMM_ERR_INVALID, "operand dimensions/nnz must be < 2^31 (int32 indices, as the reference)");
    HIPCHK(hipSetDevice(c->device));
    return SMM_OK;
}

extern "C" int smm_csr_from_host(smm_ctx *c, int64_t rows, int64_t cols, int64_t nnz, const int32_t *indptr,
                                 const int32_t *indices, const double *data, smm_csr **out)
{
    if (!out) return fail(SMM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    CHK(csr_common(c, rows, cols, nnz));
    CTX_LOCK(c);
    if (!indptr || (nnz > 0 && (!indices || !data))) return fail(SMM_ERR_INVALID, "NULL CSR array");
    int *dp = nullptr, *di = nullptr; double *dv = nullptr;
    if (dev_malloc(c, (void **)&dp, (rows + 1) * sizeof(int)) != hipSuccess ||
        dev_malloc(c, (void **)&di, (std::max<int64_t>(nnz, 1) + 2) * sizeof(int)) != hipSuccess ||     // + slack: the wide symbolic walk reads columns in pairs
        dev_malloc(c, (void **)&dv, std::max<int64_t>(nnz, 1) * sizeof(double)) != hipSuccess) {
        (void)hipFree(dp); (void)hipFree(di); (void)hipFree(dv);
        return fail(SMM_ERR_ALLOC, "hipMalloc of a CSR operand failed");
    }
    HIPCHK(hipMemcpyAsync(dp, indptr, (rows + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        HIPCHK(hipMemcpyAsync(di, indices, nnz * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dv, data, nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    smm_csr *m = new smm_csr();
    m->ctx = c; m->rows = rows; m->cols = cols; m->nnz = nnz;
    m->ptr = dp; m->idx = di; m->val = dv; m->owned = true;
    int rc = validate(c, m);
    if (rc != SMM_OK) { smm_csr_destroy(m); return rc; }
    *out = m;
    return SMM_OK;
}

extern "C" int smm_csr_from_device(smm_ctx *c, int64_t rows, int64_t cols, int64_t nnz, const int32_t *d_indptr,
                                   const int32_t *d_indices, const double *d_data, smm_csr **out)
{
    if (!out) return fail(SMM_ERR_INVALID, "out is NULL");
    *out = nullptr;
    CHK(csr_common(c, rows, cols, nnz));
    CTX_LOCK(c);
    if (!d_indptr || (nnz > 0 && (!d_indices || !d_data))) return fail(SMM_ERR_INVALID, "NULL CSR array");
    smm_csr *m = new smm_csr();
    m->ctx = c; m->rows = rows; m->cols = cols; m->nnz = nnz;
    m->ptr = d_indptr; m->idx = d_indices; m->val = d_data; m->owned = false;
    int rc = validate(c, m);
    if (rc != SMM_OK) { delete m; return rc; }
    *out = m;
    return SMM_OK;
}

extern "C" void smm_csr_destroy(smm_csr *m)
{
    if (!m) return;
    CTX_LOCK(m->ctx);
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->owned) { (void)hipFree((void *)m->ptr); (void)hipFree((void *)m->idx); (void)hipFree((void *)m->val); }
    for (auto &e : m->segs) (void)hipFree(e.seg);
    for (auto &e : m->locs) (void)hipFree(e.loc);
    for (auto &e : m->slabs) { (void)hipFree(e.soff); (void)hipFree(e.scol); (void)hipFree(e.sval); }
    for (auto &e : m->packs) { (void)hipFree(e.desc); (void)hipFree(e.pay); }
    for (auto &e : m->ccs) { (void)hipFree(e.cptr); (void)hipFree(e.stream); }
    (void)hipFree(m->idx16); (void)hipFree(m->idx_pad);
    (void)hipFree(m->ell_off); (void)hipFree(m->ell_col); (void)hipFree(m->ell_val);
    (void)hipFree(m->ring_off); (void)hipFree(m->ring_col); (void)hipFree(m->ring_val); (void)hipFree(m->ring_hdr);
    delete m;
}
extern "C" int64_t smm_csr_rows(const smm_csr *m) { return m ? m->rows : -1; }
extern "C" int64_t smm_csr_cols(const smm_csr *m) { return m ? m->cols : -1; }
extern "C" int64_t smm_csr_nnz(const smm_csr *m) { return m ? m->nnz : -1; }
extern "C" int smm_csr_is_canonical(smm_ctx *c, smm_csr *m)
{
    if (!c || !m) return fail(SMM_ERR_INVALID, "NULL argument");
    CTX_LOCK(c);
    CHK(validate(c, m));
    return (m->vflags & (CSR_UNSORTED | CSR_HAS_EQUAL)) ? 0 : 1;
}

extern "C" int64_t smm_csr_device_bytes(const smm_csr *m)
{
    if (!m) return -1;
    CTX_LOCK(m->ctx);
    int64_t own = 0;
    if (m->owned) own = (m->rows + 1) * (int64_t)sizeof(int) + (std::max<int64_t>(m->nnz, 1) + 2) * (int64_t)sizeof(int) + std::max<int64_t>(m->nnz, 1) * (int64_t)sizeof(double);
    return own + m->derived_bytes + m->ell_bytes + m->ring_bytes;
}

// Tile geometry: nct coarse tiles of wc = nw*wf columns; fine tile t covers [t*wf,(t+1)*wf).
struct Geom { int nct, wc, wf, n_ft, nw; };
static Geom make_geom(const smm_ctx *c, int64_t ncols, const smm_csr *b, bool exact)
{
    Geom g;
    if (!exact) {
        // shared-tile walk: one coarse tile per workgroup, waves split the chunks; the tile
        // index has one entry per coarse tile
        g.nw = c->waves_shared;
        const int64_t cols = std::max<int64_t>(ncols, 1);
        g.nct = (int)((cols + c->lds_cols_shared - 1) / c->lds_cols_shared);
        g.wc = (int)((cols + g.nct - 1) / g.nct);
        g.wf = g.wc;
        g.n_ft = g.nct;
        return g;
    }
    // exact walk: wave w of the workgroup owns fine tile w of the coarse tile; LDS also holds
    // sizeof(ExactScratch) per wave behind the accumulators
    g.nw = c->waves;
    const int64_t cols = std::max<int64_t>(ncols, 1);
    const int64_t lds_max = ((int64_t)160 * 1024 - (int64_t)g.nw * (int64_t)sizeof(ExactScratch) - 64 * 8) / 8 - 2;
    const int64_t wc_max = std::max<int64_t>(std::min<int64_t>(c->lds_cols, lds_max), g.nw);
    g.nct = (int)((cols + wc_max - 1) / wc_max);
    const int64_t per = (cols + g.nct - 1) / g.nct;
    g.wf = (int)((per + g.nw - 1) / g.nw);
    g.wc = g.wf * g.nw;
    if (g.wc > wc_max) {                      // rounding up to a multiple of nw overshot the LDS budget
        g.nct += 1;
        g.wf = (int)(((cols + g.nct - 1) / g.nct + g.nw - 1) / g.nw);
        g.wc = g.wf * g.nw;
    }
    g.n_ft = g.nct * g.nw;
    return g;
}

static int ensure_seg(smm_ctx *c, smm_csr *b, const Geom &g, const int **out)
{
    for (auto &e : b->segs)
        if (e.wf == g.wf && e.n_ft == g.n_ft) { *out = e.seg; return SMM_OK; }
    const int64_t total = b->rows * (int64_t)(g.n_ft + 1);
    int *seg = nullptr;
    if (dev_malloc(c, (void **)&seg, std::max<int64_t>(total, 1) * sizeof(int)) != hipSuccess)
        return fail(SMM_ERR_ALLOC, "hipMalloc of the tile index failed");
    if (total > 0) {
        LAUNCH(c, "smm_segptr", smm_segptr, (total + 255) / 256, 256, 0, (int)b->rows, g.n_ft, g.wf, b->ptr, b->idx, seg);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(seg); return fail(SMM_ERR_HIP, "smm_segptr: %s", hipGetErrorString(e)); }
    }
    b->segs.push_back({g.wf, g.n_ft, seg});
    b->derived_bytes += std::max<int64_t>(total, 1) * (int64_t)sizeof(int);
    *out = seg;
    return SMM_OK;
}

static int ensure_idx16(smm_ctx *c, smm_csr *b)
{
    if (b->idx16) return SMM_OK;
    if (b->cols >= 65535) return fail(SMM_ERR_INVALID, "16-bit column copy needs < 65535 columns");
    // + 2 entries of slack: the wide symbolic walk reads columns in pairs
    if (dev_malloc(c, (void **)&b->idx16, (std::max<int64_t>(b->nnz, 1) + 2) * sizeof(unsigned short)) != hipSuccess)
        return fail(SMM_ERR_ALLOC, "hipMalloc of the 16-bit column copy failed");
    if (b->nnz > 0) {
        LAUNCH(c, "smm_idx16", smm_idx16, std::min<int64_t>((b->nnz + 255) / 256, 65536), 256, 0, (int)b->nnz, b->idx, b->idx16);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(b->idx16); b->idx16 = nullptr; return fail(SMM_ERR_HIP, "smm_idx16: %s", hipGetErrorString(e)); }
    }
    b->derived_bytes += (std::max<int64_t>(b->nnz, 1) + 2) * (int64_t)sizeof(unsigned short);
    return SMM_OK;
}

// The wide 32-bit symbolic walk loads columns in pairs, so the pair of a row's last, odd entry reaches one int
// past the array.  Arrays this library allocated carry that slack; a BORROWED array (smm_csr_from_device) ends
// where the caller's allocation may end, so the walk reads a padded copy of it instead (made once per handle).
static int idx_with_slack(smm_ctx *c, smm_csr *b, const int **out)
{
    if (b->owned) { *out = b->idx; return SMM_OK; }
    if (!b->idx_pad) {
        const size_t n = (size_t)std::max<int64_t>(b->nnz, 1) + 2;
        if (dev_malloc(c, (void **)&b->idx_pad, n * sizeof(int)) != hipSuccess)
            return fail(SMM_ERR_ALLOC, "hipMalloc of the padded column copy failed");
        hipError_t e = hipMemsetAsync(b->idx_pad + (n - 2), 0, 2 * sizeof(int), c->stream);
        if (e == hipSuccess && b->nnz > 0)
            e = hipMemcpyAsync(b->idx_pad, b->idx, (size_t)b->nnz * sizeof(int), hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) { (void)hipFree(b->idx_pad); b->idx_pad = nullptr; return fail(SMM_ERR_HIP, "padded column copy: %s", hipGetErrorString(e)); }
        b->derived_bytes += (int64_t)(n * sizeof(int));
    }
    *out = b->idx_pad;
    return SMM_OK;
}

static int ensure_loc(smm_ctx *c, smm_csr *b, const Geom &g, const short **out)
{
    for (auto &e : b->locs)
        if (e.wc == g.wc) { *out = e.loc; return SMM_OK; }
    if (g.wc > 32767) return fail(SMM_ERR_INVALID, "coarse tile wider than 32767 columns");
    short *loc = nullptr;
    if (dev_malloc(c, (void **)&loc, std::max<int64_t>(b->nnz, 1) * sizeof(short)) != hipSuccess)
        return fail(SMM_ERR_ALLOC, "hipMalloc of the tile-local column array failed");
    if (b->nnz > 0) {
        LAUNCH(c, "smm_loc16", smm_loc16, std::min<int64_t>((b->nnz + 255) / 256, 65536), 256, 0, (int)b->nnz, g.wc, b->idx, loc);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { (void)hipFree(loc); return fail(SMM_ERR_HIP, "smm_loc16: %s", hipGetErrorString(e)); }
    }
    b->locs.push_back({g.wc, loc});
    b->derived_bytes += std::max<int64_t>(b->nnz, 1) * (int64_t)sizeof(short);
    *out = loc;
    return SMM_OK;
}

// Sliced-ELL re-layout of H for triple-product stage 2 (see smm_triple_stage2); one copy is cached per
// handle, for one chunk width and one order of the steps (spread = any order inside a segment, the default
// mode's; otherwise stored order with idle steps, SMM_EXACT's -- smm_ell_fill).
static int ensure_ell(smm_ctx *c, smm_csr *h, int nchunks, int chunk, bool spread)
{
    if (h->ell_val && h->ell_chunk == chunk && h->ell_nchunks == nchunks && h->ell_spread == spread) return SMM_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(h->ell_off); (void)hipFree(h->ell_col); (void)hipFree(h->ell_val);
    h->ell_off = nullptr; h->ell_col = nullptr; h->ell_val = nullptr; h->ell_chunk = 0; h->ell_bytes = 0;
    Geom gh; gh.nw = 1; gh.nct = nchunks; gh.wf = chunk; gh.wc = chunk; gh.n_ft = nchunks;
    const int *hseg = nullptr;
    CHK(ensure_seg(c, h, gh, &hseg));
    const int n = (int)h->rows;
    const int nslices = (n + WAVE - 1) / WAVE;
    const int64_t items = (int64_t)nchunks * nslices;
    if (items >= 0x7fffffff) return fail(SMM_ERR_INVALID, "H too large for the sliced-ELL index (%lld blocks)", (long long)items);
    int64_t *cnt = nullptr;
    CHK(pool_get(c, (size_t)items, &cnt));
    if (dev_malloc(c, (void **)&h->ell_off, (size_t)(items + 1) * sizeof(int64_t)) != hipSuccess) {
        pool_free(c, cnt);
        return fail(SMM_ERR_ALLOC, "hipMalloc of the ELL index failed");
    }
    EllArgs E{};
    E.n = n; E.nchunks = nchunks; E.chunk = chunk; E.nslices = nslices;
    E.h_ptr = h->ptr; E.h_idx = h->idx; E.h_val = h->val; E.hseg = hseg;
    E.cnt = cnt; E.off = h->ell_off;
    const int grid = (int)((items + 3) / 4);
    LAUNCH(c, "smm_ell_count", smm_ell_count, grid, 256, 0, E);
    LAUNCH(c, "smm_scan", smm_scan<int64_t>, 1, 1024, 0, (int)items, (const int64_t *)cnt, h->ell_off);
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, h->ell_off + items, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    pool_free(c, cnt);
    // + one wave of slack: stage 2 requests step 0 of a block before it knows the block is empty
    if (dev_malloc(c, (void **)&h->ell_col, (total + WAVE) * sizeof(short)) != hipSuccess ||
        dev_malloc(c, (void **)&h->ell_val, (total + WAVE) * sizeof(double)) != hipSuccess)
        return fail(SMM_ERR_ALLOC, "hipMalloc of the ELL payload (%lld entries) failed", (long long)total);
    E.col = h->ell_col; E.val = h->ell_val;
    if (spread) LAUNCH(c, "smm_ell_fill", smm_ell_fill<1>, grid, 256, 0, E);
    else LAUNCH(c, "smm_ell_fill", smm_ell_fill<2>, grid, 256, 0, E);
    LAUNCH_CHECK();
    h->ell_chunk = chunk; h->ell_nchunks = nchunks; h->ell_spread = spread;
    h->ell_bytes = (items + 1) * (int64_t)sizeof(int64_t) + (total + WAVE) * (int64_t)(sizeof(short) + sizeof(double));
    return SMM_OK;
}

// Scheduled streams of H for the ring kernel of triple-product stage 2 (smm_ring.hpp); one copy cached per handle and
// order (spread = default mode: a lane picks among its next 8 entries; otherwise stored order, SMM_EXACT).
static int ring_fill(smm_ctx *c, smm_csr *h, RingBuildArgs &B, bool spread)
{
    if (spread) LAUNCH(c, "smm_ring_build", (smm_ring_build<8, true>), B.nkg, 1024, 0, B);
    else LAUNCH(c, "smm_ring_build", (smm_ring_build<1, true>), B.nkg, 1024, 0, B);
    LAUNCH_CHECK();
    return SMM_OK;
}
static int ensure_ring(smm_ctx *c, smm_csr *h, bool spread)
{
    const int npieces = (int)((h->cols + RING_PW - 1) / RING_PW);
    if (h->ring_val && h->ring_npieces == npieces && h->ring_spread == spread) return SMM_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(h->ring_off); (void)hipFree(h->ring_col); (void)hipFree(h->ring_val); (void)hipFree(h->ring_hdr);
    h->ring_off = nullptr; h->ring_col = nullptr; h->ring_val = nullptr; h->ring_hdr = nullptr; h->ring_bytes = 0;
    const int n = (int)h->rows;
    const int nkg = (n + 16 * WAVE - 1) / (16 * WAVE);
    const int64_t streams = (int64_t)nkg * 16;
    int64_t *cnt = nullptr;
    CHK(pool_get(c, (size_t)streams, &cnt));
    if (dev_malloc(c, (void **)&h->ring_off, (size_t)(streams + 1) * sizeof(int64_t)) != hipSuccess) {
        pool_free(c, cnt);
        return fail(SMM_ERR_ALLOC, "hipMalloc of the ring stream index failed");
    }
    RingBuildArgs B{};
    B.n = n; B.K = (int)h->cols; B.npieces = npieces; B.nkg = nkg;
    B.h_ptr = h->ptr; B.h_idx = h->idx; B.h_val = h->val;
    B.cnt = cnt; B.off = h->ring_off; B.err = c->d_err;
    if (spread) LAUNCH(c, "smm_ring_build", (smm_ring_build<8, false>), nkg, 1024, 0, B);
    else LAUNCH(c, "smm_ring_build", (smm_ring_build<1, false>), nkg, 1024, 0, B);
    LAUNCH(c, "smm_scan", smm_scan<int64_t>, 1, 1024, 0, (int)streams, (const int64_t *)cnt, h->ring_off);
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, h->ring_off + streams, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    pool_free(c, cnt);
    if (dev_malloc(c, (void **)&h->ring_col, (size_t)(total + 1) * WAVE * sizeof(short)) != hipSuccess ||
        dev_malloc(c, (void **)&h->ring_val, (size_t)(total + 1) * WAVE * sizeof(double)) != hipSuccess ||
        dev_malloc(c, (void **)&h->ring_hdr, (size_t)(total + streams + 64) * sizeof(unsigned)) != hipSuccess)
        return fail(SMM_ERR_ALLOC, "hipMalloc of the ring streams (%lld steps) failed", (long long)total);
    B.col = h->ring_col; B.val = h->ring_val; B.hdr = h->ring_hdr;
    CHK(ring_fill(c, h, B, spread));
    h->ring_npieces = npieces; h->ring_spread = spread;
    h->ring_bytes = (streams + 1) * (int64_t)sizeof(int64_t) + (total + 1) * WAVE * (int64_t)(sizeof(short) + sizeof(double)) +
                    (total + streams + 64) * (int64_t)sizeof(unsigned);
    return take_plan_error(c, "smm_ring_build");
}

static int check_pair(smm_ctx *c, smm_csr *a, smm_csr *b)
{
    if (!c || !a || !b) return fail(SMM_ERR_INVALID, "NULL argument");
    if (a->ctx != c || b->ctx != c) return fail(SMM_ERR_INVALID, "operand belongs to another context");
    if (a->cols != b->rows)       /* matrix_ops.py:312-313, sparse_sparse_dense.cpp:83-86 */
        return fail(SMM_ERR_INVALID, "Matrix dimensions are incompatible for multiplication (%lld x %lld times %lld x %lld)",
                    (long long)a->rows, (long long)a->cols, (long long)b->rows, (long long)b->cols);
    HIPCHK(hipSetDevice(c->device));
    CHK(validate(c, a));
    CHK(validate(c, b));
    return SMM_OK;
}

extern "C" int smm_row_products(smm_ctx *c, const smm_csr *a, const smm_csr *b, int64_t *products_host)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(check_pair(c, (smm_csr *)a, (smm_csr *)b));
    if (!products_host) return fail(SMM_ERR_INVALID, "products_host is NULL");
    if (a->rows == 0) return SMM_OK;
    int64_t *d = nullptr;
    CHK(pool_get(c, (size_t)a->rows, &d));
    const int grid = (int)std::min<int64_t>((a->rows + 3) / 4, 16384);
    LAUNCH(c, "smm_row_work", smm_row_work, grid, 256, 0, (int)a->rows, (int)b->cols, (int64_t)0, 0, a->ptr, a->idx,
           b->ptr, d, (int64_t *)nullptr);
    LAUNCH_CHECK();
    HIPCHK(hipMemcpyAsync(products_host, d, a->rows * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    pool_free(c, d);
    return SMM_OK;
}

// ------------------------------------------------------------------------------ numeric dispatch
template <int OUT, bool SYM, int NW, bool EXACT, bool SCR = false, bool L16 = false, bool SLAB = false>
static int launch_numeric_t(smm_ctx *c, NumericArgs &args)
{
    if constexpr (OUT == OUT_SPARSE && !L16) {
        if (args.list16) return launch_numeric_t<OUT, SYM, NW, EXACT, SCR, true>(c, args);
    }
    if constexpr (OUT == OUT_SPARSE && L16 && !SCR && !SLAB) {
        if (args.runs2) return launch_numeric_t<OUT, SYM, NW, EXACT, false, true, true>(c, args);
    }
    // accumulator tile (+ the exact walk's per-wave scratch behind it)
    // accumulators (+ the exact walk's per-wave scratch and the workgroup's 64-slot sink behind them)
    const size_t lds = (size_t)(((args.wc + 1) & ~1) + 2) * sizeof(double) + (EXACT ? (size_t)NW * sizeof(ExactScratch) + 64 * sizeof(double) : 0);
    auto kern = smm_numeric<OUT, SYM, NW, EXACT, SCR, L16, SLAB>;
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t grid = (int64_t)args.m * args.nct;
    if (grid > 0x7fffffff) return fail(SMM_ERR_INVALID, "too many (row, tile) units for one launch");
    args.unit_counter = nullptr; args.n_units = (unsigned)grid;
    // Persistent workgroups where every unit is real work (CSR output, no triangle): -0.3 ms at configs[1], -0.9 ms on a
    // configs[4] share, -0.6 ms under SMM_EXACT.  Dense output measured neutral (configs[2]) and, with the trivial units
    // below the diagonal, slower (stage 1 of configs[3]: 19.2 against 18.4 ms) -- those keep one unit per workgroup
    // unless SMM_NUMERIC_PERSIST=2 (profiles/r4_numeric_persist.txt).
    const bool persist = c->numeric_persist == 2 || (c->numeric_persist == 1 && OUT == OUT_SPARSE);
    if (persist && grid > c->n_cu) {
        // as many workgroups as are resident at once (more would only find the counter spent)
        const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(32 / NW, (int64_t)(160 * 1024) / (int64_t)(lds + 64)));
        args.unit_counter = (unsigned *)((char *)c->d_flags + 220);
        HIPCHK(hipMemsetAsync(args.unit_counter, 0, sizeof(unsigned), c->stream));
        grid = std::min<int64_t>(grid, per_cu * c->n_cu);
    }
#ifdef SMM_STAMPS
    unsigned long long *d_st = nullptr;
    CHK(pool_get(c, 4, &d_st));
    (void)hipMemsetAsync(d_st, 0, 32, c->stream);
    args.stamps = d_st;
#endif
    LAUNCH(c, SCR ? "smm_emit" : OUT == OUT_SPARSE ? "smm_numeric" : "smm_numeric_dense", kern, grid, NW * 64, lds, args);
    hipError_t e = hipGetLastError();
#ifdef SMM_STAMPS
    {
        unsigned long long h[4];
        (void)hipMemcpyAsync(h, d_st, 32, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        const double tot = (double)(h[0] + h[1] + h[2]);
        fprintf(stderr, "[SMM_STAMPS] units=%lld NW=%d wc=%d nct=%d exact=%d  init %.1f%%  accumulate %.1f%%  epilogue %.1f%%  (sum %.3g cycles)\n",
                (long long)grid, NW, args.wc, args.nct, (int)EXACT, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, tot);
        pool_free(c, d_st);
    }
#endif
    if (e != hipSuccess) return fail(SMM_ERR_HIP, "smm_numeric launch: %s", hipGetErrorString(e));
    return SMM_OK;
}
template <int OUT>
static int launch_numeric(smm_ctx *c, NumericArgs &args, bool sym, int nw, bool exact)
{
    if (args.m <= 0) return SMM_OK;
    args.dummy_idx = (const int *)((const char *)c->d_flags + 64);
    args.dummy_val = (const double *)((const char *)c->d_flags + 128);
    args.err = c->d_err;
#define SMM_CASE(S, N, X) \
    if (sym == S && nw == N && exact == X) return launch_numeric_t<OUT, S, N, X>(c, args);
    SMM_CASE(false, 1, true) SMM_CASE(true, 1, true) SMM_CASE(false, 2, true) SMM_CASE(true, 2, true)
    SMM_CASE(false, 4, true) SMM_CASE(true, 4, true) SMM_CASE(false, 8, true) SMM_CASE(true, 8, true)
    SMM_CASE(false, 16, true) SMM_CASE(true, 16, true)
    SMM_CASE(false, 4, false) SMM_CASE(true, 4, false) SMM_CASE(false, 8, false) SMM_CASE(true, 8, false)
    SMM_CASE(false, 16, false) SMM_CASE(true, 16, false)
#undef SMM_CASE
    return fail(SMM_ERR_INVALID, "unsupported numeric configuration");
}

constexpr size_t LIST_SLACK = (size_t)1 << 17;      // entries behind the ordered lists (see smm_spgemm_symbolic)
#ifndef SMM_CCS_UNROLL
#define SMM_CCS_UNROLL 8        // chunk loads in flight per wave of smm_symbolic_ccs (configs[1]: 2: 6.7, 4: 5.5, 8: 5.2 ms)
#endif
// the instantiation of the chunked symbolic walk: triangle, dense runs of columns in B, units per counter round trip
using ccs_kernel_t = decltype(&smm_symbolic_ccs<false, SMM_CCS_UNROLL>);
static ccs_kernel_t ccs_kernel(bool sym, bool dr, bool batch)
{
#define CCS_K(S, D, B) smm_symbolic_ccs<S, SMM_CCS_UNROLL, D, B>
    if (batch) return dr ? (sym ? CCS_K(true, true, 16) : CCS_K(false, true, 16)) : (sym ? CCS_K(true, false, 16) : CCS_K(false, false, 16));
    return dr ? (sym ? CCS_K(true, true, 1) : CCS_K(false, true, 1)) : (sym ? CCS_K(true, false, 1) : CCS_K(false, false, 1));
#undef CCS_K
}
// ------------------------------------------------------------------------------ CSR x CSR -> CSR
struct SlabGeom { int ws, n_slabs, rw; };
constexpr int SLAB_NW = 8;                       // waves per workgroup of smm_dense_slab
struct smm_plan {
    smm_ctx *ctx = nullptr;
    smm_csr *a = nullptr, *b = nullptr;
    int flags = 0;
    int64_t row_offset = 0;
    int64_t m = 0, ncols = 0, nnz = 0;
    bool b_sorted = true;
    Geom g{};
    int64_t *d_ub_off = nullptr;   // m+1
    void *d_tmp = nullptr;         // capacity-strided ordered column lists (int32, or uint16 when list16)
    bool list16 = false;
    unsigned *d_P = nullptr;       // nnz(A)
    unsigned *d_runs = nullptr;    // nnz(A) x (nct+1)
    int2 *d_tail = nullptr;        // m: where the tail of every row starts (smm_runs)
    // column slabs (B wider than one chunked-stream slab): slab-local lists, per-(slab,row) counts, smm_runs_slab's tables
    int n_slabs = 1, tps = 0, ws = 0;
    int *d_scnt = nullptr;         // n_slabs x m
    unsigned *d_dst0 = nullptr;    // nnz(A)
    uint2 *d_runs2 = nullptr;      // nct x nnz(A)
    unsigned char *d_tflag = nullptr;   // nct x m: which (tile, row) units hold entries of C
    const int *seg = nullptr;      // B's tile index and tile-local columns for geometry g (owned by b)
    const short *loc = nullptr;
    smm_csr::PackCache pack{0, 0, nullptr, nullptr, 0};   // default mode: packed payload of B for geometry g
    bool use_slab = false;         // dense-bin rows: smm_dense_slab -> scratch -> emission, instead of the tile kernel
    SlabGeom sg{};
    smm_csr::SlabCache slab{0, 0, nullptr, nullptr, nullptr};
    int *d_rowcnt = nullptr;       // m
    int64_t total_cap = 0;         // sum of the list capacities (= d_ub_off's last entry)
    int *d_lists = nullptr;        // 5 x m: rows of the small / medium / dense / tiny (16) / tiny (32) bins
    int n_bin[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t *d_cptr = nullptr;     // m+1
};

extern "C" void smm_plan_destroy(smm_plan *p)
{
    if (!p) return;
    smm_ctx *c = p->ctx;
    CTX_LOCK(c);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    pool_free(c, p->d_ub_off); pool_free(c, p->d_tmp); pool_free(c, p->d_P); pool_free(c, p->d_runs);
    pool_free(c, p->d_rowcnt); pool_free(c, p->d_cptr); pool_free(c, p->d_lists); pool_free(c, p->d_tail);
    pool_free(c, p->d_scnt); pool_free(c, p->d_dst0); pool_free(c, p->d_runs2); pool_free(c, p->d_tflag);
    delete p;
}
extern "C" int64_t smm_plan_nnz(const smm_plan *p) { return p ? p->nnz : -1; }
extern "C" int64_t smm_plan_device_bytes(const smm_plan *p)
{
    if (!p) return -1;
    smm_ctx *c = p->ctx;
    CTX_LOCK(c);
    int64_t total = 0;
    const void *blocks[] = {p->d_ub_off, p->d_tmp, p->d_P, p->d_runs, p->d_rowcnt, p->d_cptr, p->d_lists, p->d_tail,
                            p->d_scnt, p->d_dst0, p->d_runs2, p->d_tflag};
    for (const void *b : blocks) {
        auto it = c->live.find((void *)b);
        if (b && it != c->live.end()) total += (int64_t)it->second;
    }
    return total;
}

// exclusive scan of n values into out[0..n] (out[n] = total)
template <typename T>
static int scan_launch(smm_ctx *c, int64_t n, const T *in, int64_t *out)
{
    if (n <= 8 * SCAN_TILE) {
        LAUNCH(c, "smm_scan", smm_scan<T>, 1, 1024, 0, (int)n, in, out);
        LAUNCH_CHECK();
        return SMM_OK;
    }
    const int tiles = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    int64_t *sums = nullptr;
    CHK(pool_get(c, (size_t)2 * tiles + 1, &sums));
    LAUNCH(c, "smm_scan", smm_scan_tile_sums<T>, tiles, 1024, 0, (int)n, in, sums);
    LAUNCH(c, "smm_scan", smm_scan<int64_t>, 1, 1024, 0, tiles, (const int64_t *)sums, sums + tiles);
    LAUNCH(c, "smm_scan", smm_scan_tiles<T>, tiles, 1024, 0, (int)n, in, (const int64_t *)(sums + tiles), out);
    hipError_t e = hipGetLastError();
    pool_free(c, sums);         // stream-ordered reuse: the pool hands it out again only to work queued behind these launches
    if (e != hipSuccess) return fail(SMM_ERR_HIP, "scan: %s", hipGetErrorString(e));
    return SMM_OK;
}

// Packed tile-major payload of B for the shared-tile walk (smm_pack_* in smm_kernels.hpp), cached per geometry.
// (descriptors are handed out BY VALUE: a plan must not point into the operand's vector, which may grow)
static int ensure_pack(smm_ctx *c, smm_csr *b, const Geom &g, smm_csr::PackCache *out)
{
    for (auto &e : b->packs)
        if (e.wc == g.wc && e.nct == g.nct) { *out = e; return SMM_OK; }
    if (g.wc > 32767) return fail(SMM_ERR_INVALID, "coarse tile wider than 32767 columns");
    Geom gs = g; gs.wf = g.wc; gs.n_ft = g.nct;               // the coarse-tile index (shared walk: one entry per coarse tile)
    const int *seg = nullptr;
    CHK(ensure_seg(c, b, gs, &seg));
    const int64_t cells = (int64_t)g.nct * b->rows;
    if (cells + 1 >= INT32_MAX) return fail(SMM_ERR_INVALID, "too many (tile, row) pieces");
    int *units = nullptr; int64_t *off64 = nullptr;
    CHK(pool_get(c, (size_t)std::max<int64_t>(cells, 1), &units));
    int rc = pool_get(c, (size_t)cells + 1, &off64);
    if (rc != SMM_OK) { pool_free(c, units); return rc; }
    int *d_maxlen = (int *)((char *)c->d_flags + 244);
    (void)hipMemsetAsync(d_maxlen, 0, sizeof(int), c->stream);
    if (cells > 0) LAUNCH(c, "smm_pack_count", smm_pack_count, (cells + 255) / 256, 256, 0, (int)b->rows, g.nct, seg, units, d_maxlen);
    rc = scan_launch<int>(c, cells, units, off64);
    int64_t total = 0;
    int maxlen = 0;
    if (rc == SMM_OK) {
        hipError_t e = hipMemcpyAsync(&total, off64 + cells, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&maxlen, d_maxlen, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(SMM_ERR_HIP, "pack build: %s", hipGetErrorString(e));
    }
    if (rc == SMM_OK && total >= INT32_MAX) rc = fail(SMM_ERR_INVALID, "operand too large for the packed payload");
    smm_csr::PackCache e{g.wc, g.nct, nullptr, nullptr, maxlen};
    // (+ 4 units of slack: the piece walk's last lane reads up to three 8-byte units past a very short last piece)
    if (rc == SMM_OK &&
        (dev_malloc(c, (void **)&e.desc, (size_t)std::max<int64_t>(cells, 1) * sizeof(int2)) != hipSuccess ||
         dev_malloc(c, (void **)&e.pay, (size_t)(std::max<int64_t>(total, 1) + 4) * sizeof(double)) != hipSuccess)) {
        (void)hipFree(e.desc); (void)hipFree(e.pay);
        rc = fail(SMM_ERR_ALLOC, "hipMalloc of the packed payload failed");
    }
    if (rc == SMM_OK && cells > 0) {
        LAUNCH(c, "smm_pack_desc", smm_pack_desc, (cells + 255) / 256, 256, 0, (int)b->rows, g.nct, seg, (const int64_t *)off64, e.desc);
        LAUNCH(c, "smm_pack_fill", smm_pack_fill, std::min<int64_t>((b->rows + 3) / 4, 65536), 256, 0, (int)b->rows, g.nct, g.wc, b->ptr,
               b->idx, b->val, seg, (const int2 *)e.desc, e.pay);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);        // units / off64 go back to the pool
        if (he != hipSuccess) { (void)hipFree(e.desc); (void)hipFree(e.pay); rc = fail(SMM_ERR_HIP, "pack build: %s", hipGetErrorString(he)); }
    }
    pool_free(c, units); pool_free(c, off64);
    if (rc != SMM_OK) return rc;
    b->packs.push_back(e);
    b->derived_bytes += std::max<int64_t>(cells, 1) * (int64_t)sizeof(int2) + (std::max<int64_t>(total, 1) + 4) * (int64_t)sizeof(double);
    *out = e;
    return SMM_OK;
}

// Chunk-padded 16-bit column stream of B for the symbolic walk (smm_ccs_* / smm_symbolic_ccs), cached per slab
// geometry.  (Handed out BY VALUE, like the packed payload: the operand's vector may grow.)
static int ensure_ccs(smm_ctx *c, smm_csr *b, int ws, int n_slabs, smm_csr::CcsCache *out)
{
    for (auto &e : b->ccs)
        if (e.ws == ws && e.n_slabs == n_slabs) { *out = e; return SMM_OK; }
    if (ws > CCS_MAX_WS) return fail(SMM_ERR_INVALID, "column slab wider than %d columns", CCS_MAX_WS);
    Geom gs; gs.nw = 1; gs.nct = n_slabs; gs.wc = ws; gs.wf = ws; gs.n_ft = n_slabs;
    const int *seg = nullptr;
    CHK(ensure_seg(c, b, gs, &seg));
    const int64_t cells = (int64_t)n_slabs * b->rows;
    if (cells + n_slabs + 1 >= INT32_MAX) return fail(SMM_ERR_INVALID, "too many (slab, row) pieces");
    int *chunks = nullptr; int64_t *off64 = nullptr;
    CHK(pool_get(c, (size_t)std::max<int64_t>(cells, 1), &chunks));
    int rc = pool_get(c, (size_t)cells + 1, &off64);
    if (rc != SMM_OK) { pool_free(c, chunks); return rc; }
    if (cells > 0) LAUNCH(c, "smm_ccs_count", smm_ccs_count, (cells + 255) / 256, 256, 0, (int)b->rows, n_slabs, seg, chunks);
    rc = scan_launch<int>(c, cells, chunks, off64);
    int64_t total = 0;
    if (rc == SMM_OK) {
        hipError_t e = hipMemcpyAsync(&total, off64 + cells, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(SMM_ERR_HIP, "chunked stream build: %s", hipGetErrorString(e));
    }
    if (rc == SMM_OK && total + 1 >= (INT32_MAX / CCS_CHUNK)) rc = fail(SMM_ERR_INVALID, "operand too large for the chunked column stream");
    smm_csr::CcsCache e{ws, n_slabs, (ws + 31) / 32, (int)total, nullptr, nullptr, 0.0};
    unsigned long long *d_stat = nullptr;
    if (rc == SMM_OK) rc = pool_get(c, 2, &d_stat);
    if (rc == SMM_OK && hipMemsetAsync(d_stat, 0, 2 * sizeof(unsigned long long), c->stream) != hipSuccess) rc = fail(SMM_ERR_HIP, "memset");
    const int64_t ptr_entries = (int64_t)n_slabs * (b->rows + 1);
    if (rc == SMM_OK &&
        (dev_malloc(c, (void **)&e.cptr, (size_t)ptr_entries * sizeof(int)) != hipSuccess ||
         dev_malloc(c, (void **)&e.stream, (size_t)(total + 1) * CCS_CHUNK * sizeof(unsigned short)) != hipSuccess)) {
        (void)hipFree(e.cptr); (void)hipFree(e.stream);
        rc = fail(SMM_ERR_ALLOC, "hipMalloc of the chunked column stream failed");
    }
    if (rc == SMM_OK) {
        LAUNCH(c, "smm_ccs_ptr", smm_ccs_ptr, (ptr_entries + 255) / 256, 256, 0, (int)b->rows, n_slabs, (const int64_t *)off64, e.cptr);
        LAUNCH(c, "smm_ccs_fill", smm_ccs_fill, std::min<int64_t>(std::max<int64_t>((cells + 3) / 4, 1), 65536), 256, 0, (int)b->rows, n_slabs, ws,
               e.bm_words, b->idx, seg, (const int *)e.cptr, e.stream, d_stat);
        hipError_t he = hipGetLastError();
        unsigned long long stat[2] = {0, 0};
        if (he == hipSuccess) he = hipMemcpyAsync(stat, d_stat, sizeof(stat), hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);        // chunks / off64 go back to the pool
        e.same_word = stat[1] ? (double)stat[0] / (double)stat[1] : 0.0;
        if (he != hipSuccess) { (void)hipFree(e.cptr); (void)hipFree(e.stream); rc = fail(SMM_ERR_HIP, "chunked stream build: %s", hipGetErrorString(he)); }
    }
    pool_free(c, chunks); pool_free(c, off64); pool_free(c, d_stat);
    if (rc != SMM_OK) return rc;
    b->ccs.push_back(e);
    b->derived_bytes += ptr_entries * (int64_t)sizeof(int) + (total + 1) * CCS_CHUNK * (int64_t)sizeof(unsigned short);
    *out = e;
    return SMM_OK;
}

// ------------------------------------------------------------------------------ row block x column slab path

// Slab width: the slab's share of B's payload (10 bytes per entry) should sit in one XCD's 4 MiB L2 next
// to the streams that pass through it, and R rows of it must fit the LDS.  Returns false when the
// slab kernels cannot run for this operand.
static bool slab_geometry(const smm_ctx *c, const smm_csr *b, int64_t ncols, SlabGeom *g)
{
    if ((b->vflags & CSR_UNSORTED) || ncols <= 0 || b->nnz <= 0) return false;
    const int rw = c->slab_rw;
    const int R = SLAB_NW * rw;
    const int64_t lds_doubles = ((int64_t)160 * 1024 - (int64_t)SLAB_NW * (int64_t)sizeof(SlabScratch) - (int64_t)sizeof(SlabShared)) / 8;
    int64_t ws_lds = (lds_doubles / R) & ~(int64_t)1;
    int64_t ws = c->slab_ws;
    if (ws <= 0) {
        const double l2_bytes = 3.0e6;
        ws = (int64_t)(l2_bytes * (double)ncols / (10.0 * (double)b->nnz));
        if (ws < 64) ws = 64;
    }
    ws = std::min<int64_t>(std::min<int64_t>(ws, ws_lds), std::min<int64_t>(ncols, 32766));
    if (ws < 1) return false;
    int64_t ns = (ncols + ws - 1) / ws;
    if (ns >= 8 && c->slab_ws <= 0) ns = (ns + 7) & ~(int64_t)7;       // whole slabs per XCD
    ws = (ncols + ns - 1) / ns;
    ws = (ws + 1) & ~(int64_t)1;
    if (ws > ws_lds) ws = ws_lds;
    ns = (ncols + ws - 1) / ws;
    if (ns * b->rows + 1 >= INT32_MAX) return false;
    g->ws = (int)ws; g->n_slabs = (int)ns; g->rw = rw;
    return true;
}

static int ensure_slab(smm_ctx *c, smm_csr *b, const SlabGeom &g, smm_csr::SlabCache *out)
{
    for (auto &e : b->slabs)
        if (e.ws == g.ws && e.n_slabs == g.n_slabs) { *out = e; return SMM_OK; }
    Geom gs; gs.nw = 1; gs.nct = g.n_slabs; gs.wc = g.ws; gs.wf = g.ws; gs.n_ft = g.n_slabs;
    const int *seg = nullptr;
    CHK(ensure_seg(c, b, gs, &seg));
    const int64_t cells = (int64_t)g.n_slabs * b->rows;
    int *cnt = nullptr; int64_t *off64 = nullptr;
    CHK(pool_get(c, (size_t)std::max<int64_t>(cells, 1), &cnt));
    int rc = pool_get(c, (size_t)cells + 1, &off64);
    if (rc != SMM_OK) { pool_free(c, cnt); return rc; }
    smm_csr::SlabCache e{g.ws, g.n_slabs, nullptr, nullptr, nullptr};
    if (dev_malloc(c, (void **)&e.soff, (size_t)(cells + 1) * sizeof(int)) != hipSuccess ||
        dev_malloc(c, (void **)&e.scol, (size_t)std::max<int64_t>(b->nnz, 1) * sizeof(short)) != hipSuccess ||
        dev_malloc(c, (void **)&e.sval, (size_t)std::max<int64_t>(b->nnz, 1) * sizeof(double)) != hipSuccess) {
        (void)hipFree(e.soff); (void)hipFree(e.scol); (void)hipFree(e.sval);
        pool_free(c, cnt); pool_free(c, off64);
        return fail(SMM_ERR_ALLOC, "hipMalloc of the slab-major copy of B failed");
    }
    if (cells > 0) LAUNCH(c, "smm_slab_count", smm_slab_count, (cells + 255) / 256, 256, 0, (int)b->rows, g.n_slabs, seg, cnt);
    rc = scan_launch<int>(c, cells, cnt, off64);
    if (rc == SMM_OK) {
        LAUNCH(c, "smm_narrow32", smm_narrow32, std::min<int64_t>((cells + 256) / 256, 65536), 256, 0, cells + 1, (const int64_t *)off64, e.soff);
        if (b->rows > 0)
            LAUNCH(c, "smm_slab_fill", smm_slab_fill, std::min<int64_t>((b->rows + 3) / 4, 65536), 256, 0, (int)b->rows, g.n_slabs, g.ws,
                   b->ptr, b->idx, b->val, seg, (const int *)e.soff, e.scol, e.sval);
        hipError_t he = hipGetLastError();
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);        // cnt / off64 go back to the pool
        if (he != hipSuccess) rc = fail(SMM_ERR_HIP, "slab build: %s", hipGetErrorString(he));
    }
    pool_free(c, cnt); pool_free(c, off64);
    if (rc != SMM_OK) { (void)hipFree(e.soff); (void)hipFree(e.scol); (void)hipFree(e.sval); return rc; }
    b->slabs.push_back(e);
    b->derived_bytes += (cells + 1) * (int64_t)sizeof(int) + std::max<int64_t>(b->nnz, 1) * (int64_t)(sizeof(short) + sizeof(double));
    *out = e;
    return SMM_OK;
}

// ------------------------------------------------------------------------------ values-only update
// New values on an unchanged pattern: the operand's value array is overwritten in place and every cached
// copy that carries values is re-filled in place by the kernel that built it (same pattern -> same
// positions), so the pointers plans hold stay valid.  Everything is queued on the context's stream, behind
// whatever product is still running there.
static int refresh_value_copies(smm_ctx *c, smm_csr *m)
{
    for (auto &e : m->packs) {
        Geom gs; gs.nw = 1; gs.nct = e.nct; gs.wc = e.wc; gs.wf = e.wc; gs.n_ft = e.nct;
        const int *seg = nullptr;
        CHK(ensure_seg(c, m, gs, &seg));
        if (m->rows > 0 && m->nnz > 0)
            LAUNCH(c, "smm_pack_fill", smm_pack_fill, std::min<int64_t>((m->rows + 3) / 4, 65536), 256, 0, (int)m->rows, e.nct, e.wc, m->ptr,
                   m->idx, m->val, seg, (const int2 *)e.desc, e.pay);
    }
    for (auto &e : m->slabs) {
        Geom gs; gs.nw = 1; gs.nct = e.n_slabs; gs.wc = e.ws; gs.wf = e.ws; gs.n_ft = e.n_slabs;
        const int *seg = nullptr;
        CHK(ensure_seg(c, m, gs, &seg));
        if (m->rows > 0 && m->nnz > 0)
            LAUNCH(c, "smm_slab_fill", smm_slab_fill, std::min<int64_t>((m->rows + 3) / 4, 65536), 256, 0, (int)m->rows, e.n_slabs, e.ws,
                   m->ptr, m->idx, m->val, seg, (const int *)e.soff, e.scol, e.sval);
    }
    if (m->ring_val) {
        RingBuildArgs B{};
        B.n = (int)m->rows; B.K = (int)m->cols; B.npieces = m->ring_npieces; B.nkg = (int)((m->rows + 16 * WAVE - 1) / (16 * WAVE));
        B.h_ptr = m->ptr; B.h_idx = m->idx; B.h_val = m->val; B.off = m->ring_off; B.err = c->d_err;
        B.col = m->ring_col; B.val = m->ring_val; B.hdr = m->ring_hdr;
        CHK(ring_fill(c, m, B, m->ring_spread));
    }
    if (m->ell_val) {
        Geom gh; gh.nw = 1; gh.nct = m->ell_nchunks; gh.wf = m->ell_chunk; gh.wc = m->ell_chunk; gh.n_ft = m->ell_nchunks;
        const int *hseg = nullptr;
        CHK(ensure_seg(c, m, gh, &hseg));
        EllArgs E{};
        E.n = (int)m->rows; E.nchunks = m->ell_nchunks; E.chunk = m->ell_chunk; E.nslices = (int)((m->rows + WAVE - 1) / WAVE);
        E.h_ptr = m->ptr; E.h_idx = m->idx; E.h_val = m->val; E.hseg = hseg;
        E.cnt = nullptr; E.off = m->ell_off; E.col = m->ell_col; E.val = m->ell_val;
        const int64_t items = (int64_t)E.nchunks * E.nslices;
        const int grid = (int)((items + 3) / 4);
        if (items > 0) {
            if (m->ell_spread) LAUNCH(c, "smm_ell_fill", smm_ell_fill<1>, grid, 256, 0, E);
            else LAUNCH(c, "smm_ell_fill", smm_ell_fill<2>, grid, 256, 0, E);
        }
    }
    LAUNCH_CHECK();
    return SMM_OK;
}

extern "C" int smm_csr_update_values(smm_ctx *c, smm_csr *m, const double *data)
{
    if (!c || !m) return fail(SMM_ERR_INVALID, "NULL argument");
    if (m->ctx != c) return fail(SMM_ERR_INVALID, "operand belongs to another context");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    if (!m->owned) return fail(SMM_ERR_INVALID, "smm_csr_update_values: the operand borrows its arrays (smm_csr_from_device); "
                                                "rewrite them and call smm_csr_update_values_device");
    if (m->nnz == 0) return SMM_OK;
    if (!data) return fail(SMM_ERR_INVALID, "data is NULL");
    HIPCHK(hipMemcpyAsync((void *)m->val, data, (size_t)m->nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));          // the caller's host array may change as soon as this returns
    return refresh_value_copies(c, m);
}

extern "C" int smm_csr_update_values_device(smm_ctx *c, smm_csr *m, const double *d_data)
{
    if (!c || !m) return fail(SMM_ERR_INVALID, "NULL argument");
    if (m->ctx != c) return fail(SMM_ERR_INVALID, "operand belongs to another context");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    if (m->nnz == 0) return SMM_OK;
    if (d_data && d_data != m->val) {
        if (!m->owned) return fail(SMM_ERR_INVALID, "smm_csr_update_values_device: a borrowed operand is updated by rewriting its "
                                                    "own array (pass NULL or that array)");
        HIPCHK(hipMemcpyAsync((void *)m->val, d_data, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    return refresh_value_copies(c, m);
}

// rows (a row list, or rows 0..m-1 of A) x all slabs -> `out`, rows of the launch ldo apart, column order
template <bool NEGZERO>
static int launch_slab(smm_ctx *c, const smm_csr *a, const smm_csr *b, const smm_csr::SlabCache &sl, int rw, int m,
                       const int *rowlist, bool sym, int64_t row_offset, double *out, int64_t ldo)
{
    if (m <= 0) return SMM_OK;
    SlabArgs S{};
    const int R = SLAB_NW * rw;
    S.m = m; S.ncols = (int)b->cols; S.ws = sl.ws; S.n_slabs = sl.n_slabs; S.n_rb = (m + R - 1) / R; S.rowsB = (int)b->rows;
    S.row_offset = row_offset; S.rowlist = rowlist;
    S.kmax = (int)std::max<int64_t>(b->nnz - 1, 0);
    S.a_ptr = a->ptr; S.a_idx = a->idx; S.a_val = a->val;
    S.soff = sl.soff; S.scol = sl.scol; S.sval = sl.sval;
    S.dummy_idx = (const int *)((const char *)c->d_flags + 64);
    S.dummy_val = (const double *)((const char *)c->d_flags + 128);
    S.out = out; S.ldo = ldo;
    const int64_t units = (int64_t)S.n_rb * S.n_slabs;
    S.cpx = (int)((units + 7) / 8);
    const int64_t grid = (int64_t)S.cpx * 8;
    if (grid > 0x7fffffff) return fail(SMM_ERR_INVALID, "too many (row block, slab) units for one launch");
    const size_t lds = (size_t)R * ((sl.ws + 1) & ~1) * sizeof(double) + (size_t)SLAB_NW * sizeof(SlabScratch) + sizeof(SlabShared);
#define SLAB_CASE(S_, RW_)                                                                                       \
    if (sym == S_ && rw == RW_) {                                                                                \
        auto kern = smm_dense_slab<S_, SLAB_NW, RW_, NEGZERO>;                                                   \
        if (lds > 64 * 1024)                                                                                     \
            HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        LAUNCH(c, "smm_dense_slab", kern, grid, SLAB_NW * 64, lds, S);                                           \
        LAUNCH_CHECK();                                                                                          \
        return SMM_OK;                                                                                           \
    }
    SLAB_CASE(false, 2) SLAB_CASE(true, 2) SLAB_CASE(false, 4) SLAB_CASE(true, 4)
#undef SLAB_CASE
    return fail(SMM_ERR_INVALID, "unsupported slab configuration");
}

// Does the slab path pay?  It moves A's metadata once per slab and, for CSR output, the dense scratch
// twice; the tile kernel moves 10 bytes per product through the fabric.  `cells` = rows x columns of the
// launch, `products` their multiply-adds (an estimate is enough), `nnz_c` < 0 for dense output.
static bool slab_pays(const smm_ctx *c, const smm_csr *a, const SlabGeom &g, double cells, double products, double nnz_c,
                      double b_nnz_per_row, double ncols)
{
    if (c->slab_mode == 1) return false;
    if (c->slab_mode == 2) return true;
    if (cells <= 0) return false;
    // Measured (profiles/r2_c_slab_sweep.txt): with pieces of B's rows of 6-13 entries (50k x 50k, d = 0.01,
    // slabs of 570-1100 columns) the kernel is bound by the L1's line look-ups -- every piece is its own
    // 128-byte line, ~26 look-ups per 64-lane chunk against ~5 for the tile kernel -- and loses 1.4-1.8x.
    // It is therefore chosen only where a slab-sized piece of a row of B is long.
    const double piece = (double)b_nnz_per_row * g.ws / std::max<double>(1.0, ncols);
    if (piece < 24.0) return false;
    const double tile_bytes = 10.0 * products;
    double slab_bytes = 12.0 * (double)a->nnz * g.n_slabs + 8.0 * (double)a->nnz * g.n_slabs   /* A and soff per slab */
                        + 2.5 * products;                                                          /* ~3/4 of the gather hits L2 */
    if (nnz_c >= 0) slab_bytes += 16.0 * cells;                                                    /* scratch out and back */
    return slab_bytes < 0.7 * tile_bytes && products > 1e7;
}

template <bool SYM, bool SAFE, int MARK, int UNROLL, bool I16>
static int launch_symbolic_w(smm_ctx *c, smm_plan *p, int words, unsigned *gbm, int grid, int wpb, const int *rowlist,
                             const int *d_nrows, int *d_row_counter, int rbatch);
template <bool SYM, bool SAFE, int MARK, int UNROLL = 16>
static int launch_symbolic_t(smm_ctx *c, smm_plan *p, int words, unsigned *gbm, int grid, int wpb, const int *rowlist,
                             const int *d_nrows, int *d_row_counter, int rbatch = 1)
{
    return p->list16 ? launch_symbolic_w<SYM, SAFE, MARK, UNROLL, true>(c, p, words, gbm, grid, wpb, rowlist, d_nrows, d_row_counter, rbatch)
                     : launch_symbolic_w<SYM, SAFE, MARK, UNROLL, false>(c, p, words, gbm, grid, wpb, rowlist, d_nrows, d_row_counter, rbatch);
}
template <bool SYM, bool SAFE, int MARK, int UNROLL, bool I16>
static int launch_symbolic_w(smm_ctx *c, smm_plan *p, int words, unsigned *gbm, int grid, int wpb, const int *rowlist,
                             const int *d_nrows, int *d_row_counter, int rbatch)
{
    // LDS per wave: bitmap words + the guard word, or the hash slots
    const size_t lds = MARK == MARK_GLOBAL_BITMAP ? 0 : (size_t)(words + (MARK == MARK_LDS_HASH ? 0 : 1)) * wpb * sizeof(unsigned);
    auto kern = smm_symbolic<SYM, SAFE, MARK, UNROLL, I16>;
    const int *b_idx32 = p->b->idx;
#ifndef SMM_SYMW_UNROLL
#define SMM_SYMW_UNROLL 4
#endif
#ifndef SMM_SYMW_DEEP
#define SMM_SYMW_DEEP 16
#endif
    if constexpr (!SAFE && MARK != MARK_LDS_HASH) {
        // chunks of 128 entries, two columns per lane (16-bit columns: the idle column must fit 16 bits)
        // The pair of an odd chunk's last entry reaches one column past the row: the 16-bit copy and the 32-bit
        // arrays this library allocates have slack for that, a borrowed 32-bit array is read through a padded copy.
        if (c->sym_wide && (!I16 || words * 32 + 31 <= 65535)) {
            kern = smm_symbolic<SYM, false, MARK, (UNROLL == 16 ? SMM_SYMW_UNROLL : SMM_SYMW_DEEP), I16, true>;
            if (!I16) CHK(idx_with_slack(c, p->b, &b_idx32));
        }
    }
    if (lds > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LAUNCH(c, MARK == MARK_LDS_HASH ? "smm_symbolic_hash" : "smm_symbolic", kern, grid, wpb * 64, lds, (int)p->m, rowlist,
           d_nrows, p->row_offset, words, p->a->ptr, p->a->idx, p->b->ptr,
           I16 ? (const void *)p->b->idx16 : (const void *)b_idx32, p->d_ub_off, p->d_tmp, p->d_P,
           p->d_rowcnt, gbm, d_row_counter, rbatch);
    LAUNCH_CHECK();
    return SMM_OK;
}

// The plan checker (smm_plan_check_* in smm_kernels.hpp): every invariant the numeric phase relies on, verified on the
// device; violations come back as SMM_ERR_INTERNAL through the context's error word.
static int plan_check(smm_ctx *c, smm_plan *p)
{
    const int64_t m = p->m;
    if (m == 0 || !p->d_ub_off || !p->d_rowcnt) return take_plan_error(c, "smm_plan_check");      // zero operand: nothing was planned
    const smm_csr *a = p->a;
    const int grid = (int)std::min<int64_t>((m + 3) / 4, 16384);
    const bool slabs = p->d_scnt != nullptr;
    const int ns = slabs ? p->n_slabs : 1;
    const int *cnt = slabs ? p->d_scnt : p->d_rowcnt;
    if (p->list16)
        LAUNCH(c, "smm_plan_check", smm_plan_check_rows<unsigned short>, grid, 256, 0, (int)m, ns, slabs ? p->ws : (int)p->ncols, (int)p->ncols,
               (int64_t)a->nnz, p->total_cap, a->ptr, (const int64_t *)p->d_ub_off, cnt, (const int *)p->d_rowcnt, (const int64_t *)p->d_cptr,
               (const unsigned *)p->d_P, (const unsigned short *)p->d_tmp, c->d_err);
    else
        LAUNCH(c, "smm_plan_check", smm_plan_check_rows<int>, grid, 256, 0, (int)m, ns, (int)p->ncols, (int)p->ncols,
               (int64_t)a->nnz, p->total_cap, a->ptr, (const int64_t *)p->d_ub_off, cnt, (const int *)p->d_rowcnt, (const int64_t *)p->d_cptr,
               (const unsigned *)p->d_P, (const int *)p->d_tmp, c->d_err);
    LAUNCH_CHECK();
    CHK(take_plan_error(c, "smm_plan_check"));          // the table checks below walk the lists through what was just verified
    const int nd = p->n_bin[2];
    if (nd > 0 && p->d_lists) {
        const int *rows = p->d_lists + 2 * m;
        const int rgrid = (int)std::min<int64_t>((nd + 3) / 4, 16384);
        if (p->d_runs2)
            LAUNCH(c, "smm_plan_check", smm_plan_check_runs2, rgrid, 256, 0, nd, (int)m, p->n_slabs, p->tps, p->g.nct, p->g.wc, (int64_t)a->nnz, rows,
                   a->ptr, (const int64_t *)p->d_ub_off, (const int *)p->d_scnt, (const int64_t *)p->d_cptr, (const unsigned *)p->d_P,
                   (const unsigned short *)p->d_tmp, (const uint2 *)p->d_runs2, c->d_err);
        else if (p->d_runs && p->list16)
            LAUNCH(c, "smm_plan_check", smm_plan_check_runs<unsigned short>, rgrid, 256, 0, nd, p->g.nct, p->g.wc, rows, a->ptr,
                   (const int64_t *)p->d_ub_off, (const int *)p->d_rowcnt, (const unsigned *)p->d_P, (const unsigned short *)p->d_tmp,
                   (const unsigned *)p->d_runs, (const int2 *)p->d_tail, c->d_err);
        else if (p->d_runs)
            LAUNCH(c, "smm_plan_check", smm_plan_check_runs<int>, rgrid, 256, 0, nd, p->g.nct, p->g.wc, rows, a->ptr,
                   (const int64_t *)p->d_ub_off, (const int *)p->d_rowcnt, (const unsigned *)p->d_P, (const int *)p->d_tmp,
                   (const unsigned *)p->d_runs, (const int2 *)p->d_tail, c->d_err);
        LAUNCH_CHECK();
    }
    return take_plan_error(c, "smm_plan_check");
}

extern "C" int smm_plan_check(smm_ctx *c, smm_plan *p)
{
    if (!c || !p || p->ctx != c) return fail(SMM_ERR_INVALID, "bad plan/context");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    return plan_check(c, p);
}

// Test hook: damage one piece of the plan's metadata on the device (kinds: smm_plan_corrupt in smm_kernels.hpp).
extern "C" int smm_plan_inject_fault(smm_ctx *c, smm_plan *p, int kind)
{
    if (!c || !p || p->ctx != c) return fail(SMM_ERR_INVALID, "bad plan/context");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    if (kind < 1 || kind > 8) return fail(SMM_ERR_INVALID, "fault kind must be in [1,8]");
    if (p->nnz == 0 || !p->d_lists) return fail(SMM_ERR_INVALID, "fault injection needs a plan with a non-empty result");
    if ((kind <= 3 && !p->d_runs) || (kind == 6 && !p->d_runs2)) return fail(SMM_ERR_INVALID, "this plan has no such table");
    // the first row of the tile bin (the tables exist for those rows only), else of the medium / small hash bin
    const int bin = p->n_bin[2] > 0 ? 2 : (p->n_bin[1] > 0 ? 1 : 0);
    if (p->n_bin[bin] == 0 || (bin != 2 && (kind <= 3 || kind == 6))) return fail(SMM_ERR_INVALID, "no row to damage for this fault kind");
    int row = 0;
    HIPCHK(hipMemcpyAsync(&row, p->d_lists + (size_t)bin * p->m, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(smm_plan_corrupt, dim3(1), dim3(64), 0, c->stream, kind, row, p->g.nct, (int64_t)p->a->nnz, p->a->ptr, p->d_ub_off,
                       p->d_rowcnt, p->d_P, p->d_tmp, p->list16 ? 1 : 0, p->d_runs, p->d_tail, p->d_runs2);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(c->stream));
    return SMM_OK;
}

extern "C" int smm_spgemm_symbolic(smm_ctx *c, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset,
                                   smm_plan **plan, int64_t *nnz_out)
{
    if (!plan) return fail(SMM_ERR_INVALID, "plan is NULL");
    *plan = nullptr;
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(check_pair(c, a, b));
    if (a_row_offset < 0) return fail(SMM_ERR_INVALID, "negative a_row_offset");
    CHK(exact_guard(c, flags));
    smm_plan *p = new smm_plan();
    p->ctx = c; p->a = a; p->b = b; p->flags = flags; p->row_offset = a_row_offset;
    p->m = a->rows; p->ncols = b->cols;
    p->b_sorted = !(b->vflags & CSR_UNSORTED);
    p->g = make_geom(c, p->ncols, b, (flags & SMM_EXACT) != 0);
    const bool sym = flags & SMM_SYMMETRIC;
    const int64_t m = p->m;
    int rc = SMM_OK;
#define PCHK(expr) do { rc = (expr); if (rc != SMM_OK) { smm_plan_destroy(p); return rc; } } while (0)
    PCHK(pool_get(c, (size_t)m + 1, &p->d_cptr));
    if (m == 0 || a->nnz == 0 || b->nnz == 0 || p->ncols == 0) {
        // sparse_sparse_sparse.cpp:181-185: zero operand -> rowPtr of zeros only
        hipError_t e = hipMemsetAsync(p->d_cptr, 0, (m + 1) * sizeof(int64_t), c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
        p->nnz = 0;
        if (nnz_out) *nnz_out = 0;
        *plan = p;
        return SMM_OK;
    }
    // capacities of the per-row ordered lists
    int64_t *d_prod = nullptr, *d_ub = nullptr;
    PCHK(pool_get(c, (size_t)m, &d_prod));
    PCHK(pool_get(c, (size_t)m, &d_ub));
    PCHK(pool_get(c, (size_t)m + 1, &p->d_ub_off));
    const int wgrid = (int)std::min<int64_t>((m + 3) / 4, 16384);
    if (a->nnz <= 8 * m) {       // short rows on average: a row per lane instead of a row per wave; rows beyond 32 entries on a list
        int *d_long = nullptr;
        PCHK(pool_get(c, (size_t)(a->nnz / ROW_WORK_SHORT_MAX + 2), &d_long));      // [0]: count, [1 ...]: rows
        hipError_t e = hipMemsetAsync(d_long, 0, sizeof(int), c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
        LAUNCH(c, "smm_row_work", smm_row_work_short, (int)std::min<int64_t>((m + 255) / 256, 65536), 256, 0, (int)m, (int)p->ncols,
               p->row_offset, sym ? 1 : 0, a->ptr, a->idx, b->ptr, d_prod, d_ub, d_long + 1, d_long);
        LAUNCH(c, "smm_row_work", smm_row_work_listed, c->n_cu * 8, 256, 0, (const int *)d_long, (const int *)(d_long + 1), (int)p->ncols,
               p->row_offset, sym ? 1 : 0, a->ptr, a->idx, b->ptr, d_prod, d_ub);
        pool_free(c, d_long);           // stream-ordered: only work queued behind these launches can get it
    } else
        LAUNCH(c, "smm_row_work", smm_row_work, wgrid, 256, 0, (int)m, (int)p->ncols, p->row_offset, sym ? 1 : 0, a->ptr,
               a->idx, b->ptr, d_prod, d_ub);
    PCHK(scan_launch<int64_t>(c, m, d_ub, p->d_ub_off));
    // The marker of the symbolic phase is a bitmap of B's columns (ncols/8 bytes per wave, in LDS when
    // it fits): at 50 000 columns 24 waves fit a CU, at 1e6 columns one.  Rows with few products --
    // known now -- therefore take an LDS hash set instead, wherever that is the smaller marker:
    //   four classes: <= 256 / 512 / 1024 / 2048 products in 512 / 1024 / 2048 / 4096 slots (2 ... 16 KB per wave; round 4:
    //   there used to be two, and a band of half-width 8 -- 289 products, 33 columns per row -- ran 8 waves per CU in the
    //   4096-slot class: 11.6 ms; in the 1024-slot class 4.9 ms).
    const int bm_words = (int)((p->ncols + 31) / 32);
    const size_t bm_bytes = (size_t)(bm_words + 1) * sizeof(unsigned);      // + the guard word
    const bool ldsbm = bm_bytes <= 128 * 1024;
    const bool safe = (b->vflags & (CSR_HAS_EQUAL | CSR_UNSORTED)) != 0;
    constexpr int NHC = 4;
    constexpr int HS[NHC] = {512, 1024, 2048, 4096};
    int hmax[NHC];
    for (int i = 0; i < NHC; ++i) hmax[i] = (!safe && bm_bytes > (size_t)HS[i] * 4) ? HS[i] / 2 : (i ? hmax[i - 1] : 0);
    const int hmax1 = hmax[NHC - 1];
    constexpr int SB_REST = NHC, SB_TINY = NHC + 1, SB_TINY2 = NHC + 2, SB_N = NHC + 3;      // bins of the symbolic phase: hash classes, bitmap, tiny (16 / 32 lanes per row)
    // ... and TINY rows (<= 16 products from <= 16 entries of A, whatever B looks like) go four to a wave (smm_symbolic_tiny)
    const int tiny_max = c->tiny_max;
    const bool binned = hmax1 > 0 || tiny_max > 0;
    int sbin[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    sbin[SB_REST] = (int)m;
    int *d_slists = nullptr;
    int *d_scounts = (int *)((char *)c->d_flags + 256);
    if (binned) {
        PCHK(pool_get(c, (size_t)SB_N * m, &d_slists));
        hipError_t e = hipMemsetAsync(d_scounts, 0, 8 * sizeof(int), c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
        BinSpec spec{NHC, {hmax[0], hmax[1], hmax[2], hmax[3], 0, 0}, SB_TINY, SB_TINY2};
        LAUNCH(c, "smm_bin_rows", smm_bin_rows<int64_t>, std::min<int64_t>((m + 1023) / 1024, 2048), 1024, 0, (int)m, spec,
               (const int64_t *)d_ub, d_slists, d_scounts, tiny_max, (const int64_t *)d_prod, a->ptr);
    }
    int64_t total_ub = 0;
    {
        hipError_t e = hipMemcpyAsync(&total_ub, p->d_ub_off + m, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && binned)
            e = hipMemcpyAsync(sbin, d_scounts, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "row work: %s", hipGetErrorString(e)); }
    }
    pool_free(c, d_ub);             // (d_prod lives on: the numeric binning at the end applies the same tiny-row predicate)
    // Column slabs (round 3): B wider than one slab of the chunked stream (or than smm_ctx_tune_symbolic allows), sorted,
    // without repeated columns, and most rows beyond the hash-set classes -> every (slab, row) is walked on its own.
    {
        const int ws_cap = c->sym_max_ws > 0 ? c->sym_max_ws : CCS_MAX_WS;
        const bool wide = p->ncols > ws_cap && p->g.wc <= ws_cap && p->g.wc <= 32767;
        const bool dominant = c->sym_max_ws > 0 || hmax1 == 0 || 2 * (int64_t)sbin[SB_REST] >= m;
        if (c->sym_ccs && c->narrow_idx && wide && dominant && !safe && p->b_sorted && c->slab_mode != 2) {
            if (d_slists) pool_free(c, d_slists);
            pool_free(c, d_prod);
            // tiles per slab: as many as fit the slab limit (<= 8), but not so many that the marker bitmap leaves fewer
            // than 28 waves per CU (configs[4] share: 3 tiles = 60 000 columns = 21 waves 12.4 ms, 2 tiles = 31 waves 11.7 ms)
            int tps = std::max(1, std::min(8, ws_cap / p->g.wc));
            if (c->sym_max_ws == 0)
                while (tps > 1 && (160 * 1024) / ((((tps * p->g.wc + 31) / 32) + WAVE) * 4) < 28) --tps;
            p->tps = tps; p->ws = tps * p->g.wc; p->n_slabs = (p->g.nct + tps - 1) / tps;
            p->list16 = true;
            const int ns = p->n_slabs;
            smm_csr::CcsCache cc{};
            PCHK(ensure_ccs(c, b, p->ws, ns, &cc));
            Geom gs; gs.nw = 1; gs.nct = ns; gs.wc = p->ws; gs.wf = p->ws; gs.n_ft = ns;
            const int *sseg = nullptr;
            PCHK(ensure_seg(c, b, gs, &sseg));
            // capacities and offsets of the (slab, row) lists
            int64_t *d_ubs = nullptr;
            PCHK(pool_get(c, (size_t)ns * m, &d_ubs));
            pool_free(c, p->d_ub_off); p->d_ub_off = nullptr;
            PCHK(pool_get(c, (size_t)ns * m + 1, &p->d_ub_off));
            LAUNCH(c, "smm_row_work", smm_ccs_row_work, wgrid, 256, 0, (int)m, ns, p->ws, (int)p->ncols, (int)b->rows, p->row_offset,
                   sym ? 1 : 0, a->ptr, a->idx, sseg, d_ubs);
            PCHK(scan_launch<int64_t>(c, (int64_t)ns * m, d_ubs, p->d_ub_off));
            {
                hipError_t e = hipMemcpyAsync(&total_ub, p->d_ub_off + (int64_t)ns * m, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "slab row work: %s", hipGetErrorString(e)); }
            }
            pool_free(c, d_ubs);
            p->total_cap = total_ub;
            {
                // (+ slack: a list longer than its capacity -- a defect the plan checker reports -- and a sub-run table that
                // points past a list must still stay inside this allocation)
                char *tmp = nullptr;
                PCHK(pool_get(c, ((size_t)std::max<int64_t>(total_ub, 1) + LIST_SLACK) * 2, &tmp));
                p->d_tmp = tmp;
            }
            PCHK(pool_get(c, (size_t)a->nnz * ns, &p->d_P));
            PCHK(pool_get(c, (size_t)ns * m, &p->d_scnt));
            PCHK(pool_get(c, (size_t)m, &p->d_rowcnt));
            int *d_unitctr = (int *)((char *)c->d_flags + 208);
            {
                hipError_t e = hipMemsetAsync(d_unitctr, 0, sizeof(int), c->stream);
                if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
            }
            {
                const size_t wave_bytes = (size_t)(cc.bm_words + WAVE) * sizeof(unsigned);
                int cw = 4, best = 0;
                for (int cand : {4, 2, 1}) {
                    const int waves = (int)std::min<size_t>(32, ((size_t)160 * 1024 / (cand * wave_bytes)) * cand);
                    if (waves > best) { best = waves; cw = cand; }
                }
                const size_t lds = wave_bytes * cw;
                const int64_t units = (int64_t)ns * m;
                if (units >= INT32_MAX) { smm_plan_destroy(p); return fail(SMM_ERR_INVALID, "too many (slab, row) units"); }
                const int sgrid = (int)std::min<int64_t>((units + cw - 1) / cw, (int64_t)c->n_cu * 8 * (4 / cw));
                const bool dr = c->sym_dense == 2 || (c->sym_dense == 1 && cc.same_word >= 0.8);
                const bool batch = units >= (int64_t)64 * sgrid * cw && units < INT32_MAX - (1 << 24);          // many short units: 16 per counter round trip
                auto kern = ccs_kernel(sym, dr, batch);
                if (lds > 64 * 1024) {
                    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
                }
                LAUNCH(c, "smm_symbolic", kern, sgrid, cw * 64, lds, (int)m, ns, (const int *)nullptr, (const int *)nullptr, p->row_offset, cc.ws,
                       cc.bm_words, (int)b->rows, (int64_t)a->nnz, cc.guard_chunk, a->ptr, a->idx, (const int *)cc.cptr,
                       (const unsigned short *)cc.stream, (const int64_t *)p->d_ub_off, (unsigned short *)p->d_tmp, p->d_P, p->d_scnt, d_unitctr);
            }
            LAUNCH(c, "smm_slab_rowcnt", smm_slab_rowcnt, std::min<int64_t>((m + 255) / 256, 4096), 256, 0, (int)m, ns, (const int *)p->d_scnt,
                   p->d_rowcnt);
            PCHK(scan_launch<int>(c, m, p->d_rowcnt, p->d_cptr));
            {
                hipError_t e = hipMemcpyAsync(&p->nnz, p->d_cptr + m, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e == hipSuccess) e = hipGetLastError();
                if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "symbolic phase (slabs): %s", hipGetErrorString(e)); }
            }
            if (p->nnz > 0) {
                // every non-empty row goes to the tile kernel (the hash kernels read one list per row)
                PCHK(pool_get(c, (size_t)5 * m, &p->d_lists));
                int *d_counts = (int *)((char *)c->d_flags + 320);
                hipError_t e = hipMemsetAsync(d_counts, 0, 8 * sizeof(int), c->stream);
                BinSpec nspec{2, {0, 0, 0, 0, 0, 0}, 3, 4};
                LAUNCH(c, "smm_bin_rows", smm_bin_rows<int>, std::min<int64_t>((m + 1023) / 1024, 2048), 1024, 0, (int)m, nspec,
                       (const int *)p->d_rowcnt, p->d_lists, d_counts);
                if (e == hipSuccess) e = hipMemcpyAsync(p->n_bin, d_counts, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "row binning: %s", hipGetErrorString(e)); }
            }
            if (p->n_bin[2] > 0) {
                if (flags & SMM_EXACT) {
                    PCHK(ensure_seg(c, b, p->g, &p->seg));
                    PCHK(ensure_loc(c, b, p->g, &p->loc));
                } else {
                    PCHK(ensure_pack(c, b, p->g, &p->pack));
                }
                PCHK(pool_get(c, (size_t)a->nnz, &p->d_dst0));
                PCHK(pool_get(c, (size_t)a->nnz * p->g.nct, &p->d_runs2));
                PCHK(pool_get(c, (size_t)m * p->g.nct, &p->d_tflag));
                const int nd = p->n_bin[2];
                LAUNCH(c, "smm_runs", smm_runs_slab, std::min<int64_t>((nd + 3) / 4, 65536), 256, 0, nd, (int)m, ns, tps, p->g.nct, p->g.wc,
                       (int64_t)a->nnz, (const int *)(p->d_lists + 2 * m), a->ptr, (const int64_t *)p->d_ub_off, (const int *)p->d_scnt,
                       (const unsigned *)p->d_P, (const unsigned short *)p->d_tmp, p->d_dst0, p->d_runs2, p->d_tflag, c->d_err);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "smm_runs_slab: %s", hipGetErrorString(e)); }
            }
            if (c->check) PCHK(plan_check(c, p));
            if (nnz_out) *nnz_out = p->nnz;
            *plan = p;
            return SMM_OK;
        }
    }
    p->list16 = c->narrow_idx && p->ncols < 65535 && b->cols < 65535;
    if (p->list16) PCHK(ensure_idx16(c, b));
    p->total_cap = total_ub;
    {
        char *tmp = nullptr;
        PCHK(pool_get(c, ((size_t)std::max<int64_t>(total_ub, 1) + std::max<size_t>(LIST_SLACK, (size_t)p->ncols)) * (p->list16 ? 2 : 4), &tmp));
        p->d_tmp = tmp;
    }
    PCHK(pool_get(c, (size_t)a->nnz, &p->d_P));
    PCHK(pool_get(c, (size_t)m, &p->d_rowcnt));
    if (binned) {               // rows without products are in no bin: their count stays 0
        hipError_t e = hipMemsetAsync(p->d_rowcnt, 0, (size_t)m * sizeof(int), c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
    }

    // the kernels hand rows out through these counters (one per launch)
    int *d_rowctr = (int *)((char *)c->d_flags + 288);
    {
        hipError_t e = hipMemsetAsync(d_rowctr, 0, 8 * sizeof(int), c->stream);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "memset: %s", hipGetErrorString(e)); }
    }
    // tiny rows: four (<= 16 products) or two (<= 32) to a wave, no marker at all
    for (int tc = 0; tc < 2; ++tc) {
        const int bin = tc == 0 ? SB_TINY : SB_TINY2;
        const int nt = sbin[bin];
        if (nt == 0) continue;
        const int per_wg = tc == 0 ? 16 : 8;                         // rows per 256-thread workgroup
        const int tgrid = (int)std::min<int64_t>(((int64_t)nt + per_wg - 1) / per_wg, (int64_t)c->n_cu * 32);
        const int *rows3 = d_slists + (size_t)bin * m;
#define TINY_CASE(S, IT, G)                                                                                                      \
        LAUNCH(c, "smm_symbolic_tiny", (smm_symbolic_tiny<S, IT, G>), tgrid, 256, 0, nt, rows3, p->row_offset, a->ptr, a->idx,     \
               b->ptr, b->idx, (const int64_t *)p->d_ub_off, (IT *)p->d_tmp, p->d_P, p->d_rowcnt, c->d_err);
#define TINY_G_CASE(S, IT) if (tc == 0) { TINY_CASE(S, IT, TINY_G) } else { TINY_CASE(S, IT, TINY_G2) }
        if (p->list16) { if (sym) { TINY_G_CASE(true, unsigned short) } else { TINY_G_CASE(false, unsigned short) } }
        else           { if (sym) { TINY_G_CASE(true, int) } else { TINY_G_CASE(false, int) } }
#undef TINY_G_CASE
#undef TINY_CASE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "smm_symbolic_tiny: %s", hipGetErrorString(e)); }
    }
    // hash classes: one wave per row, four rows per workgroup
    for (int cls = 0; cls < NHC; ++cls) {
        if (hmax1 == 0 || sbin[cls] == 0) continue;
        const int hs = HS[cls];
        const int hgrid = (int)std::min<int64_t>((sbin[cls] + 3) / 4, (int64_t)c->n_cu * 16);
        if (sym) PCHK((launch_symbolic_t<true, false, MARK_LDS_HASH>(c, p, hs, nullptr, hgrid, 4, d_slists + (size_t)cls * m, d_scounts + cls, d_rowctr + cls)));
        else     PCHK((launch_symbolic_t<false, false, MARK_LDS_HASH>(c, p, hs, nullptr, hgrid, 4, d_slists + (size_t)cls * m, d_scounts + cls, d_rowctr + cls)));
    }
    // bitmap kernels for the rest: one wave per row; waves per workgroup are chosen so that as many
    // waves as possible fit a CU's 160 KB
    const int64_t nbm = binned ? sbin[SB_REST] : m;
    const int *bm_rows = binned ? d_slists + (size_t)SB_REST * m : nullptr;
    const int *bm_count = binned ? d_scounts + SB_REST : nullptr;
    int wpb = 4, waves_per_cu = 8;
    if (ldsbm) {
        int best = 0;
        for (int cand : {4, 2, 1}) {
            const int waves = (int)std::min<size_t>(32, ((size_t)160 * 1024 / (cand * bm_bytes)) * cand);
            if (waves > best) { best = waves; wpb = cand; }
        }
        waves_per_cu = best;
    }
    // few waves per CU (wide bitmaps): a round is one memory round trip whatever it carries -> 32 loads in flight
    const bool deep = !safe && waves_per_cu <= 8;
    unsigned *gbm = nullptr;
    // Round 3: sorted B without repeated columns and 16-bit lists -> the walk over the chunk-padded stream
    const bool use_ccs = c->sym_ccs && p->list16 && !safe && p->ncols <= CCS_MAX_WS && nbm > 0;
    if (use_ccs) {
        smm_csr::CcsCache cc{};
        PCHK(ensure_ccs(c, b, (int)p->ncols, 1, &cc));
        const size_t wave_bytes = (size_t)(cc.bm_words + WAVE) * sizeof(unsigned);
        int cw = 4, best = 0;
        for (int cand : {4, 2, 1}) {
            const int waves = (int)std::min<size_t>(32, ((size_t)160 * 1024 / (cand * wave_bytes)) * cand);
            if (waves > best) { best = waves; cw = cand; }
        }
        const size_t lds = wave_bytes * cw;
        const int sgrid = (int)std::min<int64_t>((nbm + cw - 1) / cw, (int64_t)c->n_cu * 8 * (4 / cw));
        const bool dr = c->sym_dense == 2 || (c->sym_dense == 1 && cc.same_word >= 0.8);
        const bool batch = nbm >= (int64_t)64 * sgrid * cw && nbm < INT32_MAX - (1 << 24);                // many short rows: 16 per counter round trip
        auto kern = ccs_kernel(sym, dr, batch);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
        }
        LAUNCH(c, "smm_symbolic", kern, sgrid, cw * 64, lds, (int)m, 1, bm_rows, bm_count, p->row_offset, cc.ws, cc.bm_words, (int)b->rows,
               (int64_t)a->nnz, cc.guard_chunk, a->ptr, a->idx, (const int *)cc.cptr, (const unsigned short *)cc.stream,
               (const int64_t *)p->d_ub_off, (unsigned short *)p->d_tmp, p->d_P, p->d_rowcnt, d_rowctr + SB_REST);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "smm_symbolic_ccs: %s", hipGetErrorString(e)); }
    } else if (nbm > 0) {
        int sgrid = (int)std::min<int64_t>((nbm + wpb - 1) / wpb, (int64_t)c->n_cu * 8 * (4 / wpb));
        if (!ldsbm) PCHK(pool_get(c, (size_t)sgrid * wpb * (bm_words + 1), &gbm));
        const int mark = ldsbm ? MARK_LDS_BITMAP : MARK_GLOBAL_BITMAP;
        // (round 4) many short rows: 16 per counter round trip (one atomic per row on one word is 11 ns of L2 time each)
        const int rbatch = nbm >= (int64_t)64 * sgrid * wpb && nbm < INT32_MAX - (1 << 24) ? 16 : 1;
#define SYM_CASE(S, F, L) if (sym == S && safe == F && mark == L && !deep) PCHK((launch_symbolic_t<S, F, L>(c, p, bm_words, gbm, sgrid, wpb, bm_rows, bm_count, d_rowctr + SB_REST, rbatch)));
        SYM_CASE(false, false, MARK_LDS_BITMAP) SYM_CASE(false, true, MARK_LDS_BITMAP) SYM_CASE(true, false, MARK_LDS_BITMAP)
        SYM_CASE(true, true, MARK_LDS_BITMAP) SYM_CASE(false, false, MARK_GLOBAL_BITMAP) SYM_CASE(false, true, MARK_GLOBAL_BITMAP)
        SYM_CASE(true, false, MARK_GLOBAL_BITMAP) SYM_CASE(true, true, MARK_GLOBAL_BITMAP)
#define SYM_DEEP(S, L) if (sym == S && mark == L && deep) PCHK((launch_symbolic_t<S, false, L, 32>(c, p, bm_words, gbm, sgrid, wpb, bm_rows, bm_count, d_rowctr + SB_REST, rbatch)));
        SYM_DEEP(false, MARK_LDS_BITMAP) SYM_DEEP(true, MARK_LDS_BITMAP) SYM_DEEP(false, MARK_GLOBAL_BITMAP) SYM_DEEP(true, MARK_GLOBAL_BITMAP)
#undef SYM_DEEP
#undef SYM_CASE
    }
    PCHK(scan_launch<int>(c, m, p->d_rowcnt, p->d_cptr));
    {
        hipError_t e = hipMemcpyAsync(&p->nnz, p->d_cptr + m, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "symbolic phase: %s", hipGetErrorString(e)); }
    }
    if (gbm) pool_free(c, gbm);
    if (d_slists) pool_free(c, d_slists);
    if (p->nnz <= 0) pool_free(c, d_prod);
    if (p->nnz > 0) {
        // bin the rows of C: few nonzeros -> LDS hash kernels, the rest -> dense LDS tiles
        PCHK(pool_get(c, (size_t)5 * m, &p->d_lists));
        int *d_counts = (int *)((char *)c->d_flags + 320);
        hipError_t e = hipMemsetAsync(d_counts, 0, 8 * sizeof(int), c->stream);
        BinSpec nspec{2, {c->hash_small, c->hash_medium, 0, 0, 0, 0}, 3, 4};
        LAUNCH(c, "smm_bin_rows", smm_bin_rows<int>, std::min<int64_t>((m + 1023) / 1024, 2048), 1024, 0, (int)m, nspec,
               (const int *)p->d_rowcnt, p->d_lists, d_counts, tiny_max, (const int64_t *)d_prod, a->ptr);
        if (e == hipSuccess) e = hipMemcpyAsync(p->n_bin, d_counts, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        pool_free(c, d_prod);
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "row binning: %s", hipGetErrorString(e)); }
    }
    if (p->b_sorted && p->n_bin[2] > 0) {
        const double est_products = (double)a->nnz * ((double)b->nnz / (double)std::max<int64_t>(b->rows, 1)) *
                                    ((double)p->n_bin[2] / (double)m);
        p->use_slab = slab_geometry(c, b, p->ncols, &p->sg) &&
                      slab_pays(c, a, p->sg, (double)p->n_bin[2] * (double)p->ncols, est_products, (double)p->nnz,
                                (double)b->nnz / (double)std::max<int64_t>(b->rows, 1), (double)p->ncols);
        if (p->use_slab) PCHK(ensure_slab(c, b, p->sg, &p->slab));
        // (chosen by the heuristic, not forced: the tile kernel's index is built as well, so that the numeric
        // phase can fall back to it when the slab path's dense scratch does not fit)
        if (p->use_slab && c->slab_mode == 2) { /* forced: slab only */ }
        else if (flags & SMM_EXACT) {
            PCHK(ensure_seg(c, b, p->g, &p->seg));
            PCHK(ensure_loc(c, b, p->g, &p->loc));
        } else {
            PCHK(ensure_pack(c, b, p->g, &p->pack));
        }
        PCHK(pool_get(c, (size_t)a->nnz * (p->g.nct + 1), &p->d_runs));
        PCHK(pool_get(c, (size_t)m, &p->d_tail));
        const int nd = p->n_bin[2];
        const int rgrid = (int)std::min<int64_t>((nd + 3) / 4, 65536);
        const int tail_min = (flags & SMM_EXACT) ? TAIL_MIN_EXACT : TAIL_MIN_DEFAULT;
        if (p->list16)
            LAUNCH(c, "smm_runs", smm_runs<unsigned short>, rgrid, 256, 0, nd, p->g.nct, p->g.wc, (const int *)(p->d_lists + 2 * m), a->ptr,
                   p->d_ub_off, p->d_rowcnt, p->d_P, (const unsigned short *)p->d_tmp, p->d_runs, p->d_tail, c->d_err, tail_min);
        else
            LAUNCH(c, "smm_runs", smm_runs<int>, rgrid, 256, 0, nd, p->g.nct, p->g.wc, (const int *)(p->d_lists + 2 * m), a->ptr,
                   p->d_ub_off, p->d_rowcnt, p->d_P, (const int *)p->d_tmp, p->d_runs, p->d_tail, c->d_err, tail_min);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { smm_plan_destroy(p); return fail(SMM_ERR_HIP, "smm_runs: %s", hipGetErrorString(e)); }
    }
    if (c->check) PCHK(plan_check(c, p));
#undef PCHK
    if (nnz_out) *nnz_out = p->nnz;
    *plan = p;
    return SMM_OK;
}

extern "C" int smm_spgemm_numeric(smm_ctx *c, smm_plan *p, int64_t *d_c_indptr, int32_t *d_c_indices, double *d_c_data)
{
    if (!c || !p || p->ctx != c) return fail(SMM_ERR_INVALID, "bad plan/context");
    if (!d_c_indptr) return fail(SMM_ERR_INVALID, "d_c_indptr is NULL");
    HIPCHK(hipSetDevice(c->device));
    CTX_LOCK(c);
    const int64_t m = p->m;
    HIPCHK(hipMemcpyAsync(d_c_indptr, p->d_cptr, (m + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
    if (p->nnz == 0) return SMM_OK;
    if (!d_c_indices || !d_c_data) return fail(SMM_ERR_INVALID, "output arrays are NULL but nnz > 0");
    const bool sym = p->flags & SMM_SYMMETRIC;
    const bool exact = (p->flags & SMM_EXACT) != 0;
    // rows with few nonzeros: LDS hash kernels (whole rows of B; B need not be sorted)
    if (p->n_bin[0] > 0 || p->n_bin[1] > 0) {
        HashArgs H{};
        H.row_offset = p->row_offset;
        H.a_ptr = p->a->ptr; H.a_idx = p->a->idx; H.a_val = p->a->val;
        H.b_ptr = p->b->ptr; H.b_idx = p->b->idx; H.b_val = p->b->val;
        H.c_ptr = p->d_cptr; H.c_idx = d_c_indices; H.c_val = d_c_data;
        H.ub_off = p->d_ub_off; H.tmp_idx = p->d_tmp; H.list16 = p->list16 ? 1 : 0;
        H.dummy_idx = (const int *)((const char *)c->d_flags + 64);
        H.dummy_val = (const double *)((const char *)c->d_flags + 128);
        H.err = c->d_err;
        if (p->n_bin[0] > 0) {      // one wave per row, four rows per workgroup (always reference order)
            H.nrows = p->n_bin[0]; H.rowlist = p->d_lists;
            const int grid = (int)std::min<int64_t>((H.nrows + 3) / 4, (int64_t)c->n_cu * 32);
            if (sym) LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<true, 512, 1, 4>), grid, 256, 0, H);
            else     LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<false, 512, 1, 4>), grid, 256, 0, H);
        }
        if (p->n_bin[1] > 0) {      // one workgroup per row; SMM_EXACT: a single wave keeps the order
            H.nrows = p->n_bin[1]; H.rowlist = p->d_lists + m;
            const int grid = (int)std::min<int64_t>(H.nrows, (int64_t)c->n_cu * 16);
            if (exact) {
                if (sym) LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<true, 4096, 1, 1>), grid, 64, 0, H);
                else     LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<false, 4096, 1, 1>), grid, 64, 0, H);
            } else {
                if (sym) LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<true, 4096, 4, 1>), grid, 256, 0, H);
                else     LAUNCH(c, "smm_numeric_hash", (smm_numeric_hash<false, 4096, 4, 1>), grid, 256, 0, H);
            }
        }
        LAUNCH_CHECK();
    }
    // tiny rows: four / two to a wave, first touch found among the lanes (always the reference's order of additions)
    for (int tc = 0; tc < 2; ++tc) {
        const int nt = p->n_bin[3 + tc];
        if (nt == 0) continue;
        const int per_wg = tc == 0 ? 16 : 8;
        const int tgrid = (int)std::min<int64_t>(((int64_t)nt + per_wg - 1) / per_wg, (int64_t)c->n_cu * 32);
        const int *rows3 = p->d_lists + (size_t)(3 + tc) * m;
#define TINY_NUM(S, G) LAUNCH(c, "smm_numeric_tiny", (smm_numeric_tiny<S, G>), tgrid, 256, 0, nt, rows3, p->row_offset, p->a->ptr, p->a->idx, \
                              p->a->val, p->b->ptr, p->b->idx, p->b->val, (const int64_t *)p->d_cptr, d_c_indices, d_c_data, c->d_err)
        if (tc == 0) { if (sym) TINY_NUM(true, TINY_G); else TINY_NUM(false, TINY_G); }
        else         { if (sym) TINY_NUM(true, TINY_G2); else TINY_NUM(false, TINY_G2); }
#undef TINY_NUM
        LAUNCH_CHECK();
    }
    if (p->n_bin[2] == 0) return SMM_OK;
    const int nd = p->n_bin[2];
    const int *dense_rows = p->d_lists + 2 * m;
    if (p->b_sorted) {
        NumericArgs A{};
        A.m = nd; A.ncols = (int)p->ncols; A.nct = p->g.nct; A.wc = p->g.wc; A.wf = p->g.wf; A.n_ft = p->g.n_ft;
        A.row_offset = p->row_offset;
        A.rowlist = dense_rows;
        A.a_ptr = p->a->ptr; A.a_idx = p->a->idx; A.a_val = p->a->val;
        A.b_idx = p->b->idx; A.b_val = p->b->val; A.seg = p->seg; A.b_loc = p->loc;
        A.kmax = (int)std::max<int64_t>(p->b->nnz - 1, 0);
        A.tdesc = p->pack.desc; A.tpay = p->pack.pay;
        A.piece_epl = !c->piece_walk || !p->pack.pay ? 0 : (p->pack.maxlen <= 128 ? 2 : (p->pack.maxlen <= 256 ? 4 : 0));
        A.rowsB = (int)p->b->rows;
        A.c_ptr = p->d_cptr; A.c_idx = d_c_indices; A.c_val = d_c_data;
        A.ub_off = p->d_ub_off; A.tmp_idx = p->d_tmp; A.list16 = p->list16 ? 1 : 0; A.runs = p->d_runs; A.tail = p->d_tail;
        A.runs2 = p->d_runs2; A.tflag = p->d_tflag; A.n_slabs = p->n_slabs; A.tps = p->tps; A.ws = p->ws; A.mtot = (int)m; A.nnzA = p->a->nnz;
        if (p->use_slab) {
            // values in column order into a dense scratch (one row per row of the bin), then the emission
            double *scratch = nullptr;
            int rc = pool_get(c, (size_t)nd * (size_t)p->ncols, &scratch);
            if (rc != SMM_OK) {
                if (!(p->pack.pay || p->seg)) return rc;         // slab path forced: no tile index to fall back to
                CHK(launch_numeric<OUT_SPARSE>(c, A, sym, p->g.nw, exact));
                return SMM_OK;
            }
            rc = launch_slab<true>(c, p->a, p->b, p->slab, p->sg.rw, nd, dense_rows, sym, p->row_offset, scratch, p->ncols);
            if (rc == SMM_OK) {
                A.c_dense = scratch; A.ldc = p->ncols;
                A.dummy_idx = (const int *)((const char *)c->d_flags + 64);
                A.dummy_val = (const double *)((const char *)c->d_flags + 128);
                A.err = c->d_err;
                rc = sym ? launch_numeric_t<OUT_SPARSE, true, 16, false, true>(c, A) : launch_numeric_t<OUT_SPARSE, false, 16, false, true>(c, A);
            }
            pool_free(c, scratch);          // stream-ordered: only work queued behind the emission can get it
            return rc;
        }
        CHK(launch_numeric<OUT_SPARSE>(c, A, sym, p->g.nw, exact));
    } else {
        const int cgrid = (int)std::min<int64_t>(nd, 65536);
        LAUNCH(c, "smm_copy_lists", smm_copy_lists, cgrid, 256, 0, nd, dense_rows, p->d_ub_off, p->d_cptr, (const void *)p->d_tmp,
               p->list16 ? 1 : 0, d_c_indices);
        LAUNCH_CHECK();
        const int grid = (int)std::min<int64_t>((nd + 3) / 4, (int64_t)c->n_cu * 2);
        int *slot = nullptr;
        // SMM_EXACT: the ordered variant (read-modify-write in the reference's order instead of atomics; a second map per wave)
        CHK(pool_get(c, (size_t)grid * 4 * (size_t)p->ncols * (exact ? 2 : 1), &slot));
#define GEN_CASE(S, O)                                                                                                        \
        if (sym == S && exact == O)                                                                                           \
            LAUNCH(c, "smm_numeric_general", (smm_numeric_general<S, O>), grid, 256, 0, nd, (int)p->ncols, p->row_offset,      \
                   dense_rows, p->a->ptr, p->a->idx, p->a->val, p->b->ptr, p->b->idx, p->b->val, p->d_cptr,                   \
                   (const int *)d_c_indices, d_c_data, slot);
        GEN_CASE(false, false) GEN_CASE(true, false) GEN_CASE(false, true) GEN_CASE(true, true)
#undef GEN_CASE
        LAUNCH_CHECK();
        HIPCHK(hipStreamSynchronize(c->stream));
        pool_free(c, slot);
    }
    return SMM_OK;
}

extern "C" int smm_plan_indptr_host(smm_ctx *c, smm_plan *p, int64_t *c_indptr)
{
    if (!c || !p || !c_indptr) return fail(SMM_ERR_INVALID, "NULL argument");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c_indptr, p->d_cptr, (p->m + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SMM_OK;
}

static int numeric_host(smm_ctx *c, smm_plan *p, int64_t *c_indptr, void *c_indices, double *c_data, bool wide);
extern "C" int smm_spgemm_numeric_host(smm_ctx *c, smm_plan *p, int64_t *c_indptr, int32_t *c_indices, double *c_data)
{
    return numeric_host(c, p, c_indptr, c_indices, c_data, false);
}
extern "C" int smm_spgemm_numeric_host_i64(smm_ctx *c, smm_plan *p, int64_t *c_indptr, int64_t *c_indices, double *c_data)
{
    return numeric_host(c, p, c_indptr, c_indices, c_data, true);
}
static int numeric_host(smm_ctx *c, smm_plan *p, int64_t *c_indptr, void *c_indices, double *c_data, bool wide)
{
    if (!c || !p || !c_indptr) return fail(SMM_ERR_INVALID, "NULL argument");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    int64_t *dp = nullptr; int *di = nullptr; double *dv = nullptr;
    const int64_t nnz = p->nnz;
    CHK(pool_get(c, (size_t)p->m + 1, &dp));
    int rc = pool_get(c, (size_t)std::max<int64_t>(nnz, 1), &di);
    if (rc == SMM_OK) rc = pool_get(c, (size_t)std::max<int64_t>(nnz, 1), &dv);
    if (rc == SMM_OK) rc = smm_spgemm_numeric(c, p, dp, di, dv);
    if (rc == SMM_OK && nnz > 0 && (!c_indices || !c_data)) rc = fail(SMM_ERR_INVALID, "output arrays are NULL but nnz > 0");
    if (rc == SMM_OK) rc = download(c, c_indptr, dp, (size_t)(p->m + 1) * sizeof(int64_t));
    if (rc == SMM_OK && nnz > 0) rc = download(c, c_indices, di, (size_t)nnz * sizeof(int), wide);
    if (rc == SMM_OK && nnz > 0) rc = download(c, c_data, dv, (size_t)nnz * sizeof(double));
    (void)hipStreamSynchronize(c->stream);
    pool_free(c, dp); pool_free(c, di); pool_free(c, dv);
    if (rc == SMM_OK) rc = take_plan_error(c, "smm_spgemm_numeric");       // what the kernels' clamps recorded, if anything
    return rc;
}

// ------------------------------------------------------------------------------ CSR x CSR -> dense
static int dense_into(smm_ctx *c, smm_csr *a, smm_csr *b, int flags, int64_t row_offset, double *d_c, int64_t ldc)
{
    const int64_t m = a->rows, n = b->cols;
    if (m == 0 || n == 0) return SMM_OK;
    const bool sym = flags & SMM_SYMMETRIC;
    if (a->nnz == 0 || b->nnz == 0) {
        HIPCHK(hipMemset2DAsync(d_c, ldc * sizeof(double), 0, n * sizeof(double), m, c->stream));
        return SMM_OK;
    }
    SlabGeom sg;
    if (slab_geometry(c, b, n, &sg) &&
        slab_pays(c, a, sg, (double)m * (double)n, (double)a->nnz * ((double)b->nnz / (double)std::max<int64_t>(b->rows, 1)), -1.0,
                  (double)b->nnz / (double)std::max<int64_t>(b->rows, 1), (double)n)) {
        smm_csr::SlabCache sl{0, 0, nullptr, nullptr, nullptr};
        CHK(ensure_slab(c, b, sg, &sl));
        return launch_slab<false>(c, a, b, sl, sg.rw, (int)m, nullptr, sym, row_offset, d_c, ldc);
    }
    if (!(b->vflags & CSR_UNSORTED)) {
        Geom g = make_geom(c, n, b, (flags & SMM_EXACT) != 0);
        const int *seg = nullptr; const short *loc = nullptr;
        smm_csr::PackCache pack{0, 0, nullptr, nullptr, 0};
        if (flags & SMM_EXACT) {
            CHK(ensure_seg(c, b, g, &seg));
            CHK(ensure_loc(c, b, g, &loc));
        } else {
            CHK(ensure_pack(c, b, g, &pack));
        }
        NumericArgs A{};
        A.m = (int)m; A.ncols = (int)n; A.nct = g.nct; A.wc = g.wc; A.wf = g.wf; A.n_ft = g.n_ft;
        A.row_offset = row_offset;
        A.a_ptr = a->ptr; A.a_idx = a->idx; A.a_val = a->val;
        A.b_idx = b->idx; A.b_val = b->val; A.seg = seg; A.b_loc = loc;
        A.kmax = (int)std::max<int64_t>(b->nnz - 1, 0);
        A.tdesc = pack.desc; A.tpay = pack.pay;
        // (dense output keeps the chunk walk: the piece walk measured 0.9 ms slower at configs[2] and 1.2-2 ms at stage 1 of
        // configs[3] with 1-3 pieces in flight -- profiles/r3_c_piece_walk.txt; env SMM_PIECE_WALK=2 forces it here too)
        A.piece_epl = c->piece_walk < 2 || !pack.pay ? 0 : (pack.maxlen <= 128 ? 2 : (pack.maxlen <= 256 ? 4 : 0));
        A.rowsB = (int)b->rows;
        A.c_dense = d_c; A.ldc = ldc;
        CHK(launch_numeric<OUT_DENSE>(c, A, sym, g.nw, (flags & SMM_EXACT) != 0));
    } else {
        const bool ordered = (flags & SMM_EXACT) != 0;
        const int grid = (int)std::min<int64_t>((m + 3) / 4, ordered ? (int64_t)c->n_cu * 2 : 65536);
        int *owner = nullptr;
        if (ordered) CHK(pool_get(c, (size_t)grid * 4 * (size_t)n, &owner));
#define GEN_CASE(S, O)                                                                                                        \
        if (sym == S && ordered == O)                                                                                         \
            LAUNCH(c, "smm_dense_general", (smm_dense_general<S, O>), grid, 256, 0, (int)m, (int)n, row_offset, a->ptr, a->idx, \
                   a->val, b->ptr, b->idx, b->val, d_c, ldc, owner);
        GEN_CASE(false, false) GEN_CASE(true, false) GEN_CASE(false, true) GEN_CASE(true, true)
#undef GEN_CASE
        hipError_t e = hipGetLastError();
        if (owner) { if (e == hipSuccess) e = hipStreamSynchronize(c->stream); pool_free(c, owner); }
        if (e != hipSuccess) return fail(SMM_ERR_HIP, "smm_dense_general: %s", hipGetErrorString(e));
    }
    return SMM_OK;
}

static int mirror_upper(smm_ctx *c, int64_t n, double *d_c, int64_t ldc)
{
    if (n <= 1) return SMM_OK;
    const int64_t tiles = (n + 63) / 64;
    const int64_t pairs = tiles * (tiles + 1) / 2;
    if (pairs > 0x7fffffff) return fail(SMM_ERR_INVALID, "matrix too large for the mirror epilogue");
    LAUNCH(c, "smm_mirror_upper", smm_mirror_upper, pairs, 256, 0, (int)n, d_c, ldc);
    LAUNCH_CHECK();
    return SMM_OK;
}

extern "C" int smm_spgemm_dense(smm_ctx *c, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset, double *d_c)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(check_pair(c, a, b));
    CHK(exact_guard(c, flags));
    if (!d_c && a->rows * b->cols > 0) return fail(SMM_ERR_INVALID, "d_c is NULL");
    if (flags & SMM_MIRROR) {
        if (!(flags & SMM_SYMMETRIC) || a->rows != b->cols || a_row_offset != 0)
            return fail(SMM_ERR_INVALID, "SMM_MIRROR needs SMM_SYMMETRIC and the whole square result (no row shard)");
    }
    CHK(dense_into(c, a, b, flags, a_row_offset, d_c, b->cols));
    if (flags & SMM_MIRROR) CHK(mirror_upper(c, b->cols, d_c, b->cols));
    return SMM_OK;
}

extern "C" int smm_spgemm_dense_host(smm_ctx *c, smm_csr *a, smm_csr *b, int flags, int64_t a_row_offset, double *out)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(check_pair(c, a, b));
    const int64_t total = a->rows * b->cols;
    if (total == 0) return SMM_OK;
    if (!out) return fail(SMM_ERR_INVALID, "c is NULL");
    double *d = nullptr;
    CHK(pool_get(c, (size_t)total, &d));
    int rc = smm_spgemm_dense(c, a, b, flags, a_row_offset, d);
    if (rc == SMM_OK) rc = download(c, out, d, (size_t)total * sizeof(double));
    (void)hipStreamSynchronize(c->stream);
    pool_free(c, d);
    return rc;
}

// ------------------------------------------------------------------------------ CSR mirror epilogue
//  upper-triangle CSR in HBM -> full symmetric CSR in HBM, two calls: row pointer + nnz, then fill.
extern "C" int smm_csr_mirror_symbolic(smm_ctx *c, int64_t n, const int64_t *d_indptr, const int32_t *d_indices,
                                       int64_t *d_full_indptr, int64_t *nnz_full)
{
    if (!c || !d_indptr || !d_full_indptr || !nnz_full) return fail(SMM_ERR_INVALID, "NULL argument");
    if (n < 0 || n >= INT32_MAX) return fail(SMM_ERR_INVALID, "bad dimension");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    *nnz_full = 0;
    if (n == 0) { HIPCHK(hipMemsetAsync(d_full_indptr, 0, sizeof(int64_t), c->stream)); return SMM_OK; }
    int *mcnt = nullptr; int64_t *flen = nullptr;
    CHK(pool_get(c, (size_t)n + 1, &mcnt));                       // [n] = longest mirrored segment
    int rc = pool_get(c, (size_t)n, &flen);
    if (rc != SMM_OK) { pool_free(c, mcnt); return rc; }
    hipError_t e = hipMemsetAsync(mcnt, 0, ((size_t)n + 1) * sizeof(int), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_flags, 0, sizeof(unsigned), c->stream);
    const int grid = (int)std::min<int64_t>((n + 3) / 4, 16384);
    LAUNCH(c, "smm_mirror_count", smm_mirror_count, grid, 256, 0, (int)n, d_indptr, d_indices, mcnt, c->d_flags);
    LAUNCH(c, "smm_mirror_rowlen", smm_mirror_rowlen, std::min<int64_t>((n + 255) / 256, 4096), 256, 0, (int)n, d_indptr, (const int *)mcnt,
           flen, mcnt + n);
    rc = scan_launch<int64_t>(c, n, flen, d_full_indptr);
    unsigned bad = 0; int maxseg = 0;
    if (rc == SMM_OK) {
        if (e == hipSuccess) e = hipMemcpyAsync(nnz_full, d_full_indptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, c->d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&maxseg, mcnt + n, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(SMM_ERR_HIP, "CSR mirror: %s", hipGetErrorString(e));
    }
    pool_free(c, mcnt); pool_free(c, flen);
    if (rc != SMM_OK) return rc;
    if (bad) return fail(SMM_ERR_INVALID, "CSR mirror: the input holds entries left of the diagonal or outside the n x n square");
    (void)maxseg;                 // (any length: segments beyond the LDS sort are placed by rank, smm_mirror_rank)
    return SMM_OK;
}

extern "C" int smm_csr_mirror_fill(smm_ctx *c, int64_t n, const int64_t *d_indptr, const int32_t *d_indices, const double *d_data,
                                   const int64_t *d_full_indptr, int32_t *d_full_indices, double *d_full_data)
{
    if (!c || !d_indptr || !d_full_indptr) return fail(SMM_ERR_INVALID, "NULL argument");
    if (n <= 0) return SMM_OK;
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    if (!d_indices || !d_data || !d_full_indices || !d_full_data) return fail(SMM_ERR_INVALID, "NULL CSR array");
    int *mcnt = nullptr, *cursor = nullptr;
    CHK(pool_get(c, (size_t)n, &mcnt));
    int rc = pool_get(c, (size_t)n, &cursor);
    if (rc != SMM_OK) { pool_free(c, mcnt); return rc; }
    hipError_t e = hipMemsetAsync(mcnt, 0, (size_t)n * sizeof(int), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(cursor, 0, (size_t)n * sizeof(int), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_flags, 0, sizeof(unsigned), c->stream);
    const int grid = (int)std::min<int64_t>((n + 3) / 4, 16384);
    LAUNCH(c, "smm_mirror_count", smm_mirror_count, grid, 256, 0, (int)n, d_indptr, d_indices, mcnt, c->d_flags);
    // rows that receive more mirrored entries than the LDS sort holds: their entries are staged and placed by rank
    int64_t *big = nullptr, *toff = nullptr; int *tidx = nullptr; double *tval = nullptr;
    int64_t total_big = 0;
    auto drop = [&]() { pool_free(c, mcnt); pool_free(c, cursor); pool_free(c, big); pool_free(c, toff); pool_free(c, tidx); pool_free(c, tval); };
    rc = pool_get(c, (size_t)n, &big);
    if (rc == SMM_OK) rc = pool_get(c, (size_t)n + 1, &toff);
    if (rc == SMM_OK) {
        LAUNCH(c, "smm_mirror_big", smm_mirror_big, std::min<int64_t>((n + 255) / 256, 4096), 256, 0, (int)n, (const int *)mcnt, big);
        rc = scan_launch<int64_t>(c, n, big, toff);
    }
    if (rc == SMM_OK) {
        if (e == hipSuccess) e = hipMemcpyAsync(&total_big, toff + n, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(SMM_ERR_HIP, "CSR mirror: %s", hipGetErrorString(e));
    }
    if (rc == SMM_OK && total_big > 0) {
        rc = pool_get(c, (size_t)total_big, &tidx);
        if (rc == SMM_OK) rc = pool_get(c, (size_t)total_big, &tval);
    }
    if (rc != SMM_OK) { drop(); return rc; }
    LAUNCH(c, "smm_mirror_fill", smm_mirror_fill, grid, 256, 0, (int)n, d_indptr, d_indices, d_data, d_full_indptr, (const int *)mcnt, cursor,
           d_full_indices, d_full_data, total_big > 0 ? (const int64_t *)toff : (const int64_t *)nullptr, tidx, tval);
    if (total_big > 0) {
        auto rk = smm_mirror_rank;
        const size_t rlds = (size_t)2 * RANK_WORDS * sizeof(unsigned);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
        LAUNCH(c, "smm_mirror_rank", rk, std::min<int64_t>(n, (int64_t)c->n_cu * 4), 1024, rlds, (int)n, d_full_indptr, (const int *)mcnt,
               (const int64_t *)toff, (const int *)tidx, (const double *)tval, d_full_indices, d_full_data);
    }
    LAUNCH(c, "smm_mirror_sort", smm_mirror_sort<false>, grid, 256, 0, (int)n, d_full_indptr, (const int *)mcnt, d_full_indices, d_full_data);
    {
        auto kern = smm_mirror_sort<true>;
        const size_t lds = (size_t)MIRROR_MAX_SEG * (sizeof(double) + sizeof(int));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        LAUNCH(c, "smm_mirror_sort", kern, std::min<int64_t>(n, (int64_t)c->n_cu * 4), 256, lds, (int)n, d_full_indptr, (const int *)mcnt,
               d_full_indices, d_full_data);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);     // mcnt / cursor / the staging array return to the pool
    drop();
    if (e != hipSuccess) return fail(SMM_ERR_HIP, "CSR mirror: %s", hipGetErrorString(e));
    return SMM_OK;
}

// ------------------------------------------------------------------------------ triple product
extern "C" int smm_triple_product(smm_ctx *c, smm_csr *h, smm_csr *q, int flags, int64_t row_begin, int64_t row_end,
                                  double *d_c)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    CHK(check_pair(c, h, q));
    CHK(exact_guard(c, flags));
    const int64_t n = h->rows, K = h->cols;
    // the reference indexes temp_values[K] with Q's columns (sparse_sparse_dense.cpp:178,196): Q may
    // be K x c with c <= K (columns c..K-1 of T stay 0); c > K would write out of bounds there
    if (q->cols > K) return fail(SMM_ERR_INVALID, "Q has more columns (%lld) than H (%lld)", (long long)q->cols, (long long)K);
    if (row_begin < 0 || row_end > n || row_begin > row_end) return fail(SMM_ERR_INVALID, "bad row range");
    const bool full = flags & SMM_FULL_MATRIX;
    const bool mirror = (flags & SMM_MIRROR) != 0;
    if (full && mirror) return fail(SMM_ERR_INVALID, "SMM_FULL_MATRIX and SMM_MIRROR exclude each other");
    if ((full || mirror) && (row_begin != 0 || row_end != n))
        return fail(SMM_ERR_INVALID, "SMM_FULL_MATRIX / SMM_MIRROR need the whole row range [0,n)");
    const int64_t nr = row_end - row_begin;
    if (nr == 0 || n == 0) return SMM_OK;
    if (!d_c) return fail(SMM_ERR_INVALID, "d_c is NULL");
    if (h->nnz == 0 || q->nnz == 0 || K == 0) {
        HIPCHK(hipMemset2DAsync(d_c, n * sizeof(double), 0, n * sizeof(double), nr, c->stream));
        return SMM_OK;
    }
    // stage 1: T = H[row_begin:row_end, :] * Q, dense nr x K (sparse_sparse_dense.cpp:187-198)
    double *T = nullptr;
    CHK(pool_get(c, (size_t)nr * K, &T));
    if (q->cols < K) HIPCHK(hipMemsetAsync(T, 0, (size_t)nr * K * sizeof(double), c->stream));
    smm_csr hv = *h;                       // row-range view of H (borrowed arrays)
    hv.ptr = h->ptr + row_begin; hv.rows = nr; hv.owned = false; hv.segs.clear(); hv.locs.clear(); hv.slabs.clear(); hv.packs.clear(); hv.ccs.clear(); hv.idx16 = nullptr; hv.idx_pad = nullptr;
    // indptr of the view is not rebased: kernels only use ptr[row], ptr[row+1] as absolute positions.
    int rc = dense_into(c, &hv, q, flags & SMM_EXACT, 0, T, K);
    if (rc != SMM_OK) { pool_free(c, T); return rc; }
    // stage 2
    if (h->vflags & CSR_UNSORTED) {
        // H with unsorted rows (legal CSR, e.g. an unsorted scipy product): the chunked ELL walk needs
        // sorted rows, so every (i,k) sums row k of H in its stored order, as the reference does
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)std::min<int64_t>(nr, 65535));
        {
            LaunchTimer lt_(c, "smm_triple_stage2_general");
            hipLaunchKernelGGL(smm_triple_stage2_general, grid, dim3(256), 0, c->stream, (int)n, (int)K, row_begin, row_end,
                               full ? 1 : 0, h->ptr, h->idx, h->val, (const double *)T, d_c, n);
        }
        if (full) LAUNCH(c, "smm_triple_mirror", smm_triple_mirror, (n * n + 255) / 256, 256, 0, (int)n, d_c, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        pool_free(c, T);
        if (e != hipSuccess) return fail(SMM_ERR_HIP, "triple product: %s", hipGetErrorString(e));
        if (mirror) CHK(mirror_upper(c, n, d_c, n));
        return SMM_OK;
    }
#ifndef SMM_S2_NW
#define SMM_S2_NW 16
#endif
    constexpr int NW = SMM_S2_NW;          // waves per workgroup = 64-row slices of H per k-group
    constexpr int R = 16;
    if (c->s2_ring && NW == 16 && (K + RING_PW - 1) / RING_PW < 65535) {
        // round 4: ring of column pieces, progress words instead of barriers (smm_ring.hpp)
        const bool exact_r = (flags & SMM_EXACT) != 0;
        rc = ensure_ring(c, h, !exact_r);
        if (rc != SMM_OK) { pool_free(c, T); return rc; }
        RingArgs A{};
        A.n = (int)n; A.K = (int)K; A.npieces = h->ring_npieces; A.nslices = (int)((n + WAVE - 1) / WAVE);
        A.nib = (int)((nr + R - 1) / R);
        A.row_begin = row_begin; A.row_end = row_end; A.full = full ? 1 : 0;
        A.off = h->ring_off; A.col = h->ring_col; A.val = h->ring_val; A.hdr = h->ring_hdr;
        A.T = T; A.C = d_c; A.ldc = n; A.err = c->d_err;
        const int64_t nkg = (n + 16 * WAVE - 1) / (16 * WAVE);
        A.nkg = (int)nkg;
        A.gk = (int)std::min<int64_t>(std::max(c->s2_group, 1), nkg);
        const int64_t grid2 = ((nkg + A.gk - 1) / A.gk) * A.gk * (((int64_t)A.nib + 7) / 8) * 8;
        if (grid2 > 0x7fffffff) { pool_free(c, T); return fail(SMM_ERR_INVALID, "triple product too large for one launch"); }
        const size_t lds = (size_t)RING_NB * RING_PW * (R + 2) * sizeof(double);
        auto kern = exact_r ? smm_triple_stage2_ring<R, 16, false> : smm_triple_stage2_ring<R, 16, true>;
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pool_free(c, T); return fail(SMM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
        LAUNCH(c, "smm_triple_stage2", kern, grid2, 16 * 64, lds, A);
        if (full) LAUNCH(c, "smm_triple_mirror", smm_triple_mirror, (n * n + 255) / 256, 256, 0, (int)n, d_c, n);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // T returns to the pool below
        pool_free(c, T);
        if (e != hipSuccess) return fail(SMM_ERR_HIP, "triple product: %s", hipGetErrorString(e));
#ifdef SMM_RING_STAMPS
        {
            unsigned long long st[6];
            (void)hipMemcpy(st, c->d_err + 4, sizeof(st), hipMemcpyDeviceToHost);
            (void)hipMemset(c->d_err + 4, 0, sizeof(st));
            fprintf(stderr, "[SMM_RING_STAMPS] wave-cycles %.4g: waiting for pieces %.1f%%  tail (owed shares) %.1f%%;  slow-path entries %.3g, spins %.3g, steps %.3g\n",
                    (double)st[0], 100.0 * st[1] / st[0], 100.0 * st[2] / st[0], (double)st[3], (double)st[4], (double)st[5]);
        }
#endif
        CHK(take_plan_error(c, "smm_triple_product"));              // (a bounded wait of the ring protocol ran out: never expected)
        if (mirror) CHK(mirror_upper(c, n, d_c, n));
        return SMM_OK;
    }
    constexpr int chunk_cap = NW * 64;     // one tile column per thread; [chunk][R+2] f64 = 144 KB of LDS at 16 waves
    const int nchunks = (int)((K + chunk_cap - 1) / chunk_cap);
    const int chunk = (int)((K + nchunks - 1) / nchunks);
    const bool exact = (flags & SMM_EXACT) != 0;
    rc = ensure_ell(c, h, nchunks, chunk, !exact);
    if (rc != SMM_OK) { pool_free(c, T); return rc; }
    TripleArgs A{};
    A.n = (int)n; A.K = (int)K; A.nchunks = nchunks; A.chunk = chunk; A.nslices = (int)((n + WAVE - 1) / WAVE);
    A.nib = (int)((nr + R - 1) / R);
    A.row_begin = row_begin; A.row_end = row_end; A.full = full ? 1 : 0;
    A.off = h->ell_off; A.col = h->ell_col; A.val = h->ell_val;
    A.T = T; A.C = d_c; A.ldc = n;
    const size_t lds = (size_t)(R + 2) * chunk * sizeof(double);
    const int64_t nkg = (n + NW * WAVE - 1) / (NW * WAVE);
    auto kern = exact ? smm_triple_stage2<R, NW, chunk_cap, false> : smm_triple_stage2<R, NW, chunk_cap, true>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pool_free(c, T); return fail(SMM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
    }
    A.nkg = (int)nkg;
    A.gk = (int)std::min<int64_t>(std::max(c->s2_group, 1), nkg);
    const int64_t grid2 = ((nkg + A.gk - 1) / A.gk) * A.gk * (((int64_t)A.nib + 7) / 8) * 8;
    if (grid2 > 0x7fffffff) { pool_free(c, T); return fail(SMM_ERR_INVALID, "triple product too large for one launch"); }
#ifdef SMM_S2_STAMPS
    unsigned long long *d_st = nullptr;
    if (pool_get(c, 8, &d_st) == SMM_OK) { (void)hipMemsetAsync(d_st, 0, 64, c->stream); A.stamps = d_st; }
#endif
    LAUNCH(c, "smm_triple_stage2", kern, grid2, NW * 64, lds, A);
#ifdef SMM_S2_STAMPS
    if (d_st) {
        unsigned long long hst[8];
        (void)hipMemcpyAsync(hst, d_st, 64, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)hst[i];
        fprintf(stderr, "[SMM_S2_STAMPS] wave-cycles %.4g: preload issue %.1f%%  barrier 1 %.1f%%  tile write %.1f%%  barrier 2 %.1f%%  tile load issue %.1f%%  steps %.1f%%\n",
                tot, 100 * hst[0] / tot, 100 * hst[1] / tot, 100 * hst[2] / tot, 100 * hst[3] / tot, 100 * hst[4] / tot, 100 * hst[5] / tot);
        pool_free(c, d_st);
    }
#endif
    if (full) LAUNCH(c, "smm_triple_mirror", smm_triple_mirror, (n * n + 255) / 256, 256, 0, (int)n, d_c, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // T returns to the pool below
    pool_free(c, T);
    if (e != hipSuccess) return fail(SMM_ERR_HIP, "triple product: %s", hipGetErrorString(e));
    if (mirror) CHK(mirror_upper(c, n, d_c, n));
    return SMM_OK;
}

extern "C" int smm_triple_product_host(smm_ctx *c, smm_csr *h, smm_csr *q, int flags, int64_t row_begin,
                                       int64_t row_end, double *out)
{
    if (!c || !h || !q) return fail(SMM_ERR_INVALID, "NULL argument");
    CTX_LOCK(c);
    const int64_t n = h->rows, nr = row_end - row_begin;
    if (nr <= 0 || n == 0) return smm_triple_product(c, h, q, flags, row_begin, row_end, nullptr);
    if (!out) return fail(SMM_ERR_INVALID, "c is NULL");
    double *d = nullptr;
    CHK(pool_get(c, (size_t)nr * n, &d));
    int rc = smm_triple_product(c, h, q, flags, row_begin, row_end, d);
    if (rc == SMM_OK) rc = download(c, out, d, (size_t)nr * (size_t)n * sizeof(double));
    (void)hipStreamSynchronize(c->stream);
    pool_free(c, d);
    return rc;
}

// ------------------------------------------------------------------------------ memory helpers
extern "C" int smm_device_malloc(smm_ctx *c, int64_t bytes, void **d_ptr)
{
    if (!c || !d_ptr || bytes < 0) return fail(SMM_ERR_INVALID, "bad argument");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    hipError_t e = dev_malloc(c, d_ptr, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) { return fail(SMM_ERR_ALLOC, "hipMalloc(%lld): %s", (long long)bytes, hipGetErrorString(e)); }
    return SMM_OK;
}
extern "C" int smm_device_free(smm_ctx *c, void *d_ptr)
{
    if (!c) return fail(SMM_ERR_INVALID, "ctx is NULL");
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(d_ptr));
    return SMM_OK;
}
extern "C" int smm_memcpy_d2h(smm_ctx *c, void *dst, const void *src, int64_t bytes)
{
    if (!c || bytes < 0) return fail(SMM_ERR_INVALID, "bad argument");
    if (bytes == 0) return SMM_OK;
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SMM_OK;
}
extern "C" int smm_memcpy_h2d(smm_ctx *c, void *dst, const void *src, int64_t bytes)
{
    if (!c || bytes < 0) return fail(SMM_ERR_INVALID, "bad argument");
    if (bytes == 0) return SMM_OK;
    CTX_LOCK(c);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SMM_OK;
}
