// praline_dp.hip -- host side + C ABI of libpraline_dp.so (see include/praline_dp.h).
//
// Host responsibilities (all plain C++, no Python/torch types):
//   * runtime: device binding, one launch stream, error strings
//   * arena:   packs profiles into the kernels' parity-split layout, finds the active symbols,
//              runs the profile x matrix pre-multiply on the device
//   * plan:    groups a pair list by its sequence TWO into 32-lane half tasks (length-sorted),
//              builds wave tasks, sizes the strip-boundary / traceback scratch
//   * parity entry points mirroring praline/util/cext.c:506-520 on raw (strided) host buffers
//
// There is deliberately NO CPU fallback here: every compute entry point needs a HIP device.
#include "praline_dp.h"
#define PRALINE_AUX_KERNELS 1
#include "dp_kernels.hip.h"
#include "dp_launch.hip.h"
#include "dp_arena16.h"
#include "sched.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <numeric>
#include <string>
#include <vector>
#include <unordered_map>

// --------------------------------------------------------------------------------------------
// errors + runtime
// --------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(e_ == hipErrorOutOfMemory ? PRALINE_ERR_NOMEM : PRALINE_ERR_DEVICE,     \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Runtime {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    // path plans that run in several launch chunks alternate between the main stream and this one (two scratch sets):
    // the traceback and the tail of chunk k overlap the fill of chunk k + 1
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int *h_flags = nullptr;   // page-locked: small read-backs that must not block the host when they are enqueued (256 ints)
    // device twins used by arena creation (praline_arena_finish), allocated once: a per-call buffer of this size would be
    // hipFree'd at return - a device-wide wait - and the call could never return ahead of its packing launch.  Every use
    // is ordered on `stream` (the next arena's memset / upload follows the previous arena's kernels).
    int *d_scan_flags = nullptr;          // 256 ints
    unsigned char *d_slot_of = nullptr;   // 256 bytes
};
static Runtime g_rt;

static int ensure_runtime(int device)
{
    if (g_rt.ready) {
        if (device >= 0 && device != g_rt.device)
            return fail(PRALINE_ERR_ARG, "already bound to device %d (requested %d)", g_rt.device, device);
        return PRALINE_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PRALINE_ERR_DEVICE, "no HIP device available (%s): libpraline_dp has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0) device = 0;
    if (device >= n) return fail(PRALINE_ERR_ARG, "device %d out of range (%d visible)", device, n);
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&g_rt.stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&g_rt.stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&g_rt.ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&g_rt.ev_join, hipEventDisableTiming));
    HIPCHK(hipHostMalloc((void **)&g_rt.h_flags, 256 * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipMalloc((void **)&g_rt.d_scan_flags, 256 * sizeof(int)));
    HIPCHK(hipMalloc((void **)&g_rt.d_slot_of, 256));
    g_rt.device = device;
    g_rt.ready = true;
    return PRALINE_OK;
}

static void pool_clear();

extern "C" int praline_abi_version(void) { return PRALINE_DP_ABI_VERSION; }

extern "C" int praline_device_count(int *count)
{
    if (!count) return fail(PRALINE_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(PRALINE_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return PRALINE_OK;
}

extern "C" int praline_init(int device)
{
    const int rc = ensure_runtime(device);
    if (rc == PRALINE_OK) sched_warm_threads();
    return rc;
}

extern "C" int praline_shutdown(void)
{
    if (!g_rt.ready) return PRALINE_OK;
    (void)hipStreamSynchronize(g_rt.stream);
    (void)hipStreamSynchronize(g_rt.stream2);
    pool_clear();
    (void)hipEventDestroy(g_rt.ev_fork);
    (void)hipEventDestroy(g_rt.ev_join);
    (void)hipHostFree(g_rt.h_flags);
    (void)hipFree(g_rt.d_scan_flags);
    (void)hipFree(g_rt.d_slot_of);
    (void)hipStreamDestroy(g_rt.stream2);
    (void)hipStreamDestroy(g_rt.stream);
    g_rt = Runtime();
    return PRALINE_OK;
}

extern "C" int praline_synchronize(void)
{
    if (!g_rt.ready) return PRALINE_OK;
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream2));
    return PRALINE_OK;
}

// Which arithmetic evaluates the match scores m = sum_sets P1 . S . P2^T of the batched plans (include/praline_dp.h).
static int g_match_mode = -1;   // -1: not set through the API, follow PRALINE_MM
static int match_mode()
{
    if (g_match_mode >= 0) return g_match_mode;
    if (const char *mm = getenv("PRALINE_MM")) {
        if (!strcmp(mm, "f32")) return PRALINE_MATCH_F32;
        if (!strcmp(mm, "ref")) return PRALINE_MATCH_REFERENCE;
    }
    return PRALINE_MATCH_FAST;
}

extern "C" int praline_set_match_mode(int kind)
{
    if (kind < -1 || kind > PRALINE_MATCH_REFERENCE) return fail(PRALINE_ERR_ARG, "unknown match-score mode %d", kind);
    g_match_mode = kind;
    return PRALINE_OK;
}

extern "C" int praline_get_match_mode(void) { return match_mode(); }

extern "C" int praline_pool_trim(void)
{
    if (!g_rt.ready) return PRALINE_OK;
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream2));
    pool_clear();
    return PRALINE_OK;
}

extern "C" int praline_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (int rc = ensure_runtime(-1)) return rc;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(PRALINE_ERR_NOMEM, "page-locked host allocation of %zu bytes failed", bytes);
    }
    return PRALINE_OK;
}

extern "C" int praline_host_free(void *p)
{
    if (!p) return PRALINE_OK;
    HIPCHK(hipHostFree(p));
    return PRALINE_OK;
}

extern "C" const char *praline_last_error(void) { return g_err.c_str(); }
extern "C" void *praline_stream(void) { return g_rt.ready ? (void *)g_rt.stream : nullptr; }

// Device buffers come from a small pool: plans allocate multi-GB scratch (strip boundaries, packed
// traceback, paths) and hipMalloc / hipFree of such blocks costs 100s of ms.  Released blocks are kept
// and handed out again when a request fits (block <= 2x request); praline_shutdown frees them.
// The pool is STREAM-ORDERED over the library's two streams: a released block carries two events, recorded on both
// streams at release time, and is only handed out again once both have completed - a buffer that is replaced while
// kernels of an earlier launch (on either stream) may still be using it can therefore never reach another user early.
// (Round 2's chunk-scratch race was this: a block released inside the chunk loop went to the other stream's set.)
struct PoolBlock { void *p; size_t bytes; hipEvent_t ev[2]; };
static std::vector<PoolBlock> g_pool;
static std::vector<hipEvent_t> g_pool_events;   // spare events
static size_t g_pool_bytes = 0;
// Cap of the cached (released, not yet freed) bytes: PRALINE_POOL_KEEP_MB, default 64 GiB of the 288 - two launch
// chunks' worth of path-plan scratch (24 GiB each, PRALINE_TB_BUDGET_MB): with a smaller cap every C3-size plan paid
// seconds of hipMalloc / hipFree.  praline_pool_trim() returns every cached block to the driver (call it before
// another allocator needs the memory).  The pool, like the rest of the library, is for one host thread per device.
static size_t pool_keep_bytes()
{
    static size_t keep = (size_t)-1;
    if (keep == (size_t)-1) {
        keep = (size_t)64 << 30;
        if (const char *env = getenv("PRALINE_POOL_KEEP_MB")) keep = (size_t)atoll(env) << 20;
    }
    return keep;
}

static hipEvent_t pool_event()
{
    if (!g_pool_events.empty()) { hipEvent_t e = g_pool_events.back(); g_pool_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}

static bool pool_block_ready(const PoolBlock &b)
{
    for (int k = 0; k < 2; ++k)
        if (b.ev[k] && hipEventQuery(b.ev[k]) != hipSuccess) return false;
    return true;
}

static void pool_block_retire(PoolBlock &b, bool wait)
{
    for (int k = 0; k < 2; ++k)
        if (b.ev[k]) {
            if (wait) (void)hipEventSynchronize(b.ev[k]);
            g_pool_events.push_back(b.ev[k]);
            b.ev[k] = nullptr;
        }
}

static void *pool_alloc(size_t bytes, size_t *got)
{
    // smallest cached block that fits and whose release point both streams have passed; a request of a GiB or more
    // takes ANY block that fits (a first hipMalloc of a 24 GiB scratch block costs the better part of a second - more
    // than the C3 stage it serves) and would rather wait for a block in flight than go to the driver; smaller ones
    // only take blocks of up to twice their size
    if (bytes < ((size_t)1 << 20)) {   // small requests: power-of-two size classes from 4 KiB, so that released blocks fit again
        size_t cls = 4096;
        while (cls < bytes) cls <<= 1;
        bytes = cls;
    }
    size_t best = (size_t)-1, best_busy = (size_t)-1;
    for (size_t i = 0; i < g_pool.size(); ++i) {
        if (!(g_pool[i].bytes >= bytes && (bytes >= ((size_t)1 << 30) || g_pool[i].bytes <= 2 * bytes + (bytes >= ((size_t)1 << 20) ? (1 << 20) : 0)))) continue;
        size_t &slot = pool_block_ready(g_pool[i]) ? best : best_busy;
        if (slot == (size_t)-1 || g_pool[i].bytes < g_pool[slot].bytes) slot = i;
    }
    if (best == (size_t)-1 && best_busy != (size_t)-1 && bytes >= ((size_t)1 << 30)) best = best_busy;
    if (best != (size_t)-1) {
        PoolBlock b = g_pool[best];
        g_pool.erase(g_pool.begin() + best);
        g_pool_bytes -= b.bytes;
        pool_block_retire(b, true);
        *got = b.bytes;
        return b.p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // give cached blocks back to the driver and retry once
        pool_clear();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return nullptr;
    }
    *got = bytes;
    return p;
}

static void pool_release(void *p, size_t bytes)
{
    if (!p) return;
    // (small blocks are cached too: a plan holds a dozen buffers below a MiB, and a hipFree - which waits for the device -
    // of each made praline_plan_destroy of a C3-sized plan 18-20 ms of host time; they come in power-of-two sizes,
    // pool_alloc)
    if (g_pool_bytes + bytes > pool_keep_bytes() || !g_rt.ready) { (void)hipFree(p); return; }
    PoolBlock b{p, bytes, {pool_event(), pool_event()}};
    const hipStream_t streams[2] = {g_rt.stream, g_rt.stream2};
    for (int k = 0; k < 2; ++k)
        if (!b.ev[k] || hipEventRecord(b.ev[k], streams[k]) != hipSuccess) {
            // no event to order the reuse by: fall back to the driver (synchronising free)
            pool_block_retire(b, false);
            (void)hipFree(p);
            return;
        }
    g_pool.push_back(b);
    g_pool_bytes += bytes;
}

static void pool_clear()
{
    for (auto &b : g_pool) { pool_block_retire(b, true); (void)hipFree(b.p); }
    g_pool.clear();
    g_pool_bytes = 0;
    for (hipEvent_t e : g_pool_events) (void)hipEventDestroy(e);
    g_pool_events.clear();
}

extern "C" int64_t praline_pool_cached_bytes(void) { return (int64_t)g_pool_bytes; }

// small RAII device buffer
template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;          // elements requested
    size_t cap_bytes = 0;  // bytes of the underlying block
    ~DevBuf() { release(); }
    void release() { if (p) { pool_release(p, cap_bytes); p = nullptr; n = 0; cap_bytes = 0; } }
    int alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        size_t got = 0;
        p = (T *)pool_alloc(count * sizeof(T), &got);
        if (!p) return fail(PRALINE_ERR_NOMEM, "device allocation of %zu bytes failed", count * sizeof(T));
        n = count;
        cap_bytes = got;
        return PRALINE_OK;
    }
    int upload(const T *src, size_t count, hipStream_t st)
    {
        if (count == 0) return PRALINE_OK;
        HIPCHK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
        return PRALINE_OK;
    }
    template <class A> int upload(const std::vector<T, A> &v, hipStream_t st)
    {
        int rc = alloc(v.size());
        if (rc) return rc;
        return upload(v.data(), v.size(), st);
    }
};

#define RC(expr) do { int rc_ = (expr); if (rc_ != PRALINE_OK) return rc_; } while (0)

// PRALINE_TIMING=1: host-side phase times of arena / plan creation on stderr (scripts/exp_e2e.py)
struct PhaseTimer {
    bool on;
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit PhaseTimer(const char *w) : on(getenv("PRALINE_TIMING") != nullptr), what(w), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *phase)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] %s: %-28s %8.3f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// The rest of the host side, in the order it is compiled (one translation unit: the parts share the static runtime state
// above):
#include "praline_arena.hip.h"
#include "praline_plan.hip.h"
#include "praline_plan_run.hip.h"
#include "praline_stage.hip.h"
#include "praline_raw.hip.h"
#include "praline_rawb.hip.h"
