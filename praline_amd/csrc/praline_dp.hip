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

extern "C" int praline_init(int device) { return ensure_runtime(device); }

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
    int upload(const std::vector<T> &v, hipStream_t st)
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

// --------------------------------------------------------------------------------------------
// arena
// --------------------------------------------------------------------------------------------
static const int kNstepChoices[] = {2, 8, 10, 12, 14, 16};

struct praline_arena {
    int64_t n_seqs = 0;
    int A = 0;               // alphabet size of the raw profiles
    int n_active = 0;        // symbols that can contribute to a match score
    int nstep = 0;           // MFMA steps per tile (template instance)
    int KP = 0, KS = 0;
    int64_t rows_raw = 0, rows_pad = 0;
    int max_len = 0;
    std::vector<int32_t> len, row_off_pad, row_off_raw, active;
    // host sources of the creation's asynchronous uploads (kept: praline_arena_create does not wait for them)
    bool building = false;       // between praline_arena_begin and praline_arena_finish: only praline_arena_put_rows may touch it
    unsigned inexact_bits = 0;   // some value of the first rows is not a normal float16 (praline_arena_put_rows)
    std::vector<float> h_S;
    std::vector<int32_t> h_seq_of_rowp, h_active_up;
    std::vector<unsigned char> h_slot_of;
    DevBuf<float> d_raw, d_S, d_P, d_Q;
    DevBuf<int32_t> d_len, d_row_off_pad, d_row_off_raw, d_seq_of_rowp, d_active;
    // f16 split operands for k_dp_split16 (matrix-pipe MFMA)
    int nr16 = 0;          // 16-wide k ranges (1 or 2); 0 = not available (> 32 active symbols)
    int nterm16 = 3;       // 1: every operand is exactly representable in f16, 3: hi/lo split in six MFMAs per step (NR = 2),
                           // 2: the same three terms K-packed into four MFMAs (at most 21 active symbols; dp_kernels.hip.h)
    DevBuf<char> d_P16, d_Q16;
    DevBuf<int> d_flag16;
    // one-hot arenas (ordinary sequences): active-symbol index per padded row; see k_dp_split16<.., ONEHOT>
    bool onehot = false;       // one-hot operand table in use
    bool all_onehot = false;   // every profile row is one-hot (plain sequences): required by the preprofile counting
    int s_scale_bits = -1;     // smallest k <= 8 with S * 2^k integral in every entry (-1: none); s_absmax = max |S|
    float s_absmax = 0.0f;
    DevBuf<unsigned char> d_sym8;
    // preprofile stage (k_path_counts): raw symbol of every one-hot row (255: not one-hot), int32 counts [rows_raw][A]
    DevBuf<unsigned char> d_sym_raw;
    DevBuf<int32_t> d_counts;
    int32_t *counts_ext = nullptr;   // caller-owned count buffer (praline_arena_counts_bind)
    int32_t *counts_ptr() const { return counts_ext ? counts_ext : d_counts.p; }
    // reference-order audit mode (k_match_ref): track-set partition of the alphabet axis and per-row nonzero lists
    std::vector<int32_t> set_lo;     // n_sets + 1 boundaries, default {0, A}
    DevBuf<int32_t> d_set_lo;
    DevBuf<unsigned char> d_nzidx, d_nzcnt;
    DevBuf<float> d_reft;    // T[row][i][b] (k_build_reft), ref_tb floats per (row, symbol); ref_tb = 0: not built
    int ref_tb = 0;          // nonzeros per row, rounded up to 4 / 8 / 16 / 32 (0: more)
    int reft_state = 0;      // d_reft: 0 not tried, 1 built, -1 not available (too large / too many nonzeros)
    bool ref_ready = false;
    // the same half-terms with two adjacent columns interleaved, for k_match_tile (dp_reftile.hip.h): T2[i][pair row][b][2]
    DevBuf<float> d_reft2;
    DevBuf<int64_t> d_pr_off;   // first pair row of every sequence
    int64_t pair_rows = 0;
    int reft2_state = 0;        // 0: not tried, 1: built, -1: not available for this arena (alphabet / row density)
    // resident progressive alignment (praline_arena_append_merged): integer counts of every row, capacities
    DevBuf<int32_t> d_cnt;
    bool have_cnt = false;
    int64_t cap_rows_raw = 0, cap_rows_pad = 0, cap_seqs = 0;   // 0: the buffers hold exactly what is in use
    int64_t rp_end = 0;       // padded rows taken by sequences (the zero tail follows)
    bool wide = false;       // more than 32 active symbols: no MFMA operand layouts; every plan runs the reference-order path
    // per-position gap scores (praline_arena_set_gap_scores): (open, extend) per padded row; plans created while they
    // are set read their match scores from dense tiles and can run with them (praline_plan_run_gaps)
    DevBuf<float> d_gaps;
    bool has_gaps = false;
    Arena16Dev view16() const
    {
        Arena16Dev v;
        v.sym8 = (onehot && nterm16 == 1) ? d_sym8.p : nullptr;
        // the staged stream addresses the arena with 32-bit lane offsets
        const char *ns = getenv("PRALINE_NO_STAGE");
        v.stage = (!(ns && ns[0] == '1') && (uint64_t)rows_pad * 64 * nr16 < 0xffff0000ull) ? 1 : 0;
        v.P16 = d_P16.p; v.Q16 = d_Q16.p; v.row_off = d_row_off_pad.p; v.len = d_len.p;
        v.half_bytes = 2 * nr16 * 16; v.row_bytes = 2 * v.half_bytes;
        return v;
    }
    ArenaDev view() const
    {
        ArenaDev v;
        v.P = d_P.p; v.Q = d_Q.p; v.row_off = d_row_off_pad.p; v.len = d_len.p; v.KP = KP; v.KS = KS;
        return v;
    }
};

// Every arena entry point but praline_arena_put_rows / _finish / _destroy goes through this: an arena between
// praline_arena_begin and praline_arena_finish has no tables, no operands and no lengths on the device yet.
static int arena_ready(const praline_arena *a)
{
    if (!a) return fail(PRALINE_ERR_ARG, "arena is NULL");
    if (a->building) return fail(PRALINE_ERR_ARG, "the arena is still being built (praline_arena_finish)");
    return PRALINE_OK;
}

static int arena_launch_premultiply(praline_arena *a, bool check_f16 = false)
{
    if (a->wide) return PRALINE_OK;   // no packed operands: plans on this arena read the raw profiles (k_match_ref)
    if (!check_f16) {   // the recurring call: everything in one launch
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)(a->rows_pad / 32)), dim3(256), 0, g_rt.stream, a->d_raw.p, a->d_S.p,
                           a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, a->d_active.p, a->n_active,
                           a->A, a->KP, a->KS, a->rows_pad, a->d_P.p, a->d_Q.p, a->nr16, (_Float16 *)a->d_P16.p,
                           (_Float16 *)a->d_Q16.p, (int64_t)0, a->nterm16 == 2 ? 1 : 0);
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    const int64_t total = a->rows_pad * a->KP;
    const int threads = 256;
    const int64_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(k_pack_profiles, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, a->d_raw.p,
                       a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p,
                       a->d_active.p, a->n_active, a->A, a->KP, a->KS, a->rows_pad, a->d_P.p);
    dim3 grid((unsigned)((a->rows_pad + 31) / 32), (unsigned)((a->KP + 31) / 32));
    hipLaunchKernelGGL(k_premultiply, grid, dim3(64), 0, g_rt.stream, a->d_raw.p, a->d_S.p,
                       a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p,
                       a->d_active.p, a->n_active, a->A, a->KP, a->KS, a->rows_pad, a->d_Q.p);
    if (a->nr16 > 0) {
        int *flag = check_f16 ? a->d_flag16.p : nullptr;
        if (check_f16) HIPCHK(hipMemsetAsync(a->d_flag16.p, 0, sizeof(int), g_rt.stream));
        praline_launch_split_f16(a->d_P.p, a->KP, a->KS, a->n_active, a->nr16, a->rows_pad, a->d_P16.p, flag, g_rt.stream);
        praline_launch_split_f16(a->d_Q.p, a->KP, a->KS, a->n_active, a->nr16, a->rows_pad, a->d_Q16.p, flag, g_rt.stream);
    }
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// Arena creation in three steps (praline_arena_create is begin + one put + finish): begin sizes the arena and allocates
// the raw rows, put uploads a range of rows (asynchronously: a caller that concatenates per-sequence arrays into
// page-locked staging uploads the first half while it copies the second), finish scans, packs and pre-multiplies.
static int arena_begin(int64_t n_seqs, const int32_t *lens, int32_t A, praline_arena **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens) return fail(PRALINE_ERR_ARG, "NULL or empty arena input");
    // the raw (concatenated) alphabet may be wide; what the kernels bound is the number of ACTIVE symbols (<= 32)
    if (A <= 0 || A > 254) return fail(PRALINE_ERR_ARG, "alphabet size %d not in 1..254 (concatenated track sets)", A);
    RC(ensure_runtime(-1));
    praline_arena *a = new praline_arena();
    a->n_seqs = n_seqs;
    a->A = A;
    a->set_lo = {0, A};
    a->len.assign(lens, lens + n_seqs);
    a->row_off_pad.resize(n_seqs);
    a->row_off_raw.resize(n_seqs);
    int64_t rr = 0, rp = 0;
    for (int64_t s = 0; s < n_seqs; ++s) {
        if (lens[s] <= 0) { delete a; return fail(PRALINE_ERR_ARG, "sequence %lld has length %d (must be >= 1)", (long long)s, lens[s]); }
        a->row_off_raw[s] = (int32_t)rr;
        a->row_off_pad[s] = (int32_t)rp;
        rr += lens[s];
        rp += (lens[s] + 31) / 32 * 32;
        a->max_len = std::max(a->max_len, lens[s]);
        if (rp > (int64_t)1 << 30) { delete a; return fail(PRALINE_ERR_ARG, "arena too large"); }
    }
    a->rows_raw = rr;
    a->rp_end = rp;
    // tail padding: the kernels prefetch one row past the longest sequence and read whole strips
    a->rows_pad = rp + (a->max_len + 31) / 32 * 32 + 64;
    int rc0 = PRALINE_OK;
    if ((rc0 = a->d_raw.alloc((size_t)rr * A)) || (rc0 = a->d_sym_raw.alloc((size_t)rr))) { delete a; return rc0; }
    a->building = true;
    *out = a;
    return PRALINE_OK;
}

static int arena_put(praline_arena *a, int64_t row0, int64_t n_rows, const float *rows)
{
    if (!a || !a->building) return fail(PRALINE_ERR_ARG, "rows can only be put into an arena between begin and finish");
    if (!rows || row0 < 0 || n_rows < 0 || row0 + n_rows > a->rows_raw) return fail(PRALINE_ERR_ARG, "row range %lld + %lld outside the arena's %lld rows", (long long)row0, (long long)n_rows, (long long)a->rows_raw);
    if (row0 == 0) {
        // ... whether some value is NOT a normal float16 (13 low mantissa bits set, or an exponent outside -14 .. 15): such
        // an arena needs the hi/lo split whatever S is, so the device-side exactness check (and the second packing launch
        // that follows its read-back) can be skipped - float profiles, i.e. every preprofile / profile-profile stage
        // (looked for in the first 64 K values only: float profiles show one in their first rows, and arenas that show none
        // there keep the device-side check)
        unsigned inexact_bits = 0;
        for (int64_t k = 0, n = std::min<int64_t>(n_rows * a->A, 65536); k < n && !inexact_bits; ++k) {
            unsigned u;
            memcpy(&u, &rows[k], 4);
            const unsigned e = (u >> 23) & 0xffu;
            if (u & 0x7fffffffu) inexact_bits = (u & 0x1fffu) | (unsigned)(e < 113u) | (unsigned)(e > 142u);
        }
        a->inexact_bits = inexact_bits;
    }
    // (a DMA when the caller's buffer is page-locked: praline_host_alloc)
    if (n_rows > 0) HIPCHK(hipMemcpyAsync(a->d_raw.p + row0 * a->A, rows, (size_t)n_rows * a->A * sizeof(float), hipMemcpyHostToDevice, g_rt.stream));
    return PRALINE_OK;
}

// (destroys the arena when it fails)
static int arena_finish(praline_arena *a, const float *S)
{
    if (!a || !a->building) return fail(PRALINE_ERR_ARG, "the arena is not being built");
    if (!S) { (void)hipStreamSynchronize(g_rt.stream); delete a; return fail(PRALINE_ERR_ARG, "NULL score matrix"); }
    PhaseTimer pt("arena_create");
    const int64_t n_seqs = a->n_seqs, rr = a->rows_raw;
    const int A = a->A;
    const int32_t *lens = a->len.data();
    const unsigned inexact_bits = a->inexact_bits;
    // active symbols: i contributes to m = sum_i P1[y,i] * Q2[x,i] only if some profile has mass on
    // it and row i of S is not all zero; dropping the others is exact (their terms are +-0).
    // One pass over the raw profiles gathers everything the host needs from them: which symbols carry mass, and per
    // row whether it is one-hot and on which symbol (counted branch-free so that the loop vectorises).
    std::vector<char> has_mass(A, 0), has_score(A, 0);
    bool all_onehot_rows = true;
    hipStream_t st = g_rt.stream;
    // the raw profiles go up first; the scan of their rows (mass per symbol, one-hot rows) runs on the device - one
    // small read-back instead of 0.6 ms of host time for the 11 MB of C2.  Everything the host can prepare without the
    // scan's answer is done while the upload is in flight (a DMA when the caller's buffer is page-locked:
    // praline_host_alloc).
    int *const d_flags = g_rt.d_scan_flags;   // (A <= 254; owned by the runtime: nothing is freed when this call returns)
    int *flags = g_rt.h_flags;
    {
        hipError_t e0 = hipMemsetAsync(d_flags, 0, ((size_t)A + 1) * sizeof(int), st);
        if (e0 == hipSuccess) {
            hipLaunchKernelGGL(k_scan_profiles, dim3((unsigned)((rr + 255) / 256)), dim3(256), 0, st, a->d_raw.p, rr, A, a->d_sym_raw.p, d_flags);
            e0 = hipGetLastError();
        }
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(flags, d_flags, ((size_t)A + 1) * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e0 != hipSuccess) { (void)hipStreamSynchronize(st); delete a; return fail(PRALINE_ERR_DEVICE, "arena scan: %s", hipGetErrorString(e0)); }
    }
    pt.mark("upload + device scan enqueued");
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < A; ++j)
            if (S[i * A + j] != 0.0f) has_score[i] = 1;
    for (int k = 0; k <= 8 && a->s_scale_bits < 0; ++k) {
        bool ok = true;
        for (int i = 0; i < A * A && ok; ++i) {
            const float v = S[i] * (float)(1 << k);
            ok = std::isfinite(v) && v == std::nearbyint(v);
        }
        if (ok) a->s_scale_bits = k;
    }
    for (int i = 0; i < A * A; ++i) a->s_absmax = std::max(a->s_absmax, std::fabs(S[i]));
    std::vector<int32_t> &seq_of_rowp = a->h_seq_of_rowp;
    seq_of_rowp.assign((size_t)a->rows_pad, -1);
    for (int64_t s = 0; s < n_seqs; ++s)
        std::fill(seq_of_rowp.begin() + a->row_off_pad[s], seq_of_rowp.begin() + a->row_off_pad[s] + (lens[s] + 31) / 32 * 32, (int32_t)s);
    a->h_S.assign(S, S + (size_t)A * A);
    {
        int rc0 = PRALINE_OK;
        if ((rc0 = a->d_S.alloc((size_t)A * A)) || (rc0 = a->d_S.upload(a->h_S.data(), (size_t)A * A, st)) ||
            (rc0 = a->d_len.upload(a->len, st)) || (rc0 = a->d_row_off_pad.upload(a->row_off_pad, st)) ||
            (rc0 = a->d_row_off_raw.upload(a->row_off_raw, st)) || (rc0 = a->d_seq_of_rowp.upload(seq_of_rowp, st)) ||
            (rc0 = a->d_flag16.alloc(1))) {
            (void)hipStreamSynchronize(st);
            delete a;
            return rc0;
        }
    }
    pt.mark("host tables");
    {
        const hipError_t e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena scan: %s", hipGetErrorString(e0)); }
        for (int i = 0; i < A; ++i) has_mass[i] = (char)(flags[(size_t)i] != 0);
        all_onehot_rows = flags[(size_t)A] == 0;
    }
    pt.mark("wait for the scan");
    for (int i = 0; i < A; ++i)
        if (has_mass[i] && has_score[i]) a->active.push_back(i);
    a->n_active = (int)a->active.size();
    if (const char *env = getenv("PRALINE_NO_COMPACT")) {
        if (env[0] == '1') { a->active.resize(A); std::iota(a->active.begin(), a->active.end(), 0); a->n_active = A; }
    }
    const int need = std::max(1, (a->n_active + 1) / 2);
    a->nstep = 0;
    for (int c : kNstepChoices) if (c >= need) { a->nstep = c; break; }
    if (!a->nstep) {
        // more active symbols than the MFMA operand layouts hold (32): the arena keeps the raw profiles only and its
        // plans evaluate the match scores on the vector ALU in the reference's order (k_match_ref, any alphabet <= 254)
        a->wide = true;
        a->nstep = 2;
    }
    a->KS = (a->nstep + 3) / 4 * 4;
    a->KP = 2 * a->KS;
    a->nr16 = a->wide ? 0 : (a->n_active <= 16 ? 1 : 2);

    // one-hot arenas (every row: a single 1, zeros elsewhere): active-symbol bytes for the one-hot operand table (built
    // on the device below: k_build_sym8)
    std::vector<unsigned char> &slot_of = a->h_slot_of;
    {
        const bool want_table = a->nr16 > 0 && !(getenv("PRALINE_NO_ONEHOT") && getenv("PRALINE_NO_ONEHOT")[0] == '1');
        a->all_onehot = all_onehot_rows;
        a->onehot = all_onehot_rows && want_table;
        if (a->onehot) {
            const unsigned char none = (unsigned char)(16 * a->nr16);
            slot_of.assign(256, none);
            for (int k = 0; k < a->n_active; ++k) slot_of[a->active[k]] = (unsigned char)k;
        }
    }

    int rc = PRALINE_OK;
    a->h_active_up = a->active.empty() ? std::vector<int32_t>(1, 0) : a->active;
    if ((rc = a->d_active.upload(a->h_active_up, st)) ||
        (rc = a->d_P.alloc(a->wide ? 1 : (size_t)a->rows_pad * a->KP)) || (rc = a->d_Q.alloc(a->wide ? 1 : (size_t)a->rows_pad * a->KP)) ||
        (a->onehot && (rc = a->d_sym8.alloc((size_t)a->rows_pad + 64))) ||
        (a->nr16 > 0 && ((rc = a->d_P16.alloc((size_t)a->rows_pad * 4 * a->nr16 * 16)) ||
                         (rc = a->d_Q16.alloc((size_t)a->rows_pad * 4 * a->nr16 * 16))))) {
        (void)hipStreamSynchronize(st);
        delete a;
        return rc;
    }
    if (a->onehot) {
        // (the table is read by k_build_sym8 below from the runtime's 256-byte buffer; its host source lives in the arena)
        if (hipMemcpyAsync(g_rt.d_slot_of, slot_of.data(), 256, hipMemcpyHostToDevice, st) != hipSuccess) {
            (void)hipStreamSynchronize(st);
            delete a;
            return fail(PRALINE_ERR_DEVICE, "symbol table upload failed");
        }
        const int64_t rows_out = a->rows_pad + 64;
        hipLaunchKernelGGL(k_build_sym8, dim3((unsigned)((rows_out + 255) / 256)), dim3(256), 0, st, a->d_sym_raw.p, a->d_seq_of_rowp.p,
                           a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, g_rt.d_slot_of, a->rows_pad, rows_out, a->d_sym8.p);
        if (hipGetLastError() != hipSuccess) { (void)hipStreamSynchronize(st); delete a; return fail(PRALINE_ERR_DEVICE, "k_build_sym8 launch failed"); }
    }
    pt.mark("allocations + uploads (async)");
    const bool host_knows_split = a->nr16 > 0 && inexact_bits != 0;
    if (host_knows_split) {
        // profiles that float16 cannot hold: three terms (K-packed into four MFMAs for at most 21 active symbols), one
        // fused pack / pre-multiply / split launch, no read-back
        const char *pk = getenv("PRALINE_PACKED3");
        a->nterm16 = (a->nr16 == 2 && 3 * a->n_active <= 63 && !(pk && pk[0] == '0')) ? 2 : 3;
        rc = arena_launch_premultiply(a);
    } else {
        rc = arena_launch_premultiply(a, true);
    }
    if (rc != PRALINE_OK) { (void)hipStreamSynchronize(st); delete a; return rc; }
    pt.mark("premultiply launches");
    // Float profiles: nothing to read back - the packing launch and the small uploads (their host sources live in the
    // arena) finish under whatever the caller does next on this stream (plan creation waits for its own uploads).
    hipError_t e = hipSuccess;
    if (!host_knows_split) {
        e = hipStreamSynchronize(st);
        pt.mark("stream sync");
    }
    if (e != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena upload: %s", hipGetErrorString(e)); }
    if (a->nr16 > 0 && !host_knows_split) {
        int flag = 1;
        e = hipMemcpy(&flag, a->d_flag16.p, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena flag: %s", hipGetErrorString(e)); }
        a->nterm16 = flag ? 3 : 1;
        // at most 21 active symbols: the three terms fit the 64 k slots of four MFMAs (PRALINE_PACKED3=0: keep six)
        const char *pk = getenv("PRALINE_PACKED3");
        if (a->nterm16 == 3 && a->nr16 == 2 && 3 * a->n_active <= 63 && !(pk && pk[0] == '0')) {
            a->nterm16 = 2;
            rc = arena_launch_premultiply(a);   // re-split in the packed layout
            if (rc == PRALINE_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "arena re-split failed");
            if (rc != PRALINE_OK) { delete a; return rc; }
        }
    }
    a->building = false;
    return PRALINE_OK;
}

extern "C" int praline_arena_create(int64_t n_seqs, const int32_t *lens, int32_t A, const float *profiles,
                                    const float *S, praline_arena **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens || !profiles || !S) return fail(PRALINE_ERR_ARG, "NULL or empty arena input");
    praline_arena *a = nullptr;
    RC(arena_begin(n_seqs, lens, A, &a));
    int rc = arena_put(a, 0, a->rows_raw, profiles);
    if (rc != PRALINE_OK) { (void)hipStreamSynchronize(g_rt.stream); delete a; return rc; }
    RC(arena_finish(a, S));
    *out = a;
    return PRALINE_OK;
}

extern "C" int praline_arena_begin(int64_t n_seqs, const int32_t *lens, int32_t A, praline_arena **out)
{
    return arena_begin(n_seqs, lens, A, out);
}

extern "C" int praline_arena_put_rows(praline_arena *arena, int64_t row0, int64_t n_rows, const float *rows)
{
    return arena_put(arena, row0, n_rows, rows);
}

extern "C" int praline_arena_finish(praline_arena *arena, const float *S) { return arena_finish(arena, S); }

extern "C" int praline_arena_destroy(praline_arena *arena)
{
    if (!arena) return PRALINE_OK;
    if (g_rt.ready) (void)hipStreamSynchronize(g_rt.stream);
    delete arena;
    return PRALINE_OK;
}

extern "C" int praline_arena_set_track_sets(praline_arena *arena, int32_t n_sets, const int32_t *sizes)
{
    RC(arena_ready(arena));
    if (!arena || n_sets <= 0 || !sizes) return fail(PRALINE_ERR_ARG, "bad track-set arguments");
    std::vector<int32_t> lo(1, 0);
    for (int n = 0; n < n_sets; ++n) {
        if (sizes[n] <= 0) return fail(PRALINE_ERR_ARG, "track set %d has size %d", n, sizes[n]);
        lo.push_back(lo.back() + sizes[n]);
    }
    if (lo.back() != arena->A) return fail(PRALINE_ERR_ARG, "track-set sizes sum to %d, the arena alphabet is %d", lo.back(), arena->A);
    arena->set_lo.swap(lo);
    arena->ref_ready = false;
    arena->reft2_state = 0;
    return PRALINE_OK;
}

// Per-position gap scores (GapScoreModel, praline/container/score.py:45-68): g = float32 [sum of the lengths][2] =
// (open, extend) of every position of every sequence, in arena order; NULL: back to constant gap scores.
extern "C" int praline_arena_set_gap_scores(praline_arena *arena, const float *g)
{
    RC(arena_ready(arena));
    if (!arena) return fail(PRALINE_ERR_ARG, "arena is NULL");
    RC(ensure_runtime(-1));
    if (!g) { arena->has_gaps = false; arena->d_gaps.release(); return PRALINE_OK; }
    // gap rows exist for the sequences the arena holds NOW: an arena that grows (praline_arena_set_counts /
    // praline_arena_append_merged) would leave its appended sequences without any - the two are mutually exclusive
    if (arena->have_cnt || arena->cap_seqs != 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "gap scores on a growing arena (praline_arena_set_counts) are not supported");
    const size_t rows = (size_t)arena->rows_pad + 64;
    std::vector<float> pad(rows * 2, 0.0f);
    for (int64_t q = 0; q < arena->n_seqs; ++q) {
        const float *src = g + (size_t)arena->row_off_raw[(size_t)q] * 2;
        const int L = arena->len[(size_t)q];
        for (int k = 0; k < 2 * L; ++k) {
            if (!(src[k] <= 0.0f)) return fail(PRALINE_ERR_UNSUPPORTED, "batched kernels need gap scores <= 0 (sequence %lld, position %d: %g)", (long long)q, k / 2, src[k]);
        }
        std::copy(src, src + 2 * (size_t)L, pad.begin() + (size_t)arena->row_off_pad[(size_t)q] * 2);
    }
    hipStream_t st = g_rt.stream;
    RC(arena->d_gaps.upload(pad, st));
    HIPCHK(hipStreamSynchronize(st));
    arena->has_gaps = true;
    return PRALINE_OK;
}

static int arena_ensure_ref(praline_arena *a);
static int arena_ensure_reft_table(praline_arena *a);

// reference-order match scores of the pairs chunk_pairs[0 .. n_chunk) into mref + m_off[pair]
static int launch_match_ref(praline_arena *a, const int32_t *d_pairs, const int32_t *d_chunk_pairs, size_t n_chunk, int max_l1,
                            const int64_t *d_m_off, float *d_mref, const TileOut &to = TileOut())
{
    RC(arena_ensure_ref(a));
    RC(arena_ensure_reft_table(a));
    hipStream_t st = g_rt.stream;
    const dim3 grid((unsigned)n_chunk, (unsigned)((max_l1 + PRALINE_REF_ROWS - 1) / PRALINE_REF_ROWS)), block(256);
    const int n_sets = (int)a->set_lo.size() - 1;
#define PRALINE_REFT(TB)                                                                                               \
    hipLaunchKernelGGL((k_match_reft<TB>), grid, block, 0, st, a->d_raw.p, a->A, a->d_reft.p, a->rows_raw, a->d_row_off_raw.p, a->d_len.p,  \
                       a->d_nzidx.p, a->d_nzcnt.p, a->d_set_lo.p, n_sets, d_pairs, d_chunk_pairs, d_m_off, d_mref, to)
    switch (a->reft_state == 1 ? a->ref_tb : 0) {
        case 4: PRALINE_REFT(4); break;
        case 8: PRALINE_REFT(8); break;
        case 16: PRALINE_REFT(16); break;
        case 32: PRALINE_REFT(32); break;
        default:
            hipLaunchKernelGGL(k_match_ref, grid, block, 0, st, a->d_raw.p, a->d_S.p, a->A, a->d_row_off_raw.p, a->d_len.p, a->d_nzidx.p,
                               a->d_nzcnt.p, a->d_set_lo.p, n_sets, d_pairs, d_chunk_pairs, d_m_off, d_mref, to);
    }
#undef PRALINE_REFT
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// nonzero lists + set boundaries for k_match_ref, built on first use
static int arena_ensure_ref(praline_arena *a)
{
    if (a->ref_ready) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    RC(a->d_set_lo.upload(a->set_lo, st));
    RC(a->d_nzidx.alloc((size_t)a->rows_raw * a->A));
    RC(a->d_nzcnt.alloc((size_t)a->rows_raw));
    hipLaunchKernelGGL(k_build_nz, dim3((unsigned)((a->rows_raw + 255) / 256)), dim3(256), 0, st, a->d_raw.p, a->rows_raw, a->A,
                       a->d_nzidx.p, a->d_nzcnt.p);
    HIPCHK(hipGetLastError());
    // the per-row tables of k_match_reft (half of every term prepared once per arena row) when they fit
    std::vector<unsigned char> cnt((size_t)a->rows_raw);
    HIPCHK(hipMemcpyAsync(cnt.data(), a->d_nzcnt.p, cnt.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int max_nz = 1;
    for (unsigned char c : cnt) max_nz = std::max(max_nz, (int)c);
    a->ref_tb = max_nz <= 4 ? 4 : (max_nz <= 8 ? 8 : (max_nz <= 16 ? 16 : (max_nz <= 32 ? 32 : 0)));
    a->reft_state = 0;
    a->d_reft.release();
    a->ref_ready = true;
    return PRALINE_OK;
}

// the per-row tables of k_match_reft, built on the first launch that needs them (plans on the tile kernels never do)
static int arena_ensure_reft_table(praline_arena *a)
{
    if (a->reft_state != 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    a->reft_state = -1;
    size_t table_limit = (size_t)16 << 30;
    if (const char *env = getenv("PRALINE_REF_TABLE_MB")) table_limit = (size_t)atoll(env) << 20;
    const size_t t_elems = (size_t)a->rows_raw * a->A * (size_t)a->ref_tb;
    if (a->ref_tb > 0 && t_elems * sizeof(float) <= table_limit) {
        RC(a->d_reft.alloc(t_elems));
        const int64_t n = a->rows_raw * a->A;
        hipLaunchKernelGGL(k_build_reft, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a->d_raw.p, a->d_S.p, a->A, a->rows_raw,
                           a->d_nzidx.p, a->d_nzcnt.p, a->d_set_lo.p, (int)a->set_lo.size() - 1, a->ref_tb, a->d_reft.p);
        HIPCHK(hipGetLastError());
        a->reft_state = 1;
    }
    return PRALINE_OK;
}

// the interleaved table of k_match_tile; state -1 when the arena does not qualify (more than 32 symbols, rows with more
// than 8 nonzeros, table over the limit): such plans take their tiles from k_match_reft / k_match_ref
static int arena_ensure_reft2(praline_arena *a)
{
    if (a->reft2_state != 0) return PRALINE_OK;
    RC(arena_ensure_ref(a));
    a->reft2_state = -1;
    if (a->wide || a->nr16 <= 0 || a->ref_tb <= 0 || !praline_match_tile_supported(a->A, a->ref_tb)) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    std::vector<int64_t> pr_off((size_t)a->n_seqs);
    int64_t pr = 0;
    for (int64_t q = 0; q < a->n_seqs; ++q) { pr_off[(size_t)q] = pr; pr += (a->len[(size_t)q] + 1) / 2; }
    a->pair_rows = std::max<int64_t>(pr, 1);
    RC(a->d_pr_off.upload(pr_off, st));
    RC(a->d_reft2.alloc((size_t)a->A * (size_t)a->pair_rows * (size_t)a->ref_tb * 2));
    RC(praline_launch_build_reft2(a->d_raw.p, a->d_S.p, a->A, a->d_row_off_raw.p, a->d_len.p, a->d_pr_off.p, a->pair_rows, a->d_nzidx.p,
                                  a->d_nzcnt.p, a->d_set_lo.p, (int)a->set_lo.size() - 1, a->ref_tb, a->d_reft2.p, (int)a->n_seqs, st));
    HIPCHK(hipStreamSynchronize(st));   // (pr_off goes out of scope)
    a->reft2_state = 1;
    return PRALINE_OK;
}

// ---- resident progressive alignment: clusters merged on the device, appended to the arena in place ----------
template <typename T> static int grow_buf(DevBuf<T> &b, size_t old_n, size_t new_n, int fill_byte, hipStream_t st)
{
    if (!b.p || new_n <= old_n) return PRALINE_OK;
    DevBuf<T> nb;
    RC(nb.alloc(new_n));
    HIPCHK(hipMemcpyAsync(nb.p, b.p, old_n * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(nb.p + old_n, fill_byte, (new_n - old_n) * sizeof(T), st));
    HIPCHK(hipStreamSynchronize(st));   // the old block goes back to the pool
    std::swap(b.p, nb.p);
    std::swap(b.n, nb.n);
    std::swap(b.cap_bytes, nb.cap_bytes);
    return PRALINE_OK;
}

static int arena_reserve(praline_arena *a, int64_t need_seqs, int64_t need_rows_raw, int64_t need_rows_pad)
{
    hipStream_t st = g_rt.stream;
    const int64_t cur_seqs = a->cap_seqs ? a->cap_seqs : a->n_seqs, cur_raw = a->cap_rows_raw ? a->cap_rows_raw : a->rows_raw,
                  cur_pad = a->cap_rows_pad ? a->cap_rows_pad : a->rows_pad;
    if (need_seqs > cur_seqs) {
        const int64_t n = std::max(need_seqs, 2 * cur_seqs);
        RC(grow_buf(a->d_len, (size_t)cur_seqs, (size_t)n, 0, st));
        RC(grow_buf(a->d_row_off_pad, (size_t)cur_seqs, (size_t)n, 0, st));
        RC(grow_buf(a->d_row_off_raw, (size_t)cur_seqs, (size_t)n, 0, st));
        a->cap_seqs = n;
    }
    if (need_rows_raw > cur_raw) {
        const int64_t n = std::max(need_rows_raw, 2 * cur_raw);
        RC(grow_buf(a->d_raw, (size_t)cur_raw * a->A, (size_t)n * a->A, 0, st));
        RC(grow_buf(a->d_cnt, (size_t)cur_raw * a->A, (size_t)n * a->A, 0, st));
        a->cap_rows_raw = n;
    }
    if (need_rows_pad > cur_pad) {
        const int64_t n = std::max(need_rows_pad, 2 * cur_pad);
        RC(grow_buf(a->d_seq_of_rowp, (size_t)cur_pad, (size_t)n, 0xff, st));   // -1: no sequence
        if (!a->wide) {
            RC(grow_buf(a->d_P, (size_t)cur_pad * a->KP, (size_t)n * a->KP, 0, st));
            RC(grow_buf(a->d_Q, (size_t)cur_pad * a->KP, (size_t)n * a->KP, 0, st));
            if (a->nr16 > 0) {
                RC(grow_buf(a->d_P16, (size_t)cur_pad * 4 * a->nr16 * 16, (size_t)n * 4 * a->nr16 * 16, 0, st));
                RC(grow_buf(a->d_Q16, (size_t)cur_pad * 4 * a->nr16 * 16, (size_t)n * 4 * a->nr16 * 16, 0, st));
            }
        }
        a->cap_rows_pad = n;
    }
    return PRALINE_OK;
}

extern "C" int praline_arena_set_counts(praline_arena *arena, const int32_t *counts, int64_t reserve_seqs, int64_t reserve_rows)
{
    RC(arena_ready(arena));
    if (!arena || !counts) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (arena->has_gaps)
        return fail(PRALINE_ERR_UNSUPPORTED, "the arena holds per-position gap scores (praline_arena_set_gap_scores): it cannot grow");
    RC(ensure_runtime(-1));
    praline_arena *a = arena;
    RC(a->d_cnt.alloc((size_t)a->rows_raw * a->A));
    HIPCHK(hipMemcpyAsync(a->d_cnt.p, counts, (size_t)a->rows_raw * a->A * sizeof(int32_t), hipMemcpyHostToDevice, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    a->have_cnt = true;
    if (reserve_seqs > 0 || reserve_rows > 0)
        RC(arena_reserve(a, a->n_seqs + std::max<int64_t>(reserve_seqs, 0), a->rows_raw + std::max<int64_t>(reserve_rows, 0),
                         a->rows_pad + std::max<int64_t>(reserve_rows, 0) + 32 * std::max<int64_t>(reserve_seqs, 0)));
    return PRALINE_OK;
}

extern "C" int praline_arena_premultiply(praline_arena *arena)
{
    RC(arena_ready(arena));
    if (!arena) return fail(PRALINE_ERR_ARG, "arena is NULL");
    return arena_launch_premultiply(arena);
}

// --------------------------------------------------------------------------------------------
// plan
// --------------------------------------------------------------------------------------------
struct praline_plan {
    praline_arena *arena = nullptr;
    int64_t n_pairs = 0;
    int64_t cells = 0;
    int64_t path_cap = 0;
    bool want_paths = false;
    bool has_rects = false;
    int max_rects = 0;    // rectangles of the pair with the most (lists given at creation)
    int64_t count_runs = -1;            // praline_plan_add_counts: runs of pairs with one master (-1: not looked for yet, 0: none)
    DevBuf<int64_t> d_count_runs;
    int slot_rects = -1;  // >= 0: the rectangles live in fixed slots on the device (praline_plan_mask_path_bounds), this many used
    int mask_kind = 0;   // 0 none, 1 <= PRALINE_MAX_RECTS rectangles per pair (registers), 2 any number (per-row mask words, k_build_zmask)
    int tp = 1;
    bool split = false;  // k_dp_split task layout
    bool quad = false;   // path plan on a one-hot arena in the 16-pairs-per-task layout of k_dp_quad_tb (dp_quad.hip.h)
    // path plan on a one-hot arena with an integral exchange matrix whose DP values fit int16: the 32-pair layout with planes
    // sized for k_dp_pk16_tb (dp_pk16.hip.h); a run whose gap scores do not qualify takes the strip kernels on the same tasks
    bool pk16 = false;
    bool run_pk16 = false;   // the run in progress / the last run used k_dp_pk16_tb
    std::vector<WaveTask> tasks;
    std::vector<int64_t> tb_elems;  // per task, uint4 elements
    std::vector<int64_t> aux_elems; // per task, floats
    int64_t bnd_elems = 0;
    DevBuf<WaveTask> d_tasks;
    DevBuf<WaveTask> d_tasks_chain;   // scores-only chain mode: the tasks with chain-mode boundary offsets
    int scores_chain = -1;            // -1: not decided, 0 / 1: this score plan runs in chain mode (plan_scores_chain_wanted)
    // small batches: four-wave workgroups whose waves share long tasks (k_dp_split16 WPG = 4, WgDesc)
    std::vector<WgDesc> wg;
    DevBuf<WgDesc> d_wg;
    // large batches in LOCAL mode: the same kernel with four independent tasks per workgroup (its one-wave LOCAL
    // instances need 256 VGPRs + ~130 AGPRs, the four-wave ones 185-219: two waves per SIMD)
    std::vector<WgDesc> wg_singles;
    DevBuf<WgDesc> d_wg_singles;
    // pipeline workgroups (k_dp_pipe, dp_pipe.hip.h): scores-only plans on float-profile arenas
    PipeSchedule pipe;
    DevBuf<PipeItem> d_pipe_items;
    DevBuf<WaveTask> d_pipe_tasks;
    DevBuf<int32_t> d_pipe_set_one, d_pipe_lane_pair;
    DevBuf<float2> d_pipe_bnd, d_pipe_analytic;
    // path plans (global mode): the pipeline as the forward fill of the two-pass scheme (k_dp_pipe<..., KEEP> +
    // k_trace_recompute): per-task sequences one, the float4 analytic column, scratch sizes (kept columns in d_bnd2, row
    // checkpoints in d_tb); the tasks carry aux_off / tb_off into them
    DevBuf<int32_t> d_pipe_lane_one;
    DevBuf<float4> d_pipe_analytic4;
    int64_t pipe_keep_bnd_elems = 0, pipe_keep_ck_floats = 0;
    int pipe_analytic_rows = 0;
    int pipe_analytic_mode = -1;            // mode and gap scores the analytic column was last written for (-1: never)
    float pipe_analytic_go = 0.0f, pipe_analytic_ge = 0.0f;
    DevBuf<int32_t> d_lane_one, d_lane_pair, d_pairs, d_rect_off, d_rects, d_end_cells, d_path_rows, d_paths;
    DevBuf<PairLoc> d_loc;
    DevBuf<float> d_scores, d_aux;
    DevBuf<char> d_bnd;
    DevBuf<float4> d_bnd2;      // two-pass mode: every strip's boundary column, kept for the recompute kernel
    std::vector<int64_t> bnd_off0;
    DevBuf<char> d_bnd_chain;   // chain mode: one boundary column per strip boundary
    DevBuf<int> d_chain_flags;  // chain mode: rows published per (task, strip)
    DevBuf<float4> d_chain_cand;  // chain mode, local: first-argmax candidate per (task, strip, pair)
    DevBuf<char> d_tb;
    // second scratch set of chunked path plans (chunks alternate between two streams)
    DevBuf<char> d_tb_b;
    DevBuf<float> d_aux_b;
    DevBuf<float4> d_bnd2_b;
    DevBuf<int64_t> d_slot_off, d_path_start;
    std::vector<int64_t> slot_off;
    float last_kernel_ms = 0.0f;
    int last_mode = -1;
    // plans whose DP reads its match scores from DENSE TILES (the dense-tile instances of k_dp_split16 / k_dp_split16_tb,
    // plan_run_dense) and who writes the tiles:
    //   1  k_match_tile - the reference's summation order (PRALINE_MATCH_REFERENCE) on arenas of up to 32 symbols whose rows
    //      hold at most 8 nonzeros (dp_reftile.hip.h)
    //   2  k_match_reft / k_match_ref, one cell per thread - the reference's order for every other arena (more than 32 active
    //      symbols, denser rows) and for plans with more than PRALINE_MAX_RECTS rectangles per pair on float profiles
    //   3  k_scores_tile_batch, the fp32 MFMA chain - plans created on an arena with per-position gap scores in the default
    //      match mode (both their constant-gap and their per-position runs)
    int dense_kind = 0;
    std::vector<int32_t> h_lane_pair, h_pairs;
    DevBuf<int32_t> d_chunk_pairs;
    bool ppg = false;       // created on an arena with per-position gap scores
    bool run_ppg = false;   // the run in progress uses them (praline_plan_run_gaps)
    DevBuf<float> d_dense;                  // the tiles of one launch chunk
    DevBuf<int64_t> d_dense_off;
    DevBuf<RefTileBlock> d_tile_blocks;
    DevBuf<int32_t> d_tile_grp;
    std::vector<int32_t> h_lane_one;        // [task][32], host copy (groups of tasks with the same sequences one)
    // mask_kind 2: column masks per (pair, strip, row) (k_build_zmask)
    DevBuf<unsigned> d_zmask;
    DevBuf<int64_t> d_zm_off;
    std::string last_kernel;        // the DP kernel instance the last run launched (as rocprofv3 names it)
    float *last_scores = nullptr;   // where the last praline_plan_run wrote the scores (own buffer or the caller's)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the last run's launches, on the launch stream
    ~praline_plan()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

// chain mode (one wave per task and strip) for plans of up to this many tasks.  Measured with paths, float
// profiles, ms per run task mode -> chain mode: 120 pairs 6.6 -> 0.93, 2 016 pairs 6.6 -> 1.5, 8 128 pairs
// 7.1 -> 3.0, C2 (32 640 pairs, 1 144 tasks) 9.6 -> 7.5, 2 048 tasks 5.2 -> 5.2, 4 600 tasks 10.6 -> 11.5.
// Blocks are dispatched in index order and a strip's producer has the smaller index, so a chain never waits for
// a wave that has not been dispatched, whatever fits on the chip at once.
static int64_t chain_max_tasks()
{
    if (const char *env = getenv("PRALINE_CHAIN_MAX_TASKS")) return atoll(env);
    return 2304;   // measured crossover with task mode (scripts/exp_chain.py); within +-5 % of it up to ~4000 tasks
}

#define PRALINE_TB2_PAD_ROWS PRALINE_TB2_PAD

// traceback scratch budget per launch chunk (bytes)
static size_t tb_budget_bytes()
{
    if (const char *env = getenv("PRALINE_TB_BUDGET_MB")) return (size_t)atoll(env) << 20;
    // 8 GiB (two sets of 4 GiB once a plan needs several chunks): with the chunks alternating between two streams the
    // rate is within 3 % of a 24 GiB budget (C3), and a first-use hipMalloc of the scratch costs 0.2 s instead of 0.8
    return (size_t)8 << 30;
}

static size_t reftile_budget_bytes();

// scheduler options of the pipeline workgroups for a pair list of this size (plan creation and praline_sched_prepare)
static PipeOptions pipe_options_for(int64_t n_pairs)
{
    PipeOptions po;
    // sequences two per scheduler block: 32 while the whole plan is resident at once (up to ~2.5 tasks per workgroup
    // slot: C2 1.90 ms against 2.09 with 16), 16 beyond (one rank's share of C4: 47.6 ms against 48.8 with 32 - the
    // unions of 32 columns' sequences one leave more half-filled sets; scripts/exp_pipe_block2.py, exp_c4_block.py)
    po.block_twos = n_pairs <= 40000 ? 32 : 16;
    if (const char *env = getenv("PRALINE_PIPE_BLOCK")) po.block_twos = atoi(env);
    if (const char *env = getenv("PRALINE_PIPE_SLOTS")) po.wg_slots = atoll(env);
    return po;
}

// the pipeline schedule of a scores-only plan over `pairs` (everything praline_plan_create derives from the pair list and
// the sequence lengths alone); below ~200 tasks (all pairs of ~110 sequences) the shared-wave task schedule is as fast
// or faster (scripts/exp_pipe_sweep.py): a pipeline item cannot be smaller than one task
static void pipe_schedule_for(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int max_len, PipeSchedule &pipe)
{
    int min_len = max_len;
    (void)max_len;
    sched_pair_stats(lens, n_seqs, n_pairs, pairs, nullptr, &min_len, nullptr);
    if (min_len >= 1) build_pipe_schedule(lens, n_seqs, n_pairs, pairs, pipe_options_for(n_pairs), pipe);
    int64_t min_tasks = 200;
    if (const char *env = getenv("PRALINE_PIPE_MIN_TASKS")) min_tasks = atoll(env);
    if (pipe.ok && (int64_t)pipe.tasks.size() < min_tasks) pipe = PipeSchedule();
}

// praline_sched_prepare: host-only, may run on another host thread while the arena of the same sequences is created
struct praline_sched {
    std::vector<int32_t> lens;
    int64_t n_pairs = 0;
    std::vector<int32_t> pairs;   // the pair list the schedule belongs to (compared entry by entry when the plan is created)
    PipeSchedule pipe;
};

extern "C" int praline_sched_prepare(int64_t n_seqs, const int32_t *lens, int64_t n_pairs, const int32_t *pairs, praline_sched **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens || n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(PRALINE_ERR_ARG, "bad schedule arguments");
    int max_len = 0;
    for (int64_t s = 0; s < n_seqs; ++s) {
        if (lens[s] <= 0) return fail(PRALINE_ERR_ARG, "sequence %lld has length %d (must be >= 1)", (long long)s, lens[s]);
        max_len = std::max(max_len, lens[s]);
    }
    {
        int64_t bad = -1;
        sched_pair_stats(lens, n_seqs, n_pairs, pairs, nullptr, nullptr, &bad);
        if (bad >= 0) return fail(PRALINE_ERR_ARG, "pair %lld = (%d, %d) out of range", (long long)bad, pairs[2 * bad], pairs[2 * bad + 1]);
    }
    praline_sched *sc = new praline_sched();
    sc->lens.assign(lens, lens + n_seqs);
    sc->n_pairs = n_pairs;
    if (n_pairs > 0) {
        sc->pairs.assign(pairs, pairs + 2 * n_pairs);
        pipe_schedule_for(lens, n_seqs, n_pairs, pairs, max_len, sc->pipe);
    }
    *out = sc;
    return PRALINE_OK;
}

extern "C" int praline_sched_destroy(praline_sched *sched)
{
    delete sched;
    return PRALINE_OK;
}

static int plan_create_impl(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths, const int32_t *rect_off,
                            const int32_t *rects, praline_sched *prep, praline_plan **out);

extern "C" int praline_plan_create(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths,
                                   const int32_t *rect_off, const int32_t *rects, praline_plan **out)
{
    return plan_create_impl(arena, n_pairs, pairs, want_paths, rect_off, rects, nullptr, out);
}

extern "C" int praline_plan_create_prepared(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, praline_sched *sched,
                                            praline_plan **out)
{
    return plan_create_impl(arena, n_pairs, pairs, 0, nullptr, nullptr, sched, out);
}

static int plan_create_impl(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths, const int32_t *rect_off,
                            const int32_t *rects, praline_sched *prep, praline_plan **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!arena || n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(PRALINE_ERR_ARG, "bad plan arguments");
    RC(arena_ready(arena));
    if ((rect_off != nullptr) != (rects != nullptr) && rect_off && rect_off[n_pairs] > 0)
        return fail(PRALINE_ERR_ARG, "rect_off given without rects");
    if (rect_off && !want_paths && rect_off[n_pairs] > 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "zero rectangles are only supported together with want_paths");
    RC(ensure_runtime(-1));
    PhaseTimer pt("plan_create");
    const praline_arena &a = *arena;
    bool many_rects = false;   // some pair carries more rectangles than the register-resident mask code holds
    int max_rects = 0;         // the longest rectangle list of a pair
    int64_t list_cells = 0;
    {
        int64_t bad = -1;
        sched_pair_stats(a.len.data(), a.n_seqs, n_pairs, pairs, &list_cells, nullptr, &bad);
        if (bad >= 0) return fail(PRALINE_ERR_ARG, "pair %lld = (%d, %d) out of range", (long long)bad, pairs[2 * bad], pairs[2 * bad + 1]);
    }
    if (rect_off)
        for (int64_t p = 0; p < n_pairs; ++p) {
            if (rect_off[p + 1] < rect_off[p]) return fail(PRALINE_ERR_ARG, "rect_off is not ascending at pair %lld", (long long)p);
            many_rects = many_rects || rect_off[p + 1] - rect_off[p] > PRALINE_MAX_RECTS;
            max_rects = std::max(max_rects, (int)(rect_off[p + 1] - rect_off[p]));
        }
    praline_plan *pl = new praline_plan();
    pl->arena = arena;
    pl->n_pairs = n_pairs;
    pl->want_paths = want_paths != 0;
    pl->has_rects = rect_off && rect_off[n_pairs] > 0;
    pl->max_rects = max_rects;
    pl->mask_kind = !pl->has_rects ? 0 : (many_rects ? 2 : 1);

    // ---- host scheduling (sched.cpp): tasks, launch order, workgroup descriptors ----
    SchedOptions opt;
    opt.want_paths = pl->want_paths;
    // every plan runs on the split-strip layout (32 pairs per wave, both halves on the same pairs)
    opt.split_layout = true;
    // the strip kernels hold PRALINE_MAX_RECTS rectangles per pair in registers; plans with more per pair (many
    // Waterman-Eggert iterations: rare) read per-row column masks (k_build_zmask): k_dp_quad_tb for plain sequences, the
    // dense-tile instances for every other arena
    const bool quad_ok = want_paths && a.nr16 > 0 && a.nterm16 == 1 && a.onehot && match_mode() == PRALINE_MATCH_FAST &&
                         !(getenv("PRALINE_TB_QUAD") && getenv("PRALINE_TB_QUAD")[0] == '0');
    pl->ppg = a.has_gaps;
    // who forms the match scores (praline_plan::dense_kind): the reference's order on request, for arenas without packed
    // operands (more than 32 active symbols) and for many-rectangle plans on float profiles
    if (match_mode() == PRALINE_MATCH_REFERENCE || a.wide || (many_rects && !quad_ok)) {
        pl->dense_kind = 2;
        // arenas of up to 32 symbols whose rows hold at most 8 nonzeros: k_match_tile (PRALINE_NO_REFTILE=1: the
        // one-cell-per-thread kernels, as for the other arenas - the independent second implementation the tests compare with)
        if (match_mode() == PRALINE_MATCH_REFERENCE && !a.wide && a.nr16 > 0 &&
            !(getenv("PRALINE_NO_REFTILE") && getenv("PRALINE_NO_REFTILE")[0] == '1')) {
            int rc = arena_ensure_reft2(arena);
            if (rc != PRALINE_OK) { delete pl; return rc; }
            if (a.reft2_state == 1) pl->dense_kind = 1;
        }
    } else if (pl->ppg) {
        pl->dense_kind = 3;
    }
    // alignments with paths of plain sequences (exact-mode arenas with their symbol stream): k_dp_quad_tb, 16 pairs per
    // task (PRALINE_TB_QUAD=0: the 32-pair strip kernels, as for every other arena)
    {
        const char *tq = getenv("PRALINE_TB_QUAD");
        const Arena16Dev v16q = a.view16();
        const bool quad_kind = want_paths && pl->dense_kind == 0 && a.nr16 > 0 && a.nterm16 == 1 &&
                               v16q.sym8 != nullptr && match_mode() == PRALINE_MATCH_FAST && !(tq && tq[0] == '0');
        pl->quad = quad_kind;
        // integer scoring within int16 (the exchange matrix alone is checked here, the gap scores by every run): two pairs per
        // lane, k_dp_pk16_tb (PRALINE_TB_PK16=0: never).  Plans with more than PRALINE_MAX_RECTS rectangles per pair keep
        // k_dp_quad_tb, which reads mask words.
        const char *tk16 = getenv("PRALINE_TB_PK16");
        // (plans that fill the chip: a task is one wave and holds twice the pairs of a k_dp_quad_tb task - measured on C2,
        // 32 640 pairs = 1 020 such tasks on 1 024 SIMDs: 0.99 against 1.04 TCUPS; on a C3 slice of 130 944 pairs 1.74 against
        // 1.41.  PRALINE_TB_PK16=1: every plan that qualifies)
        // (smaller plans run it in chain mode, one wave per task and strip)
        pl->pk16 = pl->quad && !many_rects && a.all_onehot && a.s_scale_bits >= 0 && a.s_scale_bits <= 8 && !(tk16 && tk16[0] == '0') &&
                   (2.0 * a.max_len + 36.0) * (double)a.s_absmax * (double)(1 << a.s_scale_bits) < 32000.0;   // (+ 36: the boundary cells of a last strip's padding columns)
        if (pl->pk16) pl->quad = false;
        // k_dp_quad_tb has no chain mode: a task is one wave from the first strip to the last.  Plans that do not fill the chip
        // with such waves (measured: one alignment of 1 400 x 1 400 26 ms against 2 ms in chain mode; 2 016 pairs of ~400 3.8
        // against 1.1 ms; C2-sized plans level) keep the 32-pair strip kernels and their chain mode - unless the plan needs the
        // mask words only k_dp_quad_tb reads (PRALINE_TB_QUAD=1: always)
        if (pl->quad && n_pairs < 32768 && !many_rects && !(tq && tq[0] == '1')) pl->quad = false;
        opt.pk16 = pl->pk16;
        opt.quad16 = pl->quad;
    }
    if (const char *env = getenv("PRALINE_XCD_GROUP")) opt.xcd_group = atoi(env);
    if (const char *env = getenv("PRALINE_NO_W2")) opt.shared_waves = env[0] != '1';
    {   // score plans on one-hot arenas run the lookup instances: three waves per SIMD (168 VGPRs, 4.75 KB of LDS per wave)
        const char *nl = getenv("PRALINE_NO_LOOKUP");
        if (!want_paths && a.onehot && a.nterm16 == 1 && a.nr16 > 0 && match_mode() == PRALINE_MATCH_FAST && !(nl && nl[0] == '1'))
            opt.wave_slots = 3072;
    }
    if (const char *env = getenv("PRALINE_W_SLOTS")) opt.wave_slots = atoll(env);
    if (const char *env = getenv("PRALINE_W_SNAKE")) opt.snake = atoi(env) != 0;
    if (const char *env = getenv("PRALINE_WG_XCD")) opt.wg_xcd = atoi(env) != 0;
    if (const char *env = getenv("PRALINE_WG_BALANCE")) opt.balance = atoi(env) != 0;
    Schedule sch;
    // scores-only plans on float-profile arenas (128-byte operand rows): pipeline workgroups (PRALINE_NO_PIPE=1: the task
    // schedule above, as for every other kind of plan)
    {
        const char *np = getenv("PRALINE_NO_PIPE");
        const Arena16Dev v16 = a.view16();
        // path plans without rectangles get the pipeline schedule BESIDE their task schedule: global runs take it as the
        // forward fill of the two-pass scheme (PRALINE_TB_PIPE=0: never), the other modes keep chain / task mode
        const char *tpp = getenv("PRALINE_TB_PIPE");
        const bool paths_ok = !want_paths || (!pl->has_rects && !(tpp && tpp[0] == '0'));
        if (paths_ok && pl->dense_kind == 0 && a.nr16 > 0 && v16.stage && v16.sym8 == nullptr &&
            praline_pipe_supported(a.nr16, a.nterm16) && match_mode() == PRALINE_MATCH_FAST && !(np && np[0] == '1') && n_pairs > 0) {
            // (a schedule prepared from the same lengths and pair list while the arena was being created: take it)
            const bool prepared = prep != nullptr && prep->n_pairs == n_pairs && prep->lens == a.len &&
                                  memcmp(prep->pairs.data(), pairs, (size_t)n_pairs * 2 * sizeof(int32_t)) == 0;
            if (prepared) pl->pipe = std::move(prep->pipe);
            else pipe_schedule_for(a.len.data(), a.n_seqs, n_pairs, pairs, a.max_len, pl->pipe);
            if (prepared) { prep->pipe = PipeSchedule(); prep->n_pairs = -1; }   // (consumed)
        }
    }
    if (pl->pipe.ok && want_paths) {
        // scratch of the KEEP forward fill: per task (nstrips + 1) kept columns of max_l1 + PRALINE_TB2_PAD rows and
        // nstrips x pipe_keep_blocks row checkpoints; plans beyond the scratch budget keep chain / task mode
        int64_t bnd_e = 0, ck_e = 0;
        for (WaveTask &wt : pl->pipe.tasks) {
            wt.aux_off = bnd_e;
            wt.tb_off = ck_e;
            bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
            const int rows_top = std::max(wt.max_l1 + 12, PRALINE_PIPE_MIN_STEPS);
            ck_e += (int64_t)wt.nstrips * (rows_top / PRALINE_KEEP_BH + 1) * PRALINE_TB2_CKPT_FLOATS;
        }
        if ((size_t)(bnd_e * 16 + ck_e * 4) > tb_budget_bytes()) pl->pipe = PipeSchedule();
        else { pl->pipe_keep_bnd_elems = bnd_e; pl->pipe_keep_ck_floats = ck_e; }
    }
    if (pl->pipe.ok && !want_paths) {
        // the pipeline schedule is all a scores-only run needs: no task schedule, no per-task boundary scratch
        sch.split = opt.split_layout;
        sch.cells = list_cells;
    } else {
        build_schedule(a.len.data(), n_pairs, pairs, opt, sch);
    }
    pt.mark("host scheduling");
    pl->tp = sch.tp;
    pl->split = sch.split;
    pl->tasks.swap(sch.tasks);
    pl->tb_elems.swap(sch.tb_elems);
    pl->aux_elems.swap(sch.aux_elems);
    pl->bnd_elems = sch.bnd_elems;
    pl->wg.swap(sch.wg);
    pl->wg_singles.swap(sch.wg_singles);
    pl->slot_off.swap(sch.slot_off);
    pl->path_cap = sch.path_cap;
    pl->cells = sch.cells;
    const std::vector<int32_t> &lane_one = sch.lane_one, &lane_pair = sch.lane_pair;
    const std::vector<PairLoc> &loc = sch.loc;
    if (pl->dense_kind != 0) {
        pl->h_lane_pair = sch.lane_pair;
        pl->h_pairs.assign(pairs, pairs + 2 * n_pairs);
    }
    if (pl->dense_kind == 1) pl->h_lane_one = sch.lane_one;
    if (!want_paths && !pl->pipe.ok && pl->h_pairs.empty() && (int64_t)pl->tasks.size() <= chain_max_tasks())
        pl->h_pairs.assign(pairs, pairs + 2 * n_pairs);   // (score plans that may run in chain mode: k_semiglobal_end reads the pairs)
    const int64_t bnd = pl->bnd_elems, cap = pl->path_cap;

    hipStream_t st = g_rt.stream;
    int rc = PRALINE_OK;
    if ((rc = pl->d_lane_one.upload(lane_one, st)) || (rc = pl->d_lane_pair.upload(lane_pair, st)) ||
        (rc = pl->d_scores.alloc((size_t)n_pairs)) ||
        (rc = pl->d_bnd.alloc((size_t)bnd * (want_paths ? sizeof(float4) : sizeof(float2))))) {
        delete pl;
        return rc;
    }
    if (pl->dense_kind != 0 && !want_paths) {   // (the per-cell match-score kernels and k_semiglobal_end read them)
        if ((rc = pl->d_pairs.upload(pl->h_pairs, st)) || (rc = pl->d_loc.upload(loc, st))) { delete pl; return rc; }
    }
    if (pl->pipe.ok) {
        if ((rc = pl->d_pipe_items.upload(pl->pipe.items, st)) || (rc = pl->d_pipe_tasks.upload(pl->pipe.tasks, st)) ||
            (rc = pl->d_pipe_set_one.upload(pl->pipe.set_one, st)) || (rc = pl->d_pipe_lane_pair.upload(pl->pipe.lane_pair, st)) ||
            (rc = pl->d_pipe_bnd.alloc((size_t)pl->pipe.bnd_elems))) {
            delete pl;
            return rc;
        }
        for (const PipeItem &pi : pl->pipe.items) pl->pipe_analytic_rows = std::max(pl->pipe_analytic_rows, pi.rsteps + 16);
        if (want_paths) {
            // (k_trace_recompute prefetches up to a block and a few rows beyond a sequence's last row)
            pl->pipe_analytic_rows += PRALINE_KEEP_BH + 16;
            // sequences one per task: the set's, for the lanes that hold a pair
            std::vector<int32_t> l1(pl->pipe.lane_pair.size(), -1);
            for (const PipeItem &pi : pl->pipe.items)
                for (int t = pi.task0; t < pi.task0 + pi.ntasks; ++t)
                    for (int q = 0; q < 32; ++q)
                        if (pl->pipe.lane_pair[(size_t)t * 32 + q] >= 0) l1[(size_t)t * 32 + q] = pl->pipe.set_one[(size_t)pi.set * 32 + q];
            if ((rc = pl->d_pipe_lane_one.upload(l1, st)) || (rc = pl->d_pipe_analytic4.alloc((size_t)pl->pipe_analytic_rows * 32))) {
                delete pl;
                return rc;
            }
            if (hipStreamSynchronize(st) != hipSuccess) { delete pl; return fail(PRALINE_ERR_DEVICE, "plan upload failed"); }   // (l1 goes out of scope)
        }
        if ((rc = pl->d_pipe_analytic.alloc((size_t)pl->pipe_analytic_rows * 32))) { delete pl; return rc; }
        // (rows the kernels never write only feed padding rows; keep them free of NaN bit patterns)
        if (hipMemsetAsync(pl->d_pipe_bnd.p, 0, (size_t)pl->pipe.bnd_elems * sizeof(float2), st) != hipSuccess) {
            delete pl;
            return fail(PRALINE_ERR_DEVICE, "plan upload: memset failed");
        }
    }
    if (want_paths) {
        std::vector<int32_t> pv(pairs, pairs + 2 * n_pairs);
        if ((rc = pl->d_pairs.upload(pv, st)) || (rc = pl->d_loc.upload(loc, st)) ||
            (rc = pl->d_end_cells.alloc((size_t)n_pairs * 4)) || (rc = pl->d_path_rows.alloc((size_t)n_pairs)) ||
            (rc = pl->d_path_start.alloc((size_t)n_pairs)) || (rc = pl->d_paths.alloc((size_t)cap * 2)) ||
            (rc = pl->d_slot_off.upload(pl->slot_off, st))) {
            delete pl;
            return rc;
        }
        if (pl->has_rects) {
            std::vector<int32_t> ro(rect_off, rect_off + n_pairs + 1), rv(rects, rects + (size_t)rect_off[n_pairs] * 4);
            if ((rc = pl->d_rect_off.upload(ro, st)) || (rc = pl->d_rects.upload(rv, st))) { delete pl; return rc; }
        }
        if (pl->mask_kind == 2) {
            std::vector<int64_t> zo((size_t)n_pairs);
            int64_t tot = 0;
            for (int64_t p = 0; p < n_pairs; ++p) {
                zo[(size_t)p] = tot;
                tot += (int64_t)((a.len[pairs[2 * p + 1]] + 31) / 32) * (a.len[pairs[2 * p]] + 1);
            }
            if ((rc = pl->d_zm_off.upload(zo, st)) || (rc = pl->d_zmask.alloc((size_t)tot))) { delete pl; return rc; }
            hipLaunchKernelGGL(k_build_zmask, dim3((unsigned)n_pairs), dim3(256), 0, st, pl->d_pairs.p, a.d_len.p, pl->d_rect_off.p,
                               pl->d_rects.p, pl->d_zm_off.p, pl->d_zmask.p);
        }
    }
    pt.mark("allocations + uploads (async)");
    hipError_t e = hipStreamSynchronize(st);
    pt.mark("stream sync");
    if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
    if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
    if (e != hipSuccess) { delete pl; return fail(PRALINE_ERR_DEVICE, "plan upload: %s", hipGetErrorString(e)); }
    *out = pl;
    return PRALINE_OK;
}

extern "C" int praline_plan_destroy(praline_plan *plan)
{
    if (!plan) return PRALINE_OK;
    // (the device blocks go back to the stream-ordered pool; the explicit waits keep the plan's host-side state from
    // outliving work that still reads it)
    PhaseTimer pt("plan_destroy");
    if (g_rt.ready) { (void)hipStreamSynchronize(g_rt.stream); (void)hipStreamSynchronize(g_rt.stream2); }
    pt.mark("wait for both streams");
    delete plan;
    pt.mark("release");
    return PRALINE_OK;
}

extern "C" int64_t praline_plan_cells(const praline_plan *plan) { return plan ? plan->cells : 0; }
extern "C" int64_t praline_plan_steps(const praline_plan *plan)
{
    if (!plan) return 0;
    if (plan->pipe.ok && !plan->want_paths) return plan->pipe.steps;   // wave steps of the pipeline launch (idle waves of the last rounds included)
    int64_t steps = 0;
    for (const WaveTask &wt : plan->tasks)
        if (wt.max_l1 > 0) steps += (int64_t)wt.nstrips * (wt.max_l1 + 1);
    return steps;
}
extern "C" int64_t praline_plan_tasks(const praline_plan *plan)
{
    if (!plan) return 0;
    if (plan->pipe.ok && !plan->want_paths) return (int64_t)plan->pipe.tasks.size();
    int64_t n = 0;
    for (const WaveTask &wt : plan->tasks) n += wt.max_l1 > 0;
    return n;
}
extern "C" int64_t praline_plan_path_capacity(const praline_plan *plan) { return plan ? plan->path_cap : 0; }
extern "C" void *praline_plan_device_scores(praline_plan *plan) { return plan ? (void *)plan->d_scores.p : nullptr; }

// --------------------------------------------------------------------------------------------
// the scores kernels of the split-strip layout: k_dp_split16 on the f16 hi/lo operands, or - PRALINE_MM=f32 - k_dp_split on the
// fp32 MFMA chain (one translation unit per MFMA step count, dp_split_instance.hip); see dp_launch.hip.h
// --------------------------------------------------------------------------------------------
static int launch_scores(int nstep, const LaunchArgs &la, bool local)
{
    if (la.a16 != nullptr) {
        const int rc = praline_launch_split16(la, *la.a16, la.nr16, la.nterm16, local);
        if (rc != PRALINE_OK) return fail(rc, "no k_dp_split16 instance for nr=%d nterm=%d", la.nr16, la.nterm16);
        return PRALINE_OK;
    }
    switch (nstep) {
        case 2: return praline_launch_split_2(la, local);
        case 8: return praline_launch_split_8(la, local);
        case 10: return praline_launch_split_10(la, local);
        case 12: return praline_launch_split_12(la, local);
        case 14: return praline_launch_split_14(la, local);
        case 16: return praline_launch_split_16(la, local);
    }
    return fail(PRALINE_ERR_UNSUPPORTED, "no k_dp_split instance for nstep=%d", nstep);
}

// end cells of the semiglobal modes + device traceback for the tasks [t0, t1) of a path plan (after their fill)
static int launch_traceback(praline_plan &pl, const LaunchArgs &la, size_t t0, size_t t1, int mode)
{
    hipStream_t st = la.stream;
    // k_traceback runs over all pairs and skips those whose task is outside [t0, t1)
    const int threads = 64;
    const int64_t blocks = (pl.n_pairs + threads - 1) / threads;
    if (mode >= PRALINE_MODE_SEMIGLOBAL_BOTH) {   // end cells of the semiglobal modes, scanned per task
        const int64_t lanes = (int64_t)(t1 - t0) * (pl.quad ? 16 : (pl.split ? 32 : 64));   // (k_dp_pk16_tb writes the strip kernels' end-cell scratch)
        hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar,
                           pl.d_tasks.p, pl.d_lane_one.p, pl.d_lane_pair.p, pl.d_pairs.p, la.aux,
                           pl.d_end_cells.p, la.scores, la.rp, (int32_t)t0, (int32_t)t1, pl.quad ? 2 : (pl.split ? 1 : 0));
    }
    hipLaunchKernelGGL(k_traceback, dim3((unsigned)blocks), dim3(threads), 0, st, la.ar, pl.d_tasks.p,
                       pl.d_loc.p, pl.d_pairs.p, (const uint4 *)la.tb, la.aux, la.rl, pl.d_end_cells.p,
                       la.scores, pl.d_slot_off.p, pl.d_paths.p, pl.d_path_start.p, pl.d_path_rows.p, pl.n_pairs,
                       la.rp, (int32_t)t0, (int32_t)t1, pl.run_pk16 ? 3 : (pl.quad ? 2 : (pl.split ? 1 : 0)));
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// dense match-score tiles per launch chunk (bytes)
static size_t reftile_budget_bytes()
{
    if (const char *env = getenv("PRALINE_REFTILE_BUDGET_MB")) return (size_t)atoll(env) << 20;
    return (size_t)32 << 30;
}

// Plans whose DP reads its match scores from dense tiles (praline_plan::dense_kind: the reference's summation order, arenas
// without packed operands, many-rectangle plans on float profiles, per-position gap scores).  Per chunk of tasks one of the
// producers writes the tiles (4 bytes per cell and padding):
//   1  k_match_tile (dp_reftile.hip.h);  2  k_match_reft / k_match_ref, one cell per thread;  3  k_scores_tile_batch (fp32 MFMA)
// and a dense-tile DP instance consumes them: k_dp_split16<1, 1, LOCAL, 4> for scores, k_dp_split16_tb<1, 3, LOCAL, MASK, .., 4,
// PPG, NOFLAGS> (+ k_traceback) for alignments with paths, with per-position gap scores (ppg) and for the scores of tasks that
// are swept in several launches.  A task whose tile exceeds the chunk budget (sequences beyond ~16 000 positions) runs alone,
// a range of strips per launch: the tile then holds that range, the boundary column and the local maximum carry over.
// One stream, one tile set: a k_match_tile workgroup fills its CU (registers and LDS), so a second stream only time-slices the
// chip (measured on C2: four chunks alternating between two streams 31 ms, one chunk 19 ms).
static int plan_run_dense(praline_plan &pl, LaunchArgs la, Arena16Dev a16, int mode, bool local)
{
    praline_arena &a = *pl.arena;
    int producer = pl.dense_kind;
    if (producer == 1) {
        if (a.reft2_state == 0) RC(arena_ensure_reft2(&a));   // (the arena changed since the plan was made)
        if (a.reft2_state != 1) producer = 2;                  // (... and no longer qualifies for k_match_tile)
    }
    hipStream_t st = g_rt.stream;
    const size_t nt = pl.tasks.size();
    const bool semiglobal = mode >= 2;
    const bool ppg = pl.run_ppg;
    const size_t m_budget = reftile_budget_bytes(), tb_budget = tb_budget_bytes();
    auto strip_floats = [&](const WaveTask &wt) { return (int64_t)(wt.max_l1 + PRALINE_DENSE_PAD) * 1024; };
    // range: the chunk is ONE task and sweeps its strips [strip_lo, strip_lo + strip_cnt); last: the task's end cells are final
    struct Chunk { size_t t0, t1, b0, b1, c0, c1; int64_t m_e, tb_e, aux_e; int strip_lo, strip_cnt, max_l1, strips; bool range, last; };
    std::vector<Chunk> chunks;
    std::vector<int64_t> dense_off(nt, 0);
    std::vector<RefTileBlock> blocks;
    std::vector<int32_t> grp;   // group records (dp_reftile.h)
    std::vector<int32_t> chunk_pairs;   // the pairs of every chunk, chunk after chunk (producers 2 and 3)
    bool any_range = false;
    for (size_t t = 0; t < nt; ++t) any_range = any_range || (size_t)(pl.tasks[t].nstrips * strip_floats(pl.tasks[t])) * 4 > m_budget;
    // plans without paths: the scores kernel, unless a task runs in strip ranges or with per-position gap scores
    const bool fill_only = !pl.want_paths && (ppg || any_range);
    // k_match_tile's workgroups of a chunk: the tasks are grouped by their 32 sequences one (the schedule gives every
    // sequence two of a set of ones its own task), a group's sequences two are laid end to end and cut into 128 columns
    auto add_blocks = [&](size_t t0, size_t t1) {
        std::unordered_map<std::string, size_t> index;
        std::vector<std::vector<int32_t>> members;
        for (size_t t = t0; t < t1; ++t) {
            const WaveTask &wt = pl.tasks[t];
            if (wt.max_l1 <= 0 || wt.two[0] < 0 || a.len[(size_t)wt.two[0]] <= 0) continue;
            const std::string key(reinterpret_cast<const char *>(pl.h_lane_one.data() + t * 32), 32 * sizeof(int32_t));
            auto it = index.find(key);
            if (it == index.end()) { it = index.emplace(key, members.size()).first; members.emplace_back(); }
            members[it->second].push_back((int32_t)(t - t0));
        }
        for (const std::vector<int32_t> &mem : members) {
            const int32_t base = (int32_t)grp.size();
            int32_t cum = 0;
            // (whole strips: the columns between the end of a sequence and the end of its last strip receive zeros - local
            // alignments must not see stale positive scores there)
            for (int32_t tr : mem) { grp.push_back(cum); cum += (a.len[(size_t)pl.tasks[t0 + (size_t)tr].two[0]] + 31) / 32 * 16; }
            grp.push_back(cum);
            grp.insert(grp.end(), mem.begin(), mem.end());
            for (int32_t c = 0; c * 64 < cum; ++c) blocks.push_back({base, (int32_t)mem.size(), c, 0});
        }
    };
    auto add_pairs = [&](size_t t0, size_t t1, int &max_l1, int &strips) {
        for (size_t t = t0; t < t1; ++t) {
            bool any = false;
            for (int l = 0; l < 32; ++l) {
                const int32_t p = pl.h_lane_pair[t * 32 + l];
                if (p < 0) continue;
                chunk_pairs.push_back(p);
                any = true;
            }
            if (any) { max_l1 = std::max(max_l1, (int)pl.tasks[t].max_l1); strips = std::max(strips, (int)pl.tasks[t].nstrips); }
        }
    };
    int64_t bnd4_e = 0;   // fill_only: the float4 boundary columns of k_dp_split16_tb (the plan's own are float2)
    for (size_t t0 = 0; t0 < nt;) {
        const WaveTask &w0 = pl.tasks[t0];
        const int64_t sf = strip_floats(w0);
        if ((size_t)(w0.nstrips * sf) * 4 > m_budget) {
            // one task, strip ranges
            const int per = (int)std::max<int64_t>(1, (int64_t)(m_budget / 4) / sf);
            const size_t c0 = chunk_pairs.size();
            int ml = 0, strips = 0;
            add_pairs(t0, t0 + 1, ml, strips);
            dense_off[t0] = 0;
            pl.tasks[t0].tb_off = 0;
            pl.tasks[t0].aux_off = 0;
            if (fill_only) { pl.tasks[t0].bnd_off = bnd4_e; bnd4_e += (int64_t)(w0.max_l1 + 24) * 32; }
            for (int lo = 0; lo < w0.nstrips; lo += per) {
                const int cnt = std::min(per, w0.nstrips - lo);
                chunks.push_back({t0, t0 + 1, blocks.size(), blocks.size(), c0, chunk_pairs.size(), (int64_t)cnt * sf,
                                  pl.want_paths ? pl.tb_elems[t0] : 0, semiglobal ? pl.aux_elems[t0] : 0, lo, cnt, ml, cnt, true,
                                  lo + cnt >= w0.nstrips});
            }
            ++t0;
            continue;
        }
        size_t t1 = t0;
        const size_t b0 = blocks.size(), c0 = chunk_pairs.size();
        int64_t m_e = 0, tb_e = 0, aux_e = 0;
        while (t1 < nt) {
            const WaveTask &wt = pl.tasks[t1];
            const int64_t m_add = wt.nstrips * strip_floats(wt), tb_add = pl.want_paths ? pl.tb_elems[t1] : 0;
            if ((size_t)m_add * 4 > m_budget) break;   // (the next task runs alone)
            if (t1 > t0 && ((size_t)(m_e + m_add) * 4 > m_budget || (size_t)(tb_e + tb_add) * 8 > tb_budget)) break;
            dense_off[t1] = m_e;
            pl.tasks[t1].tb_off = tb_e;
            pl.tasks[t1].aux_off = aux_e;
            if (fill_only) { pl.tasks[t1].bnd_off = bnd4_e; bnd4_e += (int64_t)(wt.max_l1 + 24) * 32; }
            m_e += m_add;
            tb_e += tb_add;
            aux_e += ((pl.want_paths || fill_only) && semiglobal) ? pl.aux_elems[t1] : 0;
            ++t1;
        }
        int ml = 0, strips = 0;
        if (producer == 1) add_blocks(t0, t1);
        else add_pairs(t0, t1, ml, strips);
        chunks.push_back({t0, t1, b0, blocks.size(), c0, chunk_pairs.size(), m_e, tb_e, aux_e, 0, 0x3fffffff, ml, strips, false, true});
        t0 = t1;
    }
    if (producer != 3 && (producer == 2 || any_range)) { RC(arena_ensure_ref(&a)); }
    {
        size_t need_m = 1, need_tb = 0, need_ax = 1;
        for (const Chunk &ch : chunks) {
            need_m = std::max(need_m, (size_t)ch.m_e);
            need_tb = std::max(need_tb, (size_t)ch.tb_e * 8);
            need_ax = std::max(need_ax, (size_t)ch.aux_e);
        }
        // (every buffer is sized once, before the loop: see the chunk loops of praline_plan_run)
        if (pl.d_dense.n < need_m) RC(pl.d_dense.alloc(need_m));
        if (pl.want_paths && pl.d_tb.n < need_tb) RC(pl.d_tb.alloc(need_tb));
        if ((pl.want_paths || fill_only) && pl.d_aux.n < need_ax) RC(pl.d_aux.alloc(need_ax));
    }
    if (fill_only) {
        if (pl.d_bnd_chain.n < (size_t)bnd4_e * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)bnd4_e * sizeof(float4)));
        if (pl.d_end_cells.n < (size_t)pl.n_pairs * 4) RC(pl.d_end_cells.alloc((size_t)pl.n_pairs * 4));
    }
    if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
    if (pl.d_dense_off.n < nt) RC(pl.d_dense_off.alloc(nt));
    if (pl.d_tile_blocks.n < blocks.size()) RC(pl.d_tile_blocks.alloc(std::max<size_t>(blocks.size(), 1)));
    if (pl.d_tile_grp.n < grp.size()) RC(pl.d_tile_grp.alloc(std::max<size_t>(grp.size(), 1)));
    if (pl.d_chunk_pairs.n < chunk_pairs.size()) RC(pl.d_chunk_pairs.alloc(std::max<size_t>(chunk_pairs.size(), 1)));
    HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(pl.d_dense_off.p, dense_off.data(), nt * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (!grp.empty()) HIPCHK(hipMemcpyAsync(pl.d_tile_grp.p, grp.data(), grp.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!blocks.empty())
        HIPCHK(hipMemcpyAsync(pl.d_tile_blocks.p, blocks.data(), blocks.size() * sizeof(RefTileBlock), hipMemcpyHostToDevice, st));
    if (!chunk_pairs.empty())
        HIPCHK(hipMemcpyAsync(pl.d_chunk_pairs.p, chunk_pairs.data(), chunk_pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // (the host lists go out of scope)
    {
        char kn[200];
        const char *lb = local ? "true" : "false";
        if (pl.want_paths || fill_only)
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<1, 3, %s, %s, false, false, 4, %s, %s>", lb, pl.has_rects ? "true" : "false",
                     ppg ? "true" : "false", fill_only ? "true" : "false");
        else snprintf(kn, sizeof(kn), "k_dp_split16<1, 1, %s, 4, 1, false>", lb);   // (four-wave workgroups: refined below)
        pl.last_kernel = kn;
    }
    for (const Chunk &ch : chunks) {
        // ---- the tiles ----
        const int chunk_producer = (producer == 1 && ch.range) ? 2 : producer;
        if (chunk_producer == 1) {
            RefTileArgs g;
            g.raw = a.d_raw.p;
            g.A = a.A;
            g.T2 = a.d_reft2.p;
            g.PR = a.pair_rows;
            g.row_off_raw = a.d_row_off_raw.p;
            g.len = a.d_len.p;
            g.pr_off = a.d_pr_off.p;
            g.set_lo = a.d_set_lo.p;
            g.n_sets = (int)a.set_lo.size() - 1;
            g.tasks = pl.d_tasks.p + ch.t0;
            g.lane_one = pl.d_lane_one.p + ch.t0 * 32;
            g.dense_off = pl.d_dense_off.p + ch.t0;
            g.m = pl.d_dense.p;
            g.blocks = pl.d_tile_blocks.p + ch.b0;
            g.grp = pl.d_tile_grp.p;
            g.waves = 0;
            int rc = praline_launch_match_tile(g, a.ref_tb, (unsigned)(ch.b1 - ch.b0), st);
            if (rc != PRALINE_OK) return fail(rc, "k_match_tile launch failed (A=%d, tb=%d)", a.A, a.ref_tb);
        } else if (ch.c1 > ch.c0) {
            TileOut to;
            to.loc = pl.d_loc.p;
            to.tasks = pl.d_tasks.p;
            to.dense_off = pl.d_dense_off.p;
            to.strip_lo = ch.strip_lo;
            to.strip_cnt = ch.strip_cnt;
            if (chunk_producer == 2) {
                RC(launch_match_ref(&a, pl.d_pairs.p, pl.d_chunk_pairs.p + ch.c0, ch.c1 - ch.c0, ch.max_l1, nullptr, pl.d_dense.p, to));
            } else {
                const int tiles_x = ch.strips, tiles_y = (ch.max_l1 + 31) / 32;
                if (tiles_x > 0 && tiles_y > 0) {
                    hipLaunchKernelGGL(k_scores_tile_batch, dim3((unsigned)(ch.c1 - ch.c0), (unsigned)tiles_y), dim3(64), 0, st,
                                       a.view(), pl.d_pairs.p, pl.d_chunk_pairs.p + ch.c0, nullptr, a.nstep, tiles_x, pl.d_dense.p, to);
                    HIPCHK(hipGetLastError());
                }
            }
        }
        // ---- the fill ----
        a16.dense = pl.d_dense.p;
        a16.dense_off = pl.d_dense_off.p + ch.t0;
        la.stream = st;
        la.tasks = pl.d_tasks.p + ch.t0;
        la.lane_one = pl.d_lane_one.p + ch.t0 * 32;
        la.lane_pair = pl.d_lane_pair.p + ch.t0 * 32;
        la.n_tasks = (unsigned)(ch.t1 - ch.t0);
        la.bnd = fill_only ? (void *)pl.d_bnd_chain.p : (void *)pl.d_bnd.p;
        int rc;
        if (!pl.want_paths && !fill_only) {
            la.tb = nullptr;
            la.aux = nullptr;
            la.wg = nullptr;
            la.n_wg = 0;
            if (chunks.size() == 1 && !pl.wg.empty() && !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1')) {
                // (the shared-wave descriptors index the plan's task list: one chunk only)
                if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
                la.wg = pl.d_wg.p;
                la.n_wg = (unsigned)pl.wg.size();
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_split16<1, 1, %s, 4, 4, false>", local ? "true" : "false");
                pl.last_kernel = kn;
            }
            rc = praline_launch_dense(la, a16, local);
            if (rc != PRALINE_OK) return fail(rc, "dense-tile scores launch failed");
            continue;
        }
        la.tb = (uint4 *)pl.d_tb.p;
        la.aux = pl.d_aux.p;
        la.end_cells = pl.d_end_cells.p;
        rc = praline_launch_dense_tb(la, a16, local, pl.has_rects, ppg, fill_only, ch.strip_lo, ch.strip_cnt);
        if (rc != PRALINE_OK) return fail(rc, "dense-tile fill launch failed");
        if (!ch.last) continue;
        if (pl.want_paths) {
            RC(launch_traceback(pl, la, ch.t0, ch.t1, mode));
        } else if (semiglobal) {
            const int64_t lanes = (int64_t)(ch.t1 - ch.t0) * 32;
            hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar, pl.d_tasks.p, pl.d_lane_one.p,
                               pl.d_lane_pair.p, pl.d_pairs.p, la.aux, pl.d_end_cells.p, la.scores, la.rp, (int32_t)ch.t0, (int32_t)ch.t1, 1);
            HIPCHK(hipGetLastError());
        }
    }
    return PRALINE_OK;
}

// Score plans of a FEW LONG sequences: the score kernels put at most four waves on a task (its strips form a chain), so a
// plan of a handful of tasks leaves the chip idle - a single 30 000 x 30 000 alignment took 2.7 s scores-only and 54 ms
// with paths.  Such plans run the chain-mode fill (one wave per task and strip, pipelined across workgroups) in its
// flag-free form (k_dp_split16_tb<..., CHAIN, TWOPASS>): same scores bit for bit.  The choice is an estimate from the
// schedule, fitted to scripts/exp_scores_chain.py (all pairs of N x ~mu residues; chain wins from single alignments up to
// about N = 64 x 400 and for every batch of long sequences that the pipeline workgroups do not take):
//   shared waves: the longest task's ceil(strips / 4) x rows steps at 0.55 us (one-hot lookup instances: 0.45), in rounds
//                 of 2048 waves;
//   chain:        the larger of the longest task's rows + 24 x strips steps at 0.7 us and an even share of all strip-rows
//                 over 2048 waves at 2.0 us per step (the waves of a chain wait for each other).
static bool plan_scores_chain_wanted(const praline_plan &pl)
{
    if (const char *env = getenv("PRALINE_SCORES_CHAIN")) return atoi(env) != 0;
    const size_t nt = pl.tasks.size();
    if (nt == 0 || (int64_t)nt > chain_max_tasks()) return false;
    double shared = 0.0, crit = 0.0, work = 0.0, bnd_bytes = 0.0;
    int max_strips = 0;
    for (const WaveTask &wt : pl.tasks) {
        const double rows = wt.max_l1 + 1.0;
        shared = std::max(shared, std::ceil(wt.nstrips / 4.0) * rows);
        crit = std::max(crit, rows + 24.0 * wt.nstrips);
        work += wt.nstrips * rows;
        bnd_bytes += (wt.nstrips + 1.0) * (wt.max_l1 + 24.0) * 512.0;
        max_strips = std::max(max_strips, (int)wt.nstrips);
    }
    if (max_strips < 2 || bnd_bytes > 64.0 * 1073741824.0) return false;
    const bool lookup = pl.arena->onehot && pl.arena->nterm16 == 1;
    const double t_shared = shared * std::ceil(4.0 * nt / 2048.0) * (lookup ? 0.45 : 0.55);   // us
    const double t_chain = std::max(crit * 0.7, work / 2048.0 * 2.0);
    return t_chain < 0.9 * t_shared;
}

static int plan_run_scores_chain(praline_plan &pl, LaunchArgs la, const Arena16Dev &a16, int mode, bool local)
{
    praline_arena &a = *pl.arena;
    hipStream_t st = g_rt.stream;
    const size_t nt = pl.tasks.size();
    const bool semiglobal = mode >= 2;
    std::vector<WaveTask> ct(pl.tasks.begin(), pl.tasks.end());
    int64_t bnd_e = 0, aux_e = 0;
    int max_strips = 0, rows = 0;
    for (size_t t = 0; t < nt; ++t) {
        WaveTask &wt = ct[t];
        wt.bnd_off = bnd_e;
        wt.tb_off = 0;
        wt.aux_off = aux_e;
        bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + 24) * 32;   // float4 elements, [strip boundary][row][32]
        aux_e += semiglobal ? pl.aux_elems[t] : 0;
        max_strips = std::max(max_strips, (int)wt.nstrips);
        rows = std::max(rows, (int)wt.max_l1);
    }
    const size_t n_flags = nt * (size_t)(max_strips + 1);
    if (pl.d_bnd_chain.n < (size_t)bnd_e * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)bnd_e * sizeof(float4)));
    if (pl.d_chain_flags.n < n_flags) RC(pl.d_chain_flags.alloc(n_flags));
    if (local && pl.d_chain_cand.n < n_flags * 32) RC(pl.d_chain_cand.alloc(n_flags * 32));
    if (pl.d_aux.n < (size_t)std::max<int64_t>(aux_e, 1)) RC(pl.d_aux.alloc((size_t)std::max<int64_t>(aux_e, 1)));
    if (pl.d_end_cells.n < (size_t)pl.n_pairs * 4) RC(pl.d_end_cells.alloc((size_t)pl.n_pairs * 4));
    if (pl.d_tasks_chain.n < nt) RC(pl.d_tasks_chain.alloc(nt));
    if (semiglobal && !pl.d_pairs.p) {
        if (pl.h_pairs.size() != (size_t)pl.n_pairs * 2) return fail(PRALINE_ERR_UNSUPPORTED, "score plan without its pair list");
        RC(pl.d_pairs.upload(pl.h_pairs, st));
    }
    HIPCHK(hipMemsetAsync(pl.d_chain_flags.p, 0, n_flags * sizeof(int), st));
    HIPCHK(hipMemcpyAsync(pl.d_tasks_chain.p, ct.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // ct goes out of scope
    la.tasks = pl.d_tasks_chain.p;
    la.n_tasks = (unsigned)nt;
    la.bnd = pl.d_bnd_chain.p;
    la.tb = nullptr;
    la.aux = pl.d_aux.p;
    la.end_cells = pl.d_end_cells.p;
    la.stream = st;
    int every = nt >= 512 ? 96 : (nt >= 64 ? 24 : 6);
    every = std::min(every, std::max(6, rows / 4));
    if (const char *env = getenv("PRALINE_CHAIN_EVERY")) every = std::max(6, atoi(env));
    int rc = praline_launch_scores_chain(la, a16, a.nr16, a.nterm16, local, max_strips, pl.d_chain_flags.p, pl.d_chain_cand.p, every);
    if (rc != PRALINE_OK) return fail(rc, "no scores-only chain instance for nr=%d nterm=%d", a.nr16, a.nterm16);
    if (local) {
        const int64_t lanes = (int64_t)nt * 32;
        hipLaunchKernelGGL(k_chain_local_end, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, la.tasks, la.lane_pair,
                           pl.d_chain_cand.p, (int)nt, max_strips + 1, pl.d_end_cells.p, la.scores);
    }
    if (semiglobal) {
        const int64_t lanes = (int64_t)nt * 32;
        hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar, la.tasks, la.lane_one,
                           la.lane_pair, pl.d_pairs.p, la.aux, pl.d_end_cells.p, la.scores, la.rp, (int32_t)0, (int32_t)nt, 1);
    }
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

extern "C" int praline_plan_run(praline_plan *plan, int mode, float gap_open, float gap_extend, void *d_scores)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!(gap_open <= 0.0f) || !(gap_extend <= 0.0f))
        return fail(PRALINE_ERR_UNSUPPORTED, "batched kernels need gap scores <= 0 (got %g, %g)", gap_open, gap_extend);
    RC(ensure_runtime(-1));
    praline_plan &pl = *plan;
    if (pl.n_pairs == 0) return PRALINE_OK;
    const praline_arena &a = *pl.arena;
    LaunchArgs la;
    la.wg = nullptr;
    la.n_wg = 0;
    la.ar = a.view();
    la.lane_one = pl.d_lane_one.p;
    la.lane_pair = pl.d_lane_pair.p;
    la.bnd = pl.d_bnd.p;
    la.rl.rect_off = pl.has_rects ? pl.d_rect_off.p : nullptr;
    la.rl.rects = pl.has_rects ? pl.d_rects.p : nullptr;
    la.rl.zmask = pl.mask_kind == 2 ? pl.d_zmask.p : nullptr;
    la.rl.zm_off = pl.mask_kind == 2 ? pl.d_zm_off.p : nullptr;
    la.scores = d_scores ? (float *)d_scores : pl.d_scores.p;
    la.end_cells = pl.d_end_cells.p;
    la.rp.mode = mode;
    la.rp.go1 = la.rp.go2 = gap_open;
    la.rp.ge1 = la.rp.ge2 = gap_extend;
    la.stream = g_rt.stream;
    la.split = pl.split ? 1 : 0;
    // match scores on the matrix pipe (f16 hi/lo split) unless PRALINE_MM=f32 asks for the fp32 MFMA chain
    Arena16Dev a16 = a.view16();
    la.a16 = nullptr;
    la.nr16 = a.nr16;
    la.nterm16 = a.nterm16;
    if (pl.split && a.nr16 > 0 && match_mode() != PRALINE_MATCH_F32) la.a16 = &a16;
    const bool local = mode == PRALINE_MODE_LOCAL;
    pl.last_mode = mode;
    pl.last_scores = la.scores;
    hipStream_t st = g_rt.stream;

    {
        char kn[160];
        const char *lb = local ? "true" : "false";
        if (pl.want_paths && pl.quad) snprintf(kn, sizeof(kn), "k_dp_quad_tb<%d, ...>", a.nr16);
        else if (pl.want_paths && pl.pk16) snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, ...>", a.nr16);   // (refined below: a run may take the strip kernels)
        else if (pl.want_paths) snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, ...>", a.nr16);   // refined below (nterm, chain)
        else if (la.a16 == nullptr) snprintf(kn, sizeof(kn), "k_dp_split<%d, %s>", a.nstep, lb);
        else {
            const char *nl = getenv("PRALINE_NO_LOOKUP");
            const bool lookup = !(nl && nl[0] == '1');
            const bool shared = !pl.wg.empty() && a16.stage;
            const bool table = a.nterm16 == 1 && a16.sym8 != nullptr && (!shared || lookup);   // one-hot path (lookup or operand table)
            const bool four = (shared && (!table || lookup)) || (a16.stage && !table && !pl.wg_singles.empty() &&
                              !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1'));
            snprintf(kn, sizeof(kn), "k_dp_split16<%d, %d, %s, %d, %d, false>", a.nr16, a.nterm16, lb,
                     table ? (lookup ? 3 : 1) : (a16.stage ? 2 : 0), four ? 4 : 1);
        }
        pl.last_kernel = kn;
    }
    if (pl.run_ppg) la.rp.gaps = a.d_gaps.p;
    if (pl.dense_kind != 0) {   // (names the kernel it launches)
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(plan_run_dense(pl, la, a16, mode, local));
        HIPCHK(hipEventRecord(pl.ev1, st));
        return PRALINE_OK;
    }
    if (!pl.want_paths && pl.pipe.ok) {   // (the match-score mode was read when the plan was created)
        char kn[160];
        snprintf(kn, sizeof(kn), "k_dp_pipe<%d, %d, %s, %s, false>", a.nr16, a.nterm16, local ? "true" : "false", mode >= 2 ? "true" : "false");
        pl.last_kernel = kn;
        PipeLaunch pp;
        pp.items = pl.d_pipe_items.p;
        pp.n_items = (unsigned)pl.pipe.items.size();
        pp.tasks = pl.d_pipe_tasks.p;
        pp.set_one = pl.d_pipe_set_one.p;
        pp.lane_pair = pl.d_pipe_lane_pair.p;
        pp.bnd = pl.d_pipe_bnd.p;
        pp.analytic = pl.d_pipe_analytic.p;
        pp.analytic_rows = pl.pipe_analytic_rows;
        pp.analytic_valid = pl.pipe_analytic_mode == mode && pl.pipe_analytic_go == la.rp.go1 && pl.pipe_analytic_ge == la.rp.ge1;
        pl.pipe_analytic_mode = mode; pl.pipe_analytic_go = la.rp.go1; pl.pipe_analytic_ge = la.rp.ge1;
        pp.scores = la.scores;
        pp.rp = la.rp;
        pp.stream = st;
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(praline_launch_pipe(pp, a16, a.nr16, a.nterm16));
        HIPCHK(hipEventRecord(pl.ev1, st));
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    if (!pl.want_paths && pl.split && la.a16 != nullptr && !pl.has_rects) {
        if (pl.scores_chain < 0) pl.scores_chain = plan_scores_chain_wanted(pl) ? 1 : 0;
        if (pl.scores_chain == 1) {
            char kn[160];
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, %d, %s, false, true, true, 0>", a.nr16, a.nterm16, local ? "true" : "false");
            pl.last_kernel = kn;
            HIPCHK(hipEventRecord(pl.ev0, st));
            RC(plan_run_scores_chain(pl, la, a16, mode, local));
            HIPCHK(hipEventRecord(pl.ev1, st));
            return PRALINE_OK;
        }
    }
    if (!pl.want_paths) {
        if (!pl.d_tasks.p) { RC(pl.d_tasks.upload(pl.tasks, st)); }
        la.tasks = pl.d_tasks.p;
        la.tb = nullptr;
        la.aux = nullptr;
        la.n_tasks = (unsigned)pl.tasks.size();
        if (!pl.wg.empty() && la.a16 != nullptr && a16.stage) {
            // small batch: shared-wave workgroups on the staged stream - also for one-hot arenas (measured,
            // 1024 tasks: 3520 vs 3099 GCUPS; the one-hot table path wins, by 4 %, only on a full chip)
            if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
            la.wg = pl.d_wg.p;
            la.n_wg = (unsigned)pl.wg.size();
            // one-hot arenas keep their symbol stream: the shared waves look their match scores up (BSRC = 3);
            // PRALINE_NO_LOOKUP=1: the staged operand stream as for float profiles
            const char *nl = getenv("PRALINE_NO_LOOKUP");
            if (a.nterm16 != 1 || (nl && nl[0] == '1')) a16.sym8 = nullptr;
        }
        // large batches: four independent tasks per workgroup (wg_singles) for arenas without the one-hot table
        // (measured, float profiles: +0..6 %); one-hot arenas are faster on the table path in every mode (C4 rank
        // share: 4.4 TCUPS global, 4.0 local against 3.1 on this list)
        else if (a16.sym8 == nullptr && !pl.wg_singles.empty() && la.a16 != nullptr && a16.stage &&
                 !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1')) {
            if (!pl.d_wg_singles.p) { RC(pl.d_wg_singles.upload(pl.wg_singles, st)); }
            la.wg = pl.d_wg_singles.p;
            la.n_wg = (unsigned)pl.wg_singles.size();
        }
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(launch_scores(a.nstep, la, local));
        HIPCHK(hipEventRecord(pl.ev1, st));
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }

    // ---- with paths: chunk the tasks so the packed traceback fits the scratch budget ----
    // k_dp_split16_tb's single-term instances take the tie flags from the predecessor states instead of the candidate
    // sums (dp_split16_tb.hip.h, INTS): valid when every DP value is a multiple of 2^-k that float32 holds exactly -
    // one-hot profiles, S and gap scores integral after scaling by 2^k, (L1 + L2) * max |score| * 2^k < 2^24.
    // Other exact-mode arenas run the three-term instances (their lo pieces are zero: same match scores).
    int tb_nterm = a.nterm16;
    pl.run_pk16 = false;
    float pk16_scale = 1.0f;
    if (a.nterm16 == 1) {
        double big_scaled = 1e30;
        int k_bits = 0;
        bool ints = a.all_onehot && a.s_scale_bits >= 0 && !(getenv("PRALINE_NO_INTS") && getenv("PRALINE_NO_INTS")[0] == '1');
        if (ints) {
            int k = a.s_scale_bits;
            for (; k <= 8; ++k) {
                const float sc = (float)(1 << k), g1 = gap_open * sc, g2 = gap_extend * sc;
                if (std::isfinite(g1) && std::isfinite(g2) && g1 == std::nearbyint(g1) && g2 == std::nearbyint(g2)) break;
            }
            const double big = std::max((double)a.s_absmax, std::max(std::fabs((double)gap_open), std::fabs((double)gap_extend)));
            ints = k <= 8 && (2.0 * a.max_len + 4.0) * big * (double)(1 << std::min(k, 8)) < 16777216.0;
            k_bits = k;
            big_scaled = big * (double)(1 << std::min(k, 8));
        }
        tb_nterm = ints ? 1 : 3;
        // two pairs per lane in int16 when every DP value of this run fits (dp_pk16.hip.h)
        pl.run_pk16 = pl.want_paths && pl.pk16 && ints && (2.0 * a.max_len + 36.0) * big_scaled < 32000.0;
        pk16_scale = (float)(1 << std::min(std::max(k_bits, 0), 8));
    }
    // rectangle slots per pair the packed kernel holds in registers: the lists' longest, or the slots filled so far
    // (praline_plan_mask_path_bounds), rounded up to an instance (1, 2, PRALINE_MAX_RECTS)
    int pk16_slots = !pl.has_rects ? 0 : (pl.slot_rects >= 0 ? pl.slot_rects : pl.max_rects);
    pk16_slots = pk16_slots <= 0 ? (pl.has_rects ? 1 : 0) : (pk16_slots <= 2 ? pk16_slots : PRALINE_MAX_RECTS);
    if (pl.want_paths && pl.pk16) {
        char kn[160];
        if (pl.run_pk16) snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, %s, %d, false>", a.nr16, local ? "true" : "false", pk16_slots);   // (chain mode: below)
        else snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, ...>", a.nr16);   // (gap scores off the int16 grid: the strip kernels)
        pl.last_kernel = kn;
    }
    size_t budget = tb_budget_bytes();
    const bool semiglobal = mode >= 2;
    const size_t tb_elem_bytes = pl.split ? sizeof(uint2) : sizeof(uint4);
    const int lanes_per_task = pl.quad ? 16 : (pl.split ? 32 : 64);
    size_t t0 = 0;
    const size_t nt = pl.tasks.size();
    if (!getenv("PRALINE_TB_BUDGET_MB") && nt > 0) {
        // Plans of LONG sequences (more than 8 MiB of packed traceback per task: ~700 x 700 and up) run in chain mode
        // chunk by chunk; a chunk of a few tasks leaves the chip half empty, so they get up to 48 GiB (of 288) instead
        // of 8.  Measured, all pairs with paths: 256 x ~1000 aa 38 -> 32 ms, 128 x ~2500 96 -> 70 ms, 96 x ~5000
        // 371 -> 174 ms.  (Plans of many small tasks keep 8 GiB: within 3 % of 24 GiB on C3, see tb_budget_bytes.)
        int64_t all = 0;
        for (size_t t = 0; t < nt; ++t) all += pl.tb_elems[t] * (int64_t)tb_elem_bytes;
        if ((size_t)all > budget && all / (int64_t)nt > ((int64_t)8 << 20))
            budget = (size_t)std::min<int64_t>((int64_t)48 << 30, 2 * all);   // (twice: chunked plans cut at half the budget)
    }
    HIPCHK(hipEventRecord(pl.ev0, st));
    // ---- two passes with the PIPELINE as the forward fill (k_dp_pipe<..., KEEP>: operand rows streamed once per
    // workgroup, boundary hand-off through LDS, H recurrence) and k_trace_recompute on blocks of PRALINE_KEEP_BH rows:
    // float-profile arenas, global mode, no rectangles, plans whose scratch fits the budget (praline_plan_create)
    if (pl.pipe.ok && mode == PRALINE_MODE_GLOBAL && la.a16 != nullptr && !pl.has_rects &&
        !(getenv("PRALINE_TB_PIPE") && getenv("PRALINE_TB_PIPE")[0] == '0')) {
        char kn[160];
        snprintf(kn, sizeof(kn), "k_dp_pipe<%d, %d, false, false, true>", a.nr16, a.nterm16);
        pl.last_kernel = kn;
        if (pl.d_bnd2.n < (size_t)pl.pipe_keep_bnd_elems) RC(pl.d_bnd2.alloc((size_t)pl.pipe_keep_bnd_elems));
        if (pl.d_tb.n < (size_t)pl.pipe_keep_ck_floats * 4) RC(pl.d_tb.alloc((size_t)pl.pipe_keep_ck_floats * 4));
        PipeLaunch pp;
        pp.items = pl.d_pipe_items.p;
        pp.n_items = (unsigned)pl.pipe.items.size();
        pp.tasks = pl.d_pipe_tasks.p;
        pp.set_one = pl.d_pipe_set_one.p;
        pp.lane_pair = pl.d_pipe_lane_pair.p;
        pp.bnd = pl.d_pipe_bnd.p;
        pp.analytic = pl.d_pipe_analytic.p;
        pp.analytic_rows = pl.pipe_analytic_rows;
        pp.analytic_valid = pl.pipe_analytic_mode == mode && pl.pipe_analytic_go == la.rp.go1 && pl.pipe_analytic_ge == la.rp.ge1;
        pl.pipe_analytic_mode = mode; pl.pipe_analytic_go = la.rp.go1; pl.pipe_analytic_ge = la.rp.ge1;
        pp.scores = la.scores;
        pp.rp = la.rp;
        pp.stream = st;
        int rc2 = praline_launch_pipe_keep(pp, a16, a.nr16, a.nterm16, pl.d_bnd2.p, (float *)pl.d_tb.p, pl.d_end_cells.p,
                                           pl.d_pipe_analytic4.p);
        if (rc2 != PRALINE_OK) return fail(rc2, "no kept-state pipeline instance for nr=%d nterm=%d", a.nr16, a.nterm16);
        Trace2Args ta;
        ta.slot_off = pl.d_slot_off.p;
        ta.paths = pl.d_paths.p;
        ta.path_start = pl.d_path_start.p;
        ta.path_rows = pl.d_path_rows.p;
        LaunchArgs lb = la;
        lb.tasks = pl.d_pipe_tasks.p;
        lb.n_tasks = (unsigned)pl.pipe.tasks.size();
        lb.lane_one = pl.d_pipe_lane_one.p;
        lb.lane_pair = pl.d_pipe_lane_pair.p;
        lb.tb = (uint4 *)pl.d_tb.p;
        lb.bnd = pl.d_bnd2.p;
        rc2 = praline_launch_tb2_backward(lb, a16, ta, a.nr16, a.nterm16, false, false, 2, pl.d_pipe_analytic4.p);
        if (rc2 != PRALINE_OK) return fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(pl.ev1, st));
        return PRALINE_OK;
    }
    // ---- two passes (dp_trace2.hip.h) for plans too large for chain mode: a flag-free forward fill that keeps the
    // strip boundary columns and (M, U, L) of every 32nd row, then k_trace_recompute rebuilds the flags of only the
    // 32 x 32 blocks each path crosses.  PRALINE_TB_TWOPASS=0 keeps the single pass.
    {
        int64_t single_bytes = 0;
        int all_strips = 0;
        for (size_t t = 0; t < nt; ++t) { single_bytes += pl.tb_elems[t] * (int64_t)tb_elem_bytes; all_strips = std::max(all_strips, (int)pl.tasks[t].nstrips); }
        // (plans over the budget are cut into chunks of about half the budget, each of which can run in chain mode)
        const int64_t chunk_tasks = (size_t)single_bytes <= budget ? (int64_t)nt
                                                                   : (int64_t)((double)nt * (double)(budget / 2) / (double)single_bytes) + 1;
        const bool would_chain = pl.split && la.a16 != nullptr && all_strips >= 2 && chunk_tasks <= chain_max_tasks() &&
                                 !(getenv("PRALINE_NO_CHAIN") && getenv("PRALINE_NO_CHAIN")[0] == '1');
        // Default: LOCAL plans only.  Measured on C3 (1 047 552 alignments of ~250 aa, one-hot): local 60.1 -> 47.3 ms,
        // global 54.7 -> 53.7 ms (the forward fill's extra stores and a recompute of ~half the cells eat the saving when
        // every path spans the whole matrix).  PRALINE_TB_TWOPASS=1: every mode, =2: also instead of chain mode, =0: never.
        const char *tp = getenv("PRALINE_TB_TWOPASS");
        const int tpv = tp ? atoi(tp) : -1;
        // ---- two passes with the forward fill on the staged SCORES kernel (k_dp_split16<..., KEEP>: LDS-DMA operand
        // stream, shared-wave workgroups, 9 instead of ~20 VALU operations per cell) - float-profile arenas, global
        // mode, plans of one chunk.  PRALINE_TB_KEEP=1 enables it.
        {
            const char *kp = getenv("PRALINE_TB_KEEP");
            const int kpv = kp ? atoi(kp) : -1;
            bool keep = pl.split && la.a16 != nullptr && a16.stage && kpv != 0 && tpv != 0 && mode == PRALINE_MODE_GLOBAL &&
                        !pl.has_rects && a.nterm16 != 1 && (!pl.wg.empty() || !pl.wg_singles.empty()) &&
                        kpv == 1;   // opt-in while it is being tuned (C2: 5.8 ms against 5.9 in chain mode)
            if (getenv("PRALINE_DEBUG_KEEP"))
                fprintf(stderr, "keep=%d split=%d a16=%d stage=%d kpv=%d tpv=%d mode=%d rects=%d nterm=%d wg=%zu singles=%zu nt=%zu\n", (int)keep,
                        (int)pl.split, la.a16 != nullptr, a16.stage, kpv, tpv, mode, (int)pl.has_rects, a.nterm16, pl.wg.size(), pl.wg_singles.size(), nt);
            int64_t ck_e = 0, bnd_e = 0;   // floats, float4s
            if (keep) {
                for (size_t t = 0; t < nt; ++t) {
                    const WaveTask &wt = pl.tasks[t];
                    ck_e += (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                }
                keep = (size_t)(ck_e * 4 + bnd_e * 16) <= budget;
            }
            if (keep) {
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_split16<%d, %d, false, 2, 4, true>", a.nr16, a.nterm16);
                pl.last_kernel = kn;
                ck_e = 0; bnd_e = 0;
                for (size_t t = 0; t < nt; ++t) {
                    WaveTask &wt = pl.tasks[t];
                    wt.tb_off = ck_e;
                    wt.aux_off = bnd_e;
                    ck_e += (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                }
                if (pl.d_tb.n < (size_t)ck_e * 4) RC(pl.d_tb.alloc((size_t)ck_e * 4));
                if (pl.d_bnd2.n < (size_t)bnd_e) RC(pl.d_bnd2.alloc((size_t)bnd_e));
                if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
                HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
                la.tasks = pl.d_tasks.p;
                la.n_tasks = (unsigned)nt;
                la.aux = nullptr;
                if (!pl.wg.empty()) {
                    if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
                    la.wg = pl.d_wg.p;
                    la.n_wg = (unsigned)pl.wg.size();
                } else {
                    if (!pl.d_wg_singles.p) { RC(pl.d_wg_singles.upload(pl.wg_singles, st)); }
                    la.wg = pl.d_wg_singles.p;
                    la.n_wg = (unsigned)pl.wg_singles.size();
                }
                int rc2 = praline_launch_keep_forward(la, a16, a.nr16, a.nterm16, pl.d_bnd2.p, (float *)pl.d_tb.p);
                if (rc2 != PRALINE_OK) return fail(rc2, "no kept-state forward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
                Trace2Args ta;
                ta.slot_off = pl.d_slot_off.p;
                ta.paths = pl.d_paths.p;
                ta.path_start = pl.d_path_start.p;
                ta.path_rows = pl.d_path_rows.p;
                la.tb = (uint4 *)pl.d_tb.p;
                la.bnd = pl.d_bnd2.p;
                rc2 = praline_launch_tb2_backward(la, a16, ta, a.nr16, a.nterm16, false, false, 1);
                if (rc2 != PRALINE_OK) return fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(pl.ev1, st));
                return PRALINE_OK;
            }
        }
        const bool twopass = pl.split && !pl.quad && !pl.run_pk16 && la.a16 != nullptr && tpv != 0 && (tpv == 2 || (!would_chain && (local || tpv == 1)));
        if (twopass) {
            char kn[160];
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, %d, %s, %s, false, true>", a.nr16, tb_nterm, local ? "true" : "false",
                     pl.has_rects ? "true" : "false");
            pl.last_kernel = kn;
            if (pl.bnd_off0.size() != nt) { pl.bnd_off0.resize(nt); for (size_t t = 0; t < nt; ++t) pl.bnd_off0[t] = pl.tasks[t].bnd_off; }
            // the chunk cutting below rewrites the tasks' boundary offsets; the single pass and chain mode address the
            // plan's shared boundary buffer through the scheduler's offsets: put them back on EVERY way out
            struct RestoreBnd {
                praline_plan &pl;
                ~RestoreBnd() { for (size_t t = 0; t < pl.tasks.size() && t < pl.bnd_off0.size(); ++t) pl.tasks[t].bnd_off = pl.bnd_off0[t]; }
            } restore_bnd{pl};
            if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
            int rc2 = PRALINE_OK;
            // plans that need several chunks: half the budget per chunk, two scratch sets, alternating streams
            size_t chunk_budget = budget;
            {
                int64_t all = 0;
                for (size_t t = 0; t < nt; ++t)
                    all += (int64_t)pl.tasks[t].nstrips * PRALINE_TB2_CKPT_BLOCKS(pl.tasks[t].max_l1) * PRALINE_TB2_CKPT_FLOATS * 4 +
                           (int64_t)(pl.tasks[t].nstrips + 1) * (pl.tasks[t].max_l1 + PRALINE_TB2_PAD_ROWS) * 32 * 16;
                if ((size_t)all > budget) chunk_budget = budget / 2;
            }
            // The chunks are cut first and each scratch set is allocated ONCE, for its largest chunk: a buffer that grew
            // in the middle of the loop would hand its old block back to the pool while the kernels of an earlier chunk
            // may still be using it - and the pool could give it to the OTHER set, which runs on the other stream
            // (seen with 45 000 alignments of ~1 000 x 1 300: thousands of wrong paths, different from run to run).
            struct Chunk2 { size_t t0, t1; int64_t ck_e, bnd_e, aux_e; };
            std::vector<Chunk2> chunks;
            while (t0 < nt) {
                size_t t1 = t0;
                int64_t ck_e = 0, bnd_e = 0, aux_e = 0;   // floats, float4s, floats
                while (t1 < nt) {
                    const WaveTask &wt = pl.tasks[t1];
                    const int64_t ck_add = (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    const int64_t bnd_add = (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                    if (t1 > t0 && (size_t)((ck_e + ck_add) * 4 + (bnd_e + bnd_add) * 16) > chunk_budget) break;
                    pl.tasks[t1].tb_off = ck_e;
                    pl.tasks[t1].bnd_off = bnd_e;
                    pl.tasks[t1].aux_off = aux_e;
                    ck_e += ck_add;
                    bnd_e += bnd_add;
                    aux_e += semiglobal ? pl.aux_elems[t1] : 0;
                    ++t1;
                }
                chunks.push_back({t0, t1, ck_e, bnd_e, aux_e});
                t0 = t1;
            }
            {
                int64_t need_ck[2] = {0, 0}, need_bk[2] = {0, 0}, need_ax[2] = {1, 1};
                for (size_t c = 0; c < chunks.size(); ++c) {
                    need_ck[c & 1] = std::max(need_ck[c & 1], chunks[c].ck_e * 4);
                    need_bk[c & 1] = std::max(need_bk[c & 1], chunks[c].bnd_e);
                    need_ax[c & 1] = std::max(need_ax[c & 1], chunks[c].aux_e);
                }
                // (an earlier run's kernels may still be reading a block that is replaced here: the pool is stream-ordered -
                // the old block is not handed out again before both streams have passed this point - so no host wait)
                if (pl.d_tb.n < (size_t)need_ck[0]) RC(pl.d_tb.alloc((size_t)need_ck[0]));
                if (pl.d_bnd2.n < (size_t)need_bk[0]) RC(pl.d_bnd2.alloc((size_t)need_bk[0]));
                if (pl.d_aux.n < (size_t)need_ax[0]) RC(pl.d_aux.alloc((size_t)need_ax[0]));
                if (chunks.size() > 1) {
                    if (pl.d_tb_b.n < (size_t)need_ck[1]) RC(pl.d_tb_b.alloc((size_t)need_ck[1]));
                    if (pl.d_bnd2_b.n < (size_t)need_bk[1]) RC(pl.d_bnd2_b.alloc((size_t)need_bk[1]));
                    if (pl.d_aux_b.n < (size_t)need_ax[1]) RC(pl.d_aux_b.alloc((size_t)need_ax[1]));
                }
            }
            HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));   // (before the fork)
            bool forked = false;
            for (size_t c = 0; c < chunks.size() && rc2 == PRALINE_OK; ++c) {
                const int set = (int)(c & 1);
                hipStream_t cs = set ? g_rt.stream2 : st;
                if (set && !forked) {
                    if (hipEventRecord(g_rt.ev_fork, st) != hipSuccess || hipStreamWaitEvent(g_rt.stream2, g_rt.ev_fork, 0) != hipSuccess) {
                        rc2 = fail(PRALINE_ERR_DEVICE, "stream fork failed");
                        break;
                    }
                    forked = true;
                }
                DevBuf<char> &d_ck = set ? pl.d_tb_b : pl.d_tb;
                DevBuf<float4> &d_bk = set ? pl.d_bnd2_b : pl.d_bnd2;
                DevBuf<float> &d_ax = set ? pl.d_aux_b : pl.d_aux;
                la.stream = cs;
                const size_t c0 = chunks[c].t0, t1 = chunks[c].t1;
                la.tasks = pl.d_tasks.p + c0;
                la.lane_one = pl.d_lane_one.p + c0 * 32;
                la.lane_pair = pl.d_lane_pair.p + c0 * 32;
                la.tb = (uint4 *)d_ck.p;
                la.bnd = d_bk.p;
                la.aux = d_ax.p;
                la.n_tasks = (unsigned)(t1 - c0);
                rc2 = praline_launch_tb2_forward(la, a16, a.nr16, tb_nterm, local, pl.has_rects);
                if (rc2 != PRALINE_OK) { rc2 = fail(rc2, "no two-pass forward instance for nr=%d nterm=%d", a.nr16, tb_nterm); break; }
                if (semiglobal) {
                    const int64_t lanes = (int64_t)(t1 - c0) * 32;
                    hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, cs, la.ar, pl.d_tasks.p,
                                       pl.d_lane_one.p, pl.d_lane_pair.p, pl.d_pairs.p, d_ax.p, pl.d_end_cells.p, la.scores,
                                       la.rp, (int32_t)c0, (int32_t)t1, 1);
                }
                Trace2Args ta;
                ta.slot_off = pl.d_slot_off.p;
                ta.paths = pl.d_paths.p;
                ta.path_start = pl.d_path_start.p;
                ta.path_rows = pl.d_path_rows.p;
                rc2 = praline_launch_tb2_backward(la, a16, ta, a.nr16, tb_nterm, local, pl.has_rects);
                if (rc2 != PRALINE_OK) { rc2 = fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, tb_nterm); break; }
                if (hipGetLastError() != hipSuccess) { rc2 = fail(PRALINE_ERR_DEVICE, "two-pass launch failed"); break; }
            }
            la.stream = st;
            if (forked && (hipEventRecord(g_rt.ev_join, g_rt.stream2) != hipSuccess || hipStreamWaitEvent(st, g_rt.ev_join, 0) != hipSuccess))
                rc2 = fail(PRALINE_ERR_DEVICE, "stream join failed");
            if (rc2 != PRALINE_OK) return rc2;
            // (d_tasks holds two-pass offsets now: the next single-pass run uploads its own)
            HIPCHK(hipEventRecord(pl.ev1, st));
            return PRALINE_OK;
        }
    }
    // plans that need several chunks: half the budget per chunk, two scratch sets, alternating streams (the traceback
    // and the tail of chunk k overlap the fill of chunk k + 1)
    size_t chunk_budget = budget;
    {
        int64_t all = 0;
        for (size_t t = 0; t < nt; ++t) all += pl.tb_elems[t] * (int64_t)tb_elem_bytes;
        if ((size_t)all > budget) chunk_budget = budget / 2;
    }
    // (cut first, allocate each scratch set once for its largest chunk - see the two-pass loop above)
    struct Chunk1 { size_t t0, t1; int64_t tb_e, aux_e; };
    std::vector<Chunk1> chunks;
    while (t0 < nt) {
        size_t t1 = t0;
        int64_t tb_e = 0, aux_e = 0;
        while (t1 < nt) {
            const int64_t add = pl.tb_elems[t1];
            if (t1 > t0 && (size_t)(tb_e + add) * tb_elem_bytes > chunk_budget) break;
            pl.tasks[t1].tb_off = tb_e;
            pl.tasks[t1].aux_off = aux_e;
            tb_e += add;
            aux_e += semiglobal ? pl.aux_elems[t1] : 0;
            ++t1;
        }
        chunks.push_back({t0, t1, tb_e, aux_e});
        t0 = t1;
    }
    {
        size_t need_tb[2] = {0, 0}, need_ax[2] = {1, 1};
        for (size_t c = 0; c < chunks.size(); ++c) {
            need_tb[c & 1] = std::max(need_tb[c & 1], (size_t)chunks[c].tb_e * tb_elem_bytes);
            need_ax[c & 1] = std::max(need_ax[c & 1], (size_t)chunks[c].aux_e);
        }
        // (blocks replaced here may still be read by an earlier run's kernels: stream-ordered pool, no host wait)
        if (pl.d_tb.n < need_tb[0]) RC(pl.d_tb.alloc(need_tb[0]));
        if (pl.d_aux.n < need_ax[0]) RC(pl.d_aux.alloc(need_ax[0]));
        if (chunks.size() > 1) {
            if (pl.d_tb_b.n < need_tb[1]) RC(pl.d_tb_b.alloc(need_tb[1]));
            if (pl.d_aux_b.n < need_ax[1]) RC(pl.d_aux_b.alloc(need_ax[1]));
        }
    }
    if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
    HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));   // (before the fork)
    bool forked = false;
    // every way out of the loop below joins the second stream again (an error return would otherwise leave stream2's
    // kernels unordered against whatever the main stream does next with the plan's buffers)
    struct JoinGuard {
        bool &forked; hipStream_t st;
        ~JoinGuard()
        {
            if (forked && hipEventRecord(g_rt.ev_join, g_rt.stream2) == hipSuccess) (void)hipStreamWaitEvent(st, g_rt.ev_join, 0);
            forked = false;
        }
    } join_guard{forked, st};
    // Chain mode (one wave per task AND strip, pipelined across workgroups: dp_split16_tb.hip.h) for chunks of up to
    // chain_max_tasks() tasks - single alignments, the merge steps of the progressive MSA, C2-sized batches, and the
    // chunks of plans whose packed traceback exceeds the scratch budget (long sequences: 32 640 alignments of ~1000 x
    // ~1000 were 96 ms in task mode, three chunks of 380 waves each).  Chain chunks share one set of boundary columns
    // and flags: they all run on the main stream.
    bool chain_chunks = pl.split && !pl.quad && la.a16 != nullptr && !(getenv("PRALINE_NO_CHAIN") && getenv("PRALINE_NO_CHAIN")[0] == '1');
    {
        int64_t need_bnd = 0;
        size_t need_flags = 0;
        for (size_t c = 0; c < chunks.size() && chain_chunks; ++c) {
            int max_strips = 0;
            int64_t bnd_e = 0;
            for (size_t t = chunks[c].t0; t < chunks[c].t1; ++t) {
                max_strips = std::max(max_strips, (int)pl.tasks[t].nstrips);
                bnd_e += (int64_t)(pl.tasks[t].nstrips + 1) * (pl.tasks[t].max_l1 + 24) * 32;
            }
            if (max_strips < 2 || (int64_t)(chunks[c].t1 - chunks[c].t0) > chain_max_tasks()) chain_chunks = false;
            need_bnd = std::max(need_bnd, bnd_e);
            need_flags = std::max(need_flags, (chunks[c].t1 - chunks[c].t0) * (size_t)(max_strips + 1));
        }
        if (chain_chunks) {
            if (pl.d_bnd_chain.n < (size_t)need_bnd * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)need_bnd * sizeof(float4)));
            if (pl.d_chain_flags.n < need_flags) RC(pl.d_chain_flags.alloc(need_flags));
            if (local && pl.d_chain_cand.n < need_flags * 32) RC(pl.d_chain_cand.alloc(need_flags * 32));
        }
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        const int set = (int)(c & 1);
        hipStream_t cs = (set && !chain_chunks) ? g_rt.stream2 : st;
        if (set && !forked && !chain_chunks) {
            HIPCHK(hipEventRecord(g_rt.ev_fork, st));
            HIPCHK(hipStreamWaitEvent(g_rt.stream2, g_rt.ev_fork, 0));
            forked = true;
        }
        DevBuf<char> &d_tbs = set ? pl.d_tb_b : pl.d_tb;
        DevBuf<float> &d_ax = set ? pl.d_aux_b : pl.d_aux;
        la.stream = cs;
        t0 = chunks[c].t0;
        const size_t t1 = chunks[c].t1;
        la.tasks = pl.d_tasks.p + t0;
        la.lane_one = pl.d_lane_one.p + t0 * lanes_per_task;
        la.lane_pair = pl.d_lane_pair.p + t0 * lanes_per_task;
        la.tb = (uint4 *)d_tbs.p;
        la.aux = d_ax.p;
        la.n_tasks = (unsigned)(t1 - t0);
        int max_strips = 0;
        for (size_t t = t0; t < t1; ++t) max_strips = std::max(max_strips, (int)pl.tasks[t].nstrips);
        const bool chain = chain_chunks;
        if (chain) {
            const size_t nc = t1 - t0;   // tasks of this chunk
            std::vector<WaveTask> ct(pl.tasks.begin() + (std::ptrdiff_t)t0, pl.tasks.begin() + (std::ptrdiff_t)t1);
            int64_t bnd_e = 0;
            for (WaveTask &wt : ct) {
                wt.bnd_off = bnd_e;
                bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + 24) * 32;   // float4 elements, [strip boundary][row][32]
            }
            const size_t n_flags = nc * (size_t)(max_strips + 1);
            HIPCHK(hipMemsetAsync(pl.d_chain_flags.p, 0, n_flags * sizeof(int), st));
            HIPCHK(hipMemcpyAsync(pl.d_tasks.p + t0, ct.data(), nc * sizeof(WaveTask), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));   // ct goes out of scope
            la.bnd = pl.d_bnd_chain.p;
            // rows between two publishes of a strip's progress: few for plans whose strip waves all run at once
            // (a single alignment: the next strip follows a few rows behind), many once a strip level alone
            // fills the chip (the consumers are dispatched a round later; every publish drains the stores)
            int every = nc >= 512 ? 96 : (nc >= 64 ? 24 : 6);
            {   // short sequences: at least four publishes per strip (co-resident consumers would wait for the end)
                int rows = 0;
                for (const WaveTask &wt : ct) rows = std::max(rows, (int)wt.max_l1);
                every = std::min(every, std::max(6, rows / 4));
            }
            // (k_dp_pk16_tb: two rows per step and shorter steps - measured on C2 one-hot 96 rows 1.55, 24 rows 1.60 TCUPS;
            // 2 016 pairs 12 rows; one alignment 6 rows: scripts/exp_pk16_chain.py)
            if (pl.run_pk16) every = std::min(every, nc >= 512 ? 24 : (nc >= 64 ? 12 : 6));
            if (const char *env = getenv("PRALINE_CHAIN_EVERY")) every = std::max(6, atoi(env));
            if (pl.run_pk16) {
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, %s, %d, true>", a.nr16, local ? "true" : "false", pk16_slots);
                pl.last_kernel = kn;
            }
            int rc = pl.run_pk16 ? praline_launch_pk16_tb_chain(la, a16, a.nr16, local, pk16_slots, pk16_scale, max_strips, pl.d_chain_flags.p,
                                                                pl.d_chain_cand.p, every)
                                 : praline_launch_split16_tb_chain(la, a16, a.nr16, tb_nterm, local, pl.has_rects, max_strips,
                                                                   pl.d_chain_flags.p, pl.d_chain_cand.p, every);
            if (rc != PRALINE_OK) return fail(rc, "no chain instance of the path kernel for nr=%d nterm=%d", a.nr16, tb_nterm);
            if (local) {
                const int64_t lanes = (int64_t)nc * 32;
                hipLaunchKernelGGL(k_chain_local_end, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, la.tasks,
                                   la.lane_pair, pl.d_chain_cand.p, (int)nc, max_strips + 1, pl.d_end_cells.p, la.scores);
            }
            la.bnd = pl.d_bnd.p;
        } else if (pl.quad) {
            int rc = praline_launch_quad_tb(la, a16, a.nr16, tb_nterm == 1, local, pl.mask_kind);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_quad_tb instance for nr=%d", a.nr16);
        } else if (pl.run_pk16) {
            int rc = praline_launch_pk16_tb(la, a16, a.nr16, local, pk16_slots, pk16_scale);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_pk16_tb instance for nr=%d", a.nr16);
        } else {
            int rc = praline_launch_split16_tb(la, a16, a.nr16, tb_nterm, local, pl.has_rects);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_split16_tb instance for nr=%d nterm=%d", a.nr16, tb_nterm);
        }
        HIPCHK(hipGetLastError());
        RC(launch_traceback(pl, la, t0, t1, mode));
    }
    la.stream = st;
    if (forked) {
        HIPCHK(hipEventRecord(g_rt.ev_join, g_rt.stream2));
        HIPCHK(hipStreamWaitEvent(st, g_rt.ev_join, 0));
        forked = false;
    }
    HIPCHK(hipEventRecord(pl.ev1, st));
    return PRALINE_OK;
}

// praline_plan_run with the arena's per-position gap scores (praline_arena_set_gap_scores) instead of one (open, extend):
// U[y][x] takes the scores of position y - 1 of sequence one, L[y][x] those of position x - 1 of sequence two
// (cext.c:155-158,172-175), the boundary cells follow align.py:371-385.  The plan must have been created while the
// arena held gap scores (such plans read their match scores from dense tiles, plan_run_dense).
extern "C" int praline_plan_run_gaps(praline_plan *plan, int mode, void *d_scores)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->ppg) return fail(PRALINE_ERR_UNSUPPORTED, "the plan was created before praline_arena_set_gap_scores");
    if (!plan->arena->has_gaps || !plan->arena->d_gaps.p) return fail(PRALINE_ERR_ARG, "the arena holds no gap scores");
    plan->run_ppg = true;
    const int rc = praline_plan_run(plan, mode, 0.0f, 0.0f, d_scores);
    plan->run_ppg = false;
    return rc;
}

extern "C" int praline_plan_kernel_name(const praline_plan *plan, char *buf, int64_t size)
{
    if (!plan || !buf || size <= 0) return fail(PRALINE_ERR_ARG, "NULL argument");
    snprintf(buf, (size_t)size, "%s", plan->last_kernel.c_str());
    return PRALINE_OK;
}

extern "C" int praline_plan_kernel_resources(const praline_plan *plan, int32_t *vgprs, int32_t *lds_bytes, int32_t *waves_per_simd)
{
    if (!plan || !vgprs || !lds_bytes || !waves_per_simd) return fail(PRALINE_ERR_ARG, "NULL argument");
    *vgprs = *lds_bytes = *waves_per_simd = 0;
    if (!plan->pipe.ok || plan->last_mode < 0) return PRALINE_OK;   // (reported for the pipeline workgroups only)
    int v = 0, l = 0;
    if (plan->want_paths) {
        // (path plans: the pipeline is the forward fill of global runs only)
        if (plan->last_mode != PRALINE_MODE_GLOBAL || plan->last_kernel.compare(0, 9, "k_dp_pipe") != 0) return PRALINE_OK;
        RC(praline_pipe_keep_attrs(plan->arena->nr16, plan->arena->nterm16, &v, &l));
    } else
    RC(praline_pipe_attrs(plan->arena->nr16, plan->arena->nterm16, plan->last_mode, &v, &l));
    *vgprs = v;
    *lds_bytes = l;
    // MI355X_MICROARCH.md, register files: allocation granule 8, 512 registers per lane and SIMD; 160 KiB of LDS per CU;
    // a workgroup of four waves puts one wave on every SIMD
    const int by_regs = std::min(8, 512 / std::max(8, (v + 7) / 8 * 8));
    const int by_lds = l > 0 ? (160 * 1024) / l : 8;
    *waves_per_simd = std::min(by_regs, by_lds);
    return PRALINE_OK;
}

extern "C" int praline_plan_last_timing(praline_plan *plan, float *kernel_ms)
{
    if (!plan || !kernel_ms) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (plan->n_pairs == 0) { *kernel_ms = 0.0f; return PRALINE_OK; }
    if (plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    HIPCHK(hipEventSynchronize(plan->ev1));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, plan->ev0, plan->ev1));
    plan->last_kernel_ms = ms;
    *kernel_ms = ms;
    return PRALINE_OK;
}

extern "C" int praline_plan_scores(praline_plan *plan, float *scores)
{
    if (!plan || (!scores && plan->n_pairs)) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (plan->n_pairs == 0) return PRALINE_OK;
    if (!plan->last_scores) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    // the buffer the last run wrote: the plan's own or the caller's d_scores
    HIPCHK(hipMemcpyAsync(scores, plan->last_scores, (size_t)plan->n_pairs * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_plan_paths(praline_plan *plan, int32_t *paths, int64_t *path_off, int32_t *path_rows)
{
    if (!plan || !paths || !path_off || !path_rows) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->n_pairs == 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    HIPCHK(hipMemcpyAsync(paths, plan->d_paths.p, (size_t)plan->path_cap * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(path_off, plan->d_path_start.p, (size_t)plan->n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(path_rows, plan->d_path_rows.p, (size_t)plan->n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PRALINE_OK;
}

// ---- preprofile stage on the device: counts and path bounding boxes (k_path_counts / k_path_bounds) ----------
extern "C" int praline_arena_counts_reset(praline_arena *arena)
{
    RC(arena_ready(arena));
    if (!arena->counts_ext && !arena->d_counts.p) RC(arena->d_counts.alloc((size_t)arena->rows_raw * arena->A));
    HIPCHK(hipMemsetAsync(arena->counts_ptr(), 0, (size_t)arena->rows_raw * arena->A * sizeof(int32_t), g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_counts_bind(praline_arena *arena, void *d_counts)
{
    RC(arena_ready(arena));
    arena->counts_ext = (int32_t *)d_counts;
    return PRALINE_OK;
}

extern "C" int praline_plan_add_counts(praline_plan *plan, int use_threshold, float threshold, int local)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    praline_arena &a = *plan->arena;
    if (!a.counts_ptr()) return fail(PRALINE_ERR_ARG, "praline_arena_counts_reset has not been called");
    if (!a.all_onehot)
        return fail(PRALINE_ERR_UNSUPPORTED, "preprofile counting needs one-hot profiles (plain sequences), as "
                    "ProfileBuilder needs plain tracks (praline/util/align.py:187-213)");
    if (plan->n_pairs == 0) return PRALINE_OK;
    // pair lists whose masters come in runs (the preprofile stage's order): a workgroup per run with the master's count
    // block in LDS (k_path_counts_runs).  The runs are found once per plan, from the device copy of the pair list.
    if (plan->count_runs < 0) {
        plan->count_runs = 0;
        const size_t lds_need = (size_t)a.max_len * a.A * sizeof(int32_t);
        const char *cr = getenv("PRALINE_COUNT_RUNS");   // 0: never; 1: whenever the count block fits LDS (tests); default: runs of 64 pairs and more on average
        const bool forced = cr && cr[0] == '1';
        const int64_t max_runs = forced ? plan->n_pairs : plan->n_pairs / 64;
        if (lds_need <= (size_t)64 << 10 && (plan->n_pairs >= 4096 || forced) && !(cr && cr[0] == '0')) {
            std::vector<int32_t> hp((size_t)plan->n_pairs * 2);
            HIPCHK(hipMemcpyAsync(hp.data(), plan->d_pairs.p, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
            HIPCHK(hipStreamSynchronize(g_rt.stream));
            std::vector<int64_t> runs;
            for (int64_t p = 0; p < plan->n_pairs;) {
                int64_t q = p + 1;
                while (q < plan->n_pairs && hp[(size_t)(2 * q)] == hp[(size_t)(2 * p)]) ++q;
                runs.push_back(p); runs.push_back(q);
                p = q;
                if ((int64_t)runs.size() / 2 > max_runs) break;   // (short runs: one lane per pair and global atomics)
            }
            if ((int64_t)runs.size() / 2 <= max_runs) {
                RC(plan->d_count_runs.upload(runs, g_rt.stream));
                HIPCHK(hipStreamSynchronize(g_rt.stream));   // (runs goes out of scope)
                plan->count_runs = (int64_t)runs.size() / 2;
            }
        }
    }
    if (plan->count_runs > 0) {
        hipLaunchKernelGGL(k_path_counts_runs, dim3((unsigned)plan->count_runs), dim3(256), (size_t)a.max_len * a.A * sizeof(int32_t),
                           g_rt.stream, plan->d_pairs.p, plan->last_scores, plan->d_paths.p, plan->d_path_start.p, plan->d_path_rows.p,
                           plan->d_count_runs.p, use_threshold, threshold, local, a.d_row_off_raw.p, a.d_len.p, a.d_sym_raw.p, a.A,
                           a.counts_ptr());
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    const int threads = 64;
    const int64_t blocks = (plan->n_pairs + threads - 1) / threads;
    hipLaunchKernelGGL(k_path_counts, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, plan->d_pairs.p,
                       plan->last_scores, plan->d_paths.p, plan->d_path_start.p, plan->d_path_rows.p, plan->n_pairs,
                       use_threshold, threshold, local, a.d_row_off_raw.p, a.d_len.p, a.d_sym_raw.p, a.A, a.counts_ptr());
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

extern "C" int praline_arena_counts_read(praline_arena *arena, int32_t *counts)
{
    if (!arena || !counts) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (!arena->counts_ptr()) return fail(PRALINE_ERR_ARG, "praline_arena_counts_reset has not been called");
    HIPCHK(hipMemcpyAsync(counts, arena->counts_ptr(), (size_t)arena->rows_raw * arena->A * sizeof(int32_t),
                          hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_plan_path_bounds(praline_plan *plan, int32_t *bounds)
{
    if (!plan || !bounds) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->n_pairs == 0) return PRALINE_OK;
    DevBuf<int32_t> d_bounds;
    RC(d_bounds.alloc((size_t)plan->n_pairs * 4));
    const int threads = 256;
    const int64_t blocks = (plan->n_pairs + threads - 1) / threads;
    hipLaunchKernelGGL(k_path_bounds, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, plan->d_paths.p,
                       plan->d_path_start.p, plan->d_path_rows.p, plan->n_pairs, d_bounds.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(bounds, d_bounds.p, (size_t)plan->n_pairs * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_append_merged_many(praline_arena *arena, praline_plan *plan, int64_t n, const int64_t *pair_index,
                                                int32_t *new_index, int32_t *new_len)
{
    if (!arena || !plan || !new_index || !new_len || (n > 0 && !pair_index)) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (plan->arena != arena) return fail(PRALINE_ERR_ARG, "the plan belongs to another arena");
    if (!plan->want_paths || plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "the plan has no paths (want_paths + praline_plan_run first)");
    if (plan->last_mode == PRALINE_MODE_LOCAL) return fail(PRALINE_ERR_UNSUPPORTED, "clusters are merged along global / semiglobal paths");
    if (n <= 0) return PRALINE_OK;
    for (int64_t q = 0; q < n; ++q)
        if (pair_index[q] < 0 || pair_index[q] >= plan->n_pairs) return fail(PRALINE_ERR_ARG, "pair index out of range");
    praline_arena *a = arena;
    if (!a->have_cnt) return fail(PRALINE_ERR_ARG, "praline_arena_set_counts has not been called");
    if (a->has_gaps) return fail(PRALINE_ERR_UNSUPPORTED, "the arena holds per-position gap scores: it cannot grow");
    hipStream_t st = g_rt.stream;
    // where the paths are: one round trip for the whole plan (a level of the guide tree is one plan)
    const int64_t np = plan->n_pairs;
    std::vector<int64_t> start((size_t)np);
    std::vector<int32_t> rows((size_t)np), pr((size_t)np * 2);
    HIPCHK(hipMemcpyAsync(start.data(), plan->d_path_start.p, (size_t)np * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(rows.data(), plan->d_path_rows.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pr.data(), plan->d_pairs.p, (size_t)np * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t idx0 = a->n_seqs;
    int64_t rows_raw = a->rows_raw, rp = a->rp_end;
    int new_max = a->max_len;
    for (int64_t q = 0; q < n; ++q) {
        const int cols = rows[(size_t)pair_index[q]] - 1;
        if (cols <= 0) return fail(PRALINE_ERR_DEVICE, "empty alignment path");
        rows_raw += cols;
        rp += (cols + 31) / 32 * 32;
        new_max = std::max(new_max, cols);
    }
    const int64_t new_rows_pad = rp + (new_max + 31) / 32 * 32 + 64;
    RC(arena_reserve(a, idx0 + n, rows_raw, new_rows_pad));
    if (!a->d_set_lo.p) RC(a->d_set_lo.upload(a->set_lo, st));
    const int64_t rp0 = a->rp_end;
    for (int64_t q = 0; q < n; ++q) {
        const int64_t p = pair_index[q];
        const int cols = rows[(size_t)p] - 1;
        const int64_t pad = (cols + 31) / 32 * 32;
        const int32_t off_raw = (int32_t)a->rows_raw, off_pad = (int32_t)a->rp_end;
        const int64_t idx = a->n_seqs;
        hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((pad + 255) / 256)), dim3(256), 0, st, a->d_seq_of_rowp.p + a->rp_end, pad, (int32_t)idx);
        hipLaunchKernelGGL(k_merge_clusters, dim3((unsigned)cols), dim3(64), 0, st, plan->d_paths.p + 2 * start[(size_t)p], cols, a->d_cnt.p,
                           a->d_raw.p, a->A, (int64_t)a->row_off_raw[pr[(size_t)(2 * p)]], (int64_t)a->row_off_raw[pr[(size_t)(2 * p + 1)]],
                           (int64_t)off_raw, a->d_set_lo.p, (int)a->set_lo.size() - 1);
        a->len.push_back(cols);
        a->row_off_raw.push_back(off_raw);
        a->row_off_pad.push_back(off_pad);
        a->n_seqs = idx + 1;
        a->rows_raw += cols;
        a->rp_end += pad;
        new_index[q] = (int32_t)idx;
        new_len[q] = cols;
    }
    HIPCHK(hipGetLastError());
    // the descriptors of the new sequences (the host vectors are final now)
    HIPCHK(hipMemcpyAsync(a->d_len.p + idx0, a->len.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(a->d_row_off_raw.p + idx0, a->row_off_raw.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(a->d_row_off_pad.p + idx0, a->row_off_pad.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    a->max_len = new_max;
    a->rows_pad = new_rows_pad;
    // a merged cluster is no plain sequence: the one-hot shortcuts of this arena end here
    a->onehot = false;
    a->all_onehot = false;
    if (a->nr16 > 0 && a->nterm16 == 1) a->nterm16 = 3;   // (an exact arena keeps the standard layout; its new rows need the lo pieces)
    a->ref_ready = false;
    a->reft2_state = 0;
    a->d_counts.release();
    a->counts_ext = nullptr;
    if (!a->wide) {   // packed operands of the new rows only (they are contiguous in the padded row space)
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)((a->rp_end - rp0) / 32)), dim3(256), 0, st, a->d_raw.p, a->d_S.p, a->d_seq_of_rowp.p,
                           a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, a->d_active.p, a->n_active, a->A, a->KP, a->KS,
                           a->rows_pad, a->d_P.p, a->d_Q.p, a->nr16, (_Float16 *)a->d_P16.p, (_Float16 *)a->d_Q16.p,
                           (int64_t)(rp0 / 32), a->nterm16 == 2 ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));   // the descriptor uploads read the host vectors
    return PRALINE_OK;
}

extern "C" int praline_arena_append_merged(praline_arena *arena, praline_plan *plan, int64_t pair_index, int32_t *new_index,
                                           int32_t *new_len)
{
    return praline_arena_append_merged_many(arena, plan, 1, &pair_index, new_index, new_len);
}

extern "C" int praline_plan_mask_path_bounds(praline_plan *plan)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->want_paths || plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "the plan has no paths (want_paths + praline_plan_run first)");
    if (plan->has_rects && plan->slot_rects < 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "the plan was created with its own rectangle lists");
    if (plan->n_pairs == 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    if (plan->slot_rects < 0) {
        // fixed slots: PRALINE_MAX_RECTS per pair, all empty to start with
        std::vector<int32_t> ro((size_t)plan->n_pairs + 1);
        for (int64_t p = 0; p <= plan->n_pairs; ++p) ro[(size_t)p] = (int32_t)(p * PRALINE_MAX_RECTS);
        RC(plan->d_rect_off.upload(ro, st));
        RC(plan->d_rects.alloc((size_t)plan->n_pairs * PRALINE_MAX_RECTS * 4));
        const int32_t empty[4] = {1 << 30, -1, 1 << 30, -1};
        std::vector<int32_t> rv((size_t)plan->n_pairs * PRALINE_MAX_RECTS * 4);
        for (size_t i = 0; i < rv.size(); ++i) rv[i] = empty[i & 3];
        RC(plan->d_rects.upload(rv.data(), rv.size(), st));
        HIPCHK(hipStreamSynchronize(st));
        plan->slot_rects = 0;
    }
    if (plan->slot_rects >= PRALINE_MAX_RECTS)
        return fail(PRALINE_ERR_UNSUPPORTED, "more than %d rectangles per pair: create a plan with explicit rectangle lists", PRALINE_MAX_RECTS);
    const int64_t blocks = (plan->n_pairs + 255) / 256;
    hipLaunchKernelGGL(k_path_bounds_to_rects, dim3((unsigned)blocks), dim3(256), 0, st, plan->d_paths.p, plan->d_path_start.p,
                       plan->d_path_rows.p, plan->n_pairs, plan->slot_rects, plan->d_rects.p);
    HIPCHK(hipGetLastError());
    plan->slot_rects += 1;
    plan->has_rects = true;
    plan->mask_kind = 1;
    return PRALINE_OK;
}

extern "C" int praline_batch_scores(praline_arena *arena, int mode, float gap_open, float gap_extend, int64_t n_pairs,
                                    const int32_t *pairs, float *scores)
{
    praline_plan *pl = nullptr;
    RC(praline_plan_create(arena, n_pairs, pairs, 0, nullptr, nullptr, &pl));
    int rc = praline_plan_run(pl, mode, gap_open, gap_extend, nullptr);
    if (rc == PRALINE_OK) rc = praline_plan_scores(pl, scores);
    praline_plan_destroy(pl);
    return rc;
}

// --------------------------------------------------------------------------------------------
// parity-layout entry points (strided host buffers <-> contiguous device copies)
// --------------------------------------------------------------------------------------------
static bool arr_ok(const praline_array *a) { return a && a->data; }

template <typename T> static void gather2(const praline_array &a, std::vector<T> &out)
{
    const int64_t R = a.dim[0], C = a.dim[1];
    out.resize((size_t)(R * C));
    const char *base = (const char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c) out[(size_t)(r * C + c)] = *(const T *)(base + r * a.stride[0] + c * a.stride[1]);
}

template <typename T> static void gather3(const praline_array &a, std::vector<T> &out)
{
    const int64_t R = a.dim[0], C = a.dim[1], K = a.dim[2];
    out.resize((size_t)(R * C * K));
    const char *base = (const char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < K; ++k)
                out[(size_t)((r * C + c) * K + k)] = *(const T *)(base + r * a.stride[0] + c * a.stride[1] + k * a.stride[2]);
}

template <typename T> static void scatter2(const std::vector<T> &in, const praline_array &a)
{
    const int64_t R = a.dim[0], C = a.dim[1];
    char *base = (char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c) *(T *)(base + r * a.stride[0] + c * a.stride[1]) = in[(size_t)(r * C + c)];
}

template <typename T> static void scatter3(const std::vector<T> &in, const praline_array &a)
{
    const int64_t R = a.dim[0], C = a.dim[1], K = a.dim[2];
    char *base = (char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < K; ++k)
                *(T *)(base + r * a.stride[0] + c * a.stride[1] + k * a.stride[2]) = in[(size_t)((r * C + c) * K + k)];
}

extern "C" int praline_build_scores(int num_sets, const praline_array *i1s, const praline_array *i2s,
                                    const praline_array *i1nzs, const praline_array *i2nzs, const praline_array *ss,
                                    const praline_array *m)
{
    (void)i1nzs; (void)i2nzs;  // dense contraction on the device; see praline_dp.h
    if (num_sets <= 0 || !i1s || !i2s || !ss || !arr_ok(m)) return fail(PRALINE_ERR_ARG, "NULL / empty build_scores argument");
    const int64_t L1 = i1s[0].dim[0], L2 = i2s[0].dim[0];
    if (L1 <= 0 || L2 <= 0) return fail(PRALINE_ERR_ARG, "empty sequence");
    if (m->dim[0] != L1 || m->dim[1] != L2) return fail(PRALINE_ERR_ARG, "m has shape %lldx%lld, expected %lldx%lld",
                                                     (long long)m->dim[0], (long long)m->dim[1], (long long)L1, (long long)L2);
    int64_t A = 0;
    for (int n = 0; n < num_sets; ++n) {
        if (!arr_ok(&i1s[n]) || !arr_ok(&i2s[n]) || !arr_ok(&ss[n])) return fail(PRALINE_ERR_ARG, "NULL array in set %d", n);
        if (i1s[n].dim[0] != L1 || i2s[n].dim[0] != L2) return fail(PRALINE_ERR_ARG, "set %d: profile lengths differ", n);
        if (ss[n].dim[0] != i1s[n].dim[1] || ss[n].dim[1] != i2s[n].dim[1])
            return fail(PRALINE_ERR_ARG, "set %d: score matrix shape does not match the profiles", n);
        A += std::max(i1s[n].dim[1], i2s[n].dim[1]);
    }
    if (A > 254) return fail(PRALINE_ERR_UNSUPPORTED, "concatenated alphabet size %lld > 254", (long long)A);
    // concatenate the track sets along the alphabet axis: P = [P_1 | P_2 ...], S = blockdiag(S_n)
    std::vector<float> prof((size_t)((L1 + L2) * A), 0.0f), S((size_t)(A * A), 0.0f), tmp;
    int64_t off = 0;
    for (int n = 0; n < num_sets; ++n) {
        const int64_t A1 = i1s[n].dim[1], A2 = i2s[n].dim[1];
        gather2<float>(i1s[n], tmp);
        for (int64_t r = 0; r < L1; ++r) for (int64_t c = 0; c < A1; ++c) prof[(size_t)(r * A + off + c)] = tmp[(size_t)(r * A1 + c)];
        gather2<float>(i2s[n], tmp);
        for (int64_t r = 0; r < L2; ++r) for (int64_t c = 0; c < A2; ++c) prof[(size_t)((L1 + r) * A + off + c)] = tmp[(size_t)(r * A2 + c)];
        gather2<float>(ss[n], tmp);
        for (int64_t r = 0; r < A1; ++r) for (int64_t c = 0; c < A2; ++c) S[(size_t)((off + r) * A + off + c)] = tmp[(size_t)(r * A2 + c)];
        off += std::max(A1, A2);
    }
    const int32_t lens[2] = {(int32_t)L1, (int32_t)L2};
    praline_arena *ar = nullptr;
    RC(praline_arena_create(2, lens, (int32_t)A, prof.data(), S.data(), &ar));
    DevBuf<float> d_m;
    int rc = d_m.alloc((size_t)(L1 * L2));
    if (rc == PRALINE_OK && (ar->wide || match_mode() == PRALINE_MATCH_REFERENCE)) {
        // the reference's own summation order (per track set), any alphabet: bit-identical to cext_build_scores
        std::vector<int32_t> sizes;
        for (int n = 0; n < num_sets; ++n) sizes.push_back((int32_t)std::max(i1s[n].dim[1], i2s[n].dim[1]));
        rc = praline_arena_set_track_sets(ar, num_sets, sizes.data());
        if (rc == PRALINE_OK) rc = arena_ensure_ref(ar);
        DevBuf<int32_t> d_pair, d_chunk;
        DevBuf<int64_t> d_off;
        if (rc == PRALINE_OK) rc = d_pair.upload(std::vector<int32_t>{0, 1}, g_rt.stream);
        if (rc == PRALINE_OK) rc = d_chunk.upload(std::vector<int32_t>{0}, g_rt.stream);
        if (rc == PRALINE_OK) rc = d_off.upload(std::vector<int64_t>{0}, g_rt.stream);
        if (rc == PRALINE_OK) rc = launch_match_ref(ar, d_pair.p, d_chunk.p, 1, (int)L1, d_off.p, d_m.p);
        if (rc == PRALINE_OK) {
            std::vector<float> hm((size_t)(L1 * L2));
            hipError_t e = hipMemcpyAsync(hm.data(), d_m.p, hm.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g_rt.stream);
            if (e != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "build_scores: %s", hipGetErrorString(e));
            else scatter2<float>(hm, *m);
        }
    } else if (rc == PRALINE_OK) {
        dim3 grid((unsigned)((L2 + 31) / 32), (unsigned)((L1 + 31) / 32));
        hipLaunchKernelGGL(k_scores_tile, grid, dim3(64), 0, g_rt.stream, ar->view(), 0, 1, ar->nstep, d_m.p);
        std::vector<float> hm((size_t)(L1 * L2));
        hipError_t e = hipMemcpyAsync(hm.data(), d_m.p, hm.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_rt.stream);
        if (e != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "build_scores: %s", hipGetErrorString(e));
        else scatter2<float>(hm, *m);
    }
    praline_arena_destroy(ar);
    return rc;
}

struct RawDev {
    DevBuf<float> m, g1, g2, o;
    DevBuf<uint8_t> t, z;
    int64_t L1 = 0, L2 = 0;
};

static int raw_upload(const praline_array *m, const praline_array *g1, const praline_array *g2, const praline_array *o,
                      const praline_array *t, const praline_array *z, RawDev &d)
{
    if (!arr_ok(m) || !arr_ok(g1) || !arr_ok(g2)) return fail(PRALINE_ERR_ARG, "NULL m / g1 / g2");
    const int64_t L1 = m->dim[0], L2 = m->dim[1];
    if (L1 <= 0 || L2 <= 0) return fail(PRALINE_ERR_ARG, "empty match score matrix");
    if (g1->dim[0] != L1 || g1->dim[1] != 2 || g2->dim[0] != L2 || g2->dim[1] != 2)
        return fail(PRALINE_ERR_ARG, "gap score arrays must be [L1][2] and [L2][2]");
    if (o && (o->dim[0] != L1 + 1 || o->dim[1] != L2 + 1 || o->dim[2] != 3)) return fail(PRALINE_ERR_ARG, "o must be [L1+1][L2+1][3]");
    if (t && (t->dim[0] != L1 + 1 || t->dim[1] != L2 + 1 || t->dim[2] != 3)) return fail(PRALINE_ERR_ARG, "t must be [L1+1][L2+1][3]");
    if (z && z->data && (z->dim[0] != L1 + 1 || z->dim[1] != L2 + 1)) return fail(PRALINE_ERR_ARG, "z must be [L1+1][L2+1]");
    RC(ensure_runtime(-1));
    d.L1 = L1; d.L2 = L2;
    hipStream_t st = g_rt.stream;
    std::vector<float> hm, hg1, hg2, ho;
    std::vector<uint8_t> ht, hz;
    gather2<float>(*m, hm); gather2<float>(*g1, hg1); gather2<float>(*g2, hg2);
    RC(d.m.upload(hm, st)); RC(d.g1.upload(hg1, st)); RC(d.g2.upload(hg2, st));
    const size_t cells = (size_t)((L1 + 1) * (L2 + 1));
    if (o) { gather3<float>(*o, ho); RC(d.o.upload(ho, st)); } else RC(d.o.alloc(cells * 3));
    if (t) { gather3<uint8_t>(*t, ht); RC(d.t.upload(ht, st)); } else RC(d.t.alloc(cells * 3));
    if (z && z->data) { gather2<uint8_t>(*z, hz); RC(d.z.upload(hz, st)); }
    else { RC(d.z.alloc(cells)); HIPCHK(hipMemsetAsync(d.z.p, 0, cells, st)); }
    HIPCHK(hipStreamSynchronize(st));
    return PRALINE_OK;
}

extern "C" int praline_align(int mode, const praline_array *m, const praline_array *g1, const praline_array *g2,
                             const praline_array *o, const praline_array *t, const praline_array *z)
{
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!arr_ok(o) || !arr_ok(t) || !arr_ok(z)) return fail(PRALINE_ERR_ARG, "NULL o / t / z");
    RawDev d;
    RC(raw_upload(m, g1, g2, o, t, z, d));
    hipLaunchKernelGGL(k_raw_align, dim3(1), dim3(64 * (unsigned)std::min<int64_t>(PRALINE_RAW_WAVES, std::max<int64_t>(1, (d.L2 + 63) / 64))), 0, g_rt.stream, mode == PRALINE_MODE_LOCAL ? 1 : 0, d.m.p, d.g1.p,
                       d.g2.p, d.o.p, d.t.p, d.z.p, (int)d.L1, (int)d.L2);
    HIPCHK(hipGetLastError());
    const size_t cells = (size_t)((d.L1 + 1) * (d.L2 + 1));
    std::vector<float> ho(cells * 3);
    std::vector<uint8_t> ht(cells * 3);
    HIPCHK(hipMemcpyAsync(ho.data(), d.o.p, ho.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipMemcpyAsync(ht.data(), d.t.p, ht.size(), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    scatter3<float>(ho, *o);
    scatter3<uint8_t>(ht, *t);
    return PRALINE_OK;
}

extern "C" int praline_align_global(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                    const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_GLOBAL, m, g1, g2, o, t, z); }
extern "C" int praline_align_local(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                   const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_LOCAL, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_both(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                             const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_BOTH, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_one(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                            const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_ONE, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_two(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                            const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_TWO, m, g1, g2, o, t, z); }

extern "C" int praline_raw_align(int mode, const praline_array *m, const praline_array *g1, const praline_array *g2,
                                 const praline_array *z, float *score, int32_t *path, int64_t *path_rows)
{
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!score || !path || !path_rows) return fail(PRALINE_ERR_ARG, "NULL output");
    RawDev d;
    RC(raw_upload(m, g1, g2, nullptr, nullptr, z, d));
    hipStream_t st = g_rt.stream;
    const int L1 = (int)d.L1, L2 = (int)d.L2;
    hipLaunchKernelGGL(k_raw_init, dim3(256), dim3(256), 0, st, mode, d.g1.p, d.g2.p, d.o.p, d.t.p, L1, L2);
    hipLaunchKernelGGL(k_raw_align, dim3(1), dim3(64 * (unsigned)std::min<int64_t>(PRALINE_RAW_WAVES, std::max<int64_t>(1, ((int64_t)L2 + 63) / 64))), 0, st, mode == PRALINE_MODE_LOCAL ? 1 : 0, d.m.p, d.g1.p, d.g2.p,
                       d.o.p, d.t.p, d.z.p, L1, L2);
    DevBuf<float> d_score;
    DevBuf<int32_t> d_path;
    DevBuf<int64_t> d_info;
    const size_t cap = (size_t)(L1 + L2 + 2);
    RC(d_score.alloc(1)); RC(d_path.alloc(cap * 2)); RC(d_info.alloc(2));
    hipLaunchKernelGGL(k_raw_trace, dim3(1), dim3(256), 0, st, mode, d.o.p, d.t.p, L1, L2, d_score.p, d_path.p, d_info.p);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> hp(cap * 2);
    int64_t info[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(score, d_score.p, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hp.data(), d_path.p, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(info, d_info.p, sizeof(info), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (info[1] <= 0 || info[0] < 0 || (size_t)(info[0] + info[1]) > cap) return fail(PRALINE_ERR_DEVICE, "traceback produced an invalid path");
    memcpy(path, hp.data() + 2 * info[0], (size_t)info[1] * 2 * sizeof(int32_t));
    *path_rows = info[1];
    return PRALINE_OK;
}

// --------------------------------------------------------------------------------------------
// debug: the per-lane match-score tile exactly as the fp32 MFMA chain forms it (NSTEP = arena.nstep via a
// runtime loop).  out: [64][32] floats.  Not part of the public header.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_tile(ArenaDev ar, const int32_t *lane_one, int two0, int two1, int x0,
                                                    int y, int tp, int nstep, float *out)
{
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const int srcA = lane_one[j], srcB = lane_one[32 + j];
    const float *pA = ar.P + ((int64_t)(srcA >= 0 ? ar.row_off[srcA] : 0) + (y - 1)) * ar.KP + half * ar.KS;
    const float *pB = ar.P + ((int64_t)(srcB >= 0 ? ar.row_off[srcB] : 0) + (y - 1)) * ar.KP + half * ar.KS;
    const float *qA = ar.Q + ((int64_t)ar.row_off[two0] + x0 + j) * ar.KP + half * ar.KS;
    const float *qB = ar.Q + ((int64_t)ar.row_off[two1] + x0 + j) * ar.KP + half * ar.KS;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
    for (int k = 0; k < nstep; ++k) {
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[k], pA[k], accA, 0, 0, 0);
        if (tp == 2) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(qB[k], pB[k], accB, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = accA[r], b = (tp == 2) ? accB[r] : 0.0f;
        swap_halves(a, b);
        out[lane * 32 + 8 * (r >> 2) + (r & 3)] = a;
        out[lane * 32 + 8 * (r >> 2) + 4 + (r & 3)] = b;
    }
}

extern "C" int praline_debug_tile(praline_arena *arena, const int32_t *lane_one, int two0, int two1, int x0, int y, int tp,
                                  float *out)
{
    RC(arena_ready(arena));
    RC(ensure_runtime(-1));
    DevBuf<int32_t> d_l;
    DevBuf<float> d_o;
    std::vector<int32_t> lv(lane_one, lane_one + 64);
    RC(d_l.upload(lv, g_rt.stream));
    RC(d_o.alloc(64 * 32));
    hipLaunchKernelGGL(k_debug_tile, dim3(1), dim3(64), 0, g_rt.stream, arena->view(), d_l.p, two0, two1, x0, y, tp,
                       arena->nstep, d_o.p);
    HIPCHK(hipMemcpyAsync(out, d_o.p, 64 * 32 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

// --------------------------------------------------------------------------------------------
// diagnostics: the dense match-score matrix of one arena pair exactly as the kernels evaluate it
// --------------------------------------------------------------------------------------------
extern "C" int praline_arena_match_scores(praline_arena *arena, int32_t one, int32_t two, int kind, float *m)
{
    if (!arena || !m) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (one < 0 || one >= arena->n_seqs || two < 0 || two >= arena->n_seqs) return fail(PRALINE_ERR_ARG, "index out of range");
    RC(ensure_runtime(-1));
    const int L1 = arena->len[one], L2 = arena->len[two];
    DevBuf<float> d_m;
    RC(d_m.alloc((size_t)L1 * L2));
    if (kind == 2) {   // the reference's summation order (what PRALINE_MATCH_REFERENCE plans and wide arenas use)
        RC(arena_ensure_ref(arena));
        DevBuf<int32_t> d_pair, d_chunk;
        DevBuf<int64_t> d_off;
        RC(d_pair.upload(std::vector<int32_t>{one, two}, g_rt.stream));
        RC(d_chunk.upload(std::vector<int32_t>{0}, g_rt.stream));
        RC(d_off.upload(std::vector<int64_t>{0}, g_rt.stream));
        RC(launch_match_ref(arena, d_pair.p, d_chunk.p, 1, L1, d_off.p, d_m.p));
        HIPCHK(hipMemcpyAsync(m, d_m.p, (size_t)L1 * L2 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
        HIPCHK(hipStreamSynchronize(g_rt.stream));
        return PRALINE_OK;
    }
    if (arena->wide) return fail(PRALINE_ERR_UNSUPPORTED, "this arena has more than 32 active symbols: only kind 2 (reference order) exists");
    if (kind == 0) {
        dim3 grid((unsigned)((L2 + 31) / 32), (unsigned)((L1 + 31) / 32));
        hipLaunchKernelGGL(k_scores_tile, grid, dim3(64), 0, g_rt.stream, arena->view(), one, two, arena->nstep, d_m.p);
    } else if (kind == 1) {
        if (arena->nr16 == 0) return fail(PRALINE_ERR_UNSUPPORTED, "no f16 operands for this arena");
        int rc = praline_launch_scores_tile16(arena->view16(), arena->nr16, arena->nterm16, one, two, L1, L2, d_m.p, g_rt.stream);
        if (rc != PRALINE_OK) return fail(rc, "no k_scores_tile16 instance");
    } else return fail(PRALINE_ERR_ARG, "kind must be 0 (fp32 MFMA chain), 1 (f16 split) or 2 (reference order)");
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m, d_m.p, (size_t)L1 * L2 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_info(const praline_arena *arena, int32_t *n_active, int32_t *mfma_steps_f32, int32_t *f16_ranges,
                                  int32_t *f16_terms)
{
    RC(arena_ready(arena));
    if (n_active) *n_active = arena->n_active;
    if (mfma_steps_f32) *mfma_steps_f32 = arena->nstep;
    if (f16_ranges) *f16_ranges = arena->nr16;
    if (f16_terms) *f16_terms = arena->nterm16;
    return PRALINE_OK;
}

extern "C" int praline_plan_tile_producer(const praline_plan *plan)
{
    if (!plan) return -1;
    if (plan->dense_kind == 1 && plan->arena->reft2_state != 1) return 2;   // (the arena no longer qualifies for k_match_tile)
    return plan->dense_kind;
}

// Which match-score arithmetic praline_plan_run uses for this plan: 0 = fp32 MFMA chain, 1 = f16 split.
extern "C" int praline_plan_match_kind(const praline_plan *plan)
{
    if (!plan) return -1;
    if (plan->dense_kind == 3) return 0;   // (per-position gap plans: the fp32 MFMA chain for both kinds of run)
    if (plan->dense_kind != 0) return 2;
    if (plan->split && plan->arena->nr16 > 0) {
        if (plan->want_paths) return 1;  // k_dp_split16_tb
        if (match_mode() != PRALINE_MATCH_F32) return 1;
    }
    return 0;
}
