// praline_rawb.hip.h -- part of praline_dp.hip (one translation unit; included there): a batch of RawPairwiseAligner requests
// (praline/component/align.py:254-447) resident on the device - the requests' own m, g1, g2 and zero cells, one launch of
// k_rawb_fill (dp_rawb.hip.h) for all of them, scores and paths back in one copy each.
#include "dp_rawb.h"
#include <memory>
#include <numeric>

struct praline_raw_batch {
    int64_t n = 0, cells = 0, path_rows_cap = 0;
    int waves = 1;
    bool mask = false;
    bool modes_dirty = true;
    std::vector<RawReq> reqs;          // launch order
    std::vector<int32_t> place;        // request index -> position in reqs
    DevBuf<RawReq> d_reqs;
    DevBuf<float> d_m, d_scores;
    DevBuf<float2> d_g1, d_g2;
    DevBuf<uint16_t> d_z;
    DevBuf<float4> d_top, d_wrap, d_edge, d_best;
    DevBuf<uint8_t> d_t;
    DevBuf<int32_t> d_paths, d_error, d_zero_req, d_zero_idx;
    DevBuf<int64_t> d_info;
    int64_t n_zero = 0;
    bool ran = false;
    float last_ms = 0.0f;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<int64_t> h_info;
    ~praline_raw_batch()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
    RawBatchDev view()
    {
        RawBatchDev d;
        d.reqs = d_reqs.p; d.n = (int)n; d.m = d_m.p; d.g1 = d_g1.p; d.g2 = d_g2.p; d.z = d_z.p; d.top = d_top.p; d.wrap = d_wrap.p;
        d.edge = d_edge.p; d.best = d_best.p; d.t = d_t.p; d.paths = d_paths.p; d.path_info = d_info.p; d.scores = d_scores.p;
        d.error = d_error.p;
        return d;
    }
};

// the requests' arrays: one after the other (m, g1, g2) or one pointer per request (mv, g1v, g2v)
struct RawSource {
    const float *m = nullptr, *g1 = nullptr, *g2 = nullptr;
    const float *const *mv = nullptr, *const *g1v = nullptr, *const *g2v = nullptr;
};

static int raw_batch_create_impl(int64_t n, const int32_t *l1, const int32_t *l2, const RawSource &src, const int64_t *zero_off,
                                 const int32_t *zero_idx, praline_raw_batch **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    const bool listed = src.mv != nullptr;
    if (n <= 0 || n > (int64_t)INT32_MAX || !l1 || !l2 || (listed ? (!src.g1v || !src.g2v) : (!src.m || !src.g1 || !src.g2)))
        return fail(PRALINE_ERR_ARG, "bad raw batch arguments");
    if (listed)
        for (int64_t r = 0; r < n; ++r)
            if (!src.mv[r] || !src.g1v[r] || !src.g2v[r]) return fail(PRALINE_ERR_ARG, "request %lld: NULL array", (long long)r);
    if ((zero_off != nullptr) != (zero_idx != nullptr) && zero_off && zero_off[n] > 0) return fail(PRALINE_ERR_ARG, "zero_off given without zero_idx");
    for (int64_t r = 0; r < n; ++r) {
        if (l1[r] < 1 || l2[r] < 1) return fail(PRALINE_ERR_ARG, "request %lld: m has shape %d x %d (both must be >= 1)", (long long)r, l1[r], l2[r]);
        if ((int64_t)l1[r] * l2[r] > ((int64_t)1 << 30)) return fail(PRALINE_ERR_UNSUPPORTED, "request %lld: more than 2^30 cells", (long long)r);
        if (zero_off && zero_off[r + 1] < zero_off[r]) return fail(PRALINE_ERR_ARG, "zero_off is not ascending at request %lld", (long long)r);
    }
    RC(ensure_runtime(-1));
    std::unique_ptr<praline_raw_batch> b(new praline_raw_batch());
    b->n = n;
    // launch order: the requests with the most cells first
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return (int64_t)l1[x] * l2[x] > (int64_t)l1[y] * l2[y]; });
    std::vector<int64_t> m_off((size_t)n + 1, 0), g1_off((size_t)n + 1, 0), g2_off((size_t)n + 1, 0);
    for (int64_t r = 0; r < n; ++r) {
        m_off[(size_t)r + 1] = m_off[(size_t)r] + (int64_t)l1[r] * l2[r];
        g1_off[(size_t)r + 1] = g1_off[(size_t)r] + l1[r];
        g2_off[(size_t)r + 1] = g2_off[(size_t)r] + l2[r];
    }
    b->cells = m_off[(size_t)n];
    b->reqs.resize((size_t)n);
    b->place.resize((size_t)n);
    int64_t n_blocks = 0, rows4 = 0, edge4 = 0, path_rows = 0, best4 = 0;
    std::vector<int64_t> block0((size_t)n);
    int max_strips = 1;
    for (int64_t q = 0; q < n; ++q) {
        const int32_t r = order[(size_t)q];
        RawReq &rq = b->reqs[(size_t)q];
        b->place[(size_t)r] = (int32_t)q;
        rq.L1 = l1[r]; rq.L2 = l2[r]; rq.mode = 0; rq.nstrips = (l1[r] + 63) / 64;
        rq.ncs = (l2[r] + 63 + 15) / 16 + 1; rq.pad0 = 0;
        rq.index = r; rq.pad1 = 0;
        const int64_t blocks = (int64_t)rq.nstrips * rq.ncs;     // (strip, chunk) blocks of the request
        block0[(size_t)q] = n_blocks;
        rq.m_off = n_blocks * 256;
        rq.src_off = m_off[(size_t)r];
        rq.g1_off = g1_off[(size_t)r]; rq.g2_off = g2_off[(size_t)r];
        rq.t_off = n_blocks * 64;
        rq.z_off = n_blocks * 64;
        n_blocks += blocks;
        rq.top_off = rows4; rows4 += l2[r] + PRALINE_RAWB_ROW_PAD;
        rq.edge_off = edge4; edge4 += l1[r] + l2[r] + 2;
        rq.best_off = best4; best4 += (int64_t)std::min(rq.nstrips, PRALINE_RAWB_WAVES) * 64;   // (a record per lane of the waves that take part)
        rq.path_off = path_rows; path_rows += l1[r] + l2[r] + 2;
        max_strips = std::max(max_strips, rq.nstrips);
    }
    b->waves = std::min(PRALINE_RAWB_WAVES, max_strips);
    b->path_rows_cap = path_rows;
    b->n_zero = zero_off ? zero_off[n] - zero_off[0] : 0;
    b->mask = b->n_zero > 0;
    hipStream_t st = g_rt.stream;
    if (n_blocks > (int64_t)1 << 31) return fail(PRALINE_ERR_UNSUPPORTED, "raw batch of %lld chunk blocks (split the list)", (long long)n_blocks);
    RC(b->d_m.alloc((size_t)n_blocks * 1024));
    RC(b->d_g1.alloc((size_t)(g1_off[(size_t)n] + PRALINE_RAWB_ROW_PAD)));
    RC(b->d_g2.alloc((size_t)(g2_off[(size_t)n] + PRALINE_RAWB_ROW_PAD)));
    RC(b->d_z.alloc((size_t)std::max<int64_t>(b->mask ? n_blocks * 64 : 0, 2)));
    RC(b->d_top.alloc((size_t)rows4));
    RC(b->d_wrap.alloc(max_strips > PRALINE_RAWB_WAVES ? (size_t)rows4 : 1));
    RC(b->d_edge.alloc((size_t)edge4));
    RC(b->d_best.alloc((size_t)best4));
    RC(b->d_t.alloc((size_t)n_blocks * 1024));
    RC(b->d_paths.alloc((size_t)path_rows * 2));
    RC(b->d_info.alloc((size_t)n * 2));
    RC(b->d_scores.alloc((size_t)n));
    RC(b->d_error.alloc(1));
    RC(b->d_reqs.alloc((size_t)n));
    // the inputs: one copy each (host or device pointers).  The padding behind g1 / g2 is read by the prefetches and never
    // used: it only has to exist - it is cleared once so that no run reads uninitialised memory.
    HIPCHK(hipMemsetAsync(b->d_g1.p + g1_off[(size_t)n], 0, PRALINE_RAWB_ROW_PAD * sizeof(float2), st));
    HIPCHK(hipMemsetAsync(b->d_g2.p + g2_off[(size_t)n], 0, PRALINE_RAWB_ROW_PAD * sizeof(float2), st));
    if (listed) {
        for (int64_t r = 0; r < n; ++r) {
            HIPCHK(hipMemcpyAsync(b->d_g1.p + g1_off[(size_t)r], src.g1v[r], (size_t)l1[r] * sizeof(float2), hipMemcpyDefault, st));
            HIPCHK(hipMemcpyAsync(b->d_g2.p + g2_off[(size_t)r], src.g2v[r], (size_t)l2[r] * sizeof(float2), hipMemcpyDefault, st));
        }
    } else {
        HIPCHK(hipMemcpyAsync(b->d_g1.p, src.g1, (size_t)g1_off[(size_t)n] * sizeof(float2), hipMemcpyDefault, st));
        HIPCHK(hipMemcpyAsync(b->d_g2.p, src.g2, (size_t)g2_off[(size_t)n] * sizeof(float2), hipMemcpyDefault, st));
    }
    HIPCHK(hipMemsetAsync(b->d_error.p, 0, sizeof(int32_t), st));
    HIPCHK(hipMemcpyAsync(b->d_reqs.p, b->reqs.data(), (size_t)n * sizeof(RawReq), hipMemcpyHostToDevice, st));
    {   // m: the caller's rows into the arena's aligned rows (dp_rawb.h)
        DevBuf<float> tmp;
        DevBuf<int64_t> d_block0;
        RC(tmp.alloc((size_t)b->cells));
        RC(d_block0.upload(block0, st));
        if (listed) {
            for (int64_t r = 0; r < n; ++r)
                HIPCHK(hipMemcpyAsync(tmp.p + m_off[(size_t)r], src.mv[r], (size_t)l1[r] * l2[r] * sizeof(float), hipMemcpyDefault, st));
        } else {
            HIPCHK(hipMemcpyAsync(tmp.p, src.m, (size_t)b->cells * sizeof(float), hipMemcpyDefault, st));
        }
        praline_launch_rawb_stage(b->view(), tmp.p, b->d_m.p, d_block0.p, n_blocks, st);
        HIPCHK(hipGetLastError());
    }
    if (b->mask) {
        // zero cells -> mask bits, on the device
        std::vector<int32_t> zreq((size_t)b->n_zero);
        for (int64_t r = 0; r < n; ++r)
            for (int64_t e = zero_off[r]; e < zero_off[r + 1]; ++e) zreq[(size_t)(e - zero_off[0])] = b->place[(size_t)r];
        RC(b->d_zero_req.upload(zreq, st));
        RC(b->d_zero_idx.alloc((size_t)b->n_zero * 2));
        HIPCHK(hipMemcpyAsync(b->d_zero_idx.p, zero_idx + 2 * zero_off[0], (size_t)b->n_zero * 2 * sizeof(int32_t), hipMemcpyDefault, st));
        HIPCHK(hipMemsetAsync(b->d_z.p, 0, (size_t)n_blocks * 64 * sizeof(uint16_t), st));
        praline_launch_rawb_zero(b->view(), b->d_zero_req.p, b->d_zero_idx.p, b->n_zero, st);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventCreate(&b->ev0));
    HIPCHK(hipEventCreate(&b->ev1));
    HIPCHK(hipStreamSynchronize(st));   // (the caller's buffers and the staging vectors above are free again)
    *out = b.release();
    return PRALINE_OK;
}

extern "C" int praline_raw_batch_create(int64_t n, const int32_t *l1, const int32_t *l2, const float *m, const float *g1, const float *g2,
                                        const int64_t *zero_off, const int32_t *zero_idx, praline_raw_batch **out)
{
    RawSource src;
    src.m = m; src.g1 = g1; src.g2 = g2;
    return raw_batch_create_impl(n, l1, l2, src, zero_off, zero_idx, out);
}

// the same with one pointer per request (m[r], g1[r], g2[r]: host or device memory each): no concatenation on the caller's side
extern "C" int praline_raw_batch_create_v(int64_t n, const int32_t *l1, const int32_t *l2, const float *const *m, const float *const *g1,
                                          const float *const *g2, const int64_t *zero_off, const int32_t *zero_idx, praline_raw_batch **out)
{
    if (!m) return fail(PRALINE_ERR_ARG, "bad raw batch arguments");
    RawSource src;
    src.mv = m; src.g1v = g1; src.g2v = g2;
    return raw_batch_create_impl(n, l1, l2, src, zero_off, zero_idx, out);
}

// One run of every request: `modes` [n] per request (PRALINE_MODE_*), or NULL and `mode` for all.  Asynchronous on the library
// stream; praline_raw_batch_results waits.
extern "C" int praline_raw_batch_run(praline_raw_batch *b, const int32_t *modes, int mode)
{
    if (!b) return fail(PRALINE_ERR_ARG, "batch is NULL");
    for (int64_t r = 0; r < b->n; ++r) {
        const int mo = modes ? modes[r] : mode;
        if (mo < 0 || mo > 4) return fail(PRALINE_ERR_ARG, "request %lld: unknown alignment mode %d", (long long)r, mo);
        RawReq &rq = b->reqs[(size_t)b->place[(size_t)r]];
        if (rq.mode != mo) { rq.mode = mo; b->modes_dirty = true; }
    }
    RC(ensure_runtime(-1));
    hipStream_t st = g_rt.stream;
    if (b->modes_dirty) {
        HIPCHK(hipMemcpyAsync(b->d_reqs.p, b->reqs.data(), (size_t)b->n * sizeof(RawReq), hipMemcpyHostToDevice, st));
        b->modes_dirty = false;
    }
    const RawBatchDev d = b->view();
    HIPCHK(hipEventRecord(b->ev0, st));
    praline_launch_rawb_init(d, st);
    bool any_local = false, any_other = false;
    for (const RawReq &rq : b->reqs) { any_local = any_local || rq.mode == 1; any_other = any_other || rq.mode != 1; }
    praline_launch_rawb_fill(d, b->waves, b->mask, any_other, any_local, st);
    praline_launch_rawb_trace(d, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev1, st));
    b->ran = true;
    return PRALINE_OK;
}

// scores [n] and the row count of every path [n] (either may be NULL); waits for the run
extern "C" int praline_raw_batch_results(praline_raw_batch *b, float *scores, int64_t *path_rows)
{
    if (!b) return fail(PRALINE_ERR_ARG, "batch is NULL");
    if (!b->ran) return fail(PRALINE_ERR_ARG, "praline_raw_batch_results before praline_raw_batch_run");
    hipStream_t st = g_rt.stream;
    int32_t err = 0;
    b->h_info.resize((size_t)b->n * 2);
    if (scores) HIPCHK(hipMemcpyAsync(scores, b->d_scores.p, (size_t)b->n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(b->h_info.data(), b->d_info.p, (size_t)b->n * 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&err, b->d_error.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1);
    if (err) return fail(PRALINE_ERR_DEVICE, "k_rawb_fill: a wave gave up waiting for its neighbour strip (internal error)");
    if (path_rows)
        for (int64_t r = 0; r < b->n; ++r) path_rows[r] = b->h_info[(size_t)(2 * r + 1)];
    return PRALINE_OK;
}

// the paths, one after the other in request order: int32 [sum of path_rows][2] (call praline_raw_batch_results first)
extern "C" int praline_raw_batch_paths(praline_raw_batch *b, int32_t *paths, int64_t cap_rows)
{
    if (!b || !paths) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (!b->ran || b->h_info.size() != (size_t)b->n * 2) return fail(PRALINE_ERR_ARG, "praline_raw_batch_paths before praline_raw_batch_results");
    int64_t total = 0;
    for (int64_t r = 0; r < b->n; ++r) total += b->h_info[(size_t)(2 * r + 1)];
    if (cap_rows < total) return fail(PRALINE_ERR_ARG, "paths holds %lld rows, %lld needed", (long long)cap_rows, (long long)total);
    std::vector<int32_t> all((size_t)b->path_rows_cap * 2);
    HIPCHK(hipMemcpy(all.data(), b->d_paths.p, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t w = 0;
    for (int64_t r = 0; r < b->n; ++r) {
        const RawReq &rq = b->reqs[(size_t)b->place[(size_t)r]];
        const int64_t first = b->h_info[(size_t)(2 * r)], rows = b->h_info[(size_t)(2 * r + 1)];
        memcpy(paths + 2 * w, all.data() + 2 * (rq.path_off + first), (size_t)rows * 2 * sizeof(int32_t));
        w += rows;
    }
    return PRALINE_OK;
}

extern "C" int64_t praline_raw_batch_cells(const praline_raw_batch *b) { return b ? b->cells : 0; }

// device time of the last run (init + fill + end cells and paths), valid after praline_raw_batch_results
extern "C" int praline_raw_batch_last_timing(const praline_raw_batch *b, float *kernel_ms)
{
    if (!b || !kernel_ms) return fail(PRALINE_ERR_ARG, "NULL argument");
    *kernel_ms = b->last_ms;
    return PRALINE_OK;
}

extern "C" void praline_raw_batch_destroy(praline_raw_batch *b)
{
    if (!b) return;
    if (g_rt.ready) (void)hipStreamSynchronize(g_rt.stream);
    delete b;
}
