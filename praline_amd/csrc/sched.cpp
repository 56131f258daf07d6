// sched.cpp -- pair list -> wavefront tasks -> launch order and workgroup descriptors.  Pure host logic.
//
// 1. Pairs are grouped by their sequence TWO (a task's 32 pairs share it: it is the A operand of the MFMA tile),
//    sorted by the length of sequence one inside a group and cut into 32-pair half tasks.
// 2. Tasks are ordered longest first and placed on the XCDs in groups of neighbours (same partners -> same
//    operand rows in that XCD's L2).
// 3. Small batches get four-wave workgroups whose waves share long tasks; large batches a list with four
//    independent tasks per workgroup.
#include "sched.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <chrono>
#include <mutex>
#include <atomic>
#include <pthread.h>
#include <condition_variable>
#include <functional>
#include <memory>

// ---- large scratch and schedule blocks, kept between calls (sched.h: NoInitAlloc) ----
namespace {
struct BigBlocks {
    struct Block { void *p; size_t cap; };
    std::mutex m;
    std::vector<Block> kept, lent;   // kept: oldest first
    size_t kept_bytes = 0, limit = (size_t)256 << 20;
    BigBlocks()
    {
        if (const char *env = getenv("PRALINE_SCHED_CACHE_MB")) limit = (size_t)std::max(0ll, atoll(env)) << 20;
        if (const char *env = getenv("PRALINE_KEEP_HOST_MEMORY"))
            if (env[0] == '0') limit = 0;
    }
    static BigBlocks &get() { static BigBlocks *g = new BigBlocks(); return *g; }   // (never destroyed: blocks may outlive main)
};
}  // namespace

void *sched_big_alloc(size_t bytes)
{
    BigBlocks &g = BigBlocks::get();
    {
        std::lock_guard<std::mutex> lk(g.m);
        // the smallest kept block that holds the request without being half as large again
        int best = -1;
        for (size_t i = 0; i < g.kept.size(); ++i)
            if (g.kept[i].cap >= bytes && g.kept[i].cap <= bytes + bytes / 2 && (best < 0 || g.kept[i].cap < g.kept[(size_t)best].cap)) best = (int)i;
        if (best >= 0) {
            const BigBlocks::Block b = g.kept[(size_t)best];
            g.kept.erase(g.kept.begin() + best);
            g.kept_bytes -= b.cap;
            g.lent.push_back(b);
            return b.p;
        }
    }
    const size_t unit = (size_t)2 << 20;
    const size_t cap = (bytes + unit - 1) / unit * unit;
    void *p = ::operator new(cap);
    std::lock_guard<std::mutex> lk(g.m);
    g.lent.push_back(BigBlocks::Block{p, cap});
    return p;
}

void sched_big_free(void *p)
{
    if (!p) return;
    BigBlocks &g = BigBlocks::get();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g.m);
        size_t i = 0;
        while (i < g.lent.size() && g.lent[i].p != p) ++i;
        if (i == g.lent.size()) { drop.push_back(p); }   // (not one of ours: cannot happen through NoInitAlloc)
        else {
            const BigBlocks::Block b = g.lent[i];
            g.lent.erase(g.lent.begin() + (long)i);
            if (b.cap > g.limit) drop.push_back(b.p);
            else {
                // make room: the oldest kept blocks go first
                while (g.kept_bytes + b.cap > g.limit && !g.kept.empty()) {
                    drop.push_back(g.kept.front().p);
                    g.kept_bytes -= g.kept.front().cap;
                    g.kept.erase(g.kept.begin());
                }
                g.kept.push_back(b);
                g.kept_bytes += b.cap;
            }
        }
    }
    for (void *q : drop) ::operator delete(q);
}

namespace {

// Host threads of the scheduler's passes over a large pair list (the counting and scatter loops are independent per slice
// of the list).  Small lists stay on the calling thread; PRALINE_SCHED_THREADS overrides (1: serial).
int sched_threads(int64_t n_pairs)
{
    if (n_pairs < (1 << 16)) return 1;
    int hw = (int)std::thread::hardware_concurrency();
    if (const char *env = getenv("PRALINE_SCHED_THREADS")) hw = atoi(env);
    return std::max(1, std::min(16, hw));
}

// A small persistent pool (created on first use, lives as long as the library): a pass over the pair list is a few
// hundred microseconds of work per thread, which spawning threads per pass would eat.  One caller at a time (the library is
// for one host thread per device; praline_sched_prepare may run beside it on another thread: the mutex serialises them).
// A schedule is fifteen to twenty such passes back to back, and waking fifteen sleepers through one condition variable cost
// 100-250 us per pass - as much as the passes themselves for a million pairs: a worker therefore spins on the generation
// counter for a short while after a pass (the next one usually follows within microseconds) before it goes to sleep, and
// the caller spins as briefly for the last worker.  EVERY pool thread acknowledges every generation (those beyond the
// pass's thread count without doing work), so the job description is never replaced while a worker may still read it.
class SchedPool {
public:
    static SchedPool &get()
    {
        static SchedPool p;
        static const bool once = (pthread_atfork(nullptr, nullptr, [] { SchedPool::get().after_fork(); }), true);
        (void)once;
        return p;
    }
    void warm(int nt)
    {
        std::lock_guard<std::mutex> whole(one_caller_);
        ensure(nt - 1);
    }
    void run(int nt, const std::function<void(int, int)> &fn)
    {
        std::lock_guard<std::mutex> whole(one_caller_);
        ensure(nt - 1);
        job_ = &fn; job_nt_ = nt;
        pending_.store((int)th_.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);   // (under the mutex: a worker between its predicate and its wait cannot miss it)
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        fn(0, nt);
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0;) {
            if (++spins < kSpins) { cpu_relax(); continue; }
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }
private:
    // In a forked child the pool's threads do not exist (praline_init starts them, so a host program that forks workers after
    // it would otherwise wait for their acknowledgements for ever): forget them - the thread objects are leaked, they
    // cannot be joined - and start new ones on demand.
    void after_fork()
    {
        new std::vector<std::thread>(std::move(th_));   // (leaked on purpose)
        th_.clear();
        pending_.store(0, std::memory_order_relaxed);
        job_ = nullptr;
        // (the condition variables count waiters that do not exist here - a broadcast would wait for them to leave -, and a
        // mutex may have been held by one of them: fresh ones in place)
        new (&cv_) std::condition_variable();
        new (&done_) std::condition_variable();
        new (&m_) std::mutex();
        new (&one_caller_) std::mutex();
    }
    static constexpr int kSpins = 20000;   // ~100-200 us of polling
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    ~SchedPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true, std::memory_order_relaxed); gen_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void ensure(int n)
    {
        while ((int)th_.size() < n) {
            const int id = (int)th_.size() + 1;
            const uint64_t seen0 = gen_.load(std::memory_order_acquire);
            th_.emplace_back([this, id, seen0]() {
                uint64_t seen = seen0;
                for (;;) {
                    uint64_t g = seen;
                    for (int spins = 0; (g = gen_.load(std::memory_order_acquire)) == seen && spins < kSpins; ++spins) cpu_relax();
                    if (g == seen) {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                        g = gen_.load(std::memory_order_acquire);
                    }
                    seen = g;
                    if (stop_.load(std::memory_order_relaxed)) return;
                    // (job_ / job_nt_ were written before the generation was published and stay until every thread has acknowledged)
                    const std::function<void(int, int)> *job = job_;
                    const int nt = job_nt_;
                    if (job && id < nt) (*job)(id, nt);
                    if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                        std::lock_guard<std::mutex> lk(m_);
                        done_.notify_one();
                    }
                }
            });
        }
    }
    std::mutex one_caller_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int, int)> *job_ = nullptr;
    int job_nt_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace

void sched_warm_threads() { SchedPool::get().warm(sched_threads((int64_t)1 << 20)); }

namespace {

template <class F> void run_threads(int nt, F &&fn)   // fn(thread, n_threads)
{
    if (nt <= 1) { fn(0, 1); return; }
    const std::function<void(int, int)> f = [&fn](int t, int n) { fn(t, n); };
    SchedPool::get().run(nt, f);
}

inline void slice_of(int64_t n, int t, int nt, int64_t &lo, int64_t &hi)
{
    lo = n * t / nt;
    hi = n * (t + 1) / nt;
}


struct HalfTask {
    int32_t two;
    int32_t max_l1;
    int32_t one[32];
    int32_t pair[32];
};

HalfTask empty_half(int32_t two)
{
    HalfTask h;
    h.two = two;
    h.max_l1 = 0;
    for (int q = 0; q < 32; ++q) { h.one[q] = -1; h.pair[q] = -1; }
    return h;
}

// step 1: group by sequence two, sort by len(one) descending, cut into 32-lane half tasks
RawVec<HalfTask> cut_half_tasks(const int32_t *lens, int64_t n_pairs, const int32_t *pairs, std::vector<int32_t> &idx, int width = 32)
{
    // order = pair indices by (sequence two ascending, length of sequence one descending, index ascending): two stable
    // counting sorts, least significant key first - by length (descending), then by sequence two.  (A comparison sort of
    // the whole list cost 15 of the 19 ms of a 261 632-pair plan, per-group std::stable_sort still 40 of the 98 ms of
    // C3's 1 047 552 pairs; this is linear.)
    const int nt = sched_threads(n_pairs);
    const bool timing = getenv("PRALINE_SCHED_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sched]   %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    RawVec<int32_t> order((size_t)n_pairs);   // (pair indices: a pair list holds fewer than 2^31 pairs)
    int32_t max_two = -1, max_len = 0;
    {
        std::vector<int32_t> mt((size_t)nt, -1), ml((size_t)nt, 0);
        run_threads(nt, [&](int t, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, t, n, lo, hi);
            int32_t a = -1, b = 0;
            for (int64_t i = lo; i < hi; ++i) { a = std::max(a, pairs[2 * i + 1]); b = std::max(b, lens[pairs[2 * i]]); }
            mt[(size_t)t] = a; ml[(size_t)t] = b;
        });
        for (int t = 0; t < nt; ++t) { max_two = std::max(max_two, mt[(size_t)t]); max_len = std::max(max_len, ml[(size_t)t]); }
    }
    // a stable counting sort whose passes run per slice of the input: thread t's share of bucket b starts after the shares
    // of the threads before it, so equal keys keep their input order whatever the number of threads
    auto counting_sort = [&](int64_t n_buckets, auto key_of, auto item_of, auto put) {
        std::vector<std::vector<int64_t>> cnt((size_t)nt);
        run_threads(nt, [&](int t, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, t, n, lo, hi);
            std::vector<int64_t> &c = cnt[(size_t)t];
            c.assign((size_t)n_buckets, 0);
            for (int64_t q = lo; q < hi; ++q) ++c[(size_t)key_of(item_of(q))];
        });
        int64_t run = 0;
        for (int64_t b = 0; b < n_buckets; ++b)
            for (int t = 0; t < nt; ++t) { const int64_t c = cnt[(size_t)t][(size_t)b]; cnt[(size_t)t][(size_t)b] = run; run += c; }
        run_threads(nt, [&](int t, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, t, n, lo, hi);
            std::vector<int64_t> &c = cnt[(size_t)t];
            for (int64_t q = lo; q < hi; ++q) { const int64_t i = item_of(q); put(c[(size_t)key_of(i)]++, i); }
        });
    };
    {
        RawVec<int32_t> by_len((size_t)n_pairs);
        // bucket b = max_len - len: longer first
        counting_sort((int64_t)max_len + 1, [&](int64_t i) { return (int64_t)(max_len - lens[pairs[2 * i]]); },
                      [&](int64_t q) { return q; }, [&](int64_t at, int64_t i) { by_len[(size_t)at] = (int32_t)i; });
        counting_sort((int64_t)max_two + 1, [&](int64_t i) { return (int64_t)pairs[2 * i + 1]; },
                      [&](int64_t q) { return (int64_t)by_len[(size_t)q]; }, [&](int64_t at, int64_t i) { order[(size_t)at] = (int32_t)i; });
    }
    mark("two counting sorts");
    // half tasks: every sequence two's run of `order` in pieces of 32, the columns cut on several threads
    std::vector<int64_t> col_start;      // first position in `order` of every run of equal sequence two, and the end
    {
        std::vector<std::vector<int64_t>> part((size_t)nt);
        run_threads(nt, [&](int t, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, t, n, lo, hi);
            for (int64_t q = lo; q < hi; ++q)
                if (q == 0 || pairs[2 * order[(size_t)q] + 1] != pairs[2 * order[(size_t)q - 1] + 1]) part[(size_t)t].push_back(q);
        });
        for (const std::vector<int64_t> &v : part) col_start.insert(col_start.end(), v.begin(), v.end());
        col_start.push_back(n_pairs);
    }
    const int64_t n_cols = (int64_t)col_start.size() - 1;
    std::vector<int64_t> half0((size_t)n_cols + 1, 0);
    for (int64_t c = 0; c < n_cols; ++c) half0[(size_t)c + 1] = half0[(size_t)c] + (col_start[(size_t)c + 1] - col_start[(size_t)c] + width - 1) / width;
    RawVec<HalfTask> halves((size_t)half0[(size_t)n_cols]);   // (every entry is written below)
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_cols, t, n, lo, hi);
        for (int64_t c = lo; c < hi; ++c) {
            int64_t i = col_start[(size_t)c];
            const int64_t end = col_start[(size_t)c + 1];
            const int32_t two = pairs[2 * order[(size_t)i] + 1];
            for (int64_t hix = half0[(size_t)c]; i < end; ++hix) {
                HalfTask h = empty_half(two);
                int k = 0;
                for (; i < end && k < width; ++i, ++k) {
                    h.one[k] = pairs[2 * order[(size_t)i]];
                    h.pair[k] = (int32_t)order[(size_t)i];
                    h.max_l1 = std::max(h.max_l1, lens[h.one[k]]);
                }
                halves[(size_t)hix] = h;
            }
        }
    });
    mark("halves cut");
    // longest work first - by strips of the sequence two, then by the longest sequence one, ties in creation order -; equal-
    // shaped halves end up adjacent (paired into one wave when TP = 2).  Two stable counting sorts of the INDICES (least
    // significant key first); the 264-byte structs are gathered once, by build_schedule.
    idx.resize(halves.size());
    {
        int max_l = 0, max_s = 0;
        for (const HalfTask &h : halves) { max_l = std::max(max_l, (int)h.max_l1); max_s = std::max(max_s, (lens[h.two] + 31) / 32); }
        std::vector<int32_t> tmp(halves.size());
        std::vector<int64_t> start((size_t)max_l + 2, 0);
        for (const HalfTask &h : halves) ++start[(size_t)(max_l - h.max_l1) + 1];
        for (size_t t = 1; t < start.size(); ++t) start[t] += start[t - 1];
        for (size_t i = 0; i < halves.size(); ++i) tmp[(size_t)start[(size_t)(max_l - halves[i].max_l1)]++] = (int32_t)i;
        start.assign((size_t)max_s + 2, 0);
        for (const HalfTask &h : halves) ++start[(size_t)(max_s - (lens[h.two] + 31) / 32) + 1];
        for (size_t t = 1; t < start.size(); ++t) start[t] += start[t - 1];
        for (size_t q = 0; q < tmp.size(); ++q) {
            const HalfTask &h = halves[(size_t)tmp[q]];
            idx[(size_t)start[(size_t)(max_s - (lens[h.two] + 31) / 32)]++] = tmp[q];
        }
    }
    mark("halves sorted");
    return halves;
}

// step 2: XCD-aware placement.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, each with a
// private 4 MB L2).  A task streams the profile rows of its 32 sequences one; neighbours in the longest-first
// order are mostly the same length class of partners of different sequences two, i.e. largely the SAME rows.
// Groups of G consecutive tasks are therefore placed on one XCD (they run at the same time and share those rows
// in its L2), and the groups rotate over the XCDs so that every XCD still gets the same cost mix.  Placement
// only affects speed, never results.  G ~ tasks / 128, i.e. ~16 group rounds per XCD (measured, float profiles,
// GCUPS: 4 336 tasks: none 1984, G = 4 2115, 16 2393, 32 2384, 64 2314; 33 049 tasks (one rank of C4): none
// 1674, 16 1817, 64 2242, 256 2634, 1024 2613, 4096 1677; whole length classes per XCD on C2: 30-40 % slower).
// (on the sorted order `idx` of the half tasks: -1 = padding block)
void place_on_xcds(std::vector<int32_t> &idx, int group)
{
    if (idx.empty()) return;
    int G = group >= 0 ? group : (int)std::min<size_t>(1024, std::max<size_t>(16, idx.size() / 128));
    if (G <= 1 || idx.size() < (size_t)(16 * G)) return;
    const std::vector<int64_t> src = xcd_group_order((int64_t)idx.size(), G);
    std::vector<int32_t> placed(src.size(), -1);
    for (size_t b = 0; b < src.size(); ++b)
        if (src[b] >= 0) placed[b] = idx[(size_t)src[b]];
    idx.swap(placed);
}

// step 3a: shared waves.  Every task gets W = 1, 2 or 4 waves - the smallest W that brings its per-wave cost
// under c*, c* the smallest value for which all workgroups fit the wave slots.
struct Cand { int64_t cost; int task; int iter, nstrips, wmax; };

int64_t wave_cost(const Cand &c, int W)   // rank 0's strips plus the last rank's start delay, in steps
{
    const int n0 = (c.nstrips + W - 1) / W;
    return (int64_t)(n0 * c.iter + (W - 1) * PRALINE_MW_LAG) * 12;
}

int waves_for(const Cand &c, int64_t cstar)
{
    int W = 1;
    while (W < c.wmax && wave_cost(c, W) > cstar) W *= 2;
    return W;
}

int barriers_of(const Cand &c, int W)   // barriers every wave of the share group executes (max over ranks)
{
    int total = 0;
    for (int r = 0; r < W; ++r) {
        const int nr = c.nstrips > r ? (c.nstrips - r + W - 1) / W : 0;
        total = std::max(total, r * PRALINE_MW_LAG + nr * c.iter);
    }
    return total;
}

// busy steps of rank r of a task shared by W waves (its strips r, r + W, ...)
int64_t rank_steps(const Cand &c, int W, int r)
{
    const int nr = c.nstrips > r ? (c.nstrips - r + W - 1) / W : 0;
    return (int64_t)nr * (12 * c.iter + 1);
}

struct Built { int64_t cost; WgDesc d; int64_t w[4]; int64_t crit; };   // w: busy steps of the four waves; crit: longest start delay + work

WgDesc blank_wg(int share)
{
    WgDesc d;
    d.task[0] = d.task[1] = d.task[2] = d.task[3] = -1;
    d.share = share;
    d.barriers = 0;
    d.pad[0] = d.pad[1] = 0;
    return d;
}

Built blank_built()
{
    Built b;
    b.cost = 0;
    b.d = blank_wg(1);
    b.w[0] = b.w[1] = b.w[2] = b.w[3] = 0;
    b.crit = 0;
    return b;
}

// workgroups of one set of candidates (cost-descending) under the wave counts W: share 4 alone, share 2 in pairs,
// singles in fours; sorted by cost, longest first
std::vector<Built> build_wgs(const std::vector<Cand> &cand, const std::vector<int> &set, const std::vector<int> &W)
{
    std::vector<int> by_w[5];
    for (int i : set) by_w[W[(size_t)i]].push_back(i);
    std::vector<Built> built;
    built.reserve(by_w[4].size() + by_w[2].size() / 2 + by_w[1].size() / 4 + 2);
    for (int i : by_w[4]) {
        const Cand &c = cand[(size_t)i];
        Built b = blank_built();
        b.cost = wave_cost(c, 4); b.d = blank_wg(4);
        b.d.task[0] = c.task; b.d.barriers = barriers_of(c, 4);
        for (int r = 0; r < 4; ++r) b.w[r] = rank_steps(c, 4, r);
        b.crit = (int64_t)b.d.barriers * 12;
        built.push_back(b);
    }
    for (size_t k = 0; k < by_w[2].size(); k += 2) {
        const Cand &c0 = cand[(size_t)by_w[2][k]];
        const Cand *c1 = k + 1 < by_w[2].size() ? &cand[(size_t)by_w[2][k + 1]] : nullptr;
        Built b = blank_built();
        b.cost = wave_cost(c0, 2); b.d = blank_wg(2);
        b.d.task[0] = c0.task; b.d.task[2] = c1 ? c1->task : -1;
        b.d.barriers = std::max(barriers_of(c0, 2), c1 ? barriers_of(*c1, 2) : 0);
        b.w[0] = rank_steps(c0, 2, 0); b.w[1] = rank_steps(c0, 2, 1);
        if (c1) { b.w[2] = rank_steps(*c1, 2, 0); b.w[3] = rank_steps(*c1, 2, 1); }
        b.crit = (int64_t)b.d.barriers * 12;
        built.push_back(b);
    }
    for (size_t k = 0; k < by_w[1].size(); k += 4) {
        Built b = blank_built();
        b.cost = cand[(size_t)by_w[1][k]].cost;
        for (int q = 0; q < 4; ++q)
            if (k + q < by_w[1].size()) {
                const Cand &c = cand[(size_t)by_w[1][k + q]];
                b.d.task[q] = c.task;
                b.w[q] = c.cost;
            }
        b.crit = b.cost;
        built.push_back(b);
    }
    std::stable_sort(built.begin(), built.end(), [](const Built &x, const Built &y) { return x.cost > y.cost; });
    return built;
}

// Launch order.  A CU holds two of these workgroups and the dispatcher deals the blocks round-robin - block b to XCD
// b % 8 and, inside the XCD, to CU (b / 8) % 32 - so blocks b and b + 256 end up on the same CU (same SIMDs;
// confirmed with the trace build): the longest go first in descending order, then the SHORTEST in ascending order
// (the longest shares its SIMDs with the shortest), then whatever is left in the middle.
std::vector<Built> snake_order(const std::vector<Built> &built, size_t round, bool on)
{
    std::vector<Built> order;
    const size_t nb = built.size();
    order.reserve(nb);
    if (on && nb > round) {
        const size_t tail = std::min<size_t>(round, nb - round);
        for (size_t i = 0; i < round; ++i) order.push_back(built[i]);
        for (size_t i = 0; i < tail; ++i) order.push_back(built[nb - 1 - i]);
        for (size_t i = round; i < nb - tail; ++i) order.push_back(built[i]);
    } else {
        order = built;
    }
    return order;
}

// The launch list for the wave counts W, and what it costs.  Model (checked against the C2 launch: 2.09 ms predicted,
// 2.08 measured): all workgroups are resident at once, two per CU (launch positions b and b + 256), wave i of both on
// SIMD i; a SIMD issues one wave's VALU work at a time, so it is busy for the SUM of its two waves' steps, and the
// launch lasts as long as the busiest SIMD (or the longest chain of start delays + work).  Returns the modelled
// makespan in steps, -1 when the workgroups do not fit the resident slots.
int64_t assemble(const std::vector<Cand> &cand, const std::vector<int> &W, const std::vector<int> (&per_xcd)[8], const SchedOptions &opt,
                 std::vector<WgDesc> *out)
{
    const size_t per_xcd_slots = (size_t)(opt.wave_slots / 32);   // resident workgroups per XCD (2 per CU)
    std::vector<Built> order;
    bool fits = true;   // (a list that does not fit is still a valid launch list: its tail starts when slots free up)
    if (!opt.wg_xcd) {
        std::vector<int> all;
        all.reserve(cand.size());
        for (size_t i = 0; i < cand.size(); ++i) all.push_back((int)i);
        order = snake_order(build_wgs(cand, all, W), 256, opt.snake);
        fits = order.size() <= 8 * per_xcd_slots;
    } else {
        std::vector<Built> lists[8];
        size_t longest = 0;
        for (int x = 0; x < 8; ++x) {
            lists[x] = snake_order(build_wgs(cand, per_xcd[x], W), 32, opt.snake);
            longest = std::max(longest, lists[x].size());
        }
        fits = longest <= per_xcd_slots;
        order.reserve(8 * longest);
        for (size_t q = 0; q < longest; ++q)
            for (int x = 0; x < 8; ++x) order.push_back(q < lists[x].size() ? lists[x][q] : blank_built());
    }
    int64_t makespan = 0;
    const size_t nb = order.size();
    for (size_t b = 0; b < nb && b < 256; ++b)
        for (int i = 0; i < 4; ++i) {
            const int64_t sum = order[b].w[i] + (b + 256 < nb ? order[b + 256].w[i] : 0);
            makespan = std::max(makespan, sum);
        }
    for (const Built &b : order) makespan = std::max(makespan, b.crit);
    if (out) {
        out->clear();
        out->reserve(nb);
        for (const Built &b : order) out->push_back(b.d);
    }
    return fits ? makespan : -1;
}

std::vector<WgDesc> share_waves(const std::vector<WaveTask> &tasks, const SchedOptions &opt)
{
    std::vector<Cand> cand;
    for (size_t t = 0; t < tasks.size(); ++t) {
        const WaveTask &wt = tasks[t];
        if (wt.max_l1 <= 0) continue;  // placement padding
        Cand c;
        c.task = (int)t;
        c.iter = (wt.max_l1 - 1) / 12 + 1;
        c.nstrips = wt.nstrips;
        c.cost = (int64_t)wt.nstrips * (12 * c.iter + 1);
        // rank r runs PRALINE_MW_LAG iterations behind rank r - 1; the wrap-around hand-off (last rank -> rank 0's
        // next strip) then has iter - (W - 1) LAG iterations, which must also be >= LAG
        c.wmax = (c.nstrips >= 4 && c.iter >= 4 * PRALINE_MW_LAG) ? 4 : (c.nstrips >= 2 && c.iter >= 2 * PRALINE_MW_LAG) ? 2 : 1;
        cand.push_back(c);
    }
    std::vector<WgDesc> out;
    if (cand.empty() || (int64_t)cand.size() >= opt.wave_slots) return out;
    std::sort(cand.begin(), cand.end(), [](const Cand &x, const Cand &y) { return x.cost != y.cost ? x.cost > y.cost : x.task < y.task; });
    const size_t n = cand.size();

    // XCD-aware: every XCD has its own 4 MB L2, and all these workgroups are resident at once, so WHERE a task runs
    // decides whether the operand rows it streams are already in that L2.  Tasks of neighbouring sequences two have
    // nearly the same partners (the all-pairs stage: {i < j}), so the sequences two are cut into eight contiguous runs
    // of equal cost, one per XCD; each XCD's workgroups are formed and snake-ordered on their own and the eight lists
    // are interleaved (launch index 8 q + x runs on XCD x).  Placement only affects speed, never results.
    std::vector<int> per_xcd[8];
    if (opt.wg_xcd) {
        int32_t max_two = 0;
        for (const Cand &c : cand) max_two = std::max(max_two, tasks[(size_t)c.task].two[0]);
        std::vector<int64_t> col_cost((size_t)max_two + 1, 0);
        int64_t total = 0;
        for (const Cand &c : cand) { col_cost[(size_t)tasks[(size_t)c.task].two[0]] += c.cost; total += c.cost; }
        std::vector<int> col_xcd((size_t)max_two + 1, 0);
        int64_t run = 0;
        for (size_t t = 0; t < col_cost.size(); ++t) {
            col_xcd[t] = (int)std::min<int64_t>(7, (run + col_cost[t] / 2) * 8 / std::max<int64_t>(total, 1));
            run += col_cost[t];
        }
        for (size_t i = 0; i < n; ++i) per_xcd[col_xcd[(size_t)tasks[(size_t)cand[i].task].two[0]]].push_back((int)i);   // (cost-descending)
    }

    // (a) every task gets W = 1, 2 or 4 waves - the smallest W that brings its per-wave cost under c*, c* the smallest
    // value for which all workgroups fit the wave slots
    auto slots_for = [&](int64_t cstar) {
        int64_t n1 = 0, n2 = 0, n4 = 0;
        for (const Cand &c : cand) { const int w = waves_for(c, cstar); (w == 1 ? n1 : w == 2 ? n2 : n4)++; }
        return 4 * (n4 + (n2 + 1) / 2 + (n1 + 3) / 4);
    };
    int64_t lo = 1, hi = cand[0].cost;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        if (slots_for(mid) <= opt.wave_slots - (opt.wg_xcd ? 48 : 0)) hi = mid; else lo = mid + 1;   // (per-XCD rounding of the workgroups)
    }
    std::vector<int> W(n), best_W;
    bool any = false;
    for (size_t i = 0; i < n; ++i) { W[i] = waves_for(cand[i], lo); any = any || W[i] > 1; }
    if (!any) return out;
    bool balance = opt.balance;
    if (const char *env = getenv("PRALINE_WG_BALANCE")) balance = atoi(env) != 0;   // (also read here: the CPU test library has no plan options)
    if (!balance) {
        assemble(cand, W, per_xcd, opt, &out);
        return out;
    }
    int64_t best = assemble(cand, W, per_xcd, opt, nullptr);
    best_W = W;

    // (b) EXPERIMENT (opt.balance / PRALINE_WG_BALANCE=1; off by default).  The threshold minimises the longest WAVE; what a
    // launch lasts is the busiest SIMD - the sum of the two waves it hosts (see assemble).  With more tasks than half the
    // slots the threshold leaves the cheaper tasks whole (long waves) beside the halves of the expensive ones: C2's
    // busiest SIMD carries 7 519 steps against a mean of 6 080.  A whole task wants a SHORT partner on its SIMD: quartering
    // a window of z mid-cost tasks provides them (the snake order pairs the longest workgroups with the shortest), the
    // cheapest tasks stay whole as far as the slots demand and the halves of the rest pair with each other; z and the
    // window's place are searched on the model, per XCD list.  Modelled on C2: 7 519 -> 6 869 steps (-8.6 %; an
    // exhaustive search of this family without the XCD split: 6 640).  MEASURED: no gain (2.14 ms against 2.07-2.14) -
    // the sum model misses what the quartered tasks cost (four waves in lock step on four SIMDs, each beside a wave of
    // another task; a wave left alone on its SIMD no longer hides its own latencies), so the threshold stays the default.
    {
        std::vector<int> all_idx;
        if (!opt.wg_xcd) for (size_t i = 0; i < n; ++i) all_idx.push_back((int)i);
        const int n_sets = opt.wg_xcd ? 8 : 1;
        const int64_t set_wgs = opt.wg_xcd ? opt.wave_slots / 32 : opt.wave_slots / 4;   // resident workgroups per set
        auto try_one = [&](int dz, int place) -> int64_t {   // place: start of the window, in 16ths of the quarterable tasks
            for (int x = 0; x < n_sets; ++x) {
                const std::vector<int> &set = opt.wg_xcd ? per_xcd[x] : all_idx;   // cost-descending
                const int64_t m = (int64_t)set.size();
                std::vector<int> can4;
                int64_t n2 = 0;
                for (int i : set) { if (cand[(size_t)i].wmax == 4) can4.push_back(i); n2 += cand[(size_t)i].wmax >= 2; }
                const int64_t z0 = std::max<int64_t>(0, (2 * n2 + (m - n2) - 4 * set_wgs) / 2);
                const int64_t step = std::max<int64_t>(1, m / 96);
                const int64_t z = std::min<int64_t>((int64_t)can4.size(), std::max<int64_t>(0, z0 + dz * step));
                const int64_t first = std::min<int64_t>((int64_t)can4.size() - z, (int64_t)can4.size() * place / 16);
                for (int i : set) W[(size_t)i] = std::min(2, cand[(size_t)i].wmax);
                for (int64_t k = 0; k < z; ++k) W[(size_t)can4[(size_t)(first + k)]] = 4;
                int64_t c1 = 0, c2 = 0, c4 = 0;
                for (int i : set) (W[(size_t)i] == 1 ? c1 : W[(size_t)i] == 2 ? c2 : c4)++;
                // the cheapest halves stay whole until the workgroups fit
                for (int64_t t = m - 1; t >= 0 && c4 + (c2 + 1) / 2 + (c1 + 3) / 4 > set_wgs; --t)
                    if (W[(size_t)set[(size_t)t]] == 2) { W[(size_t)set[(size_t)t]] = 1; --c2; ++c1; }
                if (c4 + (c2 + 1) / 2 + (c1 + 3) / 4 > set_wgs) return -1;
            }
            const int64_t ms = assemble(cand, W, per_xcd, opt, nullptr);
            if (getenv("PRALINE_SCHED_DEBUG")) fprintf(stderr, "balance: dz=%d place=%d/16 -> makespan %lld (best %lld)\n", dz, place, (long long)ms, (long long)best);
            if (ms >= 0 && (best < 0 || ms < best)) { best = ms; best_W = W; }
            return ms;
        };
        if (const char *env = getenv("PRALINE_WG_BALANCE_FORCE")) {   // experiments: "dz,place" - take exactly this member of the family
            int dz = 0, place = 4;
            if (sscanf(env, "%d,%d", &dz, &place) == 2) {
                best = -1;
                try_one(dz, place);
                if (best >= 0) { assemble(cand, best_W, per_xcd, opt, &out); return out; }
            }
        }
        int best_dz = 0;
        int64_t best_here = -1;
        for (int dz = -4; dz <= 0; dz += 2) {
            const int64_t ms = try_one(dz, 4);
            if (ms >= 0 && (best_here < 0 || ms < best_here)) { best_here = ms; best_dz = dz; }
        }
        try_one(best_dz, 2);
        try_one(best_dz, 6);
    }
    if (best < 0) {
        // (cannot happen with the reserve above; keep the threshold list whatever the model says)
        for (size_t i = 0; i < n; ++i) best_W[i] = waves_for(cand[i], lo);
    }
    assemble(cand, best_W, per_xcd, opt, &out);
    return out;
}

// step 3b: four independent tasks per workgroup.  Workgroup w runs on XCD w % 8: it gets the next four tasks of
// THAT XCD's queue (placed positions 8 (4 q + r) + x, r = 0..3), so the XCD grouping of the task list survives.
std::vector<WgDesc> four_singles(const std::vector<WaveTask> &tasks)
{
    const size_t nt = tasks.size();
    std::vector<WgDesc> out((nt + 31) / 32 * 8);
    for (size_t w = 0; w < out.size(); ++w) {
        WgDesc d;
        d.share = 1; d.barriers = 0; d.pad[0] = d.pad[1] = 0;
        const size_t q = w / 8, x = w % 8;
        for (int r = 0; r < 4; ++r) {
            const size_t t = 8 * (4 * q + r) + x;
            d.task[r] = (t < nt && tasks[t].max_l1 > 0) ? (int32_t)t : -1;
        }
        out[w] = d;
    }
    return out;
}

}  // namespace

std::vector<int64_t> xcd_group_order(int64_t n0, int G)
{
    const int64_t n = (n0 + 8 * G - 1) / (8 * G) * (8 * G);
    std::vector<int64_t> src((size_t)n, -1);
    for (int64_t i = 0; i < n0; ++i) {
        const int64_t g = i / G, x = g % 8, q = (g / 8) * G + i % G;
        src[(size_t)(8 * q + x)] = i;
    }
    return src;
}

void build_schedule(const int32_t *lens, int64_t n_pairs, const int32_t *pairs, const SchedOptions &opt, Schedule &out)
{
    const bool timing = getenv("PRALINE_SCHED_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sched] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    std::vector<int32_t> order;   // launch position -> half task (-1: padding)
    // quad16 (path plans of one-hot arenas, k_dp_quad_tb): 16 pairs per task
    const bool quad = opt.quad16 && opt.split_layout && opt.want_paths;
    const int width = quad ? 16 : 32;
    const RawVec<HalfTask> half_store = cut_half_tasks(lens, n_pairs, pairs, order, width);
    mark("half tasks");
    const HalfTask pad_half = empty_half(order.empty() ? -1 : half_store[(size_t)order.back()].two);
    place_on_xcds(order, opt.xcd_group);
    mark("xcd placement");
    struct HalfView {
        const RawVec<HalfTask> &store; const std::vector<int32_t> &order; const HalfTask &pad;
        size_t size() const { return order.size(); }
        const HalfTask &operator[](size_t i) const { return order[i] >= 0 ? store[(size_t)order[i]] : pad; }
    } halves{half_store, order, pad_half};

    out.split = opt.split_layout;
    int tp = halves.size() >= 4096 ? 2 : 1;
    if (opt.want_paths) tp = 1;  // the traceback variant keeps three states per column in registers
    if (opt.tp == 1 || (opt.tp == 2 && !opt.want_paths)) tp = opt.tp;
    if (out.split) tp = 1;       // split-strip kernels: both halves of the wave work on the same 32 pairs
    out.tp = tp;
    out.lanes_per_task = quad ? 16 : (out.split ? 32 : 64);
    const int lanes = out.lanes_per_task;

    const size_t n_tasks = (halves.size() + tp - 1) / tp;
    out.tasks.assign(n_tasks, WaveTask());
    if (tp == 1 && width == lanes) {   // (every lane of every task and every pair is written below: no initial pass)
        out.lane_one.resize(n_tasks * lanes);
        out.lane_pair.resize(n_tasks * lanes);
        out.loc.resize((size_t)n_pairs);
    } else {
        out.lane_one.assign(n_tasks * lanes, -1);
        out.lane_pair.assign(n_tasks * lanes, -1);
        out.loc.assign((size_t)n_pairs, PairLoc());
    }
    out.tb_elems.assign(n_tasks, 0);
    out.aux_elems.assign(n_tasks, 0);
    const int nt = sched_threads(n_pairs);
    run_threads(nt, [&](int th, int n) {
        int64_t lo, hi;
        slice_of((int64_t)n_tasks, th, n, lo, hi);
        for (size_t t = (size_t)lo; t < (size_t)hi; ++t) {
            WaveTask &wt = out.tasks[t];
            wt.two[0] = wt.two[1] = -1;
            wt.max_l1 = 0;
            wt.nstrips = 0;
            for (int hh = 0; hh < tp; ++hh) {
                const size_t hi2 = t * tp + hh;
                if (hi2 >= halves.size()) break;
                const HalfTask &h = halves[hi2];
                wt.two[hh] = h.two;
                wt.max_l1 = std::max(wt.max_l1, h.max_l1);
                wt.nstrips = std::max(wt.nstrips, (lens[h.two] + 31) / 32);
                for (int q = 0; q < width; ++q) {
                    out.lane_one[t * lanes + hh * 32 + q] = h.one[q];
                    out.lane_pair[t * lanes + hh * 32 + q] = h.pair[q];
                    if (h.pair[q] >= 0) { out.loc[h.pair[q]].task = (int32_t)t; out.loc[h.pair[q]].lane = hh * 32 + q; }
                }
            }
            wt.tb_off = 0;
            wt.aux_off = 0;
            // traceback planes: split layout uint2 [nstrips][max_l1 + 8][64], batch layout uint4 [nstrips][max_l1 + 1][64]
            // (quad16: uint2 [nstrips][PRALINE_QUAD_STEPS(max_l1)][64] - two rows of 8 columns per lane and step)
            // (pk16: uint4 [nstrips][PRALINE_QUAD_STEPS(max_l1)][64] counted in uint2 - never less than the strip kernels' planes)
            out.tb_elems[t] = (opt.pk16 && out.split && !quad) ? (int64_t)wt.nstrips * std::max(2 * PRALINE_QUAD_STEPS(wt.max_l1), wt.max_l1 + 8) * 64 :
                              quad ? (int64_t)wt.nstrips * PRALINE_QUAD_STEPS(wt.max_l1) * 64
                                   : (out.split ? (int64_t)wt.nstrips * (wt.max_l1 + 8) * 64 : (int64_t)wt.nstrips * (wt.max_l1 + 1) * 64);
            out.aux_elems[t] = ((int64_t)(wt.max_l1 + 1) * 3 + (int64_t)wt.nstrips * 32 * 3) * lanes;
        }
    });
    int64_t bnd = 0;
    for (size_t t = 0; t < n_tasks; ++t) {
        WaveTask &wt = out.tasks[t];
        wt.bnd_off = bnd;
        // strip-boundary rows: the 12x unrolled loops of the split kernels read ahead
        bnd += quad ? (int64_t)(wt.max_l1 + 24) * 16 : (out.split ? (int64_t)(wt.max_l1 + 24) * 32 : (int64_t)(wt.max_l1 + 1) * 64);
    }
    out.bnd_elems = bnd;
    mark("tasks");

    out.wg.clear();
    out.wg_singles.clear();
    // (path plans too: the forward fill of their two-pass scheme can run on the scores kernel's workgroups)
    if (out.split && opt.shared_waves && !quad) {
        out.wg = share_waves(out.tasks, opt);
        if (out.wg.empty()) out.wg_singles = four_singles(out.tasks);
    }

    mark("workgroup lists");
    // path slots (capacity l1 + l2 + 2 rows per pair) and the cell count: a prefix sum in two passes over slices
    out.slot_off.resize((size_t)n_pairs);   // (written by the slices below)
    int64_t cap = 0, cells = 0;
    {
        std::vector<int64_t> pcap((size_t)nt, 0), pcells((size_t)nt, 0);
        run_threads(nt, [&](int th, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, th, n, lo, hi);
            int64_t c = 0, ce = 0;
            for (int64_t p = lo; p < hi; ++p) {
                const int64_t l1 = lens[pairs[2 * p]], l2 = lens[pairs[2 * p + 1]];
                out.slot_off[(size_t)p] = c;
                c += l1 + l2 + 2;
                ce += l1 * l2;
            }
            pcap[(size_t)th] = c; pcells[(size_t)th] = ce;
        });
        std::vector<int64_t> base((size_t)nt, 0);
        for (int t = 0; t < nt; ++t) { base[(size_t)t] = cap; cap += pcap[(size_t)t]; cells += pcells[(size_t)t]; }
        if (nt > 1)
            run_threads(nt, [&](int th, int n) {
                int64_t lo, hi;
                slice_of(n_pairs, th, n, lo, hi);
                const int64_t b = base[(size_t)th];
                if (b == 0) return;
                for (int64_t p = lo; p < hi; ++p) out.slot_off[(size_t)p] += b;
            });
    }
    out.path_cap = cap;
    out.cells = cells;
    mark("path slots");
}

// What plan creation needs to know about a pair list before it schedules it, in one pass over slices of the list: the DP
// cells, the shortest sequence in any pair, and the first pair with an index outside 0 .. n_seqs - 1 (-1: none).
void sched_pair_stats(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int64_t *cells, int *min_len,
                      int64_t *first_bad)
{
    const int nt = sched_threads(n_pairs);
    std::vector<int64_t> c((size_t)nt, 0), bad((size_t)nt, -1);
    std::vector<int> ml((size_t)nt, INT32_MAX);
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_pairs, t, n, lo, hi);
        int64_t cc = 0;
        int m = INT32_MAX;
        for (int64_t p = lo; p < hi; ++p) {
            const int32_t o = pairs[2 * p], w = pairs[2 * p + 1];
            if (o < 0 || o >= n_seqs || w < 0 || w >= n_seqs) { bad[(size_t)t] = p; break; }
            const int l1 = lens[o], l2 = lens[w];
            cc += (int64_t)l1 * l2;
            m = std::min(m, std::min(l1, l2));
        }
        c[(size_t)t] = cc; ml[(size_t)t] = m;
    });
    int64_t cells_all = 0, fb = -1;
    int m_all = INT32_MAX;
    for (int t = 0; t < nt; ++t) {
        cells_all += c[(size_t)t];
        m_all = std::min(m_all, ml[(size_t)t]);
        if (fb < 0 && bad[(size_t)t] >= 0) fb = bad[(size_t)t];
    }
    if (cells) *cells = cells_all;
    if (min_len) *min_len = n_pairs > 0 ? m_all : 0;
    if (first_bad) *first_bad = fb;
}

// ---- pipeline workgroups (k_dp_pipe) ---------------------------------------------------------------------------
// 1. The distinct sequences two are taken in index order, `block_twos` at a time; the sequences one of a block's pairs
//    are pooled, sorted by length (descending) and cut into sets of 32 - neighbouring sequences two of an all-pairs or
//    one-against-all list have (nearly) the same partners, so almost every lane of a (set, two) task holds a pair.
// 2. A (set, block) list of tasks is cut into workgroup items.  The four waves of an item take the strips of its task
//    list round-robin, so an item costs ceil(strips / 4) rounds of `rsteps` steps; the cut is the smallest cost bound
//    c* for which the items fit the resident workgroup slots (small batches), or an eighth of a slot's share (large
//    batches: the dispatcher evens out the rest).
namespace {

inline int pipe_rsteps(int max_l1)
{
    return std::max(PRALINE_PIPE_MIN_STEPS, (max_l1 + 1 + 11) / 12 * 12);
}

struct PipeList { int32_t set; int32_t rsteps; std::vector<int32_t> task; };   // tasks (indices into a scratch array) of one set

}  // namespace

namespace {

struct ScratchTask { int32_t two, set, nstrips; int32_t pair[32]; };

// Step 1 of build_pipe_schedule - sets of 32 sequences one per block of sequences two, one task per (set, sequence two)
// that holds a pair - in passes over the pair list that are independent per slice of it (no pair-index permutation, no
// random gathers through one: at 8.4 M pairs the permutation version spent 0.3 s in cache misses):
//   a. pairs per sequence two -> the distinct sequences two, their block and place in it;
//   b. which sequences one occur in each block (a byte table [block][sequence]);
//   c. per block: its sequences one by descending length, cut into sets of 32 -> position of every sequence in its block;
//   d. pairs per (block, sequence two, set) slot -> the tasks, numbered by (sequence two, set);
//   e. every pair into its lane of its task.
// The lists (one per set: its tasks by ascending sequence two) and set_one come out exactly as the serial version's.
// Returns false when the byte table would be too large (the caller takes the serial version); dup: the same pair twice.
bool pipe_front_sliced(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int B, int nt,
                       RawVec<ScratchTask> &st, std::vector<PipeList> &lists, std::vector<int32_t> &set_one,
                       int64_t &old_tasks, bool &dup)
{
    dup = false;
    old_tasks = 0;
    const bool timing = getenv("PRALINE_SCHED_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sched] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // a.
    std::vector<int32_t> cnt((size_t)n_seqs, 0);
    {
        std::vector<std::vector<int32_t>> part((size_t)nt);
        run_threads(nt, [&](int t, int n) {
            int64_t lo, hi;
            slice_of(n_pairs, t, n, lo, hi);
            std::vector<int32_t> &h = part[(size_t)t];
            h.assign((size_t)n_seqs, 0);
            for (int64_t i = lo; i < hi; ++i) ++h[(size_t)pairs[2 * i + 1]];
        });
        for (const std::vector<int32_t> &h : part)
            for (int64_t q = 0; q < n_seqs; ++q) cnt[(size_t)q] += h[(size_t)q];
    }
    std::vector<int32_t> twos, block_of((size_t)n_seqs, -1), place_of((size_t)n_seqs, 0);
    for (int64_t t = 0; t < n_seqs; ++t)
        if (cnt[(size_t)t] > 0) {
            block_of[(size_t)t] = (int32_t)(twos.size() / (size_t)B);
            place_of[(size_t)t] = (int32_t)(twos.size() % (size_t)B);
            twos.push_back((int32_t)t);
            old_tasks += (cnt[(size_t)t] + 31) / 32;
        }
    const int64_t n_blocks = ((int64_t)twos.size() + B - 1) / B;
    if (n_blocks * n_seqs > ((int64_t)16 << 20)) return false;
    mark("a. pairs per sequence two");
    // b.
    std::vector<uint8_t> member((size_t)(n_blocks * n_seqs), 0);
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_pairs, t, n, lo, hi);
        for (int64_t i = lo; i < hi; ++i) {
            uint8_t &m = member[(size_t)((int64_t)block_of[(size_t)pairs[2 * i + 1]] * n_seqs + pairs[2 * i])];
            if (!m) m = 1;   // (benign race: every writer stores 1)
        }
    });
    mark("b. membership");
    // c.  (one global order by descending length serves every block: a block's sequences one are its members in that order)
    std::vector<int32_t> by_len((size_t)n_seqs);
    std::iota(by_len.begin(), by_len.end(), 0);
    std::sort(by_len.begin(), by_len.end(), [&](int32_t x, int32_t y) { return lens[x] != lens[y] ? lens[x] > lens[y] : x < y; });
    std::vector<std::vector<int32_t>> uni((size_t)n_blocks);
    std::vector<int32_t> pos((size_t)(n_blocks * n_seqs), -1);
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_blocks, t, n, lo, hi);
        for (int64_t b = lo; b < hi; ++b) {
            std::vector<int32_t> &u = uni[(size_t)b];
            const uint8_t *m = member.data() + b * n_seqs;
            int32_t *pb = pos.data() + b * n_seqs;
            for (int64_t q = 0; q < n_seqs; ++q) {
                const int32_t sq = by_len[(size_t)q];
                if (m[sq]) { pb[sq] = (int32_t)u.size(); u.push_back(sq); }
            }
        }
    });
    // slots: (block b, set g, lane, place of the sequence two) -> tmp[tbase[b] + (g * 32 + lane) * n_twos(b) + place]; for a
    // fixed sequence one (= lane) consecutive sequences two of a block are consecutive words: an all-pairs list in
    // row-major order writes them one after the other
    std::vector<int64_t> set0((size_t)n_blocks + 1, 0), tbase((size_t)n_blocks + 1, 0);
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t nsets = ((int64_t)uni[(size_t)b].size() + 31) / 32;
        const int64_t n_twos = std::min<int64_t>(B, (int64_t)twos.size() - b * B);
        set0[(size_t)b + 1] = set0[(size_t)b] + nsets;
        tbase[(size_t)b + 1] = tbase[(size_t)b] + nsets * 32 * n_twos;
    }
    const int64_t n_sets = set0[(size_t)n_blocks];
    set_one.assign((size_t)n_sets * 32, -1);
    lists.assign((size_t)n_sets, PipeList());
    for (int64_t b = 0; b < n_blocks; ++b) {
        const std::vector<int32_t> &u = uni[(size_t)b];
        std::copy(u.begin(), u.end(), set_one.begin() + set0[(size_t)b] * 32);
        for (int64_t g = 0; g < set0[(size_t)b + 1] - set0[(size_t)b]; ++g) {
            lists[(size_t)(set0[(size_t)b] + g)].set = (int32_t)(set0[(size_t)b] + g);
            lists[(size_t)(set0[(size_t)b] + g)].rsteps = pipe_rsteps(lens[u[(size_t)g * 32]]);
        }
    }
    mark("c. sets");
    // d.  every pair into its word
    // (uninitialised: the threads that fill it touch its pages first)
    const int64_t n_tmp = tbase[(size_t)n_blocks];
    RawVec<int32_t> tmp_mem((size_t)std::max<int64_t>(n_tmp, 1));
    int32_t *tmp = tmp_mem.data();
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_tmp, t, n, lo, hi);
        std::fill(tmp + lo, tmp + hi, -1);
    });
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of(n_pairs, t, n, lo, hi);
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t one = pairs[2 * i], two = pairs[2 * i + 1];
            const int64_t bq = block_of[(size_t)two];
            const int64_t n_twos = std::min<int64_t>(B, (int64_t)twos.size() - bq * B);
            tmp[(size_t)(tbase[(size_t)bq] + (int64_t)pos[(size_t)(bq * n_seqs + one)] * n_twos + place_of[(size_t)two])] = (int32_t)i;
        }
    });
    mark("d. pairs into lanes");
    // e.  tasks: the (set, sequence two) columns that hold a pair, numbered block by block, set by set, by sequence two
    std::vector<int64_t> task0((size_t)n_blocks + 1, 0);
    std::vector<int64_t> filled((size_t)n_blocks, 0);
    // (blocks dealt round-robin: in an all-pairs list block b holds b + 1 sets, contiguous slices would leave the last
    // thread with twice the average)
    run_threads(nt, [&](int t, int n) {
        for (int64_t b = t; b < n_blocks; b += n) {
            const int64_t nsets = set0[(size_t)b + 1] - set0[(size_t)b];
            const int64_t n_twos = std::min<int64_t>(B, (int64_t)twos.size() - b * B);
            // (lane by lane, the sequences two of a lane are consecutive words: the column-wise walk was a stride of n_twos
            // words over 37 MB for all of C4)
            int64_t nt_b = 0;
            std::vector<uint8_t> any((size_t)n_twos);
            for (int64_t g = 0; g < nsets; ++g) {
                std::fill(any.begin(), any.end(), (uint8_t)0);
                const int32_t *w = tmp + tbase[(size_t)b] + g * 32 * n_twos;
                for (int l = 0; l < 32; ++l)
                    for (int64_t q = 0; q < n_twos; ++q) any[q] |= (uint8_t)(w[l * n_twos + q] >= 0);
                for (int64_t q = 0; q < n_twos; ++q) nt_b += any[q];
            }
            task0[(size_t)b + 1] = nt_b;
        }
    });
    for (int64_t b = 0; b < n_blocks; ++b) task0[(size_t)b + 1] += task0[(size_t)b];
    st.resize((size_t)task0[(size_t)n_blocks]);
    run_threads(nt, [&](int t, int n) {
        for (int64_t b = t; b < n_blocks; b += n) {
            const int64_t nsets = set0[(size_t)b + 1] - set0[(size_t)b];
            const int64_t n_twos = std::min<int64_t>(B, (int64_t)twos.size() - b * B);
            int64_t tk = task0[(size_t)b], got = 0;
            std::vector<ScratchTask> row((size_t)n_twos);
            std::vector<int> cnt((size_t)n_twos);
            for (int64_t g = 0; g < nsets; ++g) {
                const int32_t *w = tmp + tbase[(size_t)b] + g * 32 * n_twos;
                std::fill(cnt.begin(), cnt.end(), 0);
                for (int l = 0; l < 32; ++l)
                    for (int64_t q = 0; q < n_twos; ++q) {
                        const int32_t v = w[l * n_twos + q];
                        row[(size_t)q].pair[l] = v;
                        cnt[(size_t)q] += v >= 0;
                    }
                for (int64_t q = 0; q < n_twos; ++q) {
                    const int cnt_l = cnt[(size_t)q];
                    if (!cnt_l) continue;
                    ScratchTask &x = row[(size_t)q];
                    x.two = twos[(size_t)(b * B + q)];
                    x.set = (int32_t)(set0[(size_t)b] + g);
                    x.nstrips = (lens[x.two] + 31) / 32;
                    st[(size_t)tk] = x;
                    lists[(size_t)x.set].task.push_back((int32_t)tk);   // (a set's tasks belong to one block: no other thread touches it)
                    ++tk;
                    got += cnt_l;
                }
            }
            filled[(size_t)b] = got;
        }
    });
    // (two pairs on one word - the same pair twice in the list - leave fewer filled lanes than pairs)
    int64_t all = 0;
    for (int64_t b = 0; b < n_blocks; ++b) all += filled[(size_t)b];
    dup = all != n_pairs;
    mark("e. tasks");
    return true;
}

// the same step with one permutation of the pair list by sequence two (any list size, any number of sequences)
void pipe_front_serial(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int B,
                       RawVec<ScratchTask> &st, std::vector<PipeList> &lists, std::vector<int32_t> &set_one,
                       int64_t &old_tasks, bool &dup)
{
    dup = false;
    old_tasks = 0;
    // pairs by sequence two (counting sort, stable: list order inside a column)
    std::vector<int64_t> start((size_t)n_seqs + 1, 0);
    for (int64_t i = 0; i < n_pairs; ++i) ++start[(size_t)pairs[2 * i + 1] + 1];
    for (size_t t = 1; t < start.size(); ++t) start[t] += start[t - 1];
    std::vector<int64_t> by_two((size_t)n_pairs);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n_pairs; ++i) by_two[(size_t)fill[(size_t)pairs[2 * i + 1]]++] = i;
    }
    std::vector<int32_t> twos;
    for (int64_t t = 0; t < n_seqs; ++t)
        if (start[(size_t)t + 1] > start[(size_t)t]) twos.push_back((int32_t)t);
    std::vector<int32_t> pos_of((size_t)n_seqs, -1), stamp((size_t)n_seqs, -1);
    for (size_t b0 = 0; b0 < twos.size(); b0 += (size_t)B) {
        const size_t b1 = std::min(twos.size(), b0 + (size_t)B);
        std::vector<int32_t> uni;
        for (size_t q = b0; q < b1; ++q) {
            const int32_t two = twos[q];
            old_tasks += (start[(size_t)two + 1] - start[(size_t)two] + 31) / 32;
            for (int64_t k = start[(size_t)two]; k < start[(size_t)two + 1]; ++k) {
                const int32_t one = pairs[2 * by_two[(size_t)k]];
                if (stamp[(size_t)one] != (int32_t)b0) { stamp[(size_t)one] = (int32_t)b0; uni.push_back(one); }
            }
        }
        std::sort(uni.begin(), uni.end(), [&](int32_t x, int32_t y) { return lens[x] != lens[y] ? lens[x] > lens[y] : x < y; });
        for (size_t k = 0; k < uni.size(); ++k) pos_of[(size_t)uni[k]] = (int32_t)k;
        const size_t nsets = (uni.size() + 31) / 32, set0 = set_one.size() / 32;
        set_one.resize((set0 + nsets) * 32, -1);
        for (size_t k = 0; k < uni.size(); ++k) set_one[set0 * 32 + k] = uni[k];
        const size_t list0 = lists.size();
        lists.resize(list0 + nsets);
        for (size_t g = 0; g < nsets; ++g) {
            lists[list0 + g].set = (int32_t)(set0 + g);
            lists[list0 + g].rsteps = pipe_rsteps(lens[uni[g * 32]]);
        }
        std::vector<int32_t> task_of(nsets);
        for (size_t q = b0; q < b1; ++q) {
            const int32_t two = twos[q];
            std::fill(task_of.begin(), task_of.end(), -1);
            for (int64_t k = start[(size_t)two]; k < start[(size_t)two + 1]; ++k) {
                const int64_t pi = by_two[(size_t)k];
                const int32_t pos = pos_of[(size_t)pairs[2 * pi]];
                const size_t g = (size_t)pos / 32;
                if (task_of[g] < 0) {
                    ScratchTask t;
                    t.two = two; t.set = (int32_t)(set0 + g); t.nstrips = (lens[two] + 31) / 32;
                    for (int l = 0; l < 32; ++l) t.pair[l] = -1;
                    task_of[g] = (int32_t)st.size();
                    st.push_back(t);
                    lists[list0 + g].task.push_back(task_of[g]);
                }
                int32_t &slot = st[(size_t)task_of[g]].pair[pos % 32];
                if (slot >= 0) { dup = true; return; }   // the same pair twice in the list: not for this layout
                slot = (int32_t)pi;
            }
        }
    }
}

}  // namespace

void build_pipe_schedule(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, const PipeOptions &opt,
                         PipeSchedule &out)
{
    out = PipeSchedule();
    if (n_pairs <= 0 || n_seqs <= 0) return;
    RawVec<ScratchTask> st;
    std::vector<PipeList> lists;
    std::vector<int32_t> set_one;
    const int B = std::max(1, opt.block_twos);
    int64_t old_tasks = 0;   // what the per-column schedule would need
    bool dup = false;
    const int nt = sched_threads(n_pairs);
    const bool timing = getenv("PRALINE_SCHED_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sched] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (opt.serial_front || !pipe_front_sliced(lens, n_seqs, n_pairs, pairs, B, nt, st, lists, set_one, old_tasks, dup)) {
        st.clear(); lists.clear(); set_one.clear();
        pipe_front_serial(lens, n_seqs, n_pairs, pairs, B, st, lists, set_one, old_tasks, dup);
    }
    mark("a-e with their scratch released");
    if (dup) return;
    if (st.empty() || lens == nullptr) return;
    for (const ScratchTask &t : st)
        if (t.nstrips <= 0) return;   // empty sequence two
    if ((double)n_pairs < opt.min_fill * 32.0 * (double)st.size() || (int64_t)st.size() > old_tasks + old_tasks / 4 + 8) return;

    // ---- items: cut every list under the cost bound c* ----
    auto list_cost = [&](const PipeList &l, size_t a, size_t b) {   // tasks a .. b - 1 of the list
        int64_t q = 0;
        for (size_t k = a; k < b; ++k) q += st[(size_t)l.task[k]].nstrips;
        return (q + 3) / 4 * (int64_t)l.rsteps;
    };
    // items of one list under the bound c*: bin packing of its tasks (<= PRALINE_PIPE_MAX_TASKS per item) into bins of
    // floor(c* / rsteps) rounds = 4 x that many strips, first fit decreasing - long and short sequences two end up
    // together, so the strip totals come out near the multiples of four that the rounds are made of
    struct Cut { int32_t list; int64_t nstrips; std::vector<int32_t> task; };   // nstrips: the strip total of its tasks
    // max_tasks: tasks per item; n_single: per list, this share (in 1/1024) of its tasks - the shortest - become items
    // of their own (large batches: the short items the dispatcher fills the tail of the launch with)
    // (every list's tasks by descending strip count, once: the bisection below packs them a dozen times)
    std::vector<std::vector<int32_t>> sorted_tasks(lists.size());
    std::vector<int64_t> list_total(lists.size(), 0), list_longest(lists.size(), 0);   // cost of the whole list, of its longest task
    const int nt_lists = sched_threads((int64_t)st.size());
    run_threads(nt_lists, [&](int th, int n) {   // (the lists are independent of each other)
        int64_t lo, hi;
        slice_of((int64_t)lists.size(), th, n, lo, hi);
        for (size_t li = (size_t)lo; li < (size_t)hi; ++li) {
            sorted_tasks[li] = lists[li].task;
            std::stable_sort(sorted_tasks[li].begin(), sorted_tasks[li].end(), [&](int32_t x, int32_t y) { return st[(size_t)x].nstrips > st[(size_t)y].nstrips; });
            list_total[li] = list_cost(lists[li], 0, lists[li].task.size());
            if (!sorted_tasks[li].empty()) list_longest[li] = (st[(size_t)sorted_tasks[li][0]].nstrips + 3) / 4 * (int64_t)lists[li].rsteps;
        }
    });
    // first-fit packing of the lists li0 .. li1 - 1 (each list on its own): items of at most max_tasks tasks and cstar cost
    auto cut_range = [&](size_t li0, size_t li1, int64_t cstar, int max_tasks, int single_share, std::vector<Cut> *cuts) {
        int64_t n = 0;
        std::vector<int64_t> load;
        std::vector<int32_t> count;
        std::vector<std::vector<int32_t>> member;
        for (size_t li = li0; li < li1; ++li) {
            const PipeList &l = lists[li];
            const int64_t cap = std::max<int64_t>(1, cstar / l.rsteps) * 4;   // strips per item
            const std::vector<int32_t> &order = sorted_tasks[li];
            const size_t n_single = (order.size() * (size_t)single_share + 1023) / 1024;
            const size_t n_packed = order.size() - std::min(order.size(), n_single);
            load.clear(); count.clear();
            if (cuts) member.clear();
            for (size_t k = 0; k < order.size(); ++k) {
                const int32_t t = order[k];
                const int64_t q = st[(size_t)t].nstrips;
                size_t b = 0;
                if (k >= n_packed) b = load.size();
                else
                    while (b < load.size() && (count[b] >= max_tasks || load[b] + q > cap)) ++b;
                if (b == load.size()) { load.push_back(0); count.push_back(0); if (cuts) member.emplace_back(); }
                load[b] += q; ++count[b];
                if (k >= n_packed) count[b] = max_tasks;   // closed
                if (cuts) member[b].push_back(t);
            }
            n += (int64_t)load.size();
            if (cuts)
                for (size_t b = 0; b < member.size(); ++b) cuts->push_back(Cut{(int32_t)li, load[b], std::move(member[b])});
        }
        return n;
    };
    auto cut_count = [&](int64_t cstar, int max_tasks, int single_share, std::vector<Cut> *cuts) {
        return cut_range(0, lists.size(), cstar, max_tasks, single_share, cuts);
    };
    mark("f. lists sorted");
    auto item_cost = [&](const Cut &c) { return (c.nstrips + 3) / 4 * (int64_t)lists[(size_t)c.list].rsteps; };
    int64_t total = 0, one_max = 0;
    for (size_t li = 0; li < lists.size(); ++li) { total += list_total[li]; one_max = std::max(one_max, list_longest[li]); }
    // Small batches (up to 2.5 tasks per workgroup slot): everything resident at once - the smallest bound c* whose
    // items fit the slots.  Larger batches: items of up to k tasks, k a third of a slot's share (at most
    // PRALINE_PIPE_MAX_TASKS: the strip total of a long list wastes less of its last round), and two slots' worth of
    // single-task items from the short end of every list to even out the tail of the launch.
    std::vector<Cut> cuts;
    {
        const int64_t n_tasks = (int64_t)st.size(), slots = std::max<int64_t>(opt.wg_slots, 1);
        if (2 * n_tasks <= 5 * slots) {
            // the bound only acts through floor(c* / rsteps) and every rsteps is a multiple of 12: search the multiples of
            // 12, from the larger of the longest task and an even share upwards (galloping, then bisection)
            int64_t lo = (std::max(one_max, (total + slots - 1) / slots) + 11) / 12, hi = lo;
            const int64_t top = (std::max(one_max, total) + 11) / 12;
            while (hi < top && cut_count(12 * hi, PRALINE_PIPE_MAX_TASKS, 0, nullptr) > slots) { lo = hi + 1; hi = std::min(top, hi + std::max<int64_t>(1, hi / 8)); }
            while (lo < hi) {
                const int64_t mid = (lo + hi) / 2;
                if (cut_count(12 * mid, PRALINE_PIPE_MAX_TASKS, 0, nullptr) <= slots) hi = mid; else lo = mid + 1;
            }
            cut_count(12 * lo, PRALINE_PIPE_MAX_TASKS, 0, &cuts);
        } else {
            const int k = (int)std::min<int64_t>(PRALINE_PIPE_MAX_TASKS, std::max<int64_t>(1, n_tasks / (3 * slots)));
            const int share = k == 1 ? 0 : (int)std::min<int64_t>(1024, 2 * slots * 1024 / n_tasks);
            // (one pass; slices of the lists on several threads, their items joined in list order)
            std::vector<std::vector<Cut>> part((size_t)nt_lists);
            run_threads(nt_lists, [&](int th, int n) {
                int64_t lo, hi;
                slice_of((int64_t)lists.size(), th, n, lo, hi);
                cut_range((size_t)lo, (size_t)hi, INT64_MAX / 8, k, share, &part[(size_t)th]);
            });
            size_t total_cuts = 0;
            for (const auto &pc : part) total_cuts += pc.size();
            cuts.reserve(total_cuts);
            for (auto &pc : part)
                for (auto &c : pc) cuts.push_back(std::move(c));
        }
    }
    mark("g. items cut");
    // launch order: longest first; when everything is resident at once (two workgroups per CU: launch positions b and
    // b + 256 share a CU), the longest share their CUs with the shortest.  (Measured and dropped: keeping the items of one
    // list on one XCD - positions of equal b % 8 - so that its L2 serves their common operand rows: C2 1.90 -> 2.06 ms.)
    // (by descending cost, equal costs in item order: one key per item, the cost's complement above the item's index)
    std::vector<int64_t> cost(cuts.size());
    int64_t cost_max = 0;
    for (size_t c = 0; c < cuts.size(); ++c) { cost[c] = item_cost(cuts[c]); cost_max = std::max(cost_max, cost[c]); }
    std::vector<int32_t> order(cuts.size());
    if (cost_max < ((int64_t)1 << 32)) {
        // two stable counting passes over the halves of the complemented cost
        std::vector<int32_t> pass(cuts.size());
        std::vector<uint32_t> start(65537);
        for (int half = 0; half < 2; ++half) {
            auto digit = [&](int32_t c) { return ((0xffffffffu - (uint32_t)cost[(size_t)c]) >> (16 * half)) & 0xffffu; };
            std::fill(start.begin(), start.end(), 0u);
            for (size_t c = 0; c < cuts.size(); ++c) ++start[digit(half == 0 ? (int32_t)c : pass[c]) + 1];
            for (size_t d = 1; d < start.size(); ++d) start[d] += start[d - 1];
            for (size_t c = 0; c < cuts.size(); ++c) {
                const int32_t it = half == 0 ? (int32_t)c : pass[c];
                (half == 0 ? pass : order)[start[digit(it)]++] = it;
            }
        }
    } else {
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cost[(size_t)x] > cost[(size_t)y]; });
    }
    if ((int64_t)order.size() > 256 && (int64_t)order.size() <= opt.wg_slots) {
        std::vector<int32_t> snake(order.begin(), order.begin() + 256);
        for (size_t k = order.size(); k > 256; --k) snake.push_back(order[k - 1]);
        order.swap(snake);
    }

    mark("h. launch order");
    out.set_one.swap(set_one);
    // items in launch order: task ranges and boundary offsets by prefix sums, then the tasks' records on several threads
    out.items.resize(order.size());
    {
        int64_t t0 = 0, bnd = 0;
        for (size_t k = 0; k < order.size(); ++k) {
            const Cut &cut = cuts[(size_t)order[k]];
            const PipeList &l = lists[(size_t)cut.list];
            PipeItem &it = out.items[k];
            it.set = l.set;
            it.task0 = (int32_t)t0;
            it.ntasks = (int32_t)cut.task.size();
            it.nstrips = (int32_t)cut.nstrips;
            it.rsteps = l.rsteps;
            it.nrounds = (it.nstrips + 3) / 4;
            it.bnd_off = bnd;
            bnd += (int64_t)(it.rsteps + 16) * 32;
            out.steps += 4 * (int64_t)it.nrounds * it.rsteps;
            t0 += it.ntasks;
        }
        out.bnd_elems = bnd;
        out.tasks.resize((size_t)t0);
        out.lane_pair.resize((size_t)t0 * 32);
    }
    std::vector<int64_t> used((size_t)nt, 0);
    run_threads(nt, [&](int t, int n) {
        int64_t lo, hi;
        slice_of((int64_t)order.size(), t, n, lo, hi);
        for (int64_t k = lo; k < hi; ++k) {
            const Cut &cut = cuts[(size_t)order[(size_t)k]];
            const PipeItem &it = out.items[(size_t)k];
            const int32_t max_l1 = lens[out.set_one[(size_t)it.set * 32]];
            for (size_t q = 0; q < cut.task.size(); ++q) {
                const ScratchTask &x = st[(size_t)cut.task[q]];
                WaveTask &wt = out.tasks[(size_t)it.task0 + q];
                wt.two[0] = x.two; wt.two[1] = -1;
                wt.max_l1 = max_l1;
                wt.nstrips = x.nstrips;
                wt.bnd_off = 0; wt.tb_off = 0; wt.aux_off = 0;
                int32_t *lp = out.lane_pair.data() + ((size_t)it.task0 + q) * 32;
                for (int l = 0; l < 32; ++l) { lp[l] = x.pair[l]; used[(size_t)t] += x.pair[l] >= 0; }
            }
        }
    });
    for (int64_t v : used) out.lanes_used += v;
    mark("i. output");
    out.ok = true;
}

extern "C" int praline_sched_pipe_test(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int block_twos,
                                       int64_t wg_slots, int64_t cap_items, int64_t cap_tasks, int64_t cap_sets, int64_t *n_items_out,
                                       int64_t *n_tasks_out, int64_t *n_sets_out, int32_t *item_fields /* [cap_items][6]: set, task0, ntasks, nstrips, rsteps, nrounds */,
                                       int32_t *task_fields /* [cap_tasks][3]: two, max_l1, nstrips */, int32_t *lane_pair /* [cap_tasks][32] */,
                                       int32_t *set_one /* [cap_sets][32] */)
{
    PipeOptions opt;
    if (block_twos < 0) { opt.serial_front = true; block_twos = -block_twos; }   // (tests: the one-thread permutation version)
    if (block_twos > 0) opt.block_twos = block_twos;
    if (wg_slots > 0) opt.wg_slots = wg_slots;
    PipeSchedule s;
    build_pipe_schedule(lens, n_seqs, n_pairs, pairs, opt, s);
    *n_items_out = (int64_t)s.items.size();
    *n_tasks_out = (int64_t)s.tasks.size();
    *n_sets_out = (int64_t)s.set_one.size() / 32;
    if (!s.ok) return 1;
    if (*n_items_out > cap_items || *n_tasks_out > cap_tasks || *n_sets_out > cap_sets) return -1;
    for (size_t i = 0; i < s.items.size(); ++i) {
        const PipeItem &it = s.items[i];
        int32_t *f = item_fields + 6 * i;
        f[0] = it.set; f[1] = it.task0; f[2] = it.ntasks; f[3] = it.nstrips; f[4] = it.rsteps; f[5] = it.nrounds;
    }
    for (size_t t = 0; t < s.tasks.size(); ++t) {
        task_fields[3 * t] = s.tasks[t].two[0]; task_fields[3 * t + 1] = s.tasks[t].max_l1; task_fields[3 * t + 2] = s.tasks[t].nstrips;
    }
    std::copy(s.lane_pair.begin(), s.lane_pair.end(), lane_pair);
    std::copy(s.set_one.begin(), s.set_one.end(), set_one);
    return 0;
}

// ---- C entry point for the CPU unit tests (tests/test_scheduler_cpu.py; not part of libpraline_dp's ABI) ----
extern "C" int praline_sched_test(const int32_t *lens, int64_t n_pairs, const int32_t *pairs, int want_paths, int xcd_group,
                                  int64_t wave_slots, int64_t cap_tasks, int64_t cap_wg, int64_t *n_tasks_out,
                                  int32_t *task_fields /* [cap_tasks][4]: two, max_l1, nstrips, pad */,
                                  int32_t *lane_pair /* [cap_tasks][32] */, int64_t *n_wg_out,
                                  int32_t *wg_fields /* [cap_wg][6]: task[4], share, barriers */, int64_t *n_singles_out)
{
    SchedOptions opt;
    opt.want_paths = want_paths != 0;
    opt.xcd_group = xcd_group;
    opt.wave_slots = wave_slots;
    Schedule s;
    build_schedule(lens, n_pairs, pairs, opt, s);
    *n_tasks_out = (int64_t)s.tasks.size();
    *n_wg_out = (int64_t)s.wg.size();
    *n_singles_out = (int64_t)s.wg_singles.size();
    if ((int64_t)s.tasks.size() > cap_tasks || (int64_t)s.wg.size() > cap_wg) return -1;
    for (size_t t = 0; t < s.tasks.size(); ++t) {
        task_fields[4 * t + 0] = s.tasks[t].two[0];
        task_fields[4 * t + 1] = s.tasks[t].max_l1;
        task_fields[4 * t + 2] = s.tasks[t].nstrips;
        task_fields[4 * t + 3] = 0;
        for (int q = 0; q < 32; ++q) lane_pair[32 * t + q] = s.lane_pair[32 * t + q];
    }
    for (size_t w = 0; w < s.wg.size(); ++w) {
        for (int q = 0; q < 4; ++q) wg_fields[6 * w + q] = s.wg[w].task[q];
        wg_fields[6 * w + 4] = s.wg[w].share;
        wg_fields[6 * w + 5] = s.wg[w].barriers;
    }
    return 0;
}
