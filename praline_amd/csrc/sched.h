// sched.h -- host-side scheduling of a pair list onto wavefront tasks (no HIP in here: sched.cpp also builds with
// g++ for the CPU unit tests, tests/test_scheduler_cpu.py).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <new>
#include <vector>
#include "dp_types.h"

// Schedule arrays of millions of entries that the scheduler's threads fill completely: a vector whose resize() leaves new
// elements uninitialised (value-initialising them first was a single-threaded pass over 25 MB for a C3-sized path plan).
// Blocks of SCHED_BIG_BYTES and more - which malloc would map afresh and unmap again, however its thresholds are set:
// the pages of the 37 + 42 + 34 MB an all-of-C4 plan passes through cost more than the passes that fill them - are kept
// by the scheduler between calls (sched_big_alloc / sched_big_free, sched.cpp: up to PRALINE_SCHED_CACHE_MB, default 256;
// 0 or PRALINE_KEEP_HOST_MEMORY=0: nothing is kept).
constexpr size_t SCHED_BIG_BYTES = (size_t)32 << 20;
void *sched_big_alloc(size_t bytes);
void sched_big_free(void *p);
template <class T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(n * sizeof(T) >= SCHED_BIG_BYTES ? sched_big_alloc(n * sizeof(T)) : ::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { if (n * sizeof(T) >= SCHED_BIG_BYTES) sched_big_free(p); else ::operator delete(p); }
    template <class U> void construct(U *p) { ::new ((void *)p) U; }   // default-initialise: trivial types stay as they are
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(static_cast<A0 &&>(a0), static_cast<A &&>(a)...); }
    bool operator==(const NoInitAlloc &) const { return true; }
    bool operator!=(const NoInitAlloc &) const { return false; }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

// starts the scheduler's host threads (otherwise the first plan of a million pairs pays for them: ~1 ms)
void sched_warm_threads();

struct SchedOptions {
    bool want_paths = false;
    bool split_layout = true;     // split-strip kernels (32 pairs per task); false: the retired 64-lane task layout (scheduler unit tests only)
    int tp = 0;                   // 64-lane layout only: sequence-two groups per wave (0 = automatic)
    int xcd_group = -1;           // XCD placement group size (-1 = automatic: tasks / 128 clamped to 16..1024, 0 = none)
    bool shared_waves = true;     // build the four-wave workgroup lists
    int64_t wave_slots = 2048;    // resident wave slots assumed by the share search (256 CUs x 4 SIMDs x 2)
    bool snake = true;            // launch order: longest workgroups share a CU with the shortest
    bool wg_xcd = true;           // shared-wave workgroups: sequences two in eight contiguous runs, one per XCD
    bool pk16 = false;            // path plans that may run k_dp_pk16_tb (32 pairs per task; larger traceback planes)
    bool quad16 = false;          // path plans on one-hot arenas: 16 pairs per task (k_dp_quad_tb, dp_quad.hip.h)
    bool balance = false;         // experiment: choose the wave counts by the modelled busiest SIMD, not by the longest wave (no measured gain)
};

struct Schedule {
    bool split = true;
    int tp = 1;
    int lanes_per_task = 32;
    std::vector<WaveTask> tasks;          // launch order (XCD placement applied; max_l1 == 0: padding)
    RawVec<int32_t> lane_one, lane_pair;  // [tasks][lanes_per_task], -1 = empty lane
    RawVec<PairLoc> loc;                  // per pair
    std::vector<int64_t> tb_elems, aux_elems;   // per task scratch sizes (path plans)
    int64_t bnd_elems = 0;
    std::vector<WgDesc> wg;               // small batches: shared waves (empty when not applicable)
    std::vector<WgDesc> wg_singles;       // large batches: four independent tasks per workgroup
    RawVec<int64_t> slot_off;             // per pair: row offset of its path slot (capacity l1 + l2 + 2)
    int64_t path_cap = 0, cells = 0;
};

// Pipeline-workgroup schedule of a scores-only plan (k_dp_pipe): sets of 32 sequences one shared by the tasks of a block
// of sequences two, tasks cut into workgroup items.
struct PipeSchedule {
    bool ok = false;                      // false: the pair list does not suit the layout (the caller keeps the task schedule)
    std::vector<PipeItem> items;          // launch order
    RawVec<WaveTask> tasks;               // two[0], max_l1 (of the set), nstrips; the tasks of an item are consecutive
    std::vector<int32_t> set_one;         // [n_sets][32] arena index of each lane's sequence one (-1: no sequence)
    RawVec<int32_t> lane_pair;            // [n_tasks][32] pair index, -1 = no pair in this lane
    int64_t bnd_elems = 0;                // float2 elements of the wrap-around boundary columns
    int64_t lanes_used = 0, steps = 0;    // pairs placed; wave steps of the launch (4 x nrounds x rsteps summed over the items)
};
struct PipeOptions {
    int block_twos = 32;                  // sequences two per block (their sequences one are pooled into the sets); praline_plan_create
                                          // passes 32 for plans that are resident at once and 16 for larger ones
    int64_t wg_slots = 512;               // resident workgroups (256 CUs x 2)
    double min_fill = 0.55;               // give up below this share of occupied lanes
    bool serial_front = false;            // tests: build the sets and tasks with the one-thread permutation version
};
void build_pipe_schedule(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, const PipeOptions &opt,
                         PipeSchedule &out);

// Launch order for an ordered list of n work items: groups of G consecutive items on one XCD (block b runs on
// XCD b % 8), groups dealt round-robin over the XCDs.  Returns, per block, the item it runs (-1: padding).
std::vector<int64_t> xcd_group_order(int64_t n, int G);

// lens: length of every arena sequence; pairs: int32 [n_pairs][2] = (sequence one, sequence two), validated by
// the caller.
void build_schedule(const int32_t *lens, int64_t n_pairs, const int32_t *pairs, const SchedOptions &opt, Schedule &out);

// One (threaded) pass over a pair list: DP cells, the shortest sequence of any pair, the first pair with an index outside
// 0 .. n_seqs - 1 (-1: none; the other results are then meaningless).
void sched_pair_stats(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int64_t *cells, int *min_len,
                      int64_t *first_bad);
