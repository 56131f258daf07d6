// dp_reftile.hip.h -- reference-order match scores for the split-strip DP kernels (PRALINE_MATCH_REFERENCE plans on
// arenas of at most 32 symbols).
//
// cext_build_scores / score_match_prof_prof (praline/util/cext.c:33-97,389-420): per track set one float32 running sum
// over the nonzeros of row y of profile one (ascending, outer) and of row x of profile two (ascending, inner), each term
// evaluated as (p2 * S) * p1 with separately rounded multiplies (the reference is built -ffast-math, setup.py:28; see
// oracle/praline_oracle.c), the per-set sums added in list order to a float32 that starts at 0.  k_match_reft
// (dp_kernels.hip.h) evaluates exactly that, one thread per cell, and is bound by the L2 traffic of its table rows
// (256 bytes per cell).  Here the same terms in the same order are evaluated
//   * two cells per lane with packed fp32 (v_pk_mul_f32 / v_pk_add_f32 round each half like v_mul_f32 / v_add_f32: the
//     sums are bit-identical), the y-independent half of every term taken from a table that interleaves two adjacent
//     columns x:   T2[i][pair row][b][e] = fl(p2[x][j_b] * S[i][j_b]),  x = 2 k + e,  j_b = b-th nonzero of row x
//     (zero past the row's nonzeros, where j_b lies in another track set than i, and past the end of the sequence:
//     adding fl(0 * p1) = +-0 leaves a float32 sum that started at +0 unchanged);
//   * with the table rows of a 128-column chunk of sequence two held in LDS for the whole workgroup (A x 4 KiB), and
//     re-read from there once per SYMBOL and group of 16 rows y: a wave keeps the 16 rows' accumulators in registers,
//     walks the symbols i that are nonzero in any of them (ascending - per cell the reference's order) and runs the
//     8 multiplies + 8 chained adds of a (row, symbol) behind a scalar test of the row's bit;
//   * written in the layout the DP kernels read (dense tile): per task (32 sequences one x one sequence two)
//         m[strip][row y = 0 .. max_l1 + pad][half h][pair j][16 columns]          (floats; row 0 is not used)
//     so that lane (j, h) of the DP wave fetches the 16 scores of its row with four 16-byte loads of one 64-byte line.
// VALU bound: 64 packed instructions per cell pair; measured rate of the bare body 34 Tterm/s (scripts/micro/pk_rate.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dp_reftile.h"

typedef float rt_f2 __attribute__((ext_vector_type(2)));
typedef float rt_f4 __attribute__((ext_vector_type(4)));

// T2 of one sequence per workgroup
__global__ void k_build_reft2(const float *__restrict__ raw, const float *__restrict__ S, int A, const int32_t *__restrict__ row_off_raw,
                              const int32_t *__restrict__ len, const int64_t *__restrict__ pr_off, int64_t PR,
                              const unsigned char *__restrict__ nzidx, const unsigned char *__restrict__ nzcnt,
                              const int32_t *__restrict__ set_lo, int n_sets, int TB, float *__restrict__ T2)
{
    const int seq = blockIdx.x;
    const int L = len[seq];
    const int npr = (L + 1) >> 1;
    const int64_t r0 = row_off_raw[seq], p0 = pr_off[seq];
    for (int e = threadIdx.x; e < A * npr * 2; e += blockDim.x) {
        const int i = e / (npr * 2), rem = e - i * (npr * 2);
        const int k = rem >> 1, half = rem & 1, x = 2 * k + half;
        int s = 0;
        while (s + 1 < n_sets && i >= set_lo[s + 1]) ++s;
        const int lo = set_lo[s], hi = set_lo[s + 1];
        float *out = T2 + (((int64_t)i * PR + p0 + k) * TB) * 2 + half;
        const int n = x < L ? (int)nzcnt[r0 + x] : 0;
        const unsigned char *idx = nzidx + (r0 + x) * A;
        const float *p2 = raw + (r0 + x) * A, *srow = S + (int64_t)i * A;
        for (int b = 0; b < TB; ++b) {
            float t = 0.0f;
            if (b < n) {
                const int j = idx[b];
                if (j >= lo && j < hi) t = __fmul_rn(p2[j], srow[j]);
            }
            out[2 * b] = t;
        }
    }
}

// acc += t * v for both halves, v a wave-uniform value in the low word of an SGPR pair
__device__ __forceinline__ rt_f2 rt_term(rt_f2 acc, rt_f2 t, unsigned long long v_sgpr)
{
    rt_f2 p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(t), "s"(v_sgpr));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(p));
    return acc;
}

__device__ __forceinline__ rt_f2 rt_add(rt_f2 a, rt_f2 b)
{
    rt_f2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// One workgroup: the table rows of 128 columns (64 pair rows) in LDS - a chunk of the sequences two of a GROUP of tasks
// with the same 32 sequences one, laid end to end (so only a group's last chunk has idle lanes) -; its waves take the
// work items (pair j, G rows y of that pair's sequence one) from a shared counter.  A wave holds the item's G x 2
// accumulators in registers, keeps the item's rows transposed in LDS (sv[symbol][row]: lane r reads row r's value of the
// symbol at hand, v_readlane hands it to the packed multiplies as an SGPR) and fetches the NEXT item's rows from memory
// while it works on this one.
// LDS: tile [A][TB / 2][64] float4 | per wave: sv [A][G] floats (+ 128 bytes) | counter, prefix [33] | colrow [64].
template <int TB, bool MULTI, int G>
__global__ __launch_bounds__((MULTI || G == 32) ? 768 : 1024) void k_match_tile(RefTileArgs g)
{
    static_assert(G == 16 || G == 32, "16 or 32 rows per work item");
    constexpr int NQ = TB / 2;                  // 16-byte pieces per (symbol, pair row)
    constexpr int NX = (G * 32 + 63) / 64;      // registers of a lane's share of an item's rows (alphabets of up to 32 symbols)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int A = g.A;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rt_f4 *tile = reinterpret_cast<rt_f4 *>(lds);
    const int wave_bytes = A * G * 4 + 128;
    char *wbase = lds + (size_t)A * NQ * 1024 + (size_t)wave * wave_bytes;
    float *sv = reinterpret_cast<float *>(wbase);
    int *ctl = reinterpret_cast<int *>(lds + (size_t)A * NQ * 1024 + (size_t)g.waves * wave_bytes);   // [0]: counter, [1 .. 34): prefix
    long long *colrow = reinterpret_cast<long long *>(ctl + 64);                                     // [64]: arena pair row of column pair l, -1: none

    const RefTileBlock blk = g.blocks[blockIdx.x];
    const int32_t *cum = g.grp + blk.base, *tix = cum + blk.count + 1;
    // this lane's column pair: pair row P of the group's sequences two laid end to end -> (task, pair row inside its two)
    const int P = blk.chunk * 64 + lane;
    int lo = 0;
    {
        int hi = blk.count;                     // cum[lo] <= P < cum[hi]  (P past the end: the last task, masked below)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] <= P) lo = mid; else hi = mid;
        }
    }
    const int my_task = tix[lo];
    const WaveTask tk = g.tasks[my_task];
    const int two = tk.two[0];
    const int L2 = g.len[two];
    const int kk = P - cum[lo];                 // pair row inside the sequence
    const bool col_ok = P < cum[blk.count];
    const int nstrips = (L2 + 31) >> 5;
    const int rows_t = tk.max_l1 + PRALINE_DENSE_PAD;
    // (the group's record counts whole strips per sequence: pair rows past the sequence's end produce zeros)
    if (wave == 0) colrow[lane] = (col_ok && 2 * kk < L2) ? g.pr_off[two] + kk : -1ll;
    __syncthreads();

    // ---- the chunk's table rows ----
    for (int e = threadIdx.x; e < A * NQ * 64; e += blockDim.x) {
        const int l = e & 63, iq = e >> 6;      // iq = i * NQ + q
        const int i = iq / NQ, q = iq - i * NQ;
        const long long pr = colrow[l];
        rt_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (pr >= 0) v = *reinterpret_cast<const rt_f4 *>(g.T2 + (((int64_t)i * g.PR + pr) * TB) * 2 + q * 4);
        tile[e] = v;
    }
    const int task0 = tix[0];                   // (every task of the group has the same 32 sequences one)
    if (threadIdx.x < 32) {
        const int one = g.lane_one[task0 * 32 + threadIdx.x];
        const int ng = one >= 0 ? (g.len[one] + G - 1) / G : 0;
        // exclusive prefix over the 32 pairs
        int incl = ng;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if ((int)threadIdx.x >= off) incl += up;
        }
        ctl[1 + threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) { ctl[0] = 0; ctl[1] = 0; }
    }
    __syncthreads();
    const int n_items = ctl[1 + 32];

    // this lane's two columns and where they go
    const int x0 = 2 * kk;
    const bool store_ok = col_ok && x0 < nstrips * 32;
    float *out_lane = g.m + g.dense_off[my_task] + ((int64_t)(x0 >> 5) * rows_t * 2 + ((x0 >> 4) & 1)) * 512 + (x0 & 15);

    // work items: (pair j, row group).  `grab` takes the next one and starts the loads of its rows (nx: element e = lane
    // + 64 k of the G x A block of raw profile rows, zero past the sequence's end)
    struct Item { int j, y0, ny; };
    float nx[NX];
    auto grab = [&](Item &it) __attribute__((always_inline)) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&ctl[0], 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) { it.j = -1; it.y0 = 0; it.ny = 0; return; }
        // the pair whose groups include this item: the number of pairs whose inclusive prefix is <= item
        const int incl_l = ctl[2 + (lane & 31)];
        it.j = __builtin_popcountll(__ballot(lane < 32 && incl_l <= item));
        const int rg = item - __builtin_amdgcn_readfirstlane(ctl[1 + it.j]);
        const int one = g.lane_one[task0 * 32 + it.j];
        const int L1 = g.len[one];
        it.y0 = rg * G;
        it.ny = min(G, L1 - it.y0);
        const float *rows1 = g.raw + ((int64_t)g.row_off_raw[one] + it.y0) * A;
        const int n_valid = it.ny * A;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int e = lane + 64 * k;
            nx[k] = e < n_valid ? rows1[e] : 0.0f;
        }
    };
    Item cur, nxt;
    grab(nxt);
    while (nxt.j >= 0) {
        cur = nxt;
        // ---- the item's rows, transposed: sv[i][yy] ----
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int e = lane + 64 * k;
            if (e < G * A) {
                const int yy = e / A, i = e - yy * A;
                sv[i * G + yy] = nx[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        grab(nxt);   // (its loads arrive while this item is computed)
        // symbol i (lane i): which of the rows hold a nonzero; the work list = the symbols with any, ascending
        unsigned mask = 0;
        int my_set = 0;
        if (lane < A) {
#pragma unroll
            for (int yy = 0; yy < G; ++yy)
                mask |= ((__float_as_uint(sv[lane * G + yy]) & 0x7fffffffu) != 0u) ? (1u << yy) : 0u;   // != 0 (cext.c: nonzero lists)
            if constexpr (MULTI) {
                while (my_set + 1 < g.n_sets && lane >= g.set_lo[my_set + 1]) ++my_set;
            }
        }
        // the work list = the set bits of `hits`; MULTI: `folds` marks the symbols that are the first of a later track set
        // (the per-set sum is folded into the score before them)
        const unsigned long long hits = __ballot(mask != 0u);
        unsigned long long folds = 0ull;
        if constexpr (MULTI) {
            const unsigned long long below = hits & ((1ull << lane) - 1ull);
            const int prev = below ? 63 - __builtin_clzll(below) : lane;
            const int prev_set = __shfl(my_set, prev);
            folds = __ballot(mask != 0u && below != 0ull && prev_set != my_set);
        }
        unsigned long long rest = hits;

        rt_f2 acc[G], score[MULTI ? G : 1];
#pragma unroll
        for (int yy = 0; yy < G; ++yy) acc[yy] = rt_f2{0.0f, 0.0f};
#pragma unroll
        for (int yy = 0; yy < (MULTI ? G : 1); ++yy) score[yy] = rt_f2{0.0f, 0.0f};

        // one (symbol, G rows) entry: the symbol's table row of this lane's column pair, the rows that hold the symbol
        // and their values (lane r: row r)
        struct Entry { unsigned rows; bool live, fold; rt_f4 t[NQ]; float v; };
        auto fetch = [&](Entry &en) __attribute__((always_inline)) {
            int i = 0;
            en.rows = 0u;                                   // past the list: symbol 0, no rows
            en.live = rest != 0ull;
            en.fold = false;
            if (en.live) {
                i = __builtin_ctzll(rest);
                rest &= rest - 1ull;
                en.rows = (unsigned)__builtin_amdgcn_readlane((int)mask, i);
                if constexpr (MULTI) en.fold = ((folds >> i) & 1ull) != 0ull;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) en.t[q] = tile[(i * NQ + q) * 64 + lane];
            en.v = sv[i * G + (lane & (G - 1))];
        };
        auto work = [&](const Entry &en) __attribute__((always_inline)) {
            if constexpr (MULTI) {
                if (en.fold) {
#pragma unroll
                    for (int yy = 0; yy < G; ++yy) { score[yy] = rt_add(score[yy], acc[yy]); acc[yy] = rt_f2{0.0f, 0.0f}; }
                }
            }
            const unsigned rows = en.rows;
#pragma unroll
            for (int yy = 0; yy < G; ++yy) {
                if (rows & (1u << yy)) {
                    const unsigned long long vs = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(en.v), yy);
                    rt_f2 a = acc[yy];
#pragma unroll
                    for (int b = 0; b < TB; ++b) {
                        const rt_f4 tq = en.t[b >> 1];
                        const rt_f2 t = (b & 1) ? rt_f2{tq.z, tq.w} : rt_f2{tq.x, tq.y};
                        a = rt_term(a, t, vs);
                    }
                    acc[yy] = a;
                }
            }
        };
        Entry ea, eb;
        fetch(ea);
        while (ea.live) {
            fetch(eb);
            work(ea);
            fetch(ea);
            work(eb);
        }
        // ---- the scores of the item's rows ----
        if (store_ok) {
            float *o = out_lane + (int64_t)cur.j * 16 + (int64_t)(cur.y0 + 1) * 1024;
#pragma unroll
            for (int yy = 0; yy < G; ++yy) {
                if (yy < cur.ny) {
                    rt_f2 r = acc[yy];
                    if constexpr (MULTI) r = rt_add(score[yy], acc[yy]);
                    __builtin_nontemporal_store(r, reinterpret_cast<rt_f2 *>(o + (int64_t)yy * 1024));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // (the next item rewrites sv)
    }
}
