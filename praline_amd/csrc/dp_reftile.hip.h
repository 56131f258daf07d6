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

// acc += t * v for both halves; the second factor is the low (HI = false) or the high half of vv
template <bool HI> __device__ __forceinline__ rt_f2 rt_term(rt_f2 acc, rt_f2 t, rt_f2 vv)
{
    rt_f2 p;
    if constexpr (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p) : "v"(t), "v"(vv));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(t), "v"(vv));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(p));
    return acc;
}

__device__ __forceinline__ rt_f2 rt_add(rt_f2 a, rt_f2 b)
{
    rt_f2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// One workgroup: the table rows of 128 columns (64 pair rows) of a task's sequence two in LDS; its waves take the work
// items (pair j of the task, 16 rows y of that pair's sequence one) from a shared counter.
// LDS: tile [A][TB / 2][64] float4 | per wave: sv [A][16] floats (+ 128 bytes) | counter, prefix [33].
template <int TB, bool MULTI>
__global__ __launch_bounds__(MULTI ? 768 : 1024) void k_match_tile(RefTileArgs g)
{
    constexpr int NQ = TB / 2;                  // 16-byte pieces per (symbol, pair row)
    constexpr int G = PRALINE_REFTILE_ROWS;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int A = g.A;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    rt_f4 *tile = reinterpret_cast<rt_f4 *>(lds);
    const int wave_bytes = A * G * 4 + 128;
    char *wbase = lds + (size_t)A * NQ * 1024 + (size_t)wave * wave_bytes;
    float *sv = reinterpret_cast<float *>(wbase);
    int *ctl = reinterpret_cast<int *>(lds + (size_t)A * NQ * 1024 + (size_t)g.waves * wave_bytes);   // [0]: counter, [1 .. 34): prefix

    const RefTileBlock blk = g.blocks[blockIdx.x];
    const WaveTask tk = g.tasks[blk.task];
    const int two = tk.two[0];
    const int L2 = g.len[two];
    const int npr = (L2 + 1) >> 1;
    const int k0 = blk.chunk * 64;              // first pair row of the chunk (inside the sequence)
    const int64_t pr0 = g.pr_off[two] + k0;
    const int nstrips = (L2 + 31) >> 5;
    const int rows_t = tk.max_l1 + PRALINE_DENSE_PAD;

    // ---- the chunk's table rows ----
    for (int e = threadIdx.x; e < A * NQ * 64; e += blockDim.x) {
        const int l = e & 63, iq = e >> 6;      // iq = i * NQ + q
        const int i = iq / NQ, q = iq - i * NQ;
        rt_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (k0 + l < npr) v = *reinterpret_cast<const rt_f4 *>(g.T2 + (((int64_t)i * g.PR + pr0 + l) * TB) * 2 + q * 4);
        tile[e] = v;
    }
    if (threadIdx.x < 32) {
        const int one = g.lane_one[blk.task * 32 + threadIdx.x];
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
    const int x0 = blk.chunk * PRALINE_REFTILE_COLS + 2 * lane;
    const bool store_ok = x0 < nstrips * 32;
    float *out_lane = g.m + g.dense_off[blk.task] + ((int64_t)(x0 >> 5) * rows_t * 2 + ((x0 >> 4) & 1)) * 512 + (x0 & 15);

    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(&ctl[0], 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        // the pair whose groups include this item: the number of pairs whose inclusive prefix is <= item
        const int incl_l = ctl[2 + (lane & 31)];
        const int j = __builtin_popcountll(__ballot(lane < 32 && incl_l <= item));
        const int rg = item - __builtin_amdgcn_readfirstlane(ctl[1 + j]);
        const int one = g.lane_one[blk.task * 32 + j];
        const int L1 = g.len[one];
        const int y0 = rg * G;
        const int ny = min(G, L1 - y0);
        const float *rows1 = g.raw + ((int64_t)g.row_off_raw[one] + y0) * A;

        // ---- the 16 rows of sequence one, transposed: sv[i][yy] ----
        for (int e = lane; e < G * A; e += 64) {
            const int yy = e / A, i = e - yy * A;
            sv[i * G + yy] = yy < ny ? rows1[e] : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
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
        // the work list = the symbols with a nonzero in any of the rows, ascending: the set bits of `hits`; lane i's word =
        // its rows | (MULTI: the per-set sum is folded into the score before this symbol, the first of a later set) << 16
        const unsigned long long hits = __ballot(mask != 0u);
        unsigned word = mask;
        if constexpr (MULTI) {
            const unsigned long long below = hits & ((1ull << lane) - 1ull);
            const int prev = below ? 63 - __builtin_clzll(below) : lane;
            const int prev_set = __shfl(my_set, prev);
            if (below != 0ull && prev_set != my_set) word |= 1u << 16;
        }
        unsigned long long rest = hits;

        rt_f2 acc[G], score[MULTI ? G : 1];
#pragma unroll
        for (int yy = 0; yy < G; ++yy) acc[yy] = rt_f2{0.0f, 0.0f};
#pragma unroll
        for (int yy = 0; yy < (MULTI ? G : 1); ++yy) score[yy] = rt_f2{0.0f, 0.0f};

        // one (symbol, 16 rows) entry: the symbol's table row of this lane's column pair and its 16 row values
        struct Entry { unsigned w; rt_f4 t[NQ]; rt_f4 v[G / 4]; };
        auto fetch = [&](Entry &en) __attribute__((always_inline)) {
            int i = 0;
            en.w = 0u;                                      // past the list: symbol 0, no rows
            if (rest != 0ull) {
                i = __builtin_ctzll(rest);
                rest &= rest - 1ull;
                en.w = (unsigned)__builtin_amdgcn_readlane((int)word, i);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) en.t[q] = tile[(i * NQ + q) * 64 + lane];
#pragma unroll
            for (int q = 0; q < G / 4; ++q) en.v[q] = *reinterpret_cast<const rt_f4 *>(sv + i * G + 4 * q);
        };
        auto work = [&](const Entry &en) __attribute__((always_inline)) {
            if constexpr (MULTI) {
                if (en.w & (1u << 16)) {
#pragma unroll
                    for (int yy = 0; yy < G; ++yy) { score[yy] = rt_add(score[yy], acc[yy]); acc[yy] = rt_f2{0.0f, 0.0f}; }
                }
            }
            const unsigned rows = en.w & 0xffffu;
#pragma unroll
            for (int yy = 0; yy < G; ++yy) {
                if (rows & (1u << yy)) {
                    const rt_f4 vq = en.v[yy >> 2];
                    const rt_f2 vv = (yy & 2) ? rt_f2{vq.z, vq.w} : rt_f2{vq.x, vq.y};
                    rt_f2 a = acc[yy];
#pragma unroll
                    for (int b = 0; b < TB; ++b) {
                        const rt_f4 tq = en.t[b >> 1];
                        const rt_f2 t = (b & 1) ? rt_f2{tq.z, tq.w} : rt_f2{tq.x, tq.y};
                        a = (yy & 1) ? rt_term<true>(a, t, vv) : rt_term<false>(a, t, vv);
                    }
                    acc[yy] = a;
                }
            }
        };
        Entry ea, eb;
        fetch(ea);
        while (ea.w != 0u) {
            fetch(eb);
            work(ea);
            fetch(ea);
            work(eb);
        }
        // ---- the scores of the item's rows ----
        if (store_ok) {
            float *o = out_lane + (int64_t)j * 16 + (int64_t)(y0 + 1) * 1024;
#pragma unroll
            for (int yy = 0; yy < G; ++yy) {
                if (yy < ny) {
                    rt_f2 r = acc[yy];
                    if constexpr (MULTI) r = rt_add(score[yy], acc[yy]);
                    __builtin_nontemporal_store(r, reinterpret_cast<rt_f2 *>(o + (int64_t)yy * 1024));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // (the next item rewrites sv)
    }
}
