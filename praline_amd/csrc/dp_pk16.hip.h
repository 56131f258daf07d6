// dp_pk16.hip.h -- k_dp_pk16_tb: fill WITH packed traceback for plain sequences under INTEGER scoring, two pairs per lane in
// packed 16-bit integers.
//
// Why.  k_dp_quad_tb (dp_quad.hip.h) is bound by VALU issue: 17 float32 operations per cell (9 for the three states, 8 for
// the four tie flags), 2.2 TCUPS with every memory operation removed.  One-hot profiles with an integral exchange matrix
// and integral gap scores (after scaling by 2^k, praline_plan_run's INTS test) make every DP value an integer; when the
// largest one, (L1 + L2 + 2) * max(|S|, |open|, |extend|) * 2^k, stays below 32 000 they fit int16 and the VOP3P
// instructions (v_pk_add_i16 / v_pk_sub_i16 with clamp, v_pk_max_i16) update TWO cells per operation.  The cells of a
// register pair must be independent: the low halves belong to pair A (slot p of the task), the high halves to pair B (slot
// p + 16) - two of the task's 32 pairs, same sequence two, same column, same row index.
//
// Exactness.  Integer arithmetic on the scaled values IS the reference's float arithmetic (cext.c:99-306) as long as nothing
// saturates: the host checks the bound.  -inf is -32768; the saturating adds keep it there (gap scores are <= 0), the
// saturating subtractions of the tie flags keep their signs, and max3(md, ud, ld) of an interior cell is always finite, so
// no match score is ever added to it.  Flags, priorities, end cells: split16_tb_step's (INTS flavour: a state ties exactly
// when its predecessor does).  Scores, end cells and paths are bit-identical to the float kernels'
// (tests/test_gpu_parity.py::test_packed_int16_paths_equal_the_float_kernels).
//
// Layout.  As k_dp_quad_tb: lane l = slot p = l & 15 (pairs p and p + 16 of the task), quarter q = l >> 4 = strip columns
// 8 q + 1 .. 8 q + 8; step t: rows 2 (t - q) - 1 and 2 (t - q); hand-off from lane l - 16 with ds_bpermute (three packed
// words per row: both pairs at once); quarter 0 reads the strip's boundary column (uint4 (M, U, L, -) [row][16 slots]),
// quarter 3 writes the next one.  Match scores: int16 lookup table per strip in LDS ([symbol][32 columns]), one 16-byte read
// per (pair, row) and one v_perm_b32 per packed cell to interleave the two pairs.
// Operations per packed cell (two cells): 9 for M / U / L, 1 interleave, 4 saturating subtractions, 2 v_perm_b32 that gather
// the eight sign bytes, 4 to shift them into two accumulators = 20, i.e. 10 per cell.
//
// Traceback planes: uint4 [strip][step][64 lanes]; (.x, .y) = the odd row 2 (t - q) - 1, (.z, .w) = the even row; first
// word: match source low bits A (8) | B << 8 | high bits A << 16 | B << 24 (code = lo | hi << 1: 1 MM / 2 MU / 3 ML / 0 stop);
// second word: "U from extend" A | B << 8 | "L from extend" A << 16 | B << 24.  k_traceback reads them as layout 3.
// End-cell scratch of the semiglobal modes: floats, [y][3][32] / [x - 1][3][32] as the 32-pair strip kernels write it.
#pragma once
#include "dp_quad.hip.h"

#ifndef PRALINE_PK16_ABLATE
#define PRALINE_PK16_ABLATE 0   // timing experiments only (results invalid): 1 no flag stores, 2 no boundary column traffic, 4 no symbol loads,
                                // 8 no quarter-to-quarter hand-off, 16 no per-row bookkeeping, 32 no flag arithmetic
#endif
#ifndef PRALINE_PK16_STORE_NT
#define PRALINE_PK16_STORE_NT 1   // flag words: streaming stores
#endif
#ifndef PRALINE_PK16_MASK1_WAVES
#define PRALINE_PK16_MASK1_WAVES 3   // ... the instances with one rectangle slot per pair
#endif
#ifndef PRALINE_PK16_WAVES
#define PRALINE_PK16_WAVES 3   // waves per SIMD the instances without rectangles are compiled for (those with: 2)
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define PRALINE_PK_NEG (-32768)

__device__ __forceinline__ s16x2 pk_adds(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16x2 pk_subs(s16x2 a, s16x2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ s16x2 pk_maxs(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ unsigned pk_u(s16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ s16x2 pk_of(unsigned v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 pk_dup(int v) { const short s = (short)v; return s16x2{s, s}; }
__device__ __forceinline__ s16x2 pk_from_left(s16x2 v, int addr) { return pk_of((unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)pk_u(v))); }

__host__ __device__ constexpr int pk16_stride() { return 64 + 16; }   // bytes per table row: 32 columns of int16 + padding
__host__ __device__ constexpr int pk16_table_bytes(int NR) { return (16 * NR + 1) * pk16_stride(); }

// v[idx] for a per-lane idx in 0..7
__device__ __forceinline__ s16x2 pk_select8(const s16x2 (&v)[8], int idx)
{
    const bool b0 = idx & 1, b1 = idx & 2, b2 = idx & 4;
    const unsigned t0 = b0 ? pk_u(v[1]) : pk_u(v[0]), t1 = b0 ? pk_u(v[3]) : pk_u(v[2]), t2 = b0 ? pk_u(v[5]) : pk_u(v[4]), t3 = b0 ? pk_u(v[7]) : pk_u(v[6]);
    const unsigned u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2;
    return pk_of(b2 ? u1 : u0);
}

struct PkIn { s16x2 inM, inU, inL; };   // states of the left neighbour cell (y, first column - 1), both pairs

// the scaled integer of a float DP value (exact by the host's checks); -inf -> -32768
__device__ __forceinline__ int pk_scaled(float v, float scale) { return v == PRALINE_NEG_INF ? PRALINE_PK_NEG : (int)(v * scale); }

// One DP row of this lane's 8 columns, both pairs.  On entry Mp / Up / Lp hold the previous row, (dM, dU, dL) the states of the
// cell left of it (y - 1, first column - 1); on exit they hold this row and (dM, dU, dL) = `in`.  zA / zB: the masked columns
// of pair A / B (MK).  Returns the row's two flag words.
template <bool LOCAL, bool MK>
__device__ __forceinline__ uint2 pk16_row(const s16x2 (&m)[8], const PkIn &in, s16x2 (&Mp)[8], s16x2 (&Up)[8], s16x2 (&Lp)[8],
                                          s16x2 &dM, s16x2 &dU, s16x2 &dL, s16x2 go2, s16x2 ge2, unsigned zA, unsigned zB)
{
    s16x2 md = dM, ud = dU, ld = dL;
    s16x2 mleft = in.inM, lleft = in.inL;
    unsigned w1 = 0, w2 = 0, w3 = 0;   // bytes: (nm A, nm B, nu A, nu B), (u A, u B, l A, l B), (stop A, stop B, -, -); column c = bit c
    const s16x2 zero = {0, 0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // cext.c:185-222 on exact integers: max3 + m IS the maximum of the three candidate sums
        const s16x2 Mref = pk_maxs(pk_maxs(md, ud), ld);
        s16x2 M = pk_adds(Mref, m[c]);
        const s16x2 uo = pk_adds(Mp[c], go2), ue = pk_adds(Up[c], ge2);
        s16x2 U = pk_maxs(uo, ue);
        const s16x2 lo = pk_adds(mleft, go2), le = pk_adds(lleft, ge2);
        s16x2 Lc = pk_maxs(lo, le);
        // first-match flags (cext.c:224-295): "MM is not the maximum", "MU is not", "U from extend", "L from extend" = the
        // signs of four differences; their high bytes gathered by two v_perm_b32 and shifted into the accumulators
        if (!(PRALINE_PK16_ABLATE & 32)) {
        const unsigned d_nm = pk_u(pk_subs(md, Mref)), d_nu = pk_u(pk_subs(ud, Mref));
        const unsigned d_u = pk_u(pk_subs(uo, ue)), d_l = pk_u(pk_subs(lo, le));
        const unsigned p1 = __builtin_amdgcn_perm(d_nu, d_nm, 0x07050301u);
        const unsigned p2 = __builtin_amdgcn_perm(d_l, d_u, 0x07050301u);
        w1 = (p1 & 0x80808080u) | (w1 >> 1);
        w2 = (p2 & 0x80808080u) | (w2 >> 1);
        }
        if constexpr (LOCAL) {
            const unsigned p3 = __builtin_amdgcn_perm(0u, pk_u(M), 0x0c0c0301u);   // (selector 0x0c: a zero byte)
            w3 = (p3 & 0x00008080u) | (w3 >> 1);
            M = pk_maxs(M, zero);                                             // cext.c:208-209
        }
        if constexpr (MK) {
            // cext.c:141-149: masked cells hold zeros (stop code: row end)
            // (v_bfe_i32 spreads each pair's bit over a word, v_bfi_b32 joins the halves and clears the states: six operations)
            // (opaque to the optimiser, which turns the bit tests into compare / select pairs with their wait states)
            unsigned kA, kB, kill;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kA) : "v"(zA), "n"(c));
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kB) : "v"(zB), "n"(c));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(kill) : "s"(0x0000ffffu), "v"(kA), "v"(kB));
            M = pk_of(pk_u(M) & ~kill); U = pk_of(pk_u(U) & ~kill); Lc = pk_of(pk_u(Lc) & ~kill);
        }
        md = Mp[c]; ud = Up[c]; ld = Lp[c];
        Mp[c] = M; Up[c] = U; Lp[c] = Lc;
        mleft = M; lleft = Lc;
        // keep the flag shifts with their cells (see quad_row)
        if ((c & 1) == 1) {
            asm volatile("" : "+v"(w1), "+v"(w2));
            if constexpr (LOCAL) asm volatile("" : "+v"(w3));
        }
    }
    dM = in.inM; dU = in.inU; dL = in.inL;
    const unsigned nm = w1 & 0xffffu, nu = w1 >> 16;
    unsigned go_on = 0xffffu;
    if constexpr (LOCAL) go_on &= ~w3;
    if constexpr (MK) go_on &= ~(zA | (zB << 8));
    const unsigned hi = nm & go_on, lo_bits = (~nm | nu) & go_on;
    return make_uint2(lo_bits | (hi << 16), w2);
}

// NR: 16-wide symbol ranges of the lookup table (1: <= 16 active symbols, 2: <= 32).  MASK: zero rectangles in registers
// (MASK = 1, 2 or PRALINE_MAX_RECTS of them per pair: the first MASK slots of a pair's list - a Waterman-Eggert pass k holds
// k - 1; two pairs' rectangles cost 8 registers each, and with one per pair the kernel keeps three waves per SIMD).
// scale = 2^k: DP values are stored as value * scale.
// CHAIN (plans of few tasks: one alignment, the merge steps of the progressive MSA, C2-sized batches): one wave per task AND
// strip, 64-thread blocks in strip-major order, the strips of a task pipelined across workgroups as in k_dp_split16_tb's
// chain mode (dp_split16_tb.hip.h): every strip boundary has its own column (uint4 [strip][row][16] at tk.bnd_off), quarter 3
// stores its rows with agent-scope write-through stores and publishes the finished row count every `chain_every` rows;
// the wave of the next strip polls that count before it issues the loads of rows it has not seen published.  Local
// alignments report one first-argmax candidate per strip (chain_cand, k_chain_local_end picks per pair).
template <int NR, bool LOCAL, int MASK, bool CHAIN = false>
__global__ __launch_bounds__(CHAIN ? 64 : 256, MASK > 1 ? 2 : (MASK == 1 ? PRALINE_PK16_MASK1_WAVES : PRALINE_PK16_WAVES)) void k_dp_pk16_tb(Arena16Dev ar, const WaveTask *__restrict__ tasks,
                                                       const int32_t *__restrict__ lane_one, const int32_t *__restrict__ lane_pair,
                                                       uint4 *bnd, uint4 *__restrict__ tb, float *__restrict__ aux, RectList rl,
                                                       float *__restrict__ scores, int32_t *__restrict__ end_cells, RunParams rp,
                                                       int n_tasks, float scale, int *chain_flags = nullptr, int chain_stride = 0,
                                                       float4 *chain_cand = nullptr, int chain_every = 6)
{
    __shared__ __attribute__((aligned(16))) char lookup_all[(CHAIN ? 1 : 4) * pk16_table_bytes(NR)];   // one table per wave of the block
    const int wv = CHAIN ? 0 : (int)(threadIdx.x >> 6);
    const int task = CHAIN ? (int)(blockIdx.x % n_tasks) : (int)blockIdx.x * 4 + wv;
    const int chain_strip = CHAIN ? (int)(blockIdx.x / n_tasks) : 0;
    if (task >= n_tasks) return;          // (the waves of a block are independent: no block-level barrier below)
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, q = lane >> 4;
    const WaveTask tk = tasks[task];
    const int base = task * 32;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    const float go = rp.go1, ge = rp.ge1;
    const float inv = 1.0f / scale;
    const s16x2 go2 = pk_dup((int)(go * scale)), ge2 = pk_dup((int)(ge * scale));
    const bool neg_gaps = go < 0.0f && ge < 0.0f;

    const int oneA = lane_one[base + p], oneB = lane_one[base + 16 + p];
    const int two = tk.two[0];
    const bool haveA = oneA >= 0, haveB = oneB >= 0;
    const int pairA = haveA ? lane_pair[base + p] : -1, pairB = haveB ? lane_pair[base + 16 + p] : -1;
    const int L1A = haveA ? ar.len[oneA] : 0, L1B = haveB ? ar.len[oneB] : 0;
    const int L2 = ar.len[two];
    const int nstrips = (L2 + 31) >> 5;
    const int clast = (L2 - 1) & 31;
    const bool own_last = (clast >> 3) == q;
    int cidx = clast & 7;
    asm volatile("" : "+v"(cidx));
    const int max_l1 = tk.max_l1;
    const int nsteps = PRALINE_QUAD_STEPS(max_l1);           // plane rows per strip
    const int run_steps = (max_l1 + 1) / 2 + 3;              // quarter 3 reaches row max_l1 at this step
    const int left_addr = ((lane - 16) & 63) * 4;            // ds_bpermute source: lane l - 16

    // shortest sequence one of the task (wave-uniform): no lane can be at a last row before it
    int min_l1 = min(haveA ? L1A : 0x7fffffff, haveB ? L1B : 0x7fffffff);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) min_l1 = min(min_l1, __shfl_xor(min_l1, off));
    min_l1 = __builtin_amdgcn_readfirstlane(min_l1);

    char *lookup_tab = lookup_all + wv * pk16_table_bytes(NR);
    const char *tab_lane = lookup_tab + q * 16;              // this lane's 8 columns of a table row
    const unsigned char *psymA = ar.sym8 + (haveA ? ar.row_off[oneA] : 0), *psymB = ar.sym8 + (haveB ? ar.row_off[oneB] : 0);
    if (CHAIN && chain_strip >= nstrips) return;
    // (M, U, L) of the strip's left boundary column, [row][16]; chain mode: one column per strip boundary, read / written
    const int64_t chain_col = CHAIN ? (int64_t)(max_l1 + 24) * 16 : 0;
    uint4 *my_bnd = bnd + tk.bnd_off + chain_col * chain_strip + p;
    uint4 *my_bnd_out = bnd + tk.bnd_off + chain_col * (chain_strip + 1) + p;
    const int *chain_in = (CHAIN && chain_strip > 0) ? chain_flags + (int64_t)task * chain_stride + chain_strip - 1 : nullptr;
    int *chain_out = CHAIN ? chain_flags + (int64_t)task * chain_stride + chain_strip : nullptr;
    int chain_seen = 0, chain_next = chain_every;
    uint4 *my_tb = reinterpret_cast<uint4 *>(reinterpret_cast<uint2 *>(tb) + tk.tb_off) + lane;   // [strip][step][64] (tb_off counts uint2, even)
    float *lastcol = aux + tk.aux_off;                                       // [y][3][32]
    float *lastrow = aux + tk.aux_off + (int64_t)(max_l1 + 1) * 3 * 32;      // [x - 1][3][32]

    constexpr int NRECT = MASK > 0 ? MASK : 1;
    static_assert(MASK >= 0 && MASK <= PRALINE_MAX_RECTS, "rectangle slots held in registers");
    int rectA[NRECT][4], rectB[NRECT][4];
    if constexpr (MASK != 0) {
        auto load_rects = [&](int pair, int (&rect)[NRECT][4]) {
            int n_rects = 0, r0 = 0;
            if (pair >= 0 && rl.rect_off != nullptr) {
                r0 = rl.rect_off[pair];
                n_rects = rl.rect_off[pair + 1] - r0;
                if (n_rects > NRECT) n_rects = NRECT;
            }
#pragma unroll
            for (int r = 0; r < NRECT; ++r) {
                const bool ok = r < n_rects;
                rect[r][0] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 0] : (1 << 30);
                rect[r][1] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 1] : -1;
                rect[r][2] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 2] : (1 << 30);
                rect[r][3] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 3] : -1;
            }
        };
        load_rects(pairA, rectA);
        load_rects(pairB, rectB);
    }

    // boundary cells (praline/component/align.py:367-385)
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);

    // local: first flat argmax over o (align.py:402); o[0,0,:] are the only boundary cells that can be >= 0.  Values packed,
    // positions per pair.
    int init_best = 0, init_k = 0;
    if (LOCAL) {
        if (o001 > 0.0f) { init_best = pk_scaled(o001, scale); init_k = 1; }
        if (o002 > (float)init_best * inv) { init_best = pk_scaled(o002, scale); init_k = 2; }
    }
    s16x2 out_best = pk_dup(init_best);
    int out_yA = 0, out_xA = 0, out_kA = init_k, out_yB = 0, out_xB = 0, out_kB = init_k;
    s16x2 corner_m = pk_dup(PRALINE_PK_NEG), corner_u = corner_m, corner_l = corner_m;

    for (int s = CHAIN ? chain_strip : 0; s < (CHAIN ? chain_strip + 1 : nstrips); ++s) {
        const int x0 = s * 32;
        const int xb = x0 + 8 * q;
        const bool last_owner = (s == nstrips - 1) && own_last;

        int smA[NRECT], smB[NRECT];   // column masks of the rectangles inside this lane's 8 columns
        if constexpr (MASK != 0) {
#pragma unroll
            for (int r = 0; r < NRECT; ++r) {
                const int loA = max(rectA[r][2] - (xb + 1), 0), hiA = min(rectA[r][3] - (xb + 1), 7);
                const int loB = max(rectB[r][2] - (xb + 1), 0), hiB = min(rectB[r][3] - (xb + 1), 7);
                smA[r] = (loA <= hiA) ? (int)((0xffu >> (7 - hiA)) & (0xffu << loA)) : 0;
                smB[r] = (loB <= hiB) ? (int)((0xffu >> (7 - hiB)) & (0xffu << loB)) : 0;
            }
        }
        // this strip's table: lane (c = lane & 31, half hh) transposes the hi pieces of half hh of the pre-multiplied row
        // x0 + c (exact mode: Q2 = hi exactly), k = 16 r + 8 hh + jj  ->  lookup_tab[k][c] as scaled int16
        {
            __builtin_amdgcn_wave_barrier();   // the previous strip's reads are done
            const int c = lane & 31, hh = lane >> 5;
            const char *src = ar.Q16 + ((int64_t)ar.row_off[two] + x0 + c) * ar.row_bytes + hh * ar.half_bytes;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const half8 hv = as_half8(reinterpret_cast<const float4 *>(src)[r]);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    *reinterpret_cast<short *>(lookup_tab + (16 * r + 8 * hh + jj) * pk16_stride() + c * 2) = (short)((float)hv[jj] * scale);
            }
            if (hh == 0) *reinterpret_cast<short *>(lookup_tab + (16 * NR) * pk16_stride() + c * 2) = 0;   // padding rows: symbol 16 NR
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
        s16x2 Mp[8], Up[8], Lp[8];
        s16x2 dM = pk_dup(PRALINE_PK_NEG), dU = dM, dL = dM;
        s16x2 best_run = out_best;
        int best_yA = out_yA, best_xA = out_xA, best_kA = out_kA, best_yB = out_yB, best_xB = out_xB, best_kB = out_kB;
        // memory operations per step as in k_dp_quad_tb: at the END of step t the loads of the boundary rows of step t + 1 and
        // of the symbols of step t + 2 (both pairs), THEN the step's stores - two boundary rows (quarter 3), one flag word.
        const s16x2 neg = pk_dup(PRALINE_PK_NEG);
        PkIn nxA = {neg, neg, neg}, nxB = nxA;
        const bool col0 = s == 0 || (PRALINE_PK16_ABLATE & 2) != 0;
        auto boundary_of = [&](int ya, const f4n &la, const f4n &lb, PkIn &ra, PkIn &rb) {
            if (col0) {   // (wave-uniform)
                ra = {neg, pk_dup(pk_scaled(boundary_value(ya, go, ge, free_one), scale)), neg};
                rb = {neg, pk_dup(pk_scaled(boundary_value(ya + 1, go, ge, free_one), scale)), neg};
            } else {
                ra = {pk_of(__float_as_uint(la.x)), pk_of(__float_as_uint(la.y)), pk_of(__float_as_uint(la.z))};
                rb = {pk_of(__float_as_uint(lb.x)), pk_of(__float_as_uint(lb.y)), pk_of(__float_as_uint(lb.z))};
            }
        };
        // (never past a sequence's own rows and padding: a short sequence in a task of long ones may be the arena's last)
        auto sym_addr_cap = [&](const unsigned char *ps, int cap, int ya) { return ps + (ya >= 1 ? (ya - 1 < cap ? ya - 1 : cap) : 0); };
        auto fetch_scores = [&](unsigned swA, unsigned swB, s16x2 (&a)[8], s16x2 (&b)[8]) {
            if (PRALINE_PK16_ABLATE & 4) { swA &= 0x0f0fu; swB &= 0x0f0fu; }
            const uint4 a0 = *reinterpret_cast<const uint4 *>(tab_lane + (swA & 0xffu) * pk16_stride());   // pair A, first row
            const uint4 a1 = *reinterpret_cast<const uint4 *>(tab_lane + (swA >> 8) * pk16_stride());      // pair A, second row
            const uint4 b0 = *reinterpret_cast<const uint4 *>(tab_lane + (swB & 0xffu) * pk16_stride());
            const uint4 b1 = *reinterpret_cast<const uint4 *>(tab_lane + (swB >> 8) * pk16_stride());
            const unsigned ra[4] = {a0.x, a0.y, a0.z, a0.w}, rb[4] = {b0.x, b0.y, b0.z, b0.w};
            const unsigned sa[4] = {a1.x, a1.y, a1.z, a1.w}, sb[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a[2 * g] = pk_of(__builtin_amdgcn_perm(rb[g], ra[g], 0x05040100u));
                a[2 * g + 1] = pk_of(__builtin_amdgcn_perm(rb[g], ra[g], 0x07060302u));
                b[2 * g] = pk_of(__builtin_amdgcn_perm(sb[g], sa[g], 0x05040100u));
                b[2 * g + 1] = pk_of(__builtin_amdgcn_perm(sb[g], sa[g], 0x07060302u));
            }
        };
        // Loads run TWO steps ahead of their use, in two register sets (odd / even steps): a set holds the boundary rows and the
        // symbols of its step (the match scores are read from the LDS table when the step starts: carrying them from the end
        // of the previous step cost 16 registers); at the END of step t the set is reloaded for step t + 2, THEN the step's
        // three stores are issued.  Step t + 1 starts with vmcnt(10): everything but the stores of step t - 1, the loads and
        // the stores of step t may still be on their way - the set of step t + 1 (issued at the end of step t - 1) has landed.
        // (With the loads one step ahead, as in k_dp_quad_tb, every step exposed a memory latency that two waves per SIMD do
        // not cover: 59 % of the VALU issue slots used on a C3 slice.)
        // prologue: set 1 = rows 1, 2 and the symbols of step 1; set 0 = rows 3, 4 and the symbols of step 2
        if constexpr (CHAIN) { if (chain_in != nullptr) chain_seen = chain_wait(chain_in, 4, chain_seen); }
        f4n ld1A = quad_load_f4(my_bnd + 16), ld1B = quad_load_f4(my_bnd + 32);
        unsigned sym1A = quad_load_u16(sym_addr_cap(psymA, L1A, 1 - 2 * q)), sym1B = quad_load_u16(sym_addr_cap(psymB, L1B, 1 - 2 * q));
        f4n ld0A = quad_load_f4(my_bnd + 48), ld0B = quad_load_f4(my_bnd + 64);
        unsigned sym0A = quad_load_u16(sym_addr_cap(psymA, L1A, 3 - 2 * q)), sym0B = quad_load_u16(sym_addr_cap(psymB, L1B, 3 - 2 * q));
        // (three stores to the unused row 0, so that step 2 finds the same ten operations behind its set as every later step:
        // ONE wait instruction for all steps - two of them in two branches made the compiler merge their register operands
        // with copies of registers whose loads were still in flight)
        // stores a step issues: two boundary rows (chain mode: two instructions each) and the flag words - the ablation builds
        // that leave some out count accordingly (a wait that assumes stores which are not there lets a step read registers
        // whose loads are still in flight: garbage end cells, a traceback out of bounds)
        constexpr int NST = (CHAIN ? 5 : 3) - ((PRALINE_PK16_ABLATE & 1) ? 1 : 0) - ((PRALINE_PK16_ABLATE & 2) ? (CHAIN ? 4 : 2) : 0);
        {
            const f4n z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < NST; ++k)
                asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(CHAIN ? my_bnd_out : my_bnd), "v"(z) : "memory");
        }
        PRALINE_QUAD_WAIT(4 + NST, ld1A, ld1B, sym1A);
        asm volatile("" : "+v"(sym1B));
        uint4 *tb_st = my_tb + (int64_t)s * nsteps * 64 + 64;     // step 1

        auto step = [&](const int t, f4n &ldA, f4n &ldB, unsigned &symA_ld, unsigned &symB_ld) __attribute__((always_inline)) {
            const int ya = 2 * (t - q) - 1;          // this quarter's rows ya, ya + 1 (<= 0: it has not started yet)
            // this step's set has landed (step 1: waited for in the prologue)
            PRALINE_QUAD_WAIT(4 + 2 * NST, ldA, ldB, symA_ld);   // (10; chain mode 14: five stores per step)
            asm volatile("" : "+v"(symB_ld));
            s16x2 mA[8], mB[8];   // packed match scores of the step's two rows
            fetch_scores(symA_ld, symB_ld, mA, mB);
            if (q == 0) boundary_of(2 * t - 1, ldA, ldB, nxA, nxB);
            if (t <= 4) {   // (wave-uniform: the later steps skip the 27 conditional moves; the empty asm keeps it a branch)
                asm volatile("");
            if (t == q + 1) {
                // the quarter starts: row 0 of its columns, and the states of the cell left of them (0, xb)
#pragma unroll
                for (int c = 0; c < 8; ++c) { Mp[c] = neg; Up[c] = neg; Lp[c] = pk_dup(pk_scaled(boundary_value(xb + c + 1, go, ge, free_two), scale)); }
                dM = (xb == 0) ? pk_dup(0) : neg;
                dU = (xb == 0) ? pk_dup(pk_scaled(o001, scale)) : neg;
                dL = (xb == 0) ? pk_dup(pk_scaled(o002, scale)) : pk_dup(pk_scaled(boundary_value(xb, go, ge, free_two), scale));
            }
            }
            unsigned zAa = 0, zAb = 0, zBa = 0, zBb = 0;   // row ya: pairs A, B; row ya + 1: pairs A, B
            if constexpr (MASK != 0) {
#pragma unroll
                for (int r = 0; r < NRECT; ++r) {
                    zAa |= (ya >= rectA[r][0] && ya <= rectA[r][1]) ? (unsigned)smA[r] : 0u;
                    zAb |= (ya >= rectB[r][0] && ya <= rectB[r][1]) ? (unsigned)smB[r] : 0u;
                    zBa |= (ya + 1 >= rectA[r][0] && ya + 1 <= rectA[r][1]) ? (unsigned)smA[r] : 0u;
                    zBb |= (ya + 1 >= rectB[r][0] && ya + 1 <= rectB[r][1]) ? (unsigned)smB[r] : 0u;
                }
            }
            // (wave-uniform) can a lane be at a last row in this step?  Quarter 0 is the furthest: rows 2 t - 1, 2 t
            const bool snap_step = 2 * t >= min_l1;
            const bool semi_last = semiglobal && s == nstrips - 1;
            // ---- per-row bookkeeping (the arrays hold row yy) ----
            // (once per pair, strip and quarter: a rolled loop - unrolled it held 48 converted values and their addresses live)
            auto store_last_row = [&](int hsel) {
                float *lr = lastrow + (int64_t)xb * 3 * 32 + 16 * hsel + p;
#pragma unroll 1
                for (int c = 0; c < 8; ++c) {
                    const s16x2 vm = pk_select8(Mp, c), vu = pk_select8(Up, c), vl = pk_select8(Lp, c);
                    lr[(c * 3 + 0) * 32] = (float)(hsel ? vm.y : vm.x) * inv;
                    lr[(c * 3 + 1) * 32] = (float)(hsel ? vu.y : vu.x) * inv;
                    lr[(c * 3 + 2) * 32] = (float)(hsel ? vl.y : vl.x) * inv;
                }
            };
            auto row_tails = [&](int yy) {
                if (LOCAL) {
                    // local end cell = first maximum of o in C order (y, x, k) (align.py:402); see split16_tb_step.  Both
                    // pairs' row maxima in one register; the positions are looked for only when a maximum may move.
                    s16x2 rowH;
                    if (neg_gaps) {
                        rowH = pk_maxs(pk_maxs(pk_maxs(Mp[0], Mp[1]), pk_maxs(Mp[2], Mp[3])), pk_maxs(pk_maxs(Mp[4], Mp[5]), pk_maxs(Mp[6], Mp[7])));
                    } else {
                        rowH = pk_maxs(pk_maxs(Mp[0], Up[0]), Lp[0]);
#pragma unroll
                        for (int c = 1; c < 8; ++c) rowH = pk_maxs(rowH, pk_maxs(pk_maxs(Mp[c], Up[c]), Lp[c]));
                    }
                    // a pair's maximum moves when its row maximum is larger, or equal in an earlier row than the one on record
                    // (rows ascend inside a strip: such ties only come up against a maximum of an earlier strip)
                    const unsigned dgt = pk_u(pk_subs(best_run, rowH)) & 0x80008000u;    // sign set: that pair's row maximum is larger
                    bool maybe = dgt != 0u;
                    if (yy < max(best_yA, best_yB)) {
                        const unsigned dx = pk_u(rowH) ^ pk_u(best_run);
                        maybe = maybe || (dx & 0xffffu) == 0u || (dx >> 16) == 0u;
                    }
                    maybe = maybe && yy >= 1;
                    if (__ballot(maybe) != 0ull) {
                        if (maybe) {
                            auto update = [&](int hsel, int &by, int &bx, int &bk) {
                                const int rh = hsel ? rowH.y : rowH.x, bh = hsel ? best_run.y : best_run.x;
                                if (rh > bh || (rh == bh && yy < by)) {
                                    if (hsel) best_run.y = (short)rh; else best_run.x = (short)rh;
                                    by = yy;
                                    if (neg_gaps) bk = 0;
#pragma unroll
                                    for (int c = 7; c >= 0; --c) {   // descending: the smallest column with the maximum wins
                                        const int vm = hsel ? Mp[c].y : Mp[c].x, vu = hsel ? Up[c].y : Up[c].x, vl = hsel ? Lp[c].y : Lp[c].x;
                                        if (neg_gaps) { if (vm == rh) bx = xb + c + 1; }
                                        else if (max(max(vm, vu), vl) == rh) { bx = xb + c + 1; bk = (vm == rh) ? 0 : ((vu == rh) ? 1 : 2); }
                                    }
                                }
                            };
                            update(0, best_yA, best_xA, best_kA);
                            update(1, best_yB, best_xB, best_kB);
                        }
                    }
                }
                if (semi_last) {
                    if (last_owner && yy >= 1) {
                        const s16x2 vm = pk_select8(Mp, cidx);
                        const s16x2 vu = pk_select8(Up, cidx);
                        const s16x2 vl = pk_select8(Lp, cidx);
                        float *lc = lastcol + (int64_t)yy * 3 * 32;            // o[y, L2, :]  (align.py:408,418-422)
                        if (haveA && yy <= L1A) { lc[p] = (float)vm.x * inv; lc[32 + p] = (float)vu.x * inv; lc[64 + p] = (float)vl.x * inv; }
                        if (haveB && yy <= L1B) { lc[16 + p] = (float)vm.y * inv; lc[48 + p] = (float)vu.y * inv; lc[80 + p] = (float)vl.y * inv; }
                    }
                }
                if (snap_step) {
                    if (haveA && yy == L1A) {
                        if (LOCAL) { out_best.x = best_run.x; out_yA = best_yA; out_xA = best_xA; out_kA = best_kA; }
                        if (last_owner) {
                            corner_m.x = pk_select8(Mp, cidx).x;
                            corner_u.x = pk_select8(Up, cidx).x;
                            corner_l.x = pk_select8(Lp, cidx).x;
                        }
                        if (semiglobal) store_last_row(0);                     // o[L1, x, :]  (align.py:407,413-417)
                    }
                    if (haveB && yy == L1B) {
                        if (LOCAL) { out_best.y = best_run.y; out_yB = best_yB; out_xB = best_xB; out_kB = best_kB; }
                        if (last_owner) {
                            corner_m.y = pk_select8(Mp, cidx).y;
                            corner_u.y = pk_select8(Up, cidx).y;
                            corner_l.y = pk_select8(Lp, cidx).y;
                        }
                        if (semiglobal) store_last_row(1);
                    }
                }
            };
            auto one_row = [&](const s16x2 (&m)[8], const PkIn &in, unsigned za, unsigned zb) {
                if constexpr (MASK != 0) {
                    if (__ballot((za | zb) != 0u) != 0ull) return pk16_row<LOCAL, true>(m, in, Mp, Up, Lp, dM, dU, dL, go2, ge2, za, zb);
                }
                return pk16_row<LOCAL, false>(m, in, Mp, Up, Lp, dM, dU, dL, go2, ge2, 0u, 0u);
            };
            const uint2 wA = one_row(mA, nxA, zAa, zAb);
            // this lane's last column, row ya: the next quarter's left neighbour one step on (in flight under row ya + 1)
            const s16x2 sAm = Mp[7], sAu = Up[7], sAl = Lp[7];
#if PRALINE_PK16_ABLATE & 8
            const s16x2 rAm = sAm, rAu = sAu, rAl = sAl;
#else
            const s16x2 rAm = pk_from_left(sAm, left_addr), rAu = pk_from_left(sAu, left_addr), rAl = pk_from_left(sAl, left_addr);
#endif
            if (!(PRALINE_PK16_ABLATE & 16))
            row_tails(ya);
            const uint2 wB = one_row(mB, nxB, zBa, zBb);
            const s16x2 sBm = Mp[7], sBu = Up[7], sBl = Lp[7];
#if PRALINE_PK16_ABLATE & 8
            const s16x2 rBm = sBm, rBu = sBu, rBl = sBl;
#else
            const s16x2 rBm = pk_from_left(sBm, left_addr), rBu = pk_from_left(sBu, left_addr), rBl = pk_from_left(sBl, left_addr);
#endif
            if (!(PRALINE_PK16_ABLATE & 16))
            row_tails(ya + 1);
            // ---- end of the step: what the next steps consume, then this step's stores (see the strip prologue) ----
            if (q != 0) { nxA = {rAm, rAu, rAl}; nxB = {rBm, rBu, rBl}; }
            if constexpr (CHAIN) { if (chain_in != nullptr) chain_seen = chain_wait(chain_in, 2 * t + 4, chain_seen); }
            ldA = quad_load_f4(my_bnd + (int64_t)(2 * t + 3) * 16);      // boundary rows of quarter 0's step t + 2
            ldB = quad_load_f4(my_bnd + (int64_t)(2 * t + 4) * 16);
            symA_ld = quad_load_u16(sym_addr_cap(psymA, L1A, (PRALINE_PK16_ABLATE & 4) ? 1 : ya + 4));    // symbols of step t + 2
            symB_ld = quad_load_u16(sym_addr_cap(psymB, L1B, (PRALINE_PK16_ABLATE & 4) ? 1 : ya + 4));
            // quarter 3: this strip's last column is the next strip's boundary column (rows <= 0: the unused row 0)
            if (q == 3) {
                const int ra = ya >= 1 ? ya : 0, rb = ya >= 1 ? ya + 1 : 0;
                if constexpr (CHAIN) {
                    // (two store instructions per row: the steps' hand-counted wait below counts five stores in chain mode)
                    chain_store_row(reinterpret_cast<char *>(my_bnd_out + (int64_t)ra * 16), __uint_as_float(pk_u(sAm)), __uint_as_float(pk_u(sAu)), __uint_as_float(pk_u(sAl)));
                    chain_store_row(reinterpret_cast<char *>(my_bnd_out + (int64_t)rb * 16), __uint_as_float(pk_u(sBm)), __uint_as_float(pk_u(sBu)), __uint_as_float(pk_u(sBl)));
                } else if (!(PRALINE_PK16_ABLATE & 2) || pk_u(sAm) == 0x12345678u) {
                my_bnd[(int64_t)ra * 16] = make_uint4(pk_u(sAm), pk_u(sAu), pk_u(sAl), 0u);
                my_bnd[(int64_t)rb * 16] = make_uint4(pk_u(sBm), pk_u(sBu), pk_u(sBl), 0u);
                }
            }
            {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 w = {wA.x, wA.y, wB.x, wB.y};
                if (!(PRALINE_PK16_ABLATE & 1) || wA.x == 0x12345u) {
#if PRALINE_PK16_STORE_NT
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(tb_st));
#else
                *reinterpret_cast<u32x4 *>(tb_st) = w;
#endif
                }
            }
            tb_st += 64;
            if constexpr (CHAIN) {
                // quarter 3 has stored the rows up to 2 (t - 3); every publish drains the wave's memory operations
                const int done = 2 * (t - 3);
                if (done >= chain_next) { chain_publish(chain_out, done, lane); chain_next = done + chain_every; }
            }
        };
        // (an odd run_steps runs one step more: rows past max_l1 that nobody reports; the planes and the boundary column
        // have room for it, PRALINE_QUAD_STEPS)
        for (int t = 1; t <= run_steps; t += 2) {
            step(t, ld1A, ld1B, sym1A, sym1B);
            step(t + 1, ld0A, ld0B, sym0A, sym0B);
        }
        if constexpr (CHAIN) chain_publish(chain_out, PRALINE_CHAIN_DONE, lane);
        // (the next strip's quarter 0 reads rows that quarter 3 stored a few steps ago - same wave, in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }

    if (CHAIN && !LOCAL && chain_strip != nstrips - 1) return;   // the last strip's wave reports the end cell
    // ---- combine the four quarters: end cell (y, x, k) and score (align.py:401-431) ----
    int bestA = out_best.x, bestB = out_best.y;
    if (LOCAL) {
        // first flat argmax: larger value wins; on ties the smaller (y, x)
#pragma unroll
        for (int mk = 16; mk <= 32; mk <<= 1) {
            {
                const int pv = __shfl_xor(bestA, mk), py = __shfl_xor(out_yA, mk), px = __shfl_xor(out_xA, mk), pk = __shfl_xor(out_kA, mk);
                if (pv > bestA || (pv == bestA && (py < out_yA || (py == out_yA && px < out_xA)))) { bestA = pv; out_yA = py; out_xA = px; out_kA = pk; }
            }
            {
                const int pv = __shfl_xor(bestB, mk), py = __shfl_xor(out_yB, mk), px = __shfl_xor(out_xB, mk), pk = __shfl_xor(out_kB, mk);
                if (pv > bestB || (pv == bestB && (py < out_yB || (py == out_yB && px < out_xB)))) { bestB = pv; out_yB = py; out_xB = px; out_kB = pk; }
            }
        }
    }
    s16x2 cm = corner_m, cu = corner_u, cl = corner_l;   // only the owner quarter holds finite values
#pragma unroll
    for (int mk = 16; mk <= 32; mk <<= 1) {
        cm = pk_maxs(cm, pk_of((unsigned)__shfl_xor((int)pk_u(cm), mk)));
        cu = pk_maxs(cu, pk_of((unsigned)__shfl_xor((int)pk_u(cu), mk)));
        cl = pk_maxs(cl, pk_of((unsigned)__shfl_xor((int)pk_u(cl), mk)));
    }
    if (CHAIN && LOCAL) {
        // every strip reports its own first-argmax candidate (value, y, x, k) per pair; k_chain_local_end picks
        if (q == 0) {
            float4 *cd = chain_cand + ((int64_t)task * chain_stride + chain_strip) * 32;
            cd[p] = make_float4((float)bestA * inv, __builtin_bit_cast(float, out_yA), __builtin_bit_cast(float, out_xA), __builtin_bit_cast(float, out_kA));
            cd[16 + p] = make_float4((float)bestB * inv, __builtin_bit_cast(float, out_yB), __builtin_bit_cast(float, out_xB), __builtin_bit_cast(float, out_kB));
        }
        return;
    }
    if (q == 0) {
        auto report = [&](bool have, int pair, int L1, int vm, int vu, int vl, int best, int by, int bx, int bk) {
            if (!have) return;
            int ey = L1, ex = L2, ek = 0;
            int sc = vm;
            if (LOCAL) { ey = by; ex = bx; ek = bk; sc = best; }
            else {
                if (vu > sc) { sc = vu; ek = 1; }  // np.argmax: first maximum
                if (vl > sc) { sc = vl; ek = 2; }
            }
            end_cells[(int64_t)pair * 4 + 0] = ey;
            end_cells[(int64_t)pair * 4 + 1] = ex;
            end_cells[(int64_t)pair * 4 + 2] = ek;
            end_cells[(int64_t)pair * 4 + 3] = 0;
            scores[pair] = (float)sc * inv;  // semiglobal: k_semiglobal_end overwrites it with the row / column rule
        };
        report(haveA, pairA, L1A, cm.x, cu.x, cl.x, bestA, out_yA, out_xA, out_kA);
        report(haveB, pairB, L1B, cm.y, cu.y, cl.y, bestB, out_yB, out_xB, out_kB);
    }
}
