// dp_quad.hip.h -- k_dp_quad_tb: fill WITH packed traceback for plain sequences (one-hot arenas, exact mode) in a
// layout built for OCCUPANCY: 16 pairs per wave, each pair's 32 strip columns on four lanes of 8 columns, two DP rows
// per step.
//
// Why.  The flagged fill of k_dp_split16_tb carries 3 x 16 states, two score sets and operand registers per lane: 242-256
// VGPRs even without MFMA operands (the lookup instances), two waves per SIMD at best - and its profile on C3 (r03: 48 % of
// the wave cycles issue, 33 % wait for memory) says the waves do not cover each other's loads.  For plain sequences the
// match score of a cell is a table lookup (m[y][x] = Q2[x][symbol of row y], dp_split16.hip.h), so nothing ties a wave to
// 32 pairs x 32 columns of an MFMA tile: here a lane holds 3 x 8 states and 2 x 8 scores, ~110 registers, and four waves
// share a SIMD.
//
// Layout.  Lane l: pair p = l & 15, quarter q = l >> 4 = strip columns 8 q + 1 .. 8 q + 8.  Step t: quarter q fills the
// rows 2 (t - q) - 1 and 2 (t - q) (quarter q runs q steps behind quarter q - 1: its left neighbours' states of both
// rows arrive from lane l - 16 with ds_bpermute one step after they were formed).  Quarter 0 takes them from the strip's
// boundary column (float4 (M, U, L) [row][16 pairs], written by quarter 3 of the previous strip; strip 0: analytic).
// A step therefore updates 16 cells per lane - the same ratio of cell work to per-step overhead as the 16-column layout.
//
// The arithmetic of a cell, the tie flags and their priorities are split16_tb_step's (dp_split16_tb.hip.h; cext.c:99-306,
// praline/util/align.py:161-174); scores, end cells and paths are bit-identical to k_dp_split16_tb's
// (tests/test_gpu_parity.py::test_quad_layout_paths_equal_the_strip_kernels).
//
// Traceback planes: uint2 [strip][step][64 lanes]; .x = the flags of the odd row 2 (t - q) - 1, .y of the even row
// 2 (t - q); per row: match source low bits (8) | high bits << 8 | "U from extend" << 16 | "L from extend" << 24
// (code = lo | hi << 1: 1 MM / 2 MU / 3 ML / 0 stop).  k_traceback reads them as layout 2.
#pragma once
#include "dp_split16_tb.hip.h"

#ifndef PRALINE_QUAD_ABLATE
#define PRALINE_QUAD_ABLATE 0   // timing experiments only (results invalid): 1 no flag stores, 2 no boundary column traffic, 4 no symbol loads,
                                // 8 no quarter-to-quarter hand-off, 16 no per-row bookkeeping
#endif
#ifndef PRALINE_QUAD_WAVES
#define PRALINE_QUAD_WAVES 3   // waves per SIMD the kernel is compiled for (measured on a C3 slice: 2, 3 and 4 waves per SIMD within 3 %)
#endif

// v[idx] for a per-lane idx in 0..7
__device__ __forceinline__ float select8(const float (&v)[8], int idx)
{
    const bool b0 = idx & 1, b1 = idx & 2, b2 = idx & 4;
    const float t0 = b0 ? v[1] : v[0], t1 = b0 ? v[3] : v[2], t2 = b0 ? v[5] : v[4], t3 = b0 ? v[7] : v[6];
    const float u0 = b1 ? t1 : t0, u1 = b1 ? t3 : t2;
    return b2 ? u1 : u0;
}

// the value lane l - 16 holds (lanes 0..15 receive lane 48..63's: unused)
__device__ __forceinline__ float from_left_quarter(float v, int addr)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// Loads the compiler does not see (it would wait for them - and, the counter being in order, for every store issued before
// them - at the first use it finds): the kernel waits for them itself, with a count that leaves the step's stores in flight.
__device__ __forceinline__ f4n quad_load_f4(const void *p)
{
    f4n v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned quad_load_u16(const void *p)
{
    unsigned v;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// every memory operation of this wave but the N youngest has completed; the operands tie the loaded registers to the wait
#define PRALINE_QUAD_WAIT(N, a, b, c) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory")

struct QuadRow {
    float inM, inU, inL;      // states of the left neighbour cell (y, first column - 1)
};

// One DP row of this lane's 8 columns.  On entry Mp / Up / Lp hold the previous row, (dM, dU, dL) the states of the cell
// left of it (y - 1, first column - 1); on exit they hold this row and (dM, dU, dL) = `in`.  Returns the row's flag word.
template <bool INTS, bool LOCAL, int MASK>
__device__ __forceinline__ unsigned quad_row(const float (&m)[8], const QuadRow &in, float (&Mp)[8], float (&Up)[8], float (&Lp)[8],
                                             float &dM, float &dU, float &dL, float go, float ge, unsigned zmask)
{
    float md = dM, ud = dU, ld = dL;
    float mleft = in.inM, lleft = in.inL;
    unsigned w_nm = 0, w_nu = 0, w_stop = 0, w_u = 0, w_l = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // exact candidate sums (cext.c:185-201), maxima and first-match flags (cext.c:207-295); INTS: every value is exact in
        // float32, max3(md, ud, ld) + m IS the maximum of the three sums and a state ties exactly when its sum does
        float sMM, sMU, M, Mref;
        if constexpr (INTS) {
            sMM = md; sMU = ud;
            Mref = max3f(md, ud, ld);
            M = Mref + m[c];
        } else {
            sMM = md + m[c]; sMU = ud + m[c];
            M = max3f(sMM, sMU, ld + m[c]);
            Mref = M;
        }
        const float uo = Mp[c] + go, ue = Up[c] + ge;
        float U = __builtin_fmaxf(uo, ue);
        const float lo = mleft + go, le = lleft + ge;
        float Lc = __builtin_fmaxf(lo, le);
        w_nm = shift_in_sign(w_nm, sMM, Mref);
        w_nu = shift_in_sign(w_nu, sMU, Mref);
        if constexpr (LOCAL) {
            w_stop = __builtin_amdgcn_alignbit(w_stop, __builtin_bit_cast(unsigned, M), 31);
            M = __builtin_fmaxf(M, 0.0f);                                    // cext.c:208-209
        }
        if constexpr (MASK != 0) {
            if (zmask & (1u << c)) { M = 0.0f; U = 0.0f; Lc = 0.0f; }          // cext.c:141-149 (stop code: row end)
        }
        w_u = shift_in_sign(w_u, uo, ue);
        w_l = shift_in_sign(w_l, lo, le);
        md = Mp[c]; ud = Up[c]; ld = Lp[c];
        Mp[c] = M; Up[c] = U; Lp[c] = Lc;
        mleft = M; lleft = Lc;
        // keep the flag shifts with their cells: left alone they are sunk into the block that stores the flag word - behind
        // the branches of the row's bookkeeping - and the 64 differences of a step stay in registers until then (+ 60 VGPRs)
        if ((c & 1) == 1) {
            asm volatile("" : "+v"(w_nm), "+v"(w_nu), "+v"(w_u), "+v"(w_l));
            if constexpr (LOCAL) asm volatile("" : "+v"(w_stop));
        }
    }
    dM = in.inM; dU = in.inU; dL = in.inL;
    // column c sits in bit 7 - c of the shifted-in words
    const unsigned r_nm = __builtin_bitreverse32(w_nm) >> 24, r_nu = __builtin_bitreverse32(w_nu) >> 24;
    unsigned go_on = 0xffu;
    if constexpr (LOCAL) go_on &= ~(__builtin_bitreverse32(w_stop) >> 24);
    if constexpr (MASK != 0) go_on &= ~zmask;
    const unsigned hi = r_nm & go_on, lo_bits = (~r_nm | r_nu) & go_on;
    return lo_bits | (hi << 8) | ((__builtin_bitreverse32(w_u) >> 24) << 16) | ((__builtin_bitreverse32(w_l) >> 24) << 24);
}

// NR: 16-wide symbol ranges of the lookup table (1: <= 16 active symbols, 2: <= 32).  INTS: tie flags from the predecessor
// states (integer scoring, checked by praline_plan_run) instead of the candidate sums.  MASK: 0 none, 1 rectangles in
// registers (<= PRALINE_MAX_RECTS per pair), 2 per-row column-mask words (k_build_zmask: any number of rectangles).
template <int NR, bool INTS, bool LOCAL, int MASK>
__global__ __launch_bounds__(256, PRALINE_QUAD_WAVES) void k_dp_quad_tb(Arena16Dev ar, const WaveTask *__restrict__ tasks,
                                                                        const int32_t *__restrict__ lane_one, const int32_t *__restrict__ lane_pair,
                                                                        float4 *bnd, uint2 *__restrict__ tb, float *__restrict__ aux, RectList rl,
                                                                        float *__restrict__ scores, int32_t *__restrict__ end_cells, RunParams rp,
                                                                        int n_tasks)
{
    __shared__ __attribute__((aligned(16))) char lookup_all[4 * lookup_bytes(NR)];   // one table per wave of the block
    const int wv = (int)(threadIdx.x >> 6);
    const int task = (int)blockIdx.x * 4 + wv;
    if (task >= n_tasks) return;          // (the waves of a block are independent: no block-level barrier below)
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, q = lane >> 4;
    const WaveTask tk = tasks[task];
    const int base = task * 16;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    const float go = rp.go1, ge = rp.ge1;

    const int my_one = lane_one[base + p];
    const int two = tk.two[0];
    const bool have_pair = my_one >= 0;
    const int my_pair = have_pair ? lane_pair[base + p] : -1;
    const int L1 = have_pair ? ar.len[my_one] : 0;
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

    // shortest sequence one of the task (wave-uniform): no lane can be at its last row before it
    int min_l1 = have_pair ? L1 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) min_l1 = min(min_l1, __shfl_xor(min_l1, off));
    min_l1 = __builtin_amdgcn_readfirstlane(min_l1);

    char *lookup_tab = lookup_all + wv * lookup_bytes(NR);
    const char *tab_lane = lookup_tab + q * 32;              // this lane's 8 columns of a table row
    const unsigned char *psym = ar.sym8 + (have_pair ? ar.row_off[my_one] : 0);
    float4 *my_bnd = bnd + tk.bnd_off + p;                   // (M, U, L) of the strip's left boundary column, [row][16]
    uint2 *my_tb = tb + tk.tb_off + lane;                    // [strip][step][64]
    float *lastcol = aux + tk.aux_off + p;                                   // [y][3][16]
    float *lastrow = aux + tk.aux_off + (int64_t)(max_l1 + 1) * 3 * 16 + p;  // [x - 1][3][16]

    int rect[PRALINE_MAX_RECTS][4];
    const unsigned *my_zm = nullptr;
    if constexpr (MASK == 1) {
        int n_rects = 0, r0 = 0;
        if (my_pair >= 0 && rl.rect_off != nullptr) {
            r0 = rl.rect_off[my_pair];
            n_rects = rl.rect_off[my_pair + 1] - r0;
            if (n_rects > PRALINE_MAX_RECTS) n_rects = PRALINE_MAX_RECTS;
        }
#pragma unroll
        for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
            const bool ok = r < n_rects;
            rect[r][0] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 0] : (1 << 30);
            rect[r][1] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 1] : -1;
            rect[r][2] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 2] : (1 << 30);
            rect[r][3] = ok ? rl.rects[(int64_t)(r0 + r) * 4 + 3] : -1;
        }
    }
    if constexpr (MASK == 2) {
        // uint32 [strip][L1 + 1] per pair: bit c = cell (y, 32 strip + c + 1) lies in one of the pair's rectangles
        if (my_pair >= 0 && rl.zmask != nullptr) my_zm = rl.zmask + rl.zm_off[my_pair];
    }

    // boundary cells (praline/component/align.py:367-385)
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);

    // local: first flat argmax over o (align.py:402); o[0,0,:] are the only boundary cells that can be >= 0
    float out_best = 0.0f;
    int out_y = 0, out_x = 0, out_k = 0;
    if (LOCAL) {
        if (o001 > out_best) { out_best = o001; out_k = 1; }
        if (o002 > out_best) { out_best = o002; out_k = 2; }
    }
    float corner_m = PRALINE_NEG_INF, corner_u = PRALINE_NEG_INF, corner_l = PRALINE_NEG_INF;

    for (int s = 0; s < nstrips; ++s) {
        const int x0 = s * 32;
        const int xb = x0 + 8 * q;
        const bool last_owner = (s == nstrips - 1) && own_last;

        int srect[PRALINE_MAX_RECTS][2];   // rows of the rectangles are in rect[r][0..1]; here: their column mask inside this lane's 8 columns
        if constexpr (MASK == 1) {
#pragma unroll
            for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
                const int lo = max(rect[r][2] - (xb + 1), 0), hi = min(rect[r][3] - (xb + 1), 7);
                srect[r][0] = (lo <= hi) ? (int)((0xffu >> (7 - hi)) & (0xffu << lo)) : 0;
                srect[r][1] = 0;
            }
        }
        // this strip's table: lane (c = lane & 31, half hh) transposes the hi pieces of half hh of the pre-multiplied row
        // x0 + c (exact mode: Q2 = hi exactly), k = 16 r + 8 hh + jj  ->  lookup_tab[k][c]   (as in k_dp_split16_tb)
        {
            __builtin_amdgcn_wave_barrier();   // the previous strip's reads are done
            const int c = lane & 31, hh = lane >> 5;
            const char *src = ar.Q16 + ((int64_t)ar.row_off[two] + x0 + c) * ar.row_bytes + hh * ar.half_bytes;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const half8 hv = as_half8(reinterpret_cast<const float4 *>(src)[r]);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    *reinterpret_cast<float *>(lookup_tab + (16 * r + 8 * hh + jj) * lookup_stride() + c * 4) = (float)hv[jj];
            }
            if (hh == 0) *reinterpret_cast<float *>(lookup_tab + (16 * NR) * lookup_stride() + c * 4) = 0.0f;   // padding rows: symbol 16 NR
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
        }
        float Mp[8], Up[8], Lp[8];
        float dM = PRALINE_NEG_INF, dU = PRALINE_NEG_INF, dL = PRALINE_NEG_INF;
        float best_run = out_best;
        int best_y = out_y, best_x = out_x, best_k = out_k;
        // What this quarter receives from its left neighbour for the rows of the next step: quarters 1 .. 3 from lane l - 16
        // (ds_bpermute), quarter 0 from the strip's boundary column - the analytic column 0, or what quarter 3 of the previous
        // strip stored.  Memory operations per step, in this order and unconditionally (so that the count is the same for
        // every step and lane group): at the END of step t the loads of the boundary rows of step t + 1 (every lane loads its
        // pair's rows: the four quarters share the lines) and of the symbols of step t + 2, THEN the step's stores - two
        // boundary rows (quarter 3), one flag word.  Step t + 1 starts with vmcnt(3): its loads have landed, the three stores
        // issued behind them may still be on their way (a store has a whole step before a wave waits for it; with
        // compiler-placed waits every step waited for its own stores).  Measured on a C3 slice: + 3 %; loads three steps
        // ahead (vmcnt(15), three register sets) gave nothing more - the stores are not what the waves wait for.
        QuadRow nxA = {PRALINE_NEG_INF, PRALINE_NEG_INF, PRALINE_NEG_INF}, nxB = nxA;
        const bool col0 = s == 0 || (PRALINE_QUAD_ABLATE & 2) != 0;
        auto boundary_of = [&](int ya, const f4n &la, const f4n &lb, QuadRow &ra, QuadRow &rb) {
            if (col0) {   // (wave-uniform)
                ra = {PRALINE_NEG_INF, boundary_value(ya, go, ge, free_one), PRALINE_NEG_INF};
                rb = {PRALINE_NEG_INF, boundary_value(ya + 1, go, ge, free_one), PRALINE_NEG_INF};
            } else {
                ra = {la.x, la.y, la.z};
                rb = {lb.x, lb.y, lb.z};
            }
        };
        // Scores: the table rows of the rows' symbols (bytes ya - 1 and ya of the sequence: one aligned 16-bit load), read
        // from LDS one step before they are used.
        // (never past the sequence's own rows and padding: a short sequence in a task of long ones may be the arena's last)
        auto sym_addr = [&](int ya) { return psym + (ya >= 1 ? (ya - 1 < L1 ? ya - 1 : L1) : 0); };
        float mA[8], mB[8];
        auto fetch_scores = [&](unsigned sw, float (&a)[8], float (&b)[8]) {
            if (PRALINE_QUAD_ABLATE & 4) sw &= 0x0f0fu;
            const float4 *qa = reinterpret_cast<const float4 *>(tab_lane + (sw & 0xffu) * lookup_stride());
            const float4 *qb = reinterpret_cast<const float4 *>(tab_lane + (sw >> 8) * lookup_stride());
            const float4 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        };
        // prologue: rows 1, 2 of the boundary column, the symbols of steps 1 and 2 (waited for outright, once per strip)
        f4n ldA = quad_load_f4(my_bnd + 16), ldB = quad_load_f4(my_bnd + 32);
        unsigned sym1 = quad_load_u16(sym_addr(1 - 2 * q));
        unsigned sym_ld = quad_load_u16(sym_addr(3 - 2 * q));     // step 2's symbols
        PRALINE_QUAD_WAIT(0, ldA, ldB, sym1);
        asm volatile("" : "+v"(sym_ld));
        fetch_scores(sym1, mA, mB);                               // step 1's rows
        uint2 *tb_st = my_tb + (int64_t)s * nsteps * 64 + 64;   // step 1

        for (int t = 1; t <= run_steps; ++t) {
            const int ya = 2 * (t - q) - 1;          // this quarter's rows ya, ya + 1 (<= 0: it has not started yet)
            // the loads issued at the end of the previous step have landed (first step: the prologue's)
            if (t > 1) PRALINE_QUAD_WAIT(3, ldA, ldB, sym_ld);
            if (q == 0) boundary_of(2 * t - 1, ldA, ldB, nxA, nxB);
            const unsigned sym_now = sym_ld;         // the symbols of step t + 1
            if (t <= 4) {   // (wave-uniform: the later steps skip the 27 conditional moves; the empty asm keeps it a branch)
                asm volatile("");
            if (t == q + 1) {
                // the quarter starts: row 0 of its columns, and the states of the cell left of them (0, xb)
#pragma unroll
                for (int c = 0; c < 8; ++c) { Mp[c] = PRALINE_NEG_INF; Up[c] = PRALINE_NEG_INF; Lp[c] = boundary_value(xb + c + 1, go, ge, free_two); }
                dM = (xb == 0) ? 0.0f : PRALINE_NEG_INF;
                dU = (xb == 0) ? o001 : PRALINE_NEG_INF;
                dL = (xb == 0) ? o002 : boundary_value(xb, go, ge, free_two);
            }
            }
            unsigned zA = 0, zB = 0;
            if constexpr (MASK == 1) {
#pragma unroll
                for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
                    zA |= (ya >= rect[r][0] && ya <= rect[r][1]) ? (unsigned)srect[r][0] : 0u;
                    zB |= (ya + 1 >= rect[r][0] && ya + 1 <= rect[r][1]) ? (unsigned)srect[r][0] : 0u;
                }
            }
            if constexpr (MASK == 2) {
                if (my_zm != nullptr) {
                    if (ya >= 1 && ya <= L1) zA = (my_zm[(int64_t)s * (L1 + 1) + ya] >> (8 * q)) & 0xffu;
                    if (ya + 1 >= 1 && ya + 1 <= L1) zB = (my_zm[(int64_t)s * (L1 + 1) + ya + 1] >> (8 * q)) & 0xffu;
                }
            }
            // (wave-uniform) can a lane be at its last row in this step?  Quarter 0 is the furthest: rows 2 t - 1, 2 t
            const bool snap_step = 2 * t >= min_l1;
            const bool semi_last = semiglobal && s == nstrips - 1;
            // ---- per-row bookkeeping (the arrays hold row yy) ----
            auto row_tails = [&](int yy) {
                if (LOCAL) {
                    // local end cell = first maximum of o in C order (y, x, k) (align.py:402); see split16_tb_step
                    if (go < 0.0f && ge < 0.0f) {
                        const float rowH = max3f(max3f(Mp[0], Mp[1], Mp[2]), max3f(Mp[3], Mp[4], Mp[5]), __builtin_fmaxf(Mp[6], Mp[7]));
                        const bool better = yy >= 1 && (rowH > best_run || (rowH == best_run && yy < best_y));
                        if (__ballot(better) != 0ull) {
                            if (better) {
                                best_run = rowH; best_y = yy; best_k = 0;
#pragma unroll
                                for (int c = 7; c >= 0; --c)
                                    if (Mp[c] == rowH) best_x = xb + c + 1;
                            }
                        }
                    } else {
                        float rowH = max3f(Mp[0], Up[0], Lp[0]);
#pragma unroll
                        for (int c = 1; c < 8; ++c) rowH = __builtin_fmaxf(rowH, max3f(Mp[c], Up[c], Lp[c]));
                        const bool better = yy >= 1 && (rowH > best_run || (rowH == best_run && yy < best_y));
                        if (__ballot(better) != 0ull) {
                            if (better) {
                                best_run = rowH; best_y = yy;
#pragma unroll
                                for (int c = 7; c >= 0; --c)
                                    if (max3f(Mp[c], Up[c], Lp[c]) == rowH) {
                                        best_x = xb + c + 1;
                                        best_k = (Mp[c] == rowH) ? 0 : ((Up[c] == rowH) ? 1 : 2);
                                    }
                            }
                        }
                    }
                }
                if (semi_last) {
                    if (last_owner && have_pair && yy >= 1 && yy <= L1) {
                        float *lc = lastcol + (int64_t)yy * 3 * 16;            // o[y, L2, :]  (align.py:408,418-422)
                        lc[0] = select8(Mp, cidx); lc[16] = select8(Up, cidx); lc[32] = select8(Lp, cidx);
                    }
                }
                if (snap_step) {
                    if (have_pair && yy == L1) {
                        if (LOCAL) { out_best = best_run; out_y = best_y; out_x = best_x; out_k = best_k; }
                        if (last_owner) { corner_m = select8(Mp, cidx); corner_u = select8(Up, cidx); corner_l = select8(Lp, cidx); }
                        if (semiglobal) {                                       // o[L1, x, :]  (align.py:407,413-417)
                            float *lr = lastrow + (int64_t)xb * 3 * 16;
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                lr[(c * 3 + 0) * 16] = Mp[c]; lr[(c * 3 + 1) * 16] = Up[c]; lr[(c * 3 + 2) * 16] = Lp[c];
                            }
                        }
                    }
                }
            };
            const unsigned wA = quad_row<INTS, LOCAL, MASK>(mA, nxA, Mp, Up, Lp, dM, dU, dL, go, ge, zA);
            // this lane's last column, row ya: the next quarter's left neighbour one step on (in flight under row ya + 1)
            const float sAm = Mp[7], sAu = Up[7], sAl = Lp[7];
#if PRALINE_QUAD_ABLATE & 8
            const float rAm = sAm, rAu = sAu, rAl = sAl;
#else
            const float rAm = from_left_quarter(sAm, left_addr), rAu = from_left_quarter(sAu, left_addr), rAl = from_left_quarter(sAl, left_addr);
#endif
            if (!(PRALINE_QUAD_ABLATE & 16))
            row_tails(ya);
            const unsigned wB = quad_row<INTS, LOCAL, MASK>(mB, nxB, Mp, Up, Lp, dM, dU, dL, go, ge, zB);
            const float sBm = Mp[7], sBu = Up[7], sBl = Lp[7];
#if PRALINE_QUAD_ABLATE & 8
            const float rBm = sBm, rBu = sBu, rBl = sBl;
#else
            const float rBm = from_left_quarter(sBm, left_addr), rBu = from_left_quarter(sBu, left_addr), rBl = from_left_quarter(sBl, left_addr);
#endif
            if (!(PRALINE_QUAD_ABLATE & 16))
            row_tails(ya + 1);
            // ---- end of the step: what the next steps consume, then this step's stores (see the strip prologue) ----
            if (q != 0) { nxA = {rAm, rAu, rAl}; nxB = {rBm, rBu, rBl}; }
            fetch_scores(sym_now, mA, mB);                       // step t + 1's rows
            ldA = quad_load_f4(my_bnd + (int64_t)(2 * t + 1) * 16);      // boundary rows of quarter 0's step t + 1
            ldB = quad_load_f4(my_bnd + (int64_t)(2 * t + 2) * 16);
            sym_ld = quad_load_u16(sym_addr(ya + 4));            // symbols of step t + 2
            // quarter 3: this strip's last column is the next strip's boundary column (rows <= 0: the unused row 0)
            if (q == 3) {
                const int ra = ya >= 1 ? ya : 0, rb = ya >= 1 ? ya + 1 : 0;
                if (!(PRALINE_QUAD_ABLATE & 2) || sAm == 12345.0f) {
                    my_bnd[(int64_t)ra * 16] = make_float4(sAm, sAu, sAl, 0.0f);
                    my_bnd[(int64_t)rb * 16] = make_float4(sBm, sBu, sBl, 0.0f);
                }
            }
            if (!(PRALINE_QUAD_ABLATE & 1) || wA == 0x12345u)
            __builtin_nontemporal_store((unsigned long long)wA | ((unsigned long long)wB << 32), reinterpret_cast<unsigned long long *>(tb_st));
            tb_st += 64;
        }
        // (the next strip's quarter 0 reads rows that quarter 3 stored a few steps ago - same wave, in program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }

    // ---- combine the four quarters: end cell (y, x, k) and score (align.py:401-431) ----
    auto from_lane = [&](float v, int xor_mask) { return __shfl_xor(v, xor_mask); };
    if (LOCAL) {
        // first flat argmax: larger value wins; on ties the smaller (y, x)
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            const float pv = from_lane(out_best, m);
            const int py = __shfl_xor(out_y, m), px = __shfl_xor(out_x, m), pk = __shfl_xor(out_k, m);
            if (pv > out_best || (pv == out_best && (py < out_y || (py == out_y && px < out_x)))) {
                out_best = pv; out_y = py; out_x = px; out_k = pk;
            }
        }
    }
    float cm = corner_m, cu = corner_u, cl = corner_l;   // only the owner quarter holds finite values
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        cm = __builtin_fmaxf(cm, from_lane(cm, m));
        cu = __builtin_fmaxf(cu, from_lane(cu, m));
        cl = __builtin_fmaxf(cl, from_lane(cl, m));
    }
    if (have_pair && q == 0) {
        int ey = L1, ex = L2, ek = 0;
        float score = cm;
        if (LOCAL) { ey = out_y; ex = out_x; ek = out_k; score = out_best; }
        else {
            if (cu > score) { score = cu; ek = 1; }  // np.argmax: first maximum
            if (cl > score) { score = cl; ek = 2; }
        }
        end_cells[(int64_t)my_pair * 4 + 0] = ey;
        end_cells[(int64_t)my_pair * 4 + 1] = ex;
        end_cells[(int64_t)my_pair * 4 + 2] = ek;
        end_cells[(int64_t)my_pair * 4 + 3] = 0;
        scores[my_pair] = score;  // semiglobal: k_semiglobal_end overwrites it with the row / column rule
    }
}
