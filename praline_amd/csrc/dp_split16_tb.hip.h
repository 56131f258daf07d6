// dp_split16_tb.hip.h -- k_dp_split16_tb: fill WITH packed traceback in the split-strip layout, match
// scores on the matrix pipe (see dp_split.hip.h / dp_split16.hip.h for the layout and the pipeline).
//
// The three states are carried separately so ties resolve exactly as get_paths does
// (praline/util/align.py:161-174: first set flag in the order MM, MU, ML / UO, UE / LO, LE): the
// candidate sums are formed individually and compared with == as in praline/util/cext.c:224-295.
// Per lane and DP row two words are stored (match source; 16 bits "U from extend" | 16 bits "L from extend"
// << 16) = 8 bytes per 16 cells:
//     tb2[(strip * tb_rows + y) * 64 + lane] = { source word,  ubit_c << c | lbit_c << (16 + c) }
// source word: lo_c << c | hi_c << (16 + c), code = lo | hi << 1: 1 MM / 2 MU / 3 ML / 0 stop (local: the clamp won)
// lane j holds strip columns 1..16, lane j + 32 columns 17..32 of pair j.
// Zero rectangles (Waterman-Eggert, praline/component/preprofile.py:247-255) force M = U = L = 0 and
// stop codes (cext.c:141-149).  End cells: the global corner triple, the local first argmax and the
// semiglobal last row / last column triples (praline/component/align.py:401-431) are written to
// end_cells / aux in the layout k_traceback reads.
#pragma once
#include "dp_split16.hip.h"
#include <type_traits>

// w = (w << 1) | (open < extend): the sign bit of open - extend, shifted in with v_alignbit_b32.  The
// subtraction is opaque to the optimiser (the kernels are built -fno-honor-nans and -inf - -inf is a NaN).
__device__ __forceinline__ unsigned shift_in_sign(unsigned w, float open, float extend)
{
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(open), "v"(extend));
    return __builtin_amdgcn_alignbit(w, __builtin_bit_cast(unsigned, d), 31);
}

// ---- chain mode: one wave per (task, strip), strips of a task pipelined ACROSS workgroups --------------------
// A single alignment (every merge step of the progressive MSA) is one task: run strip after strip by one wave
// it takes nstrips x rows steps with the rest of the chip idle (400 x 400 with paths: 4.1 ms; 3000 x 3000:
// 215 ms).  In chain mode every strip has its own wave and its own boundary column in HBM; the wave of strip
// s + 1 follows the wave of strip s a few rows behind.  The hand-off follows the write-through recipe of the CDNA
// guide: the producer stores the boundary rows with agent-scope relaxed atomics (sc1, write-through, visible to
// every XCD), drains them (vmcnt(0)) and publishes the number of finished rows in a flag word with the same kind
// of store; the consumer polls that word (relaxed, agent scope), then ONE agent-scope acquire drops its stale
// cache lines and plain loads follow.  Blocks are dispatched in index order and the grid is strip-major, so the
// producer of any resident wave has been dispatched before it: the polling always makes progress.
typedef __attribute__((address_space(1))) unsigned long long chain_u64;
typedef __attribute__((address_space(1))) unsigned chain_u32;
#define PRALINE_CHAIN_DONE 0x3fffffff

__device__ __forceinline__ void chain_store_row(char *dst, float m, float u, float l)
{
    const unsigned long long mu = (unsigned long long)__builtin_bit_cast(unsigned, m) |
                                  ((unsigned long long)__builtin_bit_cast(unsigned, u) << 32);
    __hip_atomic_store((chain_u64 *)(dst), mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((chain_u32 *)(dst + 8), __builtin_bit_cast(unsigned, l), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void chain_publish(int *flag, int rows, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows are in memory before the flag says so
    if (lane == 0)
        __hip_atomic_store((chain_u32 *)(flag), (unsigned)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait until the producer has published at least `rows` rows; returns what it has published
__device__ __forceinline__ int chain_wait(const int *flag, int rows, int seen)
{
    if (rows <= seen) return seen;
    int v;
    for (;;) {
        v = (int)__hip_atomic_load((const chain_u32 *)(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= rows) break;
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return v;
}

struct TbCarry {
    float xm, xu, xl;     // upper half: states of the cell left of its first column, same row (handed over)
    float pxm, pxu, pxl;  // upper half: the same for the previous row (= its diagonal input)
    float dM, dU, dL;     // lower half: states of the boundary cell (y-1, x0)
};

// DM (single-term instances): one score tile serves both halves of the wave - the A operand is split by output row
// (aop: rows the lower half receives, aopH: the upper half's) and the accumulator takes mfma(aop, row t+1) +
// mfma(aopH, row t); see split16_step in dp_split16.hip.h.  BOLD holds row t and is refilled with row t+3 once its
// MFMAs are issued (three operand sets rotate); without DM, BOLD is BOPS itself.
// SINK: where the per-row flag words go.  0: the traceback planes in global memory (single pass);
//   1: nowhere - the flag-free FORWARD fill of the two-pass scheme (see k_trace_recompute below): no flag is formed,
//      instead the three states of every 32nd row are written to ckpt (float4 [block][3][4][64] per strip);
//   2: an LDS row of the recompute kernel, lds_flags + (row index) * 512 + lane * 8 (no end-cell bookkeeping);
//   3: nowhere, and no checkpoints either - SCORES ONLY (chain mode for score plans of a few long sequences).
// BSRC = 1 (one-hot arenas, single-term instances): the operand row of the refill is not loaded from the arena - 64
// lanes reading 64 different rows per step cost the CU's L1 one tag cycle per lane, which the flag-free forward fill
// (150 VALU per step instead of 350) no longer hides - but looked up in the one-hot operand table in LDS
// (dp_split16.hip.h) by the row's symbol, byte SB of symw.
// PPG (dense tiles only): per-position gap scores (GapScoreModel, praline/container/score.py:45-68) - U[y][x] takes (go_row,
// ge_row) = (open, extend) of position y - 1 of the lane's sequence one, L[y][x] those of position x - 1 of the shared
// sequence two, g2o[c] / g2e[c] for this lane's 16 columns (cext.c:155-158,172-175).  zm_row (dense tiles, MASK): the per-row
// column-mask words of plans with more than PRALINE_MAX_RECTS rectangles per pair (k_build_zmask) instead of the rectangles.
template <int NR, int NTERM, bool LOCAL, bool MASK, bool CHAIN = false, bool DM = false, int SINK = 0, int BSRC = 0, int SB = 0, bool PPG = false>
__device__ __forceinline__ void split16_tb_step(int yy, int L1, bool have_pair, int h, const f32x16 &CUR, f32x16 &PREV,
                                                float4 (&BOPS)[(NTERM == 1 ? 1 : 2) * NR],
                                                float4 (&BOLD)[(NTERM == 1 ? 1 : 2) * NR],
                                                const float4 (&aop)[(NTERM == 1 ? 1 : 2) * NR],
                                                const float4 (&aopH)[(NTERM == 1 ? 1 : 2) * NR], const char *&b_next,
                                                int b_stride, const char *&bnd_ld, char *&bnd_st, float4 &bnd_pref,
                                                uint2 *&tb_st, float (&Mp)[16], float (&Up)[16], float (&Lp)[16],
                                                float &cxm, float &cxu, float &cxl, float &cpxm, float &cpxu, float &cpxl,
                                                float &cdM, float &cdU, float &cdL, float &best_run, int &best_y,
                                                int &best_x, int &best_k, float go, float ge, int xb,
                                                const int (&rect)[PRALINE_MAX_RECTS][4], const int *chain_in = nullptr,
                                                int *chain_seen = nullptr, int load_row = 0, float *ckpt = nullptr,
                                                char *lds_flags = nullptr, int lds_row = 0, const char *onehot_lane = nullptr,
                                                unsigned symw = 0, const unsigned *zm_row = nullptr, int zm_rows = 0, float go_row = 0.0f,
                                                float ge_row = 0.0f, const float *g2o = nullptr, const float *g2e = nullptr)
{
    static_assert(!PPG || BSRC == 4, "per-position gap scores run on the dense-tile instances");
    constexpr int NP = (NTERM == 1) ? 1 : 2;
    // BSRC = 3 (one-hot arenas, integer scoring): the match scores are looked up in the strip's LDS table
    // (dp_split16.hip.h, lookup_stride): CUR holds this lane's 16 scores of its row, PREV receives the next row's (symbol
    // symw) - no MFMA, no operand registers, no accumulator tiles.
    // BSRC = 4 (reference-order match scores): the same, the next row's scores read from the task's dense tile
    // (dp_reftile.hip.h) at onehot_lane; any profiles - the NTERM = 3 instances, which compare the candidate sums
    constexpr bool DENSE = BSRC == 4;
    constexpr bool LOOKUP = BSRC == 3 || DENSE;
    constexpr int NM = LOOKUP ? 1 : NTERM * NR;
    constexpr bool INTS = NTERM == 1 || SINK == 3;   // (scores only: the three sums are needed for the tie flags alone - fl is monotone)
    float m[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) m[c] = (DM || LOOKUP) ? CUR[c] : (h ? PREV[c] : CUR[c]);
    if constexpr (DENSE) {
        const f4n *q = reinterpret_cast<const f4n *>(onehot_lane);
        const f4n a0 = PRALINE_DENSE_LOAD(q), a1 = PRALINE_DENSE_LOAD(q + 1), a2 = PRALINE_DENSE_LOAD(q + 2), a3 = PRALINE_DENSE_LOAD(q + 3);
        PREV[0] = a0.x; PREV[1] = a0.y; PREV[2] = a0.z; PREV[3] = a0.w; PREV[4] = a1.x; PREV[5] = a1.y; PREV[6] = a1.z; PREV[7] = a1.w;
        PREV[8] = a2.x; PREV[9] = a2.y; PREV[10] = a2.z; PREV[11] = a2.w; PREV[12] = a3.x; PREV[13] = a3.y; PREV[14] = a3.z; PREV[15] = a3.w;
    } else if constexpr (LOOKUP) {
        const float4 *q = reinterpret_cast<const float4 *>(onehot_lane + symw * lookup_stride());
        const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
        PREV[0] = a0.x; PREV[1] = a0.y; PREV[2] = a0.z; PREV[3] = a0.w; PREV[4] = a1.x; PREV[5] = a1.y; PREV[6] = a1.z; PREV[7] = a1.w;
        PREV[8] = a2.x; PREV[9] = a2.y; PREV[10] = a2.z; PREV[11] = a2.w; PREV[12] = a3.x; PREV[13] = a3.y; PREV[14] = a3.z; PREV[15] = a3.w;
    }

    const float4 bv = bnd_pref;  // states (M, U, L) of the boundary cell (yy, x0)
    if constexpr (CHAIN) {
        if (chain_in != nullptr) *chain_seen = chain_wait(chain_in, load_row, *chain_seen);   // row load_row is published
    }
    bnd_pref = *reinterpret_cast<const float4 *>(bnd_ld);
    bnd_ld += 32 * sizeof(float4);
    float md = h ? cpxm : cdM, ud = h ? cpxu : cdU, ld = h ? cpxl : cdL;  // states of (yy-1, x-1)
    float mleft = h ? cxm : bv.x, lleft = h ? cxl : bv.z;                 // states of (yy, x-1)
    // zero mask of this lane's 16 cells in this row: one bit per column, built once per row
    unsigned zmask = 0;
    if constexpr (MASK) {
        if (BSRC == 4 && zm_row != nullptr) {   // (wave-uniform) mask words: bit c of the strip's word of row yy, this half's 16 bits
            if (yy >= 1 && yy <= zm_rows) zmask = (zm_row[yy] >> (16 * h)) & 0xffffu;
        } else {
#pragma unroll
            for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
                // rect[r][2] holds this strip's 16-bit column mask of rectangle r (set by the kernel per strip)
                zmask |= (yy >= rect[r][0] && yy <= rect[r][1]) ? (unsigned)rect[r][2] : 0u;
            }
        }
    }
    const float go_u = PPG ? go_row : go, ge_u = PPG ? ge_row : ge;
    // "not MM" / "not MU" / (local) "clamp won" bits and the U-extend / L-extend bits of this lane's 16 cells; the
    // shifted-in words hold column c in bit 15 - c (reversed once per row below).
    unsigned w_nm = 0, w_nu = 0, w_stop = 0, w_u = 0, w_l = 0;
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // the row's cells, with or without the zero-rectangle handling (MK): rows in which no lane of the wave has a
    // masked cell - most rows - take the plain version (wave-level vote below)
    auto row_cells = [&](auto mk_tag) __attribute__((always_inline)) {
    constexpr bool MK = decltype(mk_tag)::value;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const int term = (NTERM == 1) ? 2 : k / NR;
        const int r = k % NR;
        const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
        const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
        if constexpr (!LOOKUP) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(BOPS[ib]), acc, 0, 0, 0);
        if constexpr (DM) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aopH[ia]), as_half8(BOLD[ib]), acc, 0, 0, 0);
        // pin the MFMA at the START of its chunk: left alone the scheduler sinks it to the end of the step and
        // then pads ~35 s_nop for the MFMA -> VALU result hazard in front of the next step's select
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = (16 * k) / NM; c < (16 * (k + 1)) / NM; ++c) {
            // exact candidate sums (cext.c:185-201), maxima and first-match flags (cext.c:207-295)
            // INTS (the single-term instances; the host selects them for integer scoring only, praline_plan_run):
            // every value is exact in float32, so max3(md, ud, ld) + m IS the maximum of the three sums and a state
            // ties exactly when its sum does - the three candidate adds are not needed.
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
            const float uo = Mp[c] + go_u, ue = Up[c] + ge_u;
            float U = __builtin_fmaxf(uo, ue);
            const float lo = mleft + (PPG ? g2o[c] : go), le = lleft + (PPG ? g2e[c] : ge);
            float Lc = __builtin_fmaxf(lo, le);
            // Match source, first match in the order MM, MU, ML: without the clamp one of the three sums IS the
            // maximum, so "MM is not it" and "MU is not it" (the signs of sMM - max and sMU - max, both <= 0) say
            // which; in local mode the clamp wins - stop - exactly when that maximum is negative (its own sign).
            // All three are shifted in like the extend bits below: no compare, no VCC write -> v_cndmask wait
            // states, no scalar mask logic.  The maximum is finite in every interior cell (one state of each
            // boundary cell is), so no NaN here.
            if constexpr (SINK != 1 && SINK != 3) {
                w_nm = shift_in_sign(w_nm, sMM, Mref);
                w_nu = shift_in_sign(w_nu, sMU, Mref);
            }
            if constexpr (LOCAL) {
                if constexpr (SINK != 1 && SINK != 3) w_stop = __builtin_amdgcn_alignbit(w_stop, __builtin_bit_cast(unsigned, M), 31);
                M = __builtin_fmaxf(M, 0.0f);                                    // cext.c:208-209
            }
            if constexpr (MK) {
                if (zmask & (1u << c)) { M = 0.0f; U = 0.0f; Lc = 0.0f; }          // cext.c:141-149 (stop code: row end)
            }
            // "from extend" bits = sign of (open - extend), shifted in with one v_alignbit each: no compare, so
            // no VCC write -> v_cndmask wait states (the compare form cost ~35 s_nop per step).  Column c lands
            // in bit 15 - c; reversed once per row below.  (-inf) - (-inf) only happens in cells no path enters.
            if constexpr (SINK != 1 && SINK != 3) {
                w_u = shift_in_sign(w_u, uo, ue);
                w_l = shift_in_sign(w_l, lo, le);
            }
            md = Mp[c]; ud = Up[c]; ld = Lp[c];
            Mp[c] = M; Up[c] = U; Lp[c] = Lc;
            mleft = M; lleft = Lc;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    };
    if constexpr (MASK) {
        if (__ballot(zmask != 0u) != 0ull) row_cells(std::true_type{});
        else row_cells(std::false_type{});
    } else {
        row_cells(std::false_type{});
    }
    if constexpr (!LOOKUP) PREV = acc;
    if constexpr (LOOKUP) {
        // (the table serves every row of the strip)
    } else if constexpr (BSRC == 1) {
        const unsigned sym = (symw >> (8 * SB)) & 0xffu;
        const float4 *bsrc = reinterpret_cast<const float4 *>(onehot_lane + sym * onehot_stride(NR));
#pragma unroll
        for (int q = 0; q < NR; ++q) BOLD[q] = bsrc[q];
    } else {
        const float4 *bsrc = reinterpret_cast<const float4 *>(b_next);
#pragma unroll
        for (int q = 0; q < NP * NR; ++q) BOLD[q] = bsrc[q];   // (BOLD is BOPS without DM)
    }
    b_next += b_stride;
    if (LOCAL && SINK != 2) {
        // local end cell = first maximum of o in C order (y, x, k) (align.py:402).  The state arrays now hold
        // this row's values: take the row maximum (2 ops per cell) and compare ONCE per row; the column and
        // state are located only inside the update block, which a wave-level vote skips almost always.
        // Inside one strip rows ascend, so a tie only wins with a smaller row (a later strip); ties between
        // the two halves of a pair are resolved when their results are combined.
        // With strictly negative gap scores U and L are always smaller than some earlier M (U = M' + go + k ge), so the
        // maximum of o is attained by M cells only (never tied by a U or L cell): 8 ops per row instead of 31.
        if (!PPG && go < 0.0f && ge < 0.0f) {   // (per-position scores may be zero: the general form)
            const float r0 = max3f(Mp[0], Mp[1], Mp[2]), r1 = max3f(Mp[3], Mp[4], Mp[5]), r2 = max3f(Mp[6], Mp[7], Mp[8]);
            const float r3 = max3f(Mp[9], Mp[10], Mp[11]), r4 = max3f(Mp[12], Mp[13], Mp[14]);
            const float rowH = max3f(max3f(r0, r1, r2), max3f(r3, r4, Mp[15]), PRALINE_NEG_INF);
            const bool better = rowH > best_run || (rowH == best_run && yy < best_y);
            if (__ballot(better) != 0ull) {
                if (better) {
                    best_run = rowH;
                    best_y = yy;
                    best_k = 0;
#pragma unroll
                    for (int c = 15; c >= 0; --c)   // descending: the smallest column with the maximum wins
                        if (Mp[c] == rowH) best_x = xb + c + 1;
                }
            }
        } else {
            float rowH = max3f(Mp[0], Up[0], Lp[0]);
#pragma unroll
            for (int c = 1; c < 16; ++c) rowH = __builtin_fmaxf(rowH, max3f(Mp[c], Up[c], Lp[c]));
            const bool better = rowH > best_run || (rowH == best_run && yy < best_y);
            if (__ballot(better) != 0ull) {
                if (better) {
                    best_run = rowH;
                    best_y = yy;
#pragma unroll
                    for (int c = 15; c >= 0; --c)   // descending: the smallest column with the maximum wins
                        if (max3f(Mp[c], Up[c], Lp[c]) == rowH) {
                            best_x = xb + c + 1;
                            best_k = (Mp[c] == rowH) ? 0 : ((Up[c] == rowH) ? 1 : 2);
                        }
                }
            }
        }
    }
    // lower half: this row's boundary states are the next row's diagonal input
    cdM = bv.x; cdU = bv.y; cdL = bv.z;
    // upper half: shift the hand-over generations, then receive the lower half's last column of this row
    cpxm = cxm; cpxu = cxu; cpxl = cxl;
    cxm = from_lower_half(Mp[15]);
    cxu = from_lower_half(Up[15]);
    cxl = from_lower_half(Lp[15]);

    if constexpr (CHAIN) {
        if (h) chain_store_row(bnd_st, Mp[15], Up[15], Lp[15]);
    } else if constexpr (SINK != 2) {   // (the recompute kernel reads the kept columns, it writes none)
        if (h) *reinterpret_cast<float4 *>(bnd_st) = make_float4(Mp[15], Up[15], Lp[15], 0.0f);
    }
    bnd_st += 32 * sizeof(float4);
    if constexpr (SINK == 1) {
        // forward fill of the two-pass scheme: the states of every 32nd row are the recompute kernel's starting points
#ifndef PRALINE_TB2_ABLATE
#define PRALINE_TB2_ABLATE 0   // timing experiments only: 1 no checkpoint stores, 2 no kept boundary columns
#endif
        if (!(PRALINE_TB2_ABLATE & 1) && yy >= 32 && (yy & 31) == 0) {
            f4n *q = reinterpret_cast<f4n *>(ckpt + (int64_t)(yy >> 5) * PRALINE_TB2_CKPT_FLOATS);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4n vm = {Mp[4 * g], Mp[4 * g + 1], Mp[4 * g + 2], Mp[4 * g + 3]};
                const f4n vu = {Up[4 * g], Up[4 * g + 1], Up[4 * g + 2], Up[4 * g + 3]};
                const f4n vl = {Lp[4 * g], Lp[4 * g + 1], Lp[4 * g + 2], Lp[4 * g + 3]};
                __builtin_nontemporal_store(vm, q + g * 64);
                __builtin_nontemporal_store(vu, q + (4 + g) * 64);
                __builtin_nontemporal_store(vl, q + (8 + g) * 64);
            }
        }
    }
    // match source as two bit planes, code = lo | hi << 1: 1 MM / 2 MU / 3 ML / 0 stop (masked cell, or the clamp won)
    if constexpr (SINK != 1 && SINK != 3) {
        const unsigned r_nm = __builtin_bitreverse32(w_nm) >> 16, r_nu = __builtin_bitreverse32(w_nu) >> 16;
        unsigned go_on = 0xffffu;
        if constexpr (LOCAL) go_on &= ~(__builtin_bitreverse32(w_stop) >> 16);
        if constexpr (MASK) go_on &= ~zmask;
        const unsigned hi = r_nm & go_on, lo_bits = (~r_nm | r_nu) & go_on;
        const unsigned w_src = lo_bits | (hi << 16);
        const unsigned w_ext = (__builtin_bitreverse32(w_u) >> 16) | (__builtin_bitreverse32(w_l) & 0xffff0000u);
        const unsigned long long w64 = (unsigned long long)w_src | ((unsigned long long)w_ext << 32);
        // Task mode: streaming store.  The planes (0.5 B per cell, GBs per launch) are read once, by k_traceback,
        // much later - they must not push the strip-boundary columns and the operand rows out of L2 (global /
        // semiglobal with paths +11 %).  Not in chain mode: there the producer drains its stores before every
        // publish and a streaming store takes longer to retire.
        if constexpr (SINK == 2) *reinterpret_cast<unsigned long long *>(lds_flags + lds_row * 512) = w64;
        else if constexpr (CHAIN) *reinterpret_cast<unsigned long long *>(tb_st) = w64;
        else __builtin_nontemporal_store(w64, reinterpret_cast<unsigned long long *>(tb_st));
    }
    tb_st += 64;
}

#ifndef PRALINE_TB_WAVES_PER_SIMD
#define PRALINE_TB_WAVES_PER_SIMD 1
#endif
#ifndef PRALINE_TB_DM
#define PRALINE_TB_DM 1
#endif
// TWOPASS (task mode only): the FORWARD fill of the two-pass scheme - no flags, no traceback planes; instead every
// strip keeps its own boundary column (bnd: float4 [nstrips + 1][max_l1 + PRALINE_TB2_PAD][32] per task) and the
// states of every 32nd row go to ckpt (float4 [nstrips][PRALINE_TB2_CKPT_BLOCKS][3][4][64] per task, at tk.tb_off):
// k_trace_recompute rebuilds the flags of just the 32 x 32 blocks each path crosses.  End cells as in the single pass.
#ifndef PRALINE_TB2_WAVES_PER_SIMD
#define PRALINE_TB2_WAVES_PER_SIMD 2
#endif
#ifndef PRALINE_TB_LOOKUP_WAVES
#define PRALINE_TB_LOOKUP_WAVES 2   // waves per SIMD of the lookup instances (no operand / accumulator registers)
#endif
#ifndef PRALINE_TB_DENSE_WAVES
#define PRALINE_TB_DENSE_WAVES 1   // waves per SIMD of the dense-tile instances
#endif
// NOFLAGS (dense tiles): the fill without flags and planes - the scores of plans without paths that run with per-position gap
// scores or over strip ranges.  strip_lo / strip_cnt (task mode): this launch sweeps the strips [strip_lo, strip_lo + strip_cnt)
// of every task - the dense tile (ar.dense) then holds just those strips; the boundary column stays in the task's scratch
// between the launches and a local alignment's running maximum travels through scores / end_cells.
template <int NR, int NTERM, bool LOCAL, bool MASK, bool CHAIN = false, bool TWOPASS = false, int BSRC = 0, bool PPG = false, bool NOFLAGS = false>
__global__ __launch_bounds__(256, TWOPASS ? PRALINE_TB2_WAVES_PER_SIMD : (BSRC == 3 ? PRALINE_TB_LOOKUP_WAVES : (BSRC == 4 ? PRALINE_TB_DENSE_WAVES : PRALINE_TB_WAVES_PER_SIMD))) void k_dp_split16_tb(Arena16Dev ar, const WaveTask *__restrict__ tasks,
                                                       const int32_t *__restrict__ lane_one,
                                                       const int32_t *__restrict__ lane_pair, float4 *bnd,
                                                       uint2 *__restrict__ tb, float *__restrict__ aux, RectList rl,
                                                       float *__restrict__ scores, int32_t *__restrict__ end_cells,
                                                       RunParams rp, int n_tasks, int *chain_flags = nullptr,
                                                       int chain_stride = 0, float4 *chain_cand = nullptr,
                                                       int chain_every = 6, int strip_lo = 0, int strip_cnt = 0x3fffffff)
{
    constexpr int NP = (NTERM == 1) ? 1 : 2;
    constexpr int NOP = NP * NR;
    constexpr bool LOOKUP = BSRC == 3;
    constexpr bool DENSE = BSRC == 4;   // match scores from the task's dense tile (ar.dense): no operands, no MFMA
    constexpr bool DM = NTERM == 1 && (PRALINE_TB_DM != 0) && !LOOKUP && !DENSE;   // see split16_tb_step
    // CHAIN && TWOPASS: chain mode WITHOUT flags and checkpoints - the scores-only fill of plans of a few long sequences
    // (one wave per task and strip where the score kernels would put four waves on a task)
    constexpr bool FWD2 = TWOPASS && !CHAIN;   // the forward fill of the two-pass scheme proper
    constexpr int SINK = TWOPASS ? (CHAIN ? 3 : 1) : (NOFLAGS ? 3 : 0);
    static_assert(!NOFLAGS || (BSRC == 4 && !CHAIN && !TWOPASS && !MASK), "the flag-free task-mode fill is wired for the dense tiles");
    static_assert(!PPG || DENSE, "per-position gap scores (rp.gaps) are wired for the dense-tile instances");
    static_assert(BSRC == 0 || (BSRC == 1 && FWD2 && DM) || (LOOKUP && NTERM == 1 && !TWOPASS) || (DENSE && NTERM == 3 && !TWOPASS),
                  "the one-hot table feeds the single-term forward fill; the lookup serves the single-pass integer-scoring fill; "
                  "dense tiles feed the single-pass fill with candidate sums");
    __shared__ __attribute__((aligned(16))) char lookup_all[LOOKUP ? 4 * lookup_bytes(NR) : 16];   // one table per wave of the block
    __shared__ __attribute__((aligned(16))) char onehot_tab[BSRC == 1 ? onehot_bytes(NR) : 16];
    if constexpr (BSRC == 1) {
        _Float16 *tab = reinterpret_cast<_Float16 *>(onehot_tab);
        constexpr int per_row = onehot_stride(NR) / 2;  // halves per table row (the last 8 are padding)
        for (int i = threadIdx.x; i < (16 * NR + 1) * per_row; i += blockDim.x) {
            const int sym = i / per_row, e = i % per_row;
            const int hh = e / (8 * NR), r = (e / 8) % NR, jj = e % 8;
            tab[i] = (e < 16 * NR && 16 * r + 8 * hh + jj == sym) ? (_Float16)1.0f : (_Float16)0.0f;
        }
        __syncthreads();
    }
    // CHAIN: one wave per block; block b = strip-major (strip, task): producers are dispatched before consumers
    const int task = CHAIN ? (int)(blockIdx.x % n_tasks) : (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int chain_strip = CHAIN ? (int)(blockIdx.x / n_tasks) : 0;
    if (task >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int j = lane & 31;
    const WaveTask tk = tasks[task];
    const int base = task * 32;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    const float go = rp.go1, ge = rp.ge1;

    const int my_one = lane_one[base + j];
    const int two = tk.two[0];
    const bool have_pair = my_one >= 0;
    const int my_pair = have_pair ? lane_pair[base + j] : -1;
    const int L1 = have_pair ? ar.len[my_one] : 0;
    const int L2 = ar.len[two];
    const int nstrips = (L2 + 31) >> 5;
    const int clast = (L2 - 1) & 31;
    const bool own_last = (clast >> 4) == h;
    int cidx = clast & 15;
    asm volatile("" : "+v"(cidx));
    const int max_l1 = tk.max_l1;
    const int tb_rows = max_l1 + 8;  // rows per strip in the traceback planes (the pipeline overshoots)

    const char *pB = ar.P16 + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * ar.row_bytes + h * ar.half_bytes;
    const int b_stride = ar.row_bytes;
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const char *qA = ar.Q16 + ((int64_t)ar.row_off[two] + acol) * ar.row_bytes + h * ar.half_bytes;
    const unsigned *pSym = (BSRC == 1) ? reinterpret_cast<const unsigned *>(ar.sym8 + (have_pair ? ar.row_off[my_one] : 0)) : nullptr;
    char *lookup_tab = lookup_all + (LOOKUP ? (int)((threadIdx.x >> 6) & 3) * lookup_bytes(NR) : 0);
    const char *onehot_lane = LOOKUP ? lookup_tab + h * 64 : onehot_tab + h * (16 * NR);
    // LOOKUP: at step T the lower half fetches the scores of row T + 1 (symbol byte T of the sequence), the upper half
    // those of row T (byte T - 1)
    const unsigned char *psym = LOOKUP ? ar.sym8 + (have_pair ? ar.row_off[my_one] : 0) - h : nullptr;
    // DENSE: this lane's 64-byte line of row y of strip s is at dense_task + s * dense_strip + y * 4096 (the upper half's
    // pointer is one row back: at step T both halves fetch "row T + 1")
    const int64_t dense_strip = (int64_t)(max_l1 + PRALINE_DENSE_PAD) * 4096;
    const char *dense_task = DENSE ? reinterpret_cast<const char *>(ar.dense + ar.dense_off[task]) + (int64_t)lane * 64 - (int64_t)h * 4096
                                   : nullptr;
    const char *dense_lane = dense_task;

    if (CHAIN && chain_strip >= nstrips) return;
    // boundary columns: float4 [y][32]; chain mode keeps one per strip boundary, [strip][y][32]
    const int64_t chain_col = CHAIN ? (int64_t)(max_l1 + 24) * 32 : (FWD2 ? (int64_t)(max_l1 + PRALINE_TB2_PAD) * 32 : 0);
    const int ckpt_blocks = PRALINE_TB2_CKPT_BLOCKS(max_l1);
    float *my_ckpt = FWD2 ? reinterpret_cast<float *>(tb) + tk.tb_off + 4 * lane : nullptr;   // float4 [strip][block][3][4][64]
    char *my_bnd = reinterpret_cast<char *>(bnd + tk.bnd_off + chain_col * chain_strip + j);          // read by this wave
    char *my_bnd_out = reinterpret_cast<char *>(bnd + tk.bnd_off + chain_col * (chain_strip + 1) + j);  // written (CHAIN)
    constexpr int BROW = 32 * (int)sizeof(float4);
    const int *chain_in = (CHAIN && chain_strip > 0) ? chain_flags + (int64_t)task * chain_stride + chain_strip - 1 : nullptr;
    int *chain_out = CHAIN ? chain_flags + (int64_t)task * chain_stride + chain_strip : nullptr;
    int chain_seen = 0;
    uint2 *my_tb = tb + tk.tb_off + lane;                            // [strip][y][64]
    float *lastcol = aux + tk.aux_off + j;                                   // [y][3][32]
    float *lastrow = aux + tk.aux_off + (int64_t)(max_l1 + 1) * 3 * 32 + j;  // [x-1][3][32]

    int rect[PRALINE_MAX_RECTS][4];
    if constexpr (MASK) {
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
    } else {
#pragma unroll
        for (int r = 0; r < PRALINE_MAX_RECTS; ++r) { rect[r][0] = 1 << 30; rect[r][1] = -1; rect[r][2] = 1 << 30; rect[r][3] = -1; }
    }
    // dense tiles, plans with more than PRALINE_MAX_RECTS rectangles per pair: the per-row mask words of k_build_zmask, uint32
    // [nstrips][L1 + 1] at zmask + zm_off[pair] (lanes without a pair read their own zero: row range 0)
    const unsigned *my_zm = nullptr;
    if constexpr (MASK && DENSE) {
        if (rl.zmask != nullptr) my_zm = (my_pair >= 0) ? rl.zmask + rl.zm_off[my_pair] : rl.zmask;
    }
    const int zm_rows = (my_pair >= 0) ? L1 : 0;
    // PPG: the gap-score rows (open, extend) of this lane's sequence one and of the shared sequence two
    const float *g1p = PPG ? rp.gaps + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * 2 : nullptr;
    const float *g2p = PPG ? rp.gaps + (int64_t)ar.row_off[two] * 2 : nullptr;
    const int g1_last = L1 > 0 ? L1 - 1 : 0;   // the last position a lane reads (the rows past its sequence are not reported)

    // boundary cells (praline/component/align.py:367-385)
    const float o001 = free_one ? 0.0f : (PPG ? boundary_value_pp(0, g1p, false) : (go - ge));
    const float o002 = free_two ? 0.0f : (PPG ? boundary_value_pp(0, g2p, false) : (go - ge));
    // o[0, x, 2]: PPG reads positions of the shared sequence two up to L2 (never past it)
    auto row0 = [&](int x) __attribute__((always_inline)) {
        if constexpr (PPG) return boundary_value_pp(x <= L2 ? x : L2, g2p, free_two);
        else return boundary_value(x, go, ge, free_two);
    };

    // strip 0 reads its boundary column like every other strip: states of (y, 0) = (-inf, o[y,0,1], -inf)
    if (h == 0 && chain_strip == 0 && strip_lo == 0)
        for (int y = 1; y <= max_l1 + 4; ++y)
            *reinterpret_cast<float4 *>(my_bnd + (int64_t)y * BROW) =
                make_float4(PRALINE_NEG_INF, PPG ? boundary_value_pp(y <= g1_last + 1 ? y : g1_last + 1, g1p, free_one) : boundary_value(y, go, ge, free_one),
                            PRALINE_NEG_INF, 0.0f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);

    // local: first flat argmax over o (align.py:402); o[0,0,:] are the only boundary cells that can be >= 0
    float out_best = 0.0f;
    int out_y = 0, out_x = 0, out_k = 0;
    if (LOCAL) {
        if (o001 > out_best) { out_best = o001; out_k = 1; }
        if (o002 > out_best) { out_best = o002; out_k = 2; }
        if (!CHAIN && strip_lo > 0 && have_pair) {   // the running maximum of the strips before this launch
            out_best = scores[my_pair];
            out_y = end_cells[(int64_t)my_pair * 4 + 0]; out_x = end_cells[(int64_t)my_pair * 4 + 1]; out_k = end_cells[(int64_t)my_pair * 4 + 2];
        }
    }
    float corner_m = PRALINE_NEG_INF, corner_u = PRALINE_NEG_INF, corner_l = PRALINE_NEG_INF;

    const int strip_end = (strip_cnt < nstrips - strip_lo) ? strip_lo + strip_cnt : nstrips;
    for (int s = CHAIN ? chain_strip : strip_lo; s < (CHAIN ? chain_strip + 1 : strip_end); ++s) {
        const int x0 = s * 32;
        const int xb = x0 + 16 * h;
        const bool last_owner = (s == nstrips - 1) && own_last;

        int srect[PRALINE_MAX_RECTS][4];   // rows of the rectangles + their column mask inside this lane's 16 columns
#pragma unroll
        for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
            const int lo = max(rect[r][2] - (xb + 1), 0), hi = min(rect[r][3] - (xb + 1), 15);
            srect[r][0] = rect[r][0];
            srect[r][1] = rect[r][1];
            srect[r][2] = (MASK && lo <= hi) ? (int)((0xffffu >> (15 - hi)) & (0xffffu << lo)) : 0;
            srect[r][3] = 0;
        }
        float4 aop[NOP];
        if constexpr (!LOOKUP && !DENSE) {
            const float4 *sa = reinterpret_cast<const float4 *>(qA + (int64_t)x0 * ar.row_bytes);
#pragma unroll
            for (int q = 0; q < NOP; ++q) aop[q] = sa[q];
        }
        float4 aopH[NOP];
        if constexpr (DM) {
            // A row j feeds output row j; rows with (j >> 2) & 1 are the ones the upper half receives (bit masks:
            // a select between float4 values becomes an indexed stack array)
            const unsigned mh = 0u - (((unsigned)j >> 2) & 1u), ml = ~mh;
#pragma unroll
            for (int q = 0; q < NOP; ++q) {
                const float4 a = aop[q];
                aopH[q] = make_float4(__uint_as_float(__float_as_uint(a.x) & mh), __uint_as_float(__float_as_uint(a.y) & mh),
                                      __uint_as_float(__float_as_uint(a.z) & mh), __uint_as_float(__float_as_uint(a.w) & mh));
                aop[q] = make_float4(__uint_as_float(__float_as_uint(a.x) & ml), __uint_as_float(__float_as_uint(a.y) & ml),
                                     __uint_as_float(__float_as_uint(a.z) & ml), __uint_as_float(__float_as_uint(a.w) & ml));
            }
        }
        float Mp[16], Up[16], Lp[16];  // states of the previous row, per column (o[0,x,:] to start with)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Mp[c] = PRALINE_NEG_INF;
            Up[c] = PRALINE_NEG_INF;
            Lp[c] = row0(xb + c + 1);
        }
        // PPG: (open, extend) of the positions xb + c of sequence two - the scores of L[y][xb + c + 1]
        float g2o[PPG ? 16 : 1], g2e[PPG ? 16 : 1];
        if constexpr (PPG) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int pos = xb + c < L2 ? xb + c : L2 - 1;   // (columns past the sequence reach no reported cell)
                g2o[c] = g2p[2 * pos]; g2e[c] = g2p[2 * pos + 1];
            }
        }
        // lower half: states of the cell (0, x0); upper half: states of (0, x0 + 16), which after the first
        // step's generation shift are the diagonal input of its row 1
        float cdM = (s == 0) ? 0.0f : PRALINE_NEG_INF;
        float cdU = (s == 0) ? o001 : PRALINE_NEG_INF;
        float cdL = (s == 0) ? o002 : row0(x0);
        float cxm = PRALINE_NEG_INF, cxu = PRALINE_NEG_INF, cxl = row0(x0 + 16);
        float cpxm = PRALINE_NEG_INF, cpxu = PRALINE_NEG_INF, cpxl = PRALINE_NEG_INF;
        float best_run = out_best;
        int best_y = out_y, best_x = out_x, best_k = out_k;

        float4 bX[NOP], bY[NOP], bZ[NOP];   // bZ: DM only (rows t, t+1, t+2 rotate through three sets)
        f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 accB = accA;
        f32x16 accC = accA;   // DENSE: three sets
        unsigned sw0 = 0;   // BSRC == 1: the symbols of rows 1 .. 4
        unsigned symA = 0, symB = 0, symC = 0;   // LOOKUP: the symbols three steps ahead (rotate like the boundary prefetch)
        if constexpr (DENSE) {
            // rows 1 and 2 of the strip (the upper half takes rows 0 and 1: its first step is undone below); three register
            // sets rotate, the step at T fetches row T + 2
            dense_lane = dense_task + (int64_t)(s - strip_lo) * dense_strip;
#pragma unroll
            for (int r = 1; r <= 2; ++r) {
                const f4n *q = reinterpret_cast<const f4n *>(dense_lane + (int64_t)r * 4096);
                const f4n a0 = PRALINE_DENSE_LOAD(q), a1 = PRALINE_DENSE_LOAD(q + 1), a2 = PRALINE_DENSE_LOAD(q + 2), a3 = PRALINE_DENSE_LOAD(q + 3);
                f32x16 &d = r == 1 ? accA : accB;
                d[0] = a0.x; d[1] = a0.y; d[2] = a0.z; d[3] = a0.w; d[4] = a1.x; d[5] = a1.y; d[6] = a1.z; d[7] = a1.w;
                d[8] = a2.x; d[9] = a2.y; d[10] = a2.z; d[11] = a2.w; d[12] = a3.x; d[13] = a3.y; d[14] = a3.z; d[15] = a3.w;
            }
        } else if constexpr (LOOKUP) {
            // this strip's table: lane (column j, half h) transposes the hi pieces of half h of the pre-multiplied row
            // x0 + j (exact mode: Q2 = hi exactly), k = 16 r + 8 h + jj  ->  lookup_tab[k][j]
            const char *src = ar.Q16 + ((int64_t)ar.row_off[two] + x0 + j) * ar.row_bytes + h * ar.half_bytes;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const half8 hv = as_half8(reinterpret_cast<const float4 *>(src)[r]);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    *reinterpret_cast<float *>(lookup_tab + (16 * r + 8 * h + jj) * lookup_stride() + j * 4) = (float)hv[jj];
            }
            if (h == 0) *reinterpret_cast<float *>(lookup_tab + (16 * NR) * lookup_stride() + j * 4) = 0.0f;   // padding rows: symbol 16 NR
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            // scores of row 1 (lower half; the upper half's first step is undone below), symbols for the steps 1, 2, 3
            const unsigned first = h ? 0u : (unsigned)psym[0];
            const float4 *q = reinterpret_cast<const float4 *>(onehot_lane + first * lookup_stride());
            const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
            accA[0] = a0.x; accA[1] = a0.y; accA[2] = a0.z; accA[3] = a0.w; accA[4] = a1.x; accA[5] = a1.y; accA[6] = a1.z; accA[7] = a1.w;
            accA[8] = a2.x; accA[9] = a2.y; accA[10] = a2.z; accA[11] = a2.w; accA[12] = a3.x; accA[13] = a3.y; accA[14] = a3.z; accA[15] = a3.w;
            symA = psym[1]; symB = psym[2]; symC = psym[3];
        } else {
            float4 b1[NOP];
            const float4 *s1 = reinterpret_cast<const float4 *>(pB);
            const float4 *s2 = reinterpret_cast<const float4 *>(pB + b_stride);
            const float4 *s3 = reinterpret_cast<const float4 *>(pB + 2 * b_stride);
            if constexpr (BSRC == 1) {
                sw0 = pSym[0];
                s1 = reinterpret_cast<const float4 *>(onehot_lane + (sw0 & 0xffu) * onehot_stride(NR));
                s2 = reinterpret_cast<const float4 *>(onehot_lane + ((sw0 >> 8) & 0xffu) * onehot_stride(NR));
                s3 = reinterpret_cast<const float4 *>(onehot_lane + ((sw0 >> 16) & 0xffu) * onehot_stride(NR));
            }
#pragma unroll
            for (int q = 0; q < NOP; ++q) { b1[q] = s1[q]; bX[q] = s2[q]; bY[q] = s3[q]; bZ[q] = s1[q]; }
#pragma unroll
            for (int k = 0; k < NTERM * NR; ++k) {
                const int term = (NTERM == 1) ? 2 : k / NR;
                const int r = k % NR;
                const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
                const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(b1[ib]), accA, 0, 0, 0);
            }
        }
        const char *b_next = pB + 3 * b_stride;
        // boundary states three rows ahead, in three rotating registers (one row = 0.7 us at one wave per SIMD is
        // not enough for a load that misses L2; measured with the 1-deep version: 27 % of the cycles waiting)
        // TWOPASS: strip s reads column s and writes column s + 1 (all of them stay for the recompute kernel)
        const char *col_in = (FWD2 && !(PRALINE_TB2_ABLATE & 2)) ? my_bnd + (int64_t)s * chain_col * (int64_t)sizeof(float4) : my_bnd;
        const char *bnd_ld = col_in + 4 * BROW;
        char *bnd_st = CHAIN ? my_bnd_out : ((FWD2 && !(PRALINE_TB2_ABLATE & 2)) ? my_bnd + (int64_t)(s + 1) * chain_col * (int64_t)sizeof(float4) : my_bnd);   // upper half stores row yy = t - 1 (row 0: dummy)
        float *ckpt_strip = FWD2 ? my_ckpt + (int64_t)s * ckpt_blocks * PRALINE_TB2_CKPT_FLOATS : nullptr;
        if constexpr (CHAIN) {
            if (chain_in != nullptr) chain_seen = chain_wait(chain_in, 3, chain_seen);
        }
        float4 bnd_prefA = *reinterpret_cast<const float4 *>(col_in + BROW);      // row 1
        float4 bnd_prefB = *reinterpret_cast<const float4 *>(col_in + 2 * BROW);  // row 2
        float4 bnd_prefC = *reinterpret_cast<const float4 *>(col_in + 3 * BROW);  // row 3
        uint2 *tb_st = my_tb + (int64_t)s * tb_rows * 64 + (h ? 0 : 64);     // row yy = t - h of step t = 1
        const unsigned *zm_strip = my_zm ? my_zm + (int64_t)s * (zm_rows + 1) : nullptr;
        auto g1_at = [&](int pos) __attribute__((always_inline)) {
            const int q = pos < 0 ? 0 : (pos > g1_last ? g1_last : pos);
            return *reinterpret_cast<const float2 *>(g1p + 2 * q);
        };
        float2 g1rowA = make_float2(0.0f, 0.0f), g1rowB = g1rowA, g1rowC = g1rowA;   // steps 1, 2, 3
        if constexpr (PPG) { g1rowA = g1_at(0 - h); g1rowB = g1_at(1 - h); g1rowC = g1_at(2 - h); }

        // BUSE holds operand row T + 1; BOLD row T (DM; otherwise BUSE again); PREF the boundary states of row T
#define PRALINE_TB_STEP(T, CUR, PREV, BUSE, BOLD, PREF)                                                              \
        split16_tb_step<NR, NTERM, LOCAL, MASK, CHAIN, DM, SINK>((T) - h, L1, have_pair, h, CUR, PREV, BUSE, BOLD, aop, aopH, \
                                                b_next, b_stride,                                                           \
                                                bnd_ld, bnd_st, PREF, tb_st, Mp, Up, Lp, cxm, cxu, cxl, cpxm, cpxu, cpxl,   \
                                                cdM, cdU, cdL, best_run, best_y, best_x, best_k, go, ge, xb, srect,         \
                                                chain_in, &chain_seen, (T) + 3, ckpt_strip)
        // one-hot table: the refill (operand row T + 3) takes its symbol from byte SBYTE of SYMW
#define PRALINE_TB_STEP_OH(T, CUR, PREV, BUSE, BOLD, PREF, SYMW, SBYTE)                                                \
        split16_tb_step<NR, NTERM, LOCAL, MASK, false, true, SINK, 1, SBYTE>((T) - h, L1, have_pair, h, CUR, PREV, BUSE, BOLD, aop, \
                                                aopH, b_next, b_stride, bnd_ld, bnd_st, PREF, tb_st, Mp, Up, Lp, cxm, cxu,  \
                                                cxl, cpxm, cpxu, cpxl, cdM, cdU, cdL, best_run, best_y, best_x, best_k, go, \
                                                ge, xb, srect, nullptr, nullptr, 0, ckpt_strip, nullptr, 0, onehot_lane, SYMW)
        // match-score lookup: the step fetches the next row's scores with SYM, which is then refilled three steps ahead
#define PRALINE_TB_STEP_LK(T, CUR, PREV, PREF, SYM)                                                                   \
        split16_tb_step<NR, NTERM, LOCAL, MASK, CHAIN, false, SINK, 3>((T) - h, L1, have_pair, h, CUR, PREV, bX, bX, aop, aopH, \
                                                b_next, b_stride, bnd_ld, bnd_st, PREF, tb_st, Mp, Up, Lp, cxm, cxu, cxl,   \
                                                cpxm, cpxu, cpxl, cdM, cdU, cdL, best_run, best_y, best_x, best_k, go, ge,  \
                                                xb, srect, chain_in, &chain_seen, (T) + 3, ckpt_strip, nullptr, 0,          \
                                                onehot_lane, SYM);                                                          \
        SYM = psym[(T) + 3];
        // dense tile: the step fetches the scores of the row two steps on (lower half: row T + 2)
        // (SFX: the rotating register set of the boundary prefetch and - PPG - of the row's gap scores: position T - h - 1 of
        // sequence one at step T, refilled three steps ahead)
#define PRALINE_TB_STEP_DN(T, CUR, PREV, SFX)                                                                         \
        split16_tb_step<NR, NTERM, LOCAL, MASK, CHAIN, false, SINK, 4, 0, PPG>((T) - h, L1, have_pair, h, CUR, PREV, bX, bX, aop, aopH, \
                                                b_next, b_stride, bnd_ld, bnd_st, bnd_pref##SFX, tb_st, Mp, Up, Lp, cxm, cxu, cxl, \
                                                cpxm, cpxu, cpxl, cdM, cdU, cdL, best_run, best_y, best_x, best_k, go, ge,  \
                                                xb, srect, chain_in, &chain_seen, (T) + 3, ckpt_strip, nullptr, 0,          \
                                                dense_lane + (int64_t)((T) + 2) * 4096, 0u, zm_strip, zm_rows, g1row##SFX.x, \
                                                g1row##SFX.y, g2o, g2e);                                                    \
        if constexpr (PPG) g1row##SFX = g1_at((T) + 3 - h - 1);
#define PRALINE_TB_TAILS(T)                                                                                          \
        {                                                                                                            \
            const int yy_ = (T) - h;                                                                                 \
            if (semiglobal && last_owner && have_pair && yy_ >= 1 && yy_ <= L1) {                                    \
                float *lc = lastcol + (int64_t)yy_ * 3 * 32;            /* o[y, L2, :]  (align.py:408,418-422) */    \
                lc[0] = select16(Mp, cidx); lc[32] = select16(Up, cidx); lc[64] = select16(Lp, cidx);                \
            }                                                                                                        \
            if (have_pair && yy_ == L1) {                                                                            \
                if (LOCAL) { out_best = best_run; out_y = best_y; out_x = best_x; out_k = best_k; }                  \
                if (last_owner) { corner_m = select16(Mp, cidx); corner_u = select16(Up, cidx); corner_l = select16(Lp, cidx); } \
                if (semiglobal) {                                       /* o[L1, x, :]  (align.py:407,413-417) */    \
                    float *lr = lastrow + (int64_t)xb * 3 * 32;                                                      \
                    _Pragma("unroll") for (int c = 0; c < 16; ++c) {                                                 \
                        lr[(c * 3 + 0) * 32] = Mp[c]; lr[(c * 3 + 1) * 32] = Up[c]; lr[(c * 3 + 2) * 32] = Lp[c];    \
                    }                                                                                                \
                }                                                                                                    \
            }                                                                                                        \
        }
        // step 1: only the lower half has a row; the upper half's garbage is undone right after
        {
            const float best_s = best_run;
            const int by = best_y, bx = best_x, bk = best_k;
            if constexpr (DENSE) { PRALINE_TB_STEP_DN(1, accA, accC, A) }
            else if constexpr (LOOKUP) { PRALINE_TB_STEP_LK(1, accA, accB, bnd_prefA, symA) }
            else if constexpr (BSRC == 1) PRALINE_TB_STEP_OH(1, accA, accB, bX, bZ, bnd_prefA, sw0, 3);
            else if constexpr (DM) PRALINE_TB_STEP(1, accA, accB, bX, bZ, bnd_prefA);
            else PRALINE_TB_STEP(1, accA, accB, bX, bX, bnd_prefA);
            if (h) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    Mp[c] = PRALINE_NEG_INF; Up[c] = PRALINE_NEG_INF; Lp[c] = row0(xb + c + 1);
                }
                best_run = best_s; best_y = by; best_x = bx; best_k = bk;
            }
            PRALINE_TB_TAILS(1)
        }
        // six steps per iteration: the accumulators ping-pong (period 2), the boundary prefetch registers and - with
        // DM - the operand sets rotate with period 3 (without DM two operand sets alternate); the steps past
        // max_l1 + 1 compute rows that nobody reports
        int chain_next = chain_every;
        if constexpr (BSRC == 1) {
            // twelve steps per iteration: the step at T refills operand row T + 3, symbol byte T + 2 of the sequence;
            // T = 2 (mod 12), so the twelve symbols are exactly the dwords 1 + 3 k, 2 + 3 k, 3 + 3 k (loaded one
            // iteration ahead)
            const unsigned *pn = pSym + 1;
            unsigned d1 = pn[0], d2 = pn[1], d3 = pn[2];
            for (int t = 2; t <= max_l1 + 1; t += 12) {
                pn += 3;
                const unsigned n1 = pn[0], n2 = pn[1], n3 = pn[2];
                PRALINE_TB_STEP_OH(t, accB, accA, bY, bX, bnd_prefB, d1, 0);
                PRALINE_TB_TAILS(t)
                PRALINE_TB_STEP_OH(t + 1, accA, accB, bZ, bY, bnd_prefC, d1, 1);
                PRALINE_TB_TAILS(t + 1)
                PRALINE_TB_STEP_OH(t + 2, accB, accA, bX, bZ, bnd_prefA, d1, 2);
                PRALINE_TB_TAILS(t + 2)
                PRALINE_TB_STEP_OH(t + 3, accA, accB, bY, bX, bnd_prefB, d1, 3);
                PRALINE_TB_TAILS(t + 3)
                PRALINE_TB_STEP_OH(t + 4, accB, accA, bZ, bY, bnd_prefC, d2, 0);
                PRALINE_TB_TAILS(t + 4)
                PRALINE_TB_STEP_OH(t + 5, accA, accB, bX, bZ, bnd_prefA, d2, 1);
                PRALINE_TB_TAILS(t + 5)
                PRALINE_TB_STEP_OH(t + 6, accB, accA, bY, bX, bnd_prefB, d2, 2);
                PRALINE_TB_TAILS(t + 6)
                PRALINE_TB_STEP_OH(t + 7, accA, accB, bZ, bY, bnd_prefC, d2, 3);
                PRALINE_TB_TAILS(t + 7)
                PRALINE_TB_STEP_OH(t + 8, accB, accA, bX, bZ, bnd_prefA, d3, 0);
                PRALINE_TB_TAILS(t + 8)
                PRALINE_TB_STEP_OH(t + 9, accA, accB, bY, bX, bnd_prefB, d3, 1);
                PRALINE_TB_TAILS(t + 9)
                PRALINE_TB_STEP_OH(t + 10, accB, accA, bZ, bY, bnd_prefC, d3, 2);
                PRALINE_TB_TAILS(t + 10)
                PRALINE_TB_STEP_OH(t + 11, accA, accB, bX, bZ, bnd_prefA, d3, 3);
                PRALINE_TB_TAILS(t + 11)
                d1 = n1; d2 = n2; d3 = n3;
            }
        } else
        for (int t = 2; t <= max_l1 + 1; t += 6) {
            if constexpr (CHAIN) {
                // the steps up to t - 1 have stored the boundary rows up to t - 2.  Every publish drains the wave's
                // stores (~1 us): plans with more waves per strip level than the chip has slots publish rarely -
                // their consumers are dispatched a round later anyway (chain_every, set by the host)
                if (t - 2 >= chain_next) { chain_publish(chain_out, t - 2, lane); chain_next = t - 2 + chain_every; }
            }
            if constexpr (DENSE) {
                PRALINE_TB_STEP_DN(t, accB, accA, B)
                PRALINE_TB_TAILS(t)
                PRALINE_TB_STEP_DN(t + 1, accC, accB, C)
                PRALINE_TB_TAILS(t + 1)
                PRALINE_TB_STEP_DN(t + 2, accA, accC, A)
                PRALINE_TB_TAILS(t + 2)
                PRALINE_TB_STEP_DN(t + 3, accB, accA, B)
                PRALINE_TB_TAILS(t + 3)
                PRALINE_TB_STEP_DN(t + 4, accC, accB, C)
                PRALINE_TB_TAILS(t + 4)
                PRALINE_TB_STEP_DN(t + 5, accA, accC, A)
                PRALINE_TB_TAILS(t + 5)
            } else if constexpr (LOOKUP) {
                PRALINE_TB_STEP_LK(t, accB, accA, bnd_prefB, symB)
                PRALINE_TB_TAILS(t)
                PRALINE_TB_STEP_LK(t + 1, accA, accB, bnd_prefC, symC)
                PRALINE_TB_TAILS(t + 1)
                PRALINE_TB_STEP_LK(t + 2, accB, accA, bnd_prefA, symA)
                PRALINE_TB_TAILS(t + 2)
                PRALINE_TB_STEP_LK(t + 3, accA, accB, bnd_prefB, symB)
                PRALINE_TB_TAILS(t + 3)
                PRALINE_TB_STEP_LK(t + 4, accB, accA, bnd_prefC, symC)
                PRALINE_TB_TAILS(t + 4)
                PRALINE_TB_STEP_LK(t + 5, accA, accB, bnd_prefA, symA)
                PRALINE_TB_TAILS(t + 5)
            } else if constexpr (DM) {
                PRALINE_TB_STEP(t, accB, accA, bY, bX, bnd_prefB);
                PRALINE_TB_TAILS(t)
                PRALINE_TB_STEP(t + 1, accA, accB, bZ, bY, bnd_prefC);
                PRALINE_TB_TAILS(t + 1)
                PRALINE_TB_STEP(t + 2, accB, accA, bX, bZ, bnd_prefA);
                PRALINE_TB_TAILS(t + 2)
                PRALINE_TB_STEP(t + 3, accA, accB, bY, bX, bnd_prefB);
                PRALINE_TB_TAILS(t + 3)
                PRALINE_TB_STEP(t + 4, accB, accA, bZ, bY, bnd_prefC);
                PRALINE_TB_TAILS(t + 4)
                PRALINE_TB_STEP(t + 5, accA, accB, bX, bZ, bnd_prefA);
                PRALINE_TB_TAILS(t + 5)
            } else {
                PRALINE_TB_STEP(t, accB, accA, bY, bY, bnd_prefB);
                PRALINE_TB_TAILS(t)
                PRALINE_TB_STEP(t + 1, accA, accB, bX, bX, bnd_prefC);
                PRALINE_TB_TAILS(t + 1)
                PRALINE_TB_STEP(t + 2, accB, accA, bY, bY, bnd_prefA);
                PRALINE_TB_TAILS(t + 2)
                PRALINE_TB_STEP(t + 3, accA, accB, bX, bX, bnd_prefB);
                PRALINE_TB_TAILS(t + 3)
                PRALINE_TB_STEP(t + 4, accB, accA, bY, bY, bnd_prefC);
                PRALINE_TB_TAILS(t + 4)
                PRALINE_TB_STEP(t + 5, accA, accB, bX, bX, bnd_prefA);
                PRALINE_TB_TAILS(t + 5)
            }
        }
#undef PRALINE_TB_STEP
#undef PRALINE_TB_STEP_LK
#undef PRALINE_TB_STEP_DN
#undef PRALINE_TB_STEP_OH
#undef PRALINE_TB_TAILS
        if constexpr (CHAIN) chain_publish(chain_out, PRALINE_CHAIN_DONE, lane);
    }
    if (CHAIN && LOCAL) {
        // every strip reports its own first-argmax candidate (value, y, x, k); k_chain_local_end picks per pair
        const float pv = partner_value(out_best, h);
        const int py = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_y), h));
        const int px = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_x), h));
        const int pk = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_k), h));
        if (pv > out_best || (pv == out_best && (py < out_y || (py == out_y && px < out_x)))) {
            out_best = pv; out_y = py; out_x = px; out_k = pk;
        }
        if (h == 0)
            chain_cand[((int64_t)task * chain_stride + chain_strip) * 32 + j] =
                make_float4(out_best, __builtin_bit_cast(float, out_y), __builtin_bit_cast(float, out_x), __builtin_bit_cast(float, out_k));
        return;
    }
    if (CHAIN && chain_strip != nstrips - 1) return;   // the last strip's wave reports the end cell

    // ---- combine the halves: end cell (y, x, k) and score (align.py:401-431) ----
    if (LOCAL) {
        // first flat argmax: larger value wins; on ties the smaller (y, x)
        const float pv = partner_value(out_best, h);
        const int py = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_y), h));
        const int px = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_x), h));
        const int pk = __builtin_bit_cast(int, partner_value(__builtin_bit_cast(float, out_k), h));
        if (pv > out_best || (pv == out_best && (py < out_y || (py == out_y && px < out_x)))) {
            out_best = pv; out_y = py; out_x = px; out_k = pk;
        }
    }
    const float cm = __builtin_fmaxf(corner_m, partner_value(corner_m, h));  // only the owner half holds finite values
    const float cu = __builtin_fmaxf(corner_u, partner_value(corner_u, h));
    const float cl = __builtin_fmaxf(corner_l, partner_value(corner_l, h));
    if (have_pair && h == 0) {
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
        scores[my_pair] = score;  // semiglobal: k_traceback overwrites it with the row / column rule
    }
}
