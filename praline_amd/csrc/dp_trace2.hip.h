// dp_trace2.hip.h -- k_trace_recompute: the BACKWARD pass of the two-pass alignments-with-paths scheme.
//
// The single-pass fill (k_dp_split16_tb, SINK = 0) forms four tie flags for EVERY cell - 17 to 21 VALU operations
// per cell against 8 to 10 for the bare three-state recurrence - although a traceback visits L1 + L2 of the L1 x L2
// cells.  Two passes: the forward fill (k_dp_split16_tb<..., TWOPASS>) forms no flags; it keeps every strip's boundary
// column and the (M, U, L) states of every 32nd row.  This kernel then walks each path backwards strip by strip and
// RECOMPUTES, with the flag logic of the single pass (split16_tb_step, SINK = 2: flag words into LDS), only the
// 32-row x 32-column blocks the path enters, each from its checkpoint row and its strip's boundary column:
//   * same MFMA instructions on the same operands -> the same match scores, bit for bit;
//   * same recurrence from the forward pass's own states -> the same M / U / L values, hence the same flags as a full
//     single-pass fill: paths are identical (praline/util/align.py:144-185 first-set-flag order, cext.c:224-295 ties).
// One wave per task (32 pairs sharing sequence two, lanes j / j + 32 = column halves as in the fill).  All pairs of a
// task cross the same strips, from the last to the first; inside a strip every lane recomputes ITS current block
// (the B operand rows and the boundary rows are per-lane anyway) until its path has left the strip.  End cells,
// boundary flags, masked cells and the semiglobal extensions are handled as in k_traceback (dp_kernels.hip.h).
#pragma once
#include "dp_split16_tb.hip.h"

#ifndef PRALINE_TB2_BWD_WAVES
#define PRALINE_TB2_BWD_WAVES 2
#endif
#define PRALINE_TB2_ROWS(BH) ((BH) + 8)   // LDS flag rows per block: the block's rows + the pipeline's overshoot

// BH: rows per recomputed block = the forward fill's checkpoint spacing (32: k_dp_split16_tb<..., TWOPASS> and
// k_dp_split16<..., KEEP>; PRALINE_KEEP_BH: k_dp_pipe<..., KEEP>, whose checkpoints are float4 [row / BH][3][4][64] per strip
// with pipe_keep_blocks() blocks).  analytic4 (k_dp_pipe forward): the column 0 of EVERY task, float4 (-inf, o[y,0,1], -inf)
// [row][32], written once per launch - that forward fill keeps no copy per task.
template <int NR, int NTERM, bool LOCAL, bool MASK, int BH = 32>
__global__ __launch_bounds__(64, MASK ? 1 : PRALINE_TB2_BWD_WAVES) void k_trace_recompute(Arena16Dev ar, const WaveTask *__restrict__ tasks,
                                                        const int32_t *__restrict__ lane_one,
                                                        const int32_t *__restrict__ lane_pair, const float4 *__restrict__ bnd,
                                                        const float *__restrict__ ckpt, RectList rl,
                                                        const int32_t *__restrict__ end_cells,
                                                        const int64_t *__restrict__ slot_off, int32_t *__restrict__ paths,
                                                        int64_t *__restrict__ path_start, int32_t *__restrict__ path_rows,
                                                        RunParams rp, int n_tasks, int keep_in_aux = 0,
                                                        const float4 *__restrict__ analytic4 = nullptr, int ckpt_blocks_arg = 0)
{
    static_assert(BH == 32 || BH % 6 == 0, "the six-step rotation covers a block exactly, or BH = 32 with its two tail steps");
    // keep_in_aux: the forward fill was k_dp_split16<..., KEEP> - the kept columns sit at tk.aux_off (tk.bnd_off is
    // that kernel's own (H, L) hand-off column)
    constexpr int NP = (NTERM == 1) ? 1 : 2;
    constexpr int NOP = NP * NR;
    constexpr bool DM = NTERM == 1 && (PRALINE_TB_DM != 0);
    __shared__ __attribute__((aligned(16))) char lds_flags_all[PRALINE_TB2_ROWS(BH) * 512];
    const int task = blockIdx.x;
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
    const int max_l1 = tk.max_l1;

    const char *pB = ar.P16 + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * ar.row_bytes + h * ar.half_bytes;
    const int b_stride = ar.row_bytes;
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const char *qA = ar.Q16 + ((int64_t)ar.row_off[two] + acol) * ar.row_bytes + h * ar.half_bytes;

    const int64_t col_elems = (int64_t)(max_l1 + PRALINE_TB2_PAD) * 32;   // float4 elements per boundary column
    const char *my_bnd = reinterpret_cast<const char *>(bnd + (keep_in_aux ? tk.aux_off : tk.bnd_off) + j);
    constexpr int BROW = 32 * (int)sizeof(float4);
    const int ckpt_blocks = ckpt_blocks_arg > 0 ? ((max_l1 + 12 > PRALINE_PIPE_MIN_STEPS ? max_l1 + 12 : PRALINE_PIPE_MIN_STEPS) / BH + 1)
                                                : PRALINE_TB2_CKPT_BLOCKS(max_l1);   // (ckpt_blocks_arg: the pipeline forward's count, pipe_keep_blocks)
    const float4 *my_ckpt = reinterpret_cast<const float4 *>(ckpt + tk.tb_off) + lane;   // float4 [strip][block][3][4][64]
    char *lds_flags = lds_flags_all + lane * 8;

    int rect[PRALINE_MAX_RECTS][4];
    int n_rects = 0;
    if constexpr (MASK) {
        int r0 = 0;
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
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);

    // ---- traceback state of this lane's pair (both halves keep a copy; the lower half writes the path) ----
    int y = 0, x = 0, k = 0;
    bool stopped = !have_pair;
    if (have_pair) { y = end_cells[(int64_t)my_pair * 4 + 0]; x = end_cells[(int64_t)my_pair * 4 + 1]; k = end_cells[(int64_t)my_pair * 4 + 2]; }
    const int64_t slot_end = have_pair ? slot_off[my_pair] + (L1 + L2 + 2) : 0;
    int64_t w = slot_end;
    const bool writer = have_pair && h == 0;
    auto emit = [&](int yy, int xx) { --w; if (writer) { paths[2 * w] = yy; paths[2 * w + 1] = xx; } };
    if (have_pair) {
        if (semiglobal) {   // suffix extension (align.py:284-295)
            if (y != L1) { for (int yy = L1; yy > y; --yy) emit(yy, x); }
            else if (x != L2) { for (int xx = L2; xx > x; --xx) emit(y, xx); }
        }
        emit(y, x);
    }

    for (int s = nstrips - 1; s >= 0; --s) {
        const int x0 = s * 32;
        const int xb = x0 + 16 * h;
        // nobody left in this strip (or to its left)?
        if (__ballot(!stopped && y >= 1 && x >= 1) == 0ull) break;
        if (__ballot(!stopped && y >= 1 && x > x0) == 0ull) continue;

        int srect[PRALINE_MAX_RECTS][4];
#pragma unroll
        for (int r = 0; r < PRALINE_MAX_RECTS; ++r) {
            const int lo = max(rect[r][2] - (xb + 1), 0), hi = min(rect[r][3] - (xb + 1), 15);
            srect[r][0] = rect[r][0];
            srect[r][1] = rect[r][1];
            srect[r][2] = (MASK && lo <= hi) ? (int)((0xffffu >> (15 - hi)) & (0xffffu << lo)) : 0;
            srect[r][3] = 0;
        }
        float4 aop[NOP];
        {
            const float4 *sa = reinterpret_cast<const float4 *>(qA + (int64_t)x0 * ar.row_bytes);
#pragma unroll
            for (int q = 0; q < NOP; ++q) aop[q] = sa[q];
        }
        float4 aopH[NOP];
        if constexpr (DM) {
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
        const char *col_in = (s == 0 && analytic4 != nullptr) ? reinterpret_cast<const char *>(analytic4 + j)
                                                              : my_bnd + (int64_t)s * col_elems * (int64_t)sizeof(float4);
        const float4 *ckpt_strip = my_ckpt + (int64_t)s * ckpt_blocks * (PRALINE_TB2_CKPT_FLOATS / 4);

        for (;;) {
            const bool act = !stopped && y >= 1 && x > x0;      // (x <= x0 + 32 holds: the strips are walked downwards)
            if (__ballot(act) == 0ull) break;
            const int yb0 = act ? ((y - 1) / BH) * BH : 0;       // this lane's block: rows yb0 + 1 .. yb0 + BH

            // ---- recompute the block with the single pass's flag logic ----
            float Mp[16], Up[16], Lp[16];
            auto load_top = [&]() {
                if (yb0 == 0) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        Mp[c] = PRALINE_NEG_INF; Up[c] = PRALINE_NEG_INF; Lp[c] = boundary_value(xb + c + 1, go, ge, free_two);
                    }
                } else {
                    const float4 *q = ckpt_strip + (int64_t)(yb0 / BH) * (PRALINE_TB2_CKPT_FLOATS / 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 vm = q[g * 64], vu = q[(4 + g) * 64], vl = q[(8 + g) * 64];
                        Mp[4 * g] = vm.x; Mp[4 * g + 1] = vm.y; Mp[4 * g + 2] = vm.z; Mp[4 * g + 3] = vm.w;
                        Up[4 * g] = vu.x; Up[4 * g + 1] = vu.y; Up[4 * g + 2] = vu.z; Up[4 * g + 3] = vu.w;
                        Lp[4 * g] = vl.x; Lp[4 * g + 1] = vl.y; Lp[4 * g + 2] = vl.z; Lp[4 * g + 3] = vl.w;
                    }
                }
            };
            load_top();
            // lower half: states of the boundary cell (yb0, x0)
            float cdM, cdU, cdL;
            if (yb0 == 0) {
                cdM = (s == 0) ? 0.0f : PRALINE_NEG_INF;
                cdU = (s == 0) ? o001 : PRALINE_NEG_INF;
                cdL = (s == 0) ? o002 : boundary_value(x0, go, ge, free_two);
            } else {
                const float4 bq = *reinterpret_cast<const float4 *>(col_in + (int64_t)yb0 * BROW);
                cdM = bq.x; cdU = bq.y; cdL = bq.z;
            }
            // upper half: states of (yb0, x0 + 16) = the lower half's last column of the top row
            float cxm = from_lower_half(Mp[15]), cxu = from_lower_half(Up[15]), cxl = from_lower_half(Lp[15]);
            float cpxm = PRALINE_NEG_INF, cpxu = PRALINE_NEG_INF, cpxl = PRALINE_NEG_INF;
            float best_run = 0.0f;
            int best_y = 0, best_x = 0, best_k = 0;

            const char *pBy = pB + (int64_t)yb0 * b_stride;
            float4 bX[NOP], bY[NOP], bZ[NOP];
            f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            f32x16 accB = accA;
            {
                float4 b1[NOP];
                const float4 *s1 = reinterpret_cast<const float4 *>(pBy);
                const float4 *s2 = reinterpret_cast<const float4 *>(pBy + b_stride);
                const float4 *s3 = reinterpret_cast<const float4 *>(pBy + 2 * b_stride);
#pragma unroll
                for (int q = 0; q < NOP; ++q) { b1[q] = s1[q]; bX[q] = s2[q]; bY[q] = s3[q]; bZ[q] = s1[q]; }
#pragma unroll
                for (int kk = 0; kk < NTERM * NR; ++kk) {
                    const int term = (NTERM == 1) ? 2 : kk / NR;
                    const int r = kk % NR;
                    const int ia = (NTERM == 2) ? kk : ((term == 0) ? NR + r : r);
                    const int ib = (NTERM == 2) ? kk : ((term == 1) ? NR + r : r);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(b1[ib]), accA, 0, 0, 0);
                }
            }
            const char *b_next = pBy + 3 * b_stride;
            const char *bnd_ld = col_in + (int64_t)(yb0 + 4) * BROW;
            char *bnd_st = nullptr;   // SINK = 2 stores no boundary
            float4 bnd_prefA = *reinterpret_cast<const float4 *>(col_in + (int64_t)(yb0 + 1) * BROW);
            float4 bnd_prefB = *reinterpret_cast<const float4 *>(col_in + (int64_t)(yb0 + 2) * BROW);
            float4 bnd_prefC = *reinterpret_cast<const float4 *>(col_in + (int64_t)(yb0 + 3) * BROW);
            uint2 *tb_st = nullptr;

#define PRALINE_TB2_STEP(T, CUR, PREV, BUSE, BOLD, PREF)                                                                \
            split16_tb_step<NR, NTERM, LOCAL, MASK, false, DM, 2>(yb0 + (T) - h, L1, have_pair, h, CUR, PREV, BUSE, BOLD, aop, aopH, \
                                                    b_next, b_stride, bnd_ld, bnd_st, PREF, tb_st, Mp, Up, Lp, cxm, cxu, cxl, \
                                                    cpxm, cpxu, cpxl, cdM, cdU, cdL, best_run, best_y, best_x, best_k, go, ge, \
                                                    xb, srect, nullptr, nullptr, 0, nullptr, lds_flags, (T) - h)
            // step 1: only the lower half has a row; the upper half's garbage is undone right after
            if constexpr (DM) PRALINE_TB2_STEP(1, accA, accB, bX, bZ, bnd_prefA);
            else PRALINE_TB2_STEP(1, accA, accB, bX, bX, bnd_prefA);
            if (h) load_top();
            // steps 2 .. BH + 1 (the upper half's row BH is computed at step BH + 1): full rounds of the six-step rotation
            // (BH = 32: five of them and the first two steps of a sixth)
            constexpr int T_ROT = (BH == 32) ? 31 : BH + 1;
            for (int t = 2; t <= T_ROT; t += 6) {
                if constexpr (DM) {
                    PRALINE_TB2_STEP(t, accB, accA, bY, bX, bnd_prefB);
                    PRALINE_TB2_STEP(t + 1, accA, accB, bZ, bY, bnd_prefC);
                    PRALINE_TB2_STEP(t + 2, accB, accA, bX, bZ, bnd_prefA);
                    PRALINE_TB2_STEP(t + 3, accA, accB, bY, bX, bnd_prefB);
                    PRALINE_TB2_STEP(t + 4, accB, accA, bZ, bY, bnd_prefC);
                    PRALINE_TB2_STEP(t + 5, accA, accB, bX, bZ, bnd_prefA);
                } else {
                    PRALINE_TB2_STEP(t, accB, accA, bY, bY, bnd_prefB);
                    PRALINE_TB2_STEP(t + 1, accA, accB, bX, bX, bnd_prefC);
                    PRALINE_TB2_STEP(t + 2, accB, accA, bY, bY, bnd_prefA);
                    PRALINE_TB2_STEP(t + 3, accA, accB, bX, bX, bnd_prefB);
                    PRALINE_TB2_STEP(t + 4, accB, accA, bY, bY, bnd_prefC);
                    PRALINE_TB2_STEP(t + 5, accA, accB, bX, bX, bnd_prefA);
                }
            }
            if constexpr (BH == 32) {
                if constexpr (DM) {
                    PRALINE_TB2_STEP(32, accB, accA, bY, bX, bnd_prefB);
                    PRALINE_TB2_STEP(33, accA, accB, bZ, bY, bnd_prefC);
                } else {
                    PRALINE_TB2_STEP(32, accB, accA, bY, bY, bnd_prefB);
                    PRALINE_TB2_STEP(33, accA, accB, bX, bX, bnd_prefC);
                }
            }
#undef PRALINE_TB2_STEP
            // the flag words of all lanes are in LDS before any lane walks them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();

            // ---- walk this pair's path through the block (align.py:155-180 on the recomputed flags) ----
            if (act) {
                while (y > yb0 && x > x0) {
                    bool masked = false;
                    if constexpr (MASK) {
#pragma unroll
                        for (int r = 0; r < PRALINE_MAX_RECTS; ++r)
                            masked = masked || (y >= rect[r][0] && y <= rect[r][1] && x >= rect[r][2] && x <= rect[r][3]);
                    }
                    const int c = (x - 1) & 31, bit = c & 15;
                    const uint2 word = *reinterpret_cast<const uint2 *>(lds_flags_all + (y - yb0) * 512 + (j + 32 * (c >> 4)) * 8);
                    const int code = (int)(((word.x >> bit) & 1u) | (((word.x >> (16 + bit)) & 1u) << 1));
                    const int ub = (int)((word.y >> bit) & 1u), lb = (int)((word.y >> (16 + bit)) & 1u);
                    if (masked || (k == 0 && code == 0)) { stopped = true; break; }   // t is 0 there (cext.c:141-149 / the clamp)
                    const int nk = k == 0 ? code - 1 : (k == 1 ? ub : 2 * lb);
                    y -= (k != 2);
                    x -= (k != 1);
                    k = nk;
                    emit(y, x);
                }
            }
            __builtin_amdgcn_wave_barrier();   // the next block overwrites the LDS rows
        }
    }

    if (have_pair) {
        // on a boundary cell: the pre-initialised flags (align.py:377,385): t[y>=1,0,1] = UE, t[0,x>=1,2] = LE
        int guard = L1 + L2 + 2;
        while (!stopped && guard-- > 0) {
            if (x == 0 && y >= 1 && k == 1 && !free_one) --y;
            else if (y == 0 && x >= 1 && k == 2 && !free_two) --x;
            else break;
            emit(y, x);
        }
        // prefix extension (align.py:270-279): (y, x) is now the first path row
        if (semiglobal) {
            if (y != 0) { for (int yy = y - 1; yy >= 0; --yy) emit(yy, 0); }
            else if (x != 0) { for (int xx = x - 1; xx >= 0; --xx) emit(0, xx); }
        }
        if (writer) {
            path_start[my_pair] = w;
            path_rows[my_pair] = (int32_t)(slot_end - w);
        }
    }
}
