// dp_split.hip.h -- k_dp_split: the scores-only throughput kernel ("split strip" layout).
//
// One wavefront owns 32 pairs that share their sequence TWO.  BOTH 32-lane halves work on the
// same 32 pairs: lane j sweeps columns 1..16 of the current 32-column strip and lane j+32 sweeps
// columns 17..32 ONE DP ROW BEHIND it (a two-lane wavefront; the diagonal and left inputs of
// column 17 are handed over with two v_permlane32_swap per row).  The A-operand rows of the MFMA
// are permuted so that accumulator register r is strip column r in the lower half and column
// 16 + r in the upper half: the 32x32 match-score tile
//     D[i][lane] = sum_k Q2[x0 + col(i)][k] * P1_lane[y][k]          (cext.c:33-97, 308-455)
// is consumed straight out of the accumulators - no LDS, no cross-lane exchange of scores.
// Software pipeline per DP row t: the MFMAs of row t+1 are issued under the VALU recurrence of
// row t (two accumulator sets ping-pong), B operands are fetched two rows ahead, the
// strip-boundary column one row ahead.
//
// Recurrence (cext.c:99-306), per cell, with H = max(M, U, L) carried per column:
//   M = H[y-1][x-1] + m ; (local: M = max(M, 0)) ; U = U[y][x] (computed one row earlier)
//   H = max3(M, U, L) ; U[y+1][x] = max(M + go1, U + ge1) ; L[y][x+1] = max(M + go2, L + ge2)
// max(a + m, b + m, c + m) == max(a, b, c) + m holds exactly in IEEE arithmetic (rounding is
// monotone), so carrying H instead of the three states is bit-identical for the scores.
//
// lane_one / lane_pair: 32 entries per task.  bnd: float2 [max_l1 + 2][32] per task.
#pragma once
#include "dp_kernels.hip.h"

__device__ __forceinline__ float from_lower_half(float v)
{
    // upper lanes (32-63) receive the value of lane-32; lower lanes receive 0.
    unsigned ua = 0u, ub = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    ua = r[0];
    return __builtin_bit_cast(float, ua);
}

__device__ __forceinline__ float partner_value(float v, int half)
{
    // the value held by lane ^ 32
    unsigned ua = __builtin_bit_cast(unsigned, v), ub = ua;
    auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    ua = r[0];
    ub = r[1];
    return __builtin_bit_cast(float, half ? ua : ub);
}

__device__ __forceinline__ float select16(const float (&v)[16], int idx)
{
    float t8[8], t4[4], t2[2];
    const bool b0 = idx & 1, b1 = idx & 2, b2 = idx & 4, b3 = idx & 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) t8[k] = b0 ? v[2 * k + 1] : v[2 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) t4[k] = b1 ? t8[2 * k + 1] : t8[2 * k];
#pragma unroll
    for (int k = 0; k < 2; ++k) t2[k] = b2 ? t4[2 * k + 1] : t4[2 * k];
    return b3 ? t2[1] : t2[0];
}

template <int NQ> __device__ __forceinline__ float f4elem(const float4 (&v)[NQ], int k)
{
    const float4 q = v[k >> 2];
    return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
}

struct SplitCtx {
    float go1, ge1, go2, ge2;
    bool free_one;
    bool semiglobal_last_owner;  // semiglobal && last strip && this half holds column L2
    int cidx;
};

// One pipeline step: DP row t for the lower half / row t-1 for the upper half.
//   CUR  : accumulators of row t   (read by the lower half)
//   PREV : accumulators of row t-1 (read by the upper half), then overwritten with row t+1
//   BOPS : B operands of row t+1, then refilled with row t+3
template <int NSTEP, bool LOCAL, int NQ>
__device__ __forceinline__ void split_step(int t, int s, int h, int L1, bool have_pair, const f32x16 &CUR,
                                           f32x16 &PREV, float4 (&BOPS)[NQ], const float (&aop)[NSTEP],
                                           const float *pB, int KP, float2 *my_bnd, float2 &bnd_pref,
                                           float (&Hp)[16], float (&Uc)[16], float &dH, float &hd_x,
                                           float &l_x, float &best, float &colmax, const SplitCtx &cx,
                                           int max_l1)
{
    // 1. this lane's 16 match scores: lower half row t, upper half row t-1
    float m[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) m[c] = h ? PREV[c] : CUR[c];

    // 2. MFMAs of row t+1 into PREV (its old contents were consumed above)
    {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NSTEP; ++k)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], f4elem<NQ>(BOPS, k), acc, 0, 0, 0);
        PREV = acc;
    }
    // 3. refill BOPS with row t+3 (0-based arena row t+2); rows past the end of a sequence read the
    //    next sequence / the zeroed tail padding - finite values that only reach masked-off rows
    {
        const float4 *src = reinterpret_cast<const float4 *>(pB + (int64_t)(t + 2) * KP);
#pragma unroll
        for (int q = 0; q < NQ; ++q) BOPS[q] = src[q];
    }
    // 4. boundary column of this row (lower half), prefetch the next one
    const float2 bv = bnd_pref;
    if (h == 0 && s > 0 && t + 1 <= max_l1) bnd_pref = my_bnd[(int64_t)(t + 1) * 32];

    // 5. the recurrence
    const int yy = t - h;
    float hd_out = PRALINE_NEG_INF, lrun_out = PRALINE_NEG_INF;
    if (have_pair && yy >= 1 && yy <= L1) {
        float hl, lin;
        if (s == 0) { hl = boundary_value(yy, cx.go1, cx.ge1, cx.free_one); lin = PRALINE_NEG_INF; }
        else { hl = bv.x; lin = bv.y; }
        float hd = h ? hd_x : dH;
        float lrun = h ? l_x : lin;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float M = hd + m[c];
            if (LOCAL) M = __builtin_fmaxf(M, 0.0f);
            const float U = Uc[c];
            const float H = max3f(M, U, lrun);
            if (LOCAL) best = __builtin_fmaxf(best, H);
            Uc[c] = __builtin_fmaxf(M + cx.go1, U + cx.ge1);
            lrun = __builtin_fmaxf(M + cx.go2, lrun + cx.ge2);
            hd = Hp[c];
            Hp[c] = H;
        }
        hd_out = hd;
        lrun_out = lrun;
        if (h == 0) dH = hl;
        else my_bnd[(int64_t)yy * 32] = make_float2(Hp[15], lrun);  // H[yy][x0+32], L[yy][x0+33]
        if (cx.semiglobal_last_owner) colmax = __builtin_fmaxf(colmax, select16(Hp, cx.cidx));
    }
    // 6. hand the inputs of column 17 (same row) to the upper half for the next step
    hd_x = from_lower_half(hd_out);
    l_x = from_lower_half(lrun_out);
}

template <int NSTEP, bool LOCAL>
__global__ __launch_bounds__(64) void k_dp_split(ArenaDev ar, const WaveTask *__restrict__ tasks,
                                                 const int32_t *__restrict__ lane_one,
                                                 const int32_t *__restrict__ lane_pair,
                                                 float2 *bnd, float *__restrict__ scores, RunParams rp)
{
    constexpr int NQ = (NSTEP + 3) / 4;
    const int lane = threadIdx.x;
    const int h = lane >> 5;
    const int j = lane & 31;
    const WaveTask tk = tasks[blockIdx.x];
    const int base = blockIdx.x * 32;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    const float go1 = rp.go1, ge1 = rp.ge1, go2 = rp.go2, ge2 = rp.ge2;

    const int my_one = lane_one[base + j];
    const int two = tk.two[0];
    const bool have_pair = my_one >= 0;
    const int L1 = have_pair ? ar.len[my_one] : 0;
    const int L2 = ar.len[two];
    const int nstrips = (L2 + 31) >> 5;
    const int clast = (L2 - 1) & 31;
    const bool own_last = (clast >> 4) == h;  // this half holds column L2 in the last strip
    const int max_l1 = tk.max_l1;

    // B operand: profile row of this lane's sequence one, k parity = h
    const float *pB = ar.P + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * ar.KP + h * ar.KS;
    // A operand: MFMA row i = j is strip column 16g + 4q + r for i = 8q + 4g + r, so that the
    // accumulator registers (r' = 4q + r) of half g are the consecutive columns 16g .. 16g + 15
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const float *qA = ar.Q + ((int64_t)ar.row_off[two] + acol) * ar.KP + h * ar.KS;

    float2 *my_bnd = bnd + tk.bnd_off + j;  // [y][32]

    // boundary cells (praline/component/align.py:367-385)
    const float o001 = free_one ? 0.0f : (go1 - ge1);
    const float o002 = free_two ? 0.0f : (go2 - ge2);
    const float h00 = max3f(0.0f, o001, o002);

    float best = 0.0f;  // local: running max of o; o[0,0,:] are the only boundary cells that can be >= 0
    if (LOCAL) best = __builtin_fmaxf(best, __builtin_fmaxf(o001, o002));
    float rowmax = (have_pair && h == 0) ? boundary_value(L1, go1, ge1, free_one) : PRALINE_NEG_INF;  // o[L1,0,1]
    float colmax = (have_pair && own_last) ? boundary_value(L2, go2, ge2, free_two) : PRALINE_NEG_INF; // o[0,L2,2]
    float corner = PRALINE_NEG_INF;

    SplitCtx cx;
    cx.go1 = go1; cx.ge1 = ge1; cx.go2 = go2; cx.ge2 = ge2;
    cx.free_one = free_one;
    cx.cidx = clast & 15;

    for (int s = 0; s < nstrips; ++s) {
        const int x0 = s * 32;
        const int xb = x0 + 16 * h;  // this lane's columns are DP columns xb+1 .. xb+16
        const bool is_last = s == nstrips - 1;
        cx.semiglobal_last_owner = semiglobal && is_last && own_last;

        float aop[NSTEP];
        {
            const float4 *sa = reinterpret_cast<const float4 *>(qA + (int64_t)x0 * ar.KP);
            float4 va[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) va[q] = sa[q];
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) aop[k] = f4elem<NQ>(va, k);
        }
        float Hp[16], Uc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Hp[c] = boundary_value(xb + c + 1, go2, ge2, free_two);  // H[0][x] = o[0,x,2]
            Uc[c] = PRALINE_NEG_INF;                                 // U[1][x]
        }
        float dH = (s == 0) ? h00 : boundary_value(x0, go2, ge2, free_two);  // lower half: H[y-1][x0]
        float hd_x = PRALINE_NEG_INF, l_x = PRALINE_NEG_INF;                   // upper half inputs

        // ---- pipeline prologue: B operands of rows 1..3, MFMAs of row 1, boundary of row 1 ----
        float4 bX[NQ], bY[NQ];
        f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 accB = accA;
        {
            float4 b1[NQ];
            const float4 *s1 = reinterpret_cast<const float4 *>(pB);
            const float4 *s2 = reinterpret_cast<const float4 *>(pB + (int64_t)ar.KP);
            const float4 *s3 = reinterpret_cast<const float4 *>(pB + (int64_t)2 * ar.KP);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { b1[q] = s1[q]; bX[q] = s2[q]; bY[q] = s3[q]; }
#pragma unroll
            for (int k = 0; k < NSTEP; ++k)
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], f4elem<NQ>(b1, k), accA, 0, 0, 0);
        }
        float2 bnd_pref = make_float2(0.0f, 0.0f);
        if (h == 0 && s > 0) bnd_pref = my_bnd[32];  // row 1

        // rows: step t handles row t (lower half) and row t-1 (upper half)
        for (int t = 1; t <= max_l1 + 1; t += 2) {
            split_step<NSTEP, LOCAL, NQ>(t, s, h, L1, have_pair, accA, accB, bX, aop, pB, ar.KP, my_bnd, bnd_pref,
                                         Hp, Uc, dH, hd_x, l_x, best, colmax, cx, max_l1);
            split_step<NSTEP, LOCAL, NQ>(t + 1, s, h, L1, have_pair, accB, accA, bY, aop, pB, ar.KP, my_bnd, bnd_pref,
                                         Hp, Uc, dH, hd_x, l_x, best, colmax, cx, max_l1);
        }

        // ---- strip epilogue: every lane's state is frozen at its last row L1 ----
        if (have_pair) {
            if (semiglobal) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    rowmax = __builtin_fmaxf(rowmax, (xb + c + 1 <= L2) ? Hp[c] : PRALINE_NEG_INF);
            }
            if (is_last && own_last) corner = select16(Hp, cx.cidx);
        }
    }

    // ---- combine the two halves of each pair and write the score (align.py:401-431) ----
    const float corner_all = __builtin_fmaxf(corner, partner_value(corner, h));
    const float rowmax_all = __builtin_fmaxf(rowmax, partner_value(rowmax, h));
    const float colmax_all = __builtin_fmaxf(colmax, partner_value(colmax, h));
    const float best_all = __builtin_fmaxf(best, partner_value(best, h));
    if (have_pair && h == 0) {
        float score;
        if (LOCAL) score = best_all;
        else if (semiglobal) score = (rowmax_all > colmax_all && free_two) ? rowmax_all : colmax_all;
        else score = corner_all;
        scores[lane_pair[base + j]] = score;
    }
}
