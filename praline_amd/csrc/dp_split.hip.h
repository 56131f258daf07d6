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
//
// Software pipeline per step t (DP row t in the lower half, row t-1 in the upper half):
//   * the NSTEP MFMAs of row t+1 are issued INTERLEAVED with the VALU recurrence of row t (two
//     accumulator sets ping-pong); for that the recurrence is branch-free and lives in the same
//     basic block as the MFMAs (sched_group_barrier pins 1 MFMA : VALU_PER_MFMA VALU);
//   * B operands are fetched two rows ahead, the strip-boundary column one row ahead.
// Branch-free means lanes keep computing (harmless, finite garbage) after their own last row L1;
// everything a pair reports is snapshotted in the step in which the lane is AT row L1.
//
// Recurrence (cext.c:99-306), per cell, with H = max(M, U, L) carried per column:
//   M = H[y-1][x-1] + m ; (local: M = max(M, 0)) ; U = U[y][x] (computed one row earlier)
//   H = max3(M, U, L) ; U[y+1][x] = max(M + go1, U + ge1) ; L[y][x+1] = max(M + go2, L + ge2)
// max(a + m, b + m, c + m) == max(a, b, c) + m holds exactly in IEEE arithmetic (rounding is
// monotone), so carrying H instead of the three states is bit-identical for the scores.
//
// lane_one / lane_pair: 32 entries per task.  bnd: float2 [max_l1 + 3][32] per task.
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

// v[idx] for a per-lane idx in 0..15.  Written with bit masks ((a & ~m) | (b & m) -> v_bfi_b32):
// a ternary over two array elements gets folded by the compiler into a dynamically indexed load,
// which drags the whole register array into LDS / scratch.
__device__ __forceinline__ float bit_select(float a, float b, unsigned mask)
{
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    return __builtin_bit_cast(float, (ua & ~mask) | (ub & mask));
}

__device__ __forceinline__ float select16(const float (&v)[16], int idx)
{
    float t8[8], t4[4], t2[2];
    const unsigned m0 = 0u - (unsigned)(idx & 1), m1 = 0u - (unsigned)((idx >> 1) & 1);
    const unsigned m2 = 0u - (unsigned)((idx >> 2) & 1), m3 = 0u - (unsigned)((idx >> 3) & 1);
#pragma unroll
    for (int k = 0; k < 8; ++k) t8[k] = bit_select(v[2 * k], v[2 * k + 1], m0);
#pragma unroll
    for (int k = 0; k < 4; ++k) t4[k] = bit_select(t8[2 * k], t8[2 * k + 1], m1);
#pragma unroll
    for (int k = 0; k < 2; ++k) t2[k] = bit_select(t4[2 * k], t4[2 * k + 1], m2);
    return bit_select(t2[0], t2[1], m3);
}

template <int NQ> __device__ __forceinline__ float f4elem(const float4 (&v)[NQ], int k)
{
    const float4 q = v[k >> 2];
    return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
}

struct SplitCtx {
    float go, ge;     // gap open / extend (PairwiseAligner uses one gap series for both sequences)
    bool last_owner;  // last strip && this half holds column L2
    bool semiglobal;
    int cidx;         // (L2 - 1) & 15, laundered into a VGPR so select16 stays a v_cndmask tree
    int xb;           // this lane's columns are DP columns xb+1 .. xb+16
    int L2;
};

struct SplitOut {      // per-pair results, snapshotted when the lane is at its last row
    float best;        // local: max over o
    float rowmax;      // semiglobal: max over o[L1, :, :]
    float colmax;      // semiglobal: max over o[:, L2, :]
    float corner;      // global: max_k o[L1, L2, k]
};

// One pipeline step: DP row t for the lower half / row t-1 for the upper half.
//   CUR  : accumulators of row t   (read by the lower half)
//   PREV : accumulators of row t-1 (read by the upper half), then overwritten with row t+1
//   BOPS : B operands of row t+1, then refilled with row t+3 from b_next
//   EXP  : ablation switches for scripts/exp_ablate.py (0 in production)
template <int NSTEP, bool LOCAL, int NQ, int EXP>
__device__ __forceinline__ void split_step(int yy, int L1, bool have_pair, int h, const f32x16 &CUR, f32x16 &PREV,
                                           float4 (&BOPS)[NQ], const float (&aop)[NSTEP], const char *&b_next, int b_stride,
                                           const char *&bnd_ld, char *&bnd_st, float2 &bnd_pref, float (&Hp)[16],
                                           float (&Uc)[16], float &dH, float &hd_x, float &l_x, float &best_run,
                                           float &col_run, SplitOut &out, const SplitCtx &cx)
{
    // ---- match scores of this lane's row: lower half row t (CUR), upper half row t-1 (PREV) ----
    float m[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) m[c] = h ? PREV[c] : CUR[c];

    // boundary column of this row (H[yy][x0], L[yy][x0+1]); prefetch the next row's
    const float2 bv = bnd_pref;
    if (!(EXP & 2)) bnd_pref = *reinterpret_cast<const float2 *>(bnd_ld);
    bnd_ld += 32 * sizeof(float2);
    float hd = h ? hd_x : dH;
    float lrun = h ? l_x : bv.y;
    __builtin_amdgcn_sched_barrier(0);

    // ---- MFMAs of row t+1 (into PREV, consumed above) interleaved with the recurrence of this row:
    //      MFMA k, then its share of the 16 columns; sched_barrier(0) pins the order, so the VALU
    //      work runs in the shadow of the 64-cycle fp32 MFMA passes ----
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NSTEP; ++k) {
        if (EXP & 16) {
            // timing-only experiment: bf16 MFMAs (separate matrix pipe) in place of the fp32 chain
            typedef short bf16x8_t __attribute__((ext_vector_type(8)));
            bf16x8_t av, bw;
#pragma unroll
            for (int e = 0; e < 8; ++e) { av[e] = (short)__builtin_bit_cast(unsigned, aop[(k + e) % NSTEP]); bw[e] = (short)__builtin_bit_cast(unsigned, f4elem<NQ>(BOPS, (k + e) % NSTEP)); }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc, 0, 0, 0);
            if (k < 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw, av, acc, 0, 0, 0);
        } else if (!(EXP & 4)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], f4elem<NQ>(BOPS, k), acc, 0, 0, 0);
        else acc[k] += aop[k] + f4elem<NQ>(BOPS, k);
#pragma unroll
        for (int c = (16 * k) / NSTEP; c < ((EXP & 8) ? 0 : (16 * (k + 1)) / NSTEP); ++c) {
            float M = hd + m[c];                               // max_k o[y-1,x-1,k] + m   (cext.c:192-222)
            if (LOCAL) M = __builtin_fmaxf(M, 0.0f);           // cext.c:208-209
            const float U = Uc[c];
            const float H = max3f(M, U, lrun);
            if (LOCAL) best_run = __builtin_fmaxf(best_run, H);
            const float Mo = M + cx.go;                        // gap opened from this cell
            Uc[c] = __builtin_fmaxf(Mo, U + cx.ge);            // U[y+1][x]   (cext.c:152-166,247-254)
            lrun = __builtin_fmaxf(Mo, lrun + cx.ge);          // L[y][x+1]   (cext.c:169-183,276-283)
            hd = Hp[c];
            Hp[c] = H;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    PREV = acc;
    // refill BOPS with row t+3; rows past the end of a sequence read the next sequence / the zeroed
    // tail padding: finite values that only feed rows nobody reports
    if (!(EXP & 1)) {
        const float4 *bsrc = reinterpret_cast<const float4 *>(b_next);
#pragma unroll
        for (int q = 0; q < NQ; ++q) BOPS[q] = bsrc[q];
    }
    b_next += b_stride;
    dH = bv.x;  // H[yy][x0] is the diagonal input of the next row (only the lower half reads dH)
    // hand the inputs of column 17 (same row) to the upper half for the next step
    hd_x = from_lower_half(hd);
    l_x = from_lower_half(lrun);

    // ---- per-half / rare tails ----
    if (h && !(EXP & 2)) *reinterpret_cast<float2 *>(bnd_st) = make_float2(Hp[15], lrun);  // H[yy][x0+32], L[yy][x0+33]
    bnd_st += 32 * sizeof(float2);
    if (cx.semiglobal && cx.last_owner) col_run = __builtin_fmaxf(col_run, select16(Hp, cx.cidx));
    if (have_pair && yy == L1) {
        // this lane has just finished the last row of its pair: snapshot what the pair reports
        if (LOCAL) out.best = best_run;
        if (cx.semiglobal) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                out.rowmax = __builtin_fmaxf(out.rowmax, (cx.xb + c + 1 <= cx.L2) ? Hp[c] : PRALINE_NEG_INF);
            out.colmax = col_run;
        }
        if (cx.last_owner) out.corner = select16(Hp, cx.cidx);
    }
}

template <int NSTEP, bool LOCAL, int EXP = 0>
__global__ __launch_bounds__(256) void k_dp_split(ArenaDev ar, const WaveTask *__restrict__ tasks,
                                                  const int32_t *__restrict__ lane_one,
                                                  const int32_t *__restrict__ lane_pair, float2 *bnd,
                                                  float *__restrict__ scores, RunParams rp, int n_tasks)
{
    constexpr int NQ = (NSTEP + 3) / 4;
    // one task per wavefront; a workgroup carries blockDim.x / 64 of them
    const int task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int j = lane & 31;
    const WaveTask tk = tasks[task];
    const int base = task * 32;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const float go = rp.go1, ge = rp.ge1;  // host guarantees go1 == go2, ge1 == ge2 for this kernel

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
    const char *pB = reinterpret_cast<const char *>(ar.P + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * ar.KP + h * ar.KS);
    const int b_stride = ar.KP * (int)sizeof(float);
    // A operand: MFMA row i = j is strip column 16g + 4q + r for i = 8q + 4g + r, so that the
    // accumulator registers (r' = 4q + r) of half g are the consecutive columns 16g .. 16g + 15
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const float *qA = ar.Q + ((int64_t)ar.row_off[two] + acol) * ar.KP + h * ar.KS;

    char *my_bnd = reinterpret_cast<char *>(bnd + tk.bnd_off + j);  // float2 [y][32]
    constexpr int BROW = 32 * (int)sizeof(float2);

    // boundary cells (praline/component/align.py:367-385)
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);
    const float h00 = max3f(0.0f, o001, o002);

    // strip 0 reads its boundary column like every other strip: fill (H[y][0], L[y][1]) = (o[y,0,1], -inf)
    if (h == 0)
        for (int y = 1; y <= max_l1 + 2; ++y)
            *reinterpret_cast<float2 *>(my_bnd + (int64_t)y * BROW) = make_float2(boundary_value(y, go, ge, free_one), PRALINE_NEG_INF);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);

    SplitCtx cx;
    cx.go = go; cx.ge = ge;
    cx.semiglobal = rp.mode >= 2;
    int cidx_v = clast & 15;
    asm volatile("" : "+v"(cidx_v));  // keep it a per-lane value (see select16); launder a LOCAL, not a struct member
    cx.cidx = cidx_v;  // keep it a per-lane value: see SplitCtx
    cx.L2 = L2;

    SplitOut out;
    // local: o[0,0,:] are the only boundary cells that can be >= 0 for gap scores <= 0
    out.best = LOCAL ? h00 : 0.0f;
    out.rowmax = (have_pair && h == 0) ? boundary_value(L1, go, ge, free_one) : PRALINE_NEG_INF;  // o[L1,0,1]
    out.colmax = (have_pair && own_last) ? boundary_value(L2, go, ge, free_two) : PRALINE_NEG_INF; // o[0,L2,2]
    out.corner = PRALINE_NEG_INF;

    for (int s = 0; s < nstrips; ++s) {
        const int x0 = s * 32;
        cx.xb = x0 + 16 * h;
        cx.last_owner = (s == nstrips - 1) && own_last;

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
            Hp[c] = boundary_value(cx.xb + c + 1, go, ge, free_two);  // H[0][x] = o[0,x,2]
            Uc[c] = PRALINE_NEG_INF;                                  // U[1][x]
        }
        float dH = (s == 0) ? h00 : boundary_value(x0, go, ge, free_two);  // lower half: H[0][x0]
        float hd_x = PRALINE_NEG_INF, l_x = PRALINE_NEG_INF;                // upper half inputs
        float best_run = out.best;     // running values; the pair's results are snapshots of them
        float col_run = out.colmax;

        // ---- pipeline prologue: B operands of rows 1..3, MFMAs of row 1, boundary of row 1 ----
        float4 bX[NQ], bY[NQ];
        f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 accB = accA;
        {
            float4 b1[NQ];
            const float4 *s1 = reinterpret_cast<const float4 *>(pB);
            const float4 *s2 = reinterpret_cast<const float4 *>(pB + b_stride);
            const float4 *s3 = reinterpret_cast<const float4 *>(pB + 2 * b_stride);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { b1[q] = s1[q]; bX[q] = s2[q]; bY[q] = s3[q]; }
#pragma unroll
            for (int k = 0; k < NSTEP; ++k)
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], f4elem<NQ>(b1, k), accA, 0, 0, 0);
        }
        const char *b_next = pB + 3 * b_stride;                   // row 4 (0-based 3): first refill
        const char *bnd_ld = my_bnd + 2 * BROW;                   // next prefetch: row 2
        char *bnd_st = my_bnd;                                    // upper half stores row yy = t - 1 (row 0: dummy)
        float2 bnd_pref = *reinterpret_cast<const float2 *>(my_bnd + BROW);  // row 1

        // step 1: only the lower half has a row; the upper half's garbage is undone right after
        {
            float Hs[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) Hs[c] = Hp[c];
            const float best_s = best_run, col_s = col_run;
            split_step<NSTEP, LOCAL, NQ, EXP>(1 - h, L1, have_pair, h, accA, accB, bX, aop, b_next, b_stride, bnd_ld, bnd_st,
                                              bnd_pref, Hp, Uc, dH, hd_x, l_x, best_run, col_run, out, cx);
            if (h) {
#pragma unroll
                for (int c = 0; c < 16; ++c) { Hp[c] = Hs[c]; Uc[c] = PRALINE_NEG_INF; }
                best_run = best_s;
                col_run = col_s;
            }
        }
        // steps 2 .. max_l1 + 1, two per iteration (accumulators and operand sets ping-pong)
        for (int t = 2; t <= max_l1 + 1; t += 2) {
            split_step<NSTEP, LOCAL, NQ, EXP>(t - h, L1, have_pair, h, accB, accA, bY, aop, b_next, b_stride, bnd_ld, bnd_st,
                                              bnd_pref, Hp, Uc, dH, hd_x, l_x, best_run, col_run, out, cx);
            split_step<NSTEP, LOCAL, NQ, EXP>(t + 1 - h, L1, have_pair, h, accA, accB, bX, aop, b_next, b_stride, bnd_ld,
                                              bnd_st, bnd_pref, Hp, Uc, dH, hd_x, l_x, best_run, col_run, out, cx);
        }
    }

    // ---- combine the two halves of each pair and write the score (align.py:401-431) ----
    const float corner_all = __builtin_fmaxf(out.corner, partner_value(out.corner, h));
    const float rowmax_all = __builtin_fmaxf(out.rowmax, partner_value(out.rowmax, h));
    const float colmax_all = __builtin_fmaxf(out.colmax, partner_value(out.colmax, h));
    const float best_all = __builtin_fmaxf(out.best, partner_value(out.best, h));
    if (have_pair && h == 0) {
        float score;
        if (LOCAL) score = best_all;
        else if (cx.semiglobal) score = (rowmax_all > colmax_all && free_two) ? rowmax_all : colmax_all;
        else score = corner_all;
        scores[lane_pair[base + j]] = score;
    }
}
