// dp_pipe.hip.h -- k_dp_pipe: the scores-only fill as a PIPELINE of the four waves of a workgroup.
//
// What round 3's measurements said about k_dp_split16 on float profiles (scripts/exp_ablate16.py, scripts/micro/
// step_cost.hip): a saturated launch is bound by its memory instructions, not by the vector ALU -
// every wave streams its own 32 operand rows (4 KiB per step, 11.5 TB/s of L2 -> LDS traffic at 2.9 TCUPS, the
// measured ceiling of that gather pattern) and round-trips the strip-boundary column through memory (one 256-byte
// store and one 256-byte DMA per step: 16 % of the launch); without both the same recurrence runs twice as fast.
//
// Here the four waves of a workgroup work on tasks that share ONE set of 32 sequences one (PipeItem, sched.cpp):
//   * wave r sweeps the strips r, r + 4, r + 8, ... of the item's concatenated strip list (the strips of task 0, then
//     of task 1, ...), PRALINE_PIPE_LAG = 2 steps behind wave r - 1: all four are within 6 rows of each other;
//   * the set's operand rows are therefore streamed ONCE per workgroup and step: wave r fetches a quarter of the row
//     (pairs 8 r .. 8 r + 7, one 1 KiB LDS-DMA) five steps ahead into a 12-row ring that all four waves read
//     (a quarter of the bytes and a quarter of the DMA instructions per wave);
//   * a strip's boundary column (H, L of its last column, row by row) is handed to the wave of the next strip through
//     a four-row LDS ring; only the hand-off from wave 3 to wave 0's next strip goes through memory (`bnd`, one round
//     = rsteps - 6 steps later);
//   * the waves meet at one s_barrier per step: it orders ring writes (every wave waits for its own DMA piece of the
//     row three steps after issuing it), ring reuse and the boundary hand-off.
// The arithmetic of a cell is split16_step's (dp_split16.hip.h) instruction for instruction - scores are bit-identical
// to k_dp_split16 (tests/test_gpu_parity.py::test_pipeline_workgroups_agree_bitwise).
//
// Step u (0-based) of a round: lower half DP row u + 1, upper half row u.  A round has rsteps = 12 k >= max_l1 + 1
// steps; stream position p = round * rsteps + u holds the operand row u + 1 of every sequence of the set and lives in
// ring slot p % 12 = u % 12: static per unrolled step, the same for every wave.
//
// LDS per workgroup: [12 x 4 KiB ring][4 x 4 KiB A tiles of the next strips][4 x 1 KiB boundary rings][results].
#pragma once
#include "dp_split16.hip.h"

__host__ __device__ constexpr int pipe_ring_bytes() { return PRALINE_PIPE_RING * 4096; }
__host__ __device__ constexpr int pipe_lds_bytes()
{
    return pipe_ring_bytes() + 4 * 4096 + 4 * 1024 + PRALINE_PIPE_MAX_TASKS * 2 * 32 * 4;
}

// barrier of the pipeline: every LDS access of this wave (the hand-off write above all) has completed before it
#define PRALINE_PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct PipeDma {
    unsigned long long src;    // wave-uniform: P16 + (row of the next position to fetch) * row bytes
    unsigned dst;              // LDS byte address of this wave's piece in the ring slot of that position
    unsigned left;             // positions left in the round before the row cursor wraps
    unsigned rsteps;
    unsigned ring_lo, ring_hi; // ring bounds for this wave's piece (dst wraps from ring_hi to ring_lo)
    unsigned gofs;             // per-lane byte offset of the piece (VGPR)
    // wave 0 only: boundary rows from `bnd`
    unsigned long long bsrc;   // address of the next boundary row to fetch
    unsigned bdst_base, brow;  // LDS ring base; next row index (ring slot = brow & 3)
    unsigned bgofs;
};

// one operand piece (1 KiB) of the next stream position
__device__ __forceinline__ void pipe_issue(PipeDma &d)
{
    unsigned keep;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 : "=&s"(keep)
                 : "v"(d.gofs), "s"(d.src), "s"(d.dst)
                 : "memory");
    d.dst += 4096;
    if (d.dst == d.ring_hi) d.dst = d.ring_lo;
    d.src += 128;
    if (--d.left == 0) {   // once per round: a real branch (the empty asm keeps it from becoming a chain of selects)
        asm volatile("");
        d.left = d.rsteps;
        d.src -= (unsigned long long)d.rsteps * 128ull;
    }
}
// wave 0: the boundary row `brow` of the wrap-around column (256 bytes: float2 per pair)
__device__ __forceinline__ void pipe_issue_bnd(PipeDma &d)
{
    unsigned keep;
    const unsigned dst = d.bdst_base + (d.brow & 3u) * 256u;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, %2\n\t"
                 : "=&s"(keep)
                 : "v"(d.bgofs), "s"(d.bsrc), "s"(dst)
                 : "memory");
    d.bsrc += 256;
    d.brow += 1;
    // the column is rewritten every round (rows 1 .. rsteps; row rsteps is never written: it only feeds padding rows)
    if (d.brow > d.rsteps) {
        asm volatile("");
        d.brow = 1;
        d.bsrc -= (unsigned long long)d.rsteps * 256ull;
    }
}

// The per-step wait: every memory operation of this wave older than the last two steps' has completed - ranks 0 and 3
// issue two per step (piece + boundary DMA / boundary store), ranks 1 and 2 one; `extra`: the four A-tile DMAs of the
// next strip were issued within those two steps.
__device__ __forceinline__ void pipe_wait(bool wait4, bool extra)
{
    if (extra) { if (wait4) PRALINE_VMCNT(8); else PRALINE_VMCNT(6); }
    else { if (wait4) PRALINE_VMCNT(4); else PRALINE_VMCNT(2); }
}

struct PipeStrip {             // wave-uniform facts of the strip being swept
    bool first;                // strip 0 of its task: the boundary column is the analytic column 0
    bool last_owner;           // (per lane) last strip && this half holds column L2
    int xb, L2;
};

// One step of an active wave.  K = u % 12.  Register roles as in split16_step (BSRC = 2 without DM):
//   CUR row u + 1's scores, PREV row u's (receives row u + 2's); BOPS operands of row u + 2, BFILL receives row u + 3.
template <int NR, int NTERM, bool LOCAL, int K>
__device__ __forceinline__ void pipe_step(int u, int L1, bool have_pair, int h, const f32x16 &CUR, f32x16 &PREV,
                                          const float4 (&BOPS)[4], float4 (&BFILL)[4], const float4 (&aop)[4],
                                          const char *ring, const unsigned (&stage_rd)[4], const char *bnd_in, char *bnd_out,
                                          char *&bnd_st, bool to_memory, bool wait4, PipeDma &dma, bool has_bnd_dma,
                                          float (&Hs)[17], float (&Uc)[16], float &dH, float &hd_x, float &l_x,
                                          float &best_run, float &col_run, float &out_best, float &out_rowmax, float &out_colmax,
                                          float &out_corner, float go, float ge, bool free_one, bool semiglobal, int cidx,
                                          const PipeStrip &sp, bool may_snap, bool extra_ops = false)
{
    static_assert(NR == 2 && (NTERM == 2 || NTERM == 3), "k_dp_pipe is built for the 128-byte operand rows of float-profile arenas");
    constexpr int NM = NTERM * NR;
    const int yy = u + 1 - h;   // this lane's DP row
    // every DMA piece older than two steps has landed (this wave's share of the rows read below), then the barrier: the
    // other waves' shares too, and the previous step's boundary hand-off
    pipe_wait(wait4, (K == 1 || K == 2) && extra_ops);
    PRALINE_PIPE_BARRIER();
    // boundary column of row u + 1: (H[y][x0], L[y][x0 + 1]); u = K (mod 12), so the ring slots are static
    float2 bv = *reinterpret_cast<const float2 *>(bnd_in + ((K + 1) & 3) * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        BFILL[q] = *reinterpret_cast<const float4 *>(ring + ((K + 2) % PRALINE_PIPE_RING) * 4096 + stage_rd[q]);
    // the half select does not depend on the boundary value: it covers the LDS latency of the read above
    f2 m2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        m2[c].x = h ? PREV[2 * c] : CUR[2 * c];
        m2[c].y = h ? PREV[2 * c + 1] : CUR[2 * c + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (sp.first) {   // column 0 of the task: o[y,0,1], no L (a real branch: one strip in ~13 pays for the float64 form)
        asm volatile("");
        bv = make_float2(boundary_value(u + 1, go, ge, free_one), PRALINE_NEG_INF);
    }
    Hs[0] = h ? hd_x : dH;
    float lrun = h ? l_x : bv.y;
    const float hd_out = Hs[16];
    const f2 go2 = {go, go}, ge2 = {ge, ge};
    f2 hs = {Hs[0], Hs[1]};
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const int term = k / NR;
        const int r = k % NR;
        const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
        const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(BOPS[ib]), acc, 0, 0, 0);
#pragma unroll
        for (int cp = (8 * k) / NM; cp < (8 * (k + 1)) / NM; ++cp) {
            f2 M = pk_add(hs, m2[cp]);                            // max_k o[y-1,x-1,k] + m      (cext.c:192-222)
            if (LOCAL) { M.x = __builtin_fmaxf(M.x, 0.0f); M.y = __builtin_fmaxf(M.y, 0.0f); }  // cext.c:208-209
            const f2 Mo = pk_add(M, go2);
            const f2 U = {Uc[2 * cp], Uc[2 * cp + 1]};
            const f2 Ug = pk_add(U, ge2);
            const float H0 = max3f(M.x, U.x, lrun);
            lrun = __builtin_fmaxf(Mo.x, lrun + ge);      // L[y][x+1]   (cext.c:169-183,276-283)
            const float H1 = max3f(M.y, U.y, lrun);
            lrun = __builtin_fmaxf(Mo.y, lrun + ge);
            if (LOCAL) best_run = max3f(best_run, H0, H1);
            Uc[2 * cp] = __builtin_fmaxf(Mo.x, Ug.x);        // U[y+1][x]   (cext.c:152-166,247-254)
            Uc[2 * cp + 1] = __builtin_fmaxf(Mo.y, Ug.y);
            if (cp < 7) { hs.x = Hs[2 * cp + 2]; hs.y = Hs[2 * cp + 3]; }
            Hs[2 * cp + 1] = H0;
            Hs[2 * cp + 2] = H1;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    PREV = acc;
    // hand the last column to the next strip: through LDS (the next wave reads it after the next barrier), wave 3
    // through memory (wave 0 fetches it a round later)
    if (to_memory) {
        if (h) {
            const unsigned long long v = (unsigned long long)__float_as_uint(Hs[16]) | ((unsigned long long)__float_as_uint(lrun) << 32);
            __builtin_nontemporal_store(v, reinterpret_cast<unsigned long long *>(bnd_st));
        }
    } else {
        if (h) *reinterpret_cast<float2 *>(bnd_out + (K & 3) * 256) = make_float2(Hs[16], lrun);   // row yy = u: slot u & 3
    }
    bnd_st += 256;
    pipe_issue(dma);
    if (has_bnd_dma) pipe_issue_bnd(dma);
    dH = bv.x;
    hd_x = from_lower_half(hd_out);
    l_x = from_lower_half(lrun);
    if (semiglobal && sp.last_owner) col_run = __builtin_fmaxf(col_run, select16s(Hs, cidx));
    if (may_snap && have_pair && yy == L1) {
        if constexpr (LOCAL) asm volatile("");   // keep this a branch (see split16_step, SNAPBR)
        if (LOCAL) out_best = best_run;
        if (semiglobal) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                out_rowmax = __builtin_fmaxf(out_rowmax, (sp.xb + c + 1 <= sp.L2) ? Hs[c + 1] : PRALINE_NEG_INF);
            out_colmax = col_run;
        }
        if (sp.last_owner) out_corner = select16s(Hs, cidx);
    }
}

// a step of a wave that has no strip (lead-in, the last round's spare waves): its share of the operand stream only
__device__ __forceinline__ void pipe_idle_step(bool wait4, PipeDma &dma, bool has_bnd_dma)
{
    pipe_wait(wait4, false);
    PRALINE_PIPE_BARRIER();
    pipe_issue(dma);
    if (has_bnd_dma) pipe_issue_bnd(dma);
}

template <int NR, int NTERM, bool LOCAL>
__global__ __launch_bounds__(256, 2) void k_dp_pipe(Arena16Dev ar, const PipeItem *__restrict__ items, const WaveTask *__restrict__ tasks,
                                                    const int32_t *__restrict__ set_one, const int32_t *__restrict__ lane_pair,
                                                    float2 *bnd, float *__restrict__ scores, RunParams rp)
{
    __shared__ __attribute__((aligned(16))) char lds[pipe_lds_bytes()];
    char *ring = lds;
    const int rank = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform (SGPR)
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    char *atile = lds + pipe_ring_bytes() + rank * 4096;
    char *bin_all = lds + pipe_ring_bytes() + 4 * 4096;          // boundary rings: [wave][4 rows][32 pairs] float2
    float *res = reinterpret_cast<float *>(lds + pipe_ring_bytes() + 4 * 4096 + 4 * 1024);   // [task][2][32]
    const PipeItem it = items[blockIdx.x];
    const int rsteps = it.rsteps, nrounds = it.nrounds, nstrips_all = it.nstrips;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    const float go = rp.go1, ge = rp.ge1;
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);
    const float h00 = max3f(0.0f, o001, o002);

    // defined LDS contents before the first DMA / hand-off; results start at -inf
    for (int i = threadIdx.x * 16; i < pipe_ring_bytes() + 4 * 4096 + 4 * 1024; i += 256 * 16)
        *reinterpret_cast<float4 *>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < PRALINE_PIPE_MAX_TASKS * 2 * 32; i += 256) res[i] = PRALINE_NEG_INF;
    __syncthreads();

    const int my_one = set_one[it.set * 32 + j];
    const bool have_one = my_one >= 0;
    const int L1 = have_one ? ar.len[my_one] : 0;
    int min_l1 = have_one ? L1 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) min_l1 = min(min_l1, __shfl_xor(min_l1, off));
    min_l1 = __builtin_amdgcn_readfirstlane(min_l1);

    // ---- this wave's share of the operand stream: piece `rank` = pairs 8 rank .. 8 rank + 7 of every row ----
    PipeDma dma;
    unsigned stage_rd[4];
    {
        constexpr int C = 8;                       // 16-byte chunks per row
        const int p = rank * 8 + lane / C;         // the pair whose row this lane fetches
        const int one_p = set_one[it.set * 32 + p];
        const unsigned row0 = one_p >= 0 ? (unsigned)ar.row_off[one_p] : 0u;
        const unsigned chunk = ((unsigned)lane % C) ^ stage_swz<C>((unsigned)p);   // = hh * 4 + slot
        dma.gofs = row0 * 128u + chunk * 16u;      // (rows hold [hh][4 slots]: memory chunk == chunk)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            stage_rd[q] = (unsigned)j * 128u + (((unsigned)(h * 4 + q)) ^ stage_swz<C>((unsigned)j)) * 16u;
        const unsigned long long pb = reinterpret_cast<unsigned long long>(ar.P16);
        dma.src = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32)) << 32) |
                  (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb);
        const unsigned ring_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)ring);
        dma.ring_lo = ring_addr + (unsigned)rank * 1024u;
        dma.ring_hi = dma.ring_lo + (unsigned)pipe_ring_bytes();
        dma.dst = dma.ring_lo;
        dma.left = (unsigned)rsteps;
        dma.rsteps = (unsigned)rsteps;
        const unsigned long long bb = reinterpret_cast<unsigned long long>(bnd + it.bnd_off);
        dma.bsrc = (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32)) << 32) |
                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb)) + 256ull;   // row 1
        dma.bdst_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)bin_all);            // wave 0's ring
        dma.brow = 1;
        dma.bgofs = (unsigned)lane * 4u;
    }
    const bool wait4 = rank == 0 || rank == 3;     // two memory operations per step (piece + boundary DMA / store)
    const bool has_bnd_dma = rank == 0;
    const bool to_memory = rank == 3;
    const char *bnd_in = bin_all + rank * 1024 + j * 8;                   // rows handed to this wave
    char *bnd_out = bin_all + ((rank + 1) & 3) * 1024 + j * 8;            // rows this wave hands on (waves 0..2)
    char *bnd_col = reinterpret_cast<char *>(bnd + it.bnd_off + j);       // wave 3: the wrap-around column, float2 [row][32]

    // positions 0..4 of the stream (and wave 0's boundary rows 1..4: garbage for the first round - its first strip is
    // a task's first strip - but the ring's accounting starts here); everything lands before the first barrier
#pragma unroll
    for (int i = 0; i < 5; ++i) pipe_issue(dma);
    if (has_bnd_dma) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pipe_issue_bnd(dma);
    }
    PRALINE_VMCNT(0);
    PRALINE_PIPE_BARRIER();
    for (int i = 0; i < PRALINE_PIPE_LAG * rank; ++i) pipe_idle_step(wait4, dma, has_bnd_dma);

    // ---- this wave's strips ----
    int ti = 0, s = rank;          // task (inside the item) and strip of this wave's current strip
    int cidx = 0;
    float4 aop[4], b0[4], b1[4];
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 accB = accA;
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const unsigned a_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)atile);
    const unsigned a_gofs = (unsigned)acol * 128u + (unsigned)h * 64u;
    bool started = false;

    for (int round = 0; round < nrounds; ++round) {
        const int q = 4 * round + rank;
        if (q >= nstrips_all) {   // no strip left for this wave (last round only)
            for (int u = 0; u < rsteps; ++u) pipe_idle_step(wait4, dma, has_bnd_dma);
            continue;
        }
        while (s >= tasks[it.task0 + ti].nstrips) { s -= tasks[it.task0 + ti].nstrips; ++ti; }
        const WaveTask tk = tasks[it.task0 + ti];
        const int two = tk.two[0];
        const int L2 = ar.len[two];
        const int nstrips = tk.nstrips;
        const int clast = (L2 - 1) & 31;
        const bool own_last = (clast >> 4) == h;
        const int x0 = s * 32;
        PipeStrip sp;
        sp.first = s == 0;
        sp.last_owner = (s == nstrips - 1) && own_last;
        sp.xb = x0 + 16 * h;
        sp.L2 = L2;
        cidx = clast & 15;
        asm volatile("" : "+v"(cidx));
        const bool have_pair = have_one && lane_pair[(it.task0 + ti) * 32 + j] >= 0;

        if (!started) {
            // pipeline prologue of the wave's first strip: A tile straight from memory, operand rows 1 and 2 from the
            // ring (positions 0 and 1 have landed: initial barrier / lead-in steps), scores of row 1
            started = true;
            const float4 *sa = reinterpret_cast<const float4 *>(ar.Q16 + ((int64_t)ar.row_off[two] + acol + x0) * ar.row_bytes + h * ar.half_bytes);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) aop[qq] = sa[qq];
            float4 br1[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                br1[qq] = *reinterpret_cast<const float4 *>(ring + 0 * 4096 + stage_rd[qq]);
                b0[qq] = *reinterpret_cast<const float4 *>(ring + 1 * 4096 + stage_rd[qq]);
            }
            // (the compiler's own vmcnt for the A tile also drains this wave's DMA pieces: once per wave)
#pragma unroll
            for (int k = 0; k < NTERM * NR; ++k) {
                const int term = k / NR;
                const int r = k % NR;
                const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
                const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(br1[ib]), accA, 0, 0, 0);
            }
        }
        // (later strips: the previous round's last step already used this strip's A tile - accA holds row 1's scores,
        // b0 row 2's operands)

        float Hs[17], Uc[16];
        Hs[0] = PRALINE_NEG_INF;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Hs[c + 1] = boundary_value(sp.xb + c + 1, go, ge, free_two);   // H[0][x] = o[0,x,2]
            Uc[c] = PRALINE_NEG_INF;                                        // U[1][x]
        }
        float dH = (s == 0) ? h00 : boundary_value(x0, go, ge, free_two);
        float hd_x = PRALINE_NEG_INF, l_x = PRALINE_NEG_INF;
        // per-strip results (folded into the task's table at the end of the strip); the column-0 / row-0 cells and
        // o[0,0] enter at finalisation
        float out_best = PRALINE_NEG_INF, out_rowmax = PRALINE_NEG_INF, out_colmax = PRALINE_NEG_INF, out_corner = PRALINE_NEG_INF;
        float best_run = LOCAL ? h00 : PRALINE_NEG_INF;
        float col_run = PRALINE_NEG_INF;
        char *bnd_st = bnd_col;   // upper half stores row yy = u (row 0: dummy)

        // the next strip of this wave (strip q + 4): its A tile is fetched into LDS during the last iteration
        int ti_n = ti, s_n = s + 4;
        const bool more = q + 4 < nstrips_all;
        if (more) {
            while (s_n >= tasks[it.task0 + ti_n].nstrips) { s_n -= tasks[it.task0 + ti_n].nstrips; ++ti_n; }
        }

#define PRALINE_PIPE_STEP(KK, CURA, PREVA, BUSE, BFIL)                                                                 \
        pipe_step<NR, NTERM, LOCAL, KK>(u0 + KK, L1, have_pair, h, CURA, PREVA, BUSE, BFIL, aop, ring, stage_rd, bnd_in, bnd_out, \
                                        bnd_st, to_memory, wait4, dma, has_bnd_dma, Hs, Uc, dH, hd_x, l_x, best_run, col_run, \
                                        out_best, out_rowmax, out_colmax, out_corner, go, ge, free_one, semiglobal, cidx, sp, \
                                        (u0 + KK + 1) >= min_l1, a_fetch)
        for (int u0 = 0; u0 < rsteps; u0 += 12) {
            const bool last_it = u0 + 12 >= rsteps;
            const bool a_fetch = last_it && more;
            if (u0 == 0) {
                // step 0: only the lower half has a row (row 1); the upper half's garbage is undone right after
                float Hsave[17];
#pragma unroll
                for (int c = 0; c < 17; ++c) Hsave[c] = Hs[c];
                const float best_s = best_run, col_s = col_run;
                PRALINE_PIPE_STEP(0, accA, accB, b0, b1);
                if (h) {
#pragma unroll
                    for (int c = 0; c < 17; ++c) Hs[c] = Hsave[c];
#pragma unroll
                    for (int c = 0; c < 16; ++c) Uc[c] = PRALINE_NEG_INF;
                    best_run = best_s;
                    col_run = col_s;
                }
            } else {
                PRALINE_PIPE_STEP(0, accA, accB, b0, b1);
            }
            if (a_fetch) {
                // A tile of strip q + 4 (rows x0' .. x0' + 31 of its sequence two, this lane's 64 bytes) -> LDS
                const unsigned long long qa = reinterpret_cast<unsigned long long>(ar.Q16) +
                                              ((unsigned long long)ar.row_off[tasks[it.task0 + ti_n].two[0]] + (unsigned long long)(s_n * 32)) * 128ull;
                const unsigned long long qs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(qa >> 32)) << 32) |
                                              (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)qa);
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    unsigned keep;
                    const unsigned go_ = a_gofs + 16u * qq, dst = a_lds + 1024u * qq;
                    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                                 "global_load_lds_dwordx4 %1, %2\n\t"
                                 : "=&s"(keep)
                                 : "v"(go_), "s"(qs), "s"(dst)
                                 : "memory");
                }
            }
            PRALINE_PIPE_STEP(1, accB, accA, b1, b0);
            PRALINE_PIPE_STEP(2, accA, accB, b0, b1);
            PRALINE_PIPE_STEP(3, accB, accA, b1, b0);
            PRALINE_PIPE_STEP(4, accA, accB, b0, b1);
            PRALINE_PIPE_STEP(5, accB, accA, b1, b0);
            PRALINE_PIPE_STEP(6, accA, accB, b0, b1);
            PRALINE_PIPE_STEP(7, accB, accA, b1, b0);
            PRALINE_PIPE_STEP(8, accA, accB, b0, b1);
            PRALINE_PIPE_STEP(9, accB, accA, b1, b0);
            PRALINE_PIPE_STEP(10, accA, accB, b0, b1);
            if (a_fetch) {
                // the last step's MFMAs compute row 1 of the NEXT strip: switch to its A tile (fetched >= 10 steps ago:
                // every per-step wait since has covered it)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) aop[qq] = *reinterpret_cast<const float4 *>(atile + 1024 * qq + lane * 16);
            }
            PRALINE_PIPE_STEP(11, accB, accA, b1, b0);
        }
#undef PRALINE_PIPE_STEP

        // ---- fold this strip's share into the task's results ----
        {
            const float corner_all = __builtin_fmaxf(out_corner, partner_value(out_corner, h));
            const float rowmax_all = __builtin_fmaxf(out_rowmax, partner_value(out_rowmax, h));
            const float colmax_all = __builtin_fmaxf(out_colmax, partner_value(out_colmax, h));
            const float best_all = __builtin_fmaxf(out_best, partner_value(out_best, h));
            if (h == 0) {
                float *r0 = res + ti * 64 + j;
                const float v0 = LOCAL ? best_all : (semiglobal ? rowmax_all : corner_all);
                if (v0 != PRALINE_NEG_INF) atomicMax(r0, v0);
                if (semiglobal && colmax_all != PRALINE_NEG_INF) atomicMax(r0 + 32, colmax_all);
            }
        }
        s += 4;
    }
    // the other waves are up to 6 steps behind
    for (int i = 0; i < PRALINE_PIPE_LAG * (3 - rank); ++i) {
        PRALINE_PIPE_BARRIER();
    }
    PRALINE_VMCNT(0);   // no DMA may be in flight when the wave ends
    __syncthreads();

    // ---- scores: wave r finalises the tasks r, r + 4, ... ----
    for (int t = rank; t < it.ntasks; t += 4) {
        if (h != 0 || !have_one) continue;
        const int pair = lane_pair[(it.task0 + t) * 32 + j];
        if (pair < 0) continue;
        const int L2 = ar.len[tasks[it.task0 + t].two[0]];
        const float v0 = res[t * 64 + j], v1 = res[t * 64 + 32 + j];
        float score;
        if (LOCAL) score = __builtin_fmaxf(v0, h00);
        else if (semiglobal) {
            // o[L1,0,1] and o[0,L2,2] are the column-0 / row-0 members of the last row / last column (align.py:406-424)
            const float rowmax = __builtin_fmaxf(v0, boundary_value(L1, go, ge, free_one));
            const float colmax = __builtin_fmaxf(v1, boundary_value(L2, go, ge, free_two));
            score = (rowmax > colmax && free_two) ? rowmax : colmax;
        } else score = v0;
        scores[pair] = score;
    }
}
