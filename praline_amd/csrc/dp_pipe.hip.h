// dp_pipe.hip.h -- k_dp_pipe: the scores-only fill as a PIPELINE of the four waves of a workgroup.
//
// What round 3's measurements said about k_dp_split16 on float profiles (scripts/exp_ablate16.py, scripts/micro/
// step_cost.hip, issue_cost.hip):
//   * a saturated launch is bound by its memory instructions, not by the vector ALU - every wave streams its own 32
//     operand rows (4 KiB per step: 11.5 TB/s of L2 -> LDS traffic at 2.9 TCUPS, the measured ceiling of that gather
//     pattern) and round-trips the strip-boundary column through memory (16 % of the launch);
//   * a wave issues ONE instruction of any kind per ~5.5 cycles, and a scalar branch costs it ~25: what a step costs
//     is its instruction count, whatever the pipes do.
//
// Here the four waves of a workgroup work on tasks that share ONE set of 32 sequences one (PipeItem, sched.cpp):
//   * wave r sweeps the strips r, r + 4, r + 8, ... of the item's concatenated strip list (the strips of task 0, then
//     of task 1, ...), PRALINE_PIPE_LAG = 2 steps behind wave r - 1: all four are within 6 rows of each other;
//   * the set's operand rows are therefore streamed ONCE per workgroup and step: wave r fetches a quarter of the row
//     (pairs 8 r .. 8 r + 7, one 1 KiB LDS-DMA) six steps ahead into a 12-row ring that all four waves read
//     (a quarter of the bytes and of the DMA instructions per wave);
//   * a strip's boundary column (H, L of its last column, row by row) is handed to the wave of the next strip through
//     a 12-row LDS ring; only the hand-off from wave 3 to wave 0's next strip goes through memory (`bnd`, one round
//     = rsteps - 6 steps later), four rows (1 KiB) per store / DMA; a task's first strip reads the analytic column 0
//     from `analytic` (written once per launch by k_pipe_analytic) the same way;
//   * the waves meet at one s_barrier per step: it orders ring writes (every wave waits for its own DMA piece of the
//     row four steps after issuing it), ring reuse and the boundary hand-off;
//   * the step itself is branch-free apart from one wave-uniform test every fourth step (boundary blocks) and the
//     snapshot test, which only exists in the iterations that can contain a sequence's last row (SNAP).
// The arithmetic of a cell is split16_step's (dp_split16.hip.h) instruction for instruction - scores are bit-identical
// to k_dp_split16 (tests/test_gpu_parity.py::test_pipeline_workgroups_agree_bitwise).
//
// Step u (0-based) of a round: lower half DP row u + 1, upper half row u.  A round has rsteps = 12 k >= max_l1 + 1
// steps; stream position p = round * rsteps + u holds the operand row u + 1 of every sequence of the set and lives in
// ring slot p % 12 = u % 12: static per unrolled step, the same for every wave.  Boundary rows live in slot row % 12 of
// the consumer's ring.
//
// LDS per workgroup (80 KiB: two workgroups per CU): [12 x 4 KiB operand ring][4 x 4 KiB A tiles of the next strips]
// [4 x 3 KiB boundary rings][3 KiB wave 3's outgoing rows][1 KiB results].
#pragma once
#include "dp_split16.hip.h"

__host__ __device__ constexpr int pipe_ring_bytes() { return PRALINE_PIPE_RING * 4096; }
__host__ __device__ constexpr int pipe_bring_bytes() { return 12 * 256; }
__host__ __device__ constexpr int pipe_res_bytes() { return PRALINE_PIPE_MAX_TASKS * 2 * 32 * 4; }
__host__ __device__ constexpr int pipe_lds_bytes()
{
    return pipe_ring_bytes() + 4 * 4096 + 5 * pipe_bring_bytes() + pipe_res_bytes();
}
static_assert(pipe_lds_bytes() <= 80 * 1024, "two pipeline workgroups per CU");

// barrier of the pipeline: every LDS access of this wave (the hand-off write above all) has completed before it
#define PRALINE_PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PRALINE_PIPE_AHEAD 6   // the operand stream runs this many positions ahead of wave 0's step

// KEEP: the pipeline as the FORWARD fill of the two-pass alignments with paths (dp_trace2.hip.h; global mode).  Besides its
// (H, L) hand-off a strip writes what k_trace_recompute starts from, in the layout of k_dp_split16<..., KEEP>:
//   * its last column as three states, float4 (M, U, L, 0) [strip + 1][row][32 pairs] per task at tk.aux_off - the
//     recurrence's own intermediates (M before the maximum, U and L on entry to the cell), bit for bit the values the
//     three-state kernels carry: fl(max3(Mp, Up, Lp) + m) = max3 of the three rounded sums;
//   * the (M, U, L) states of every PRALINE_KEEP_BH-th row, float4 [row / BH][3][4][64] per strip at tk.tb_off.  BH is a
//     multiple of 12, so the two steps per block that hold such a row (lower half: u = 11 mod 12, upper half: u = 0 mod 12)
//     are static positions of the 12x unrolled loop: only they carry the (wave-uniform) test;
//   * M and U of the corner cell (the end state k is the first of M, U, L that equals the score, align.py:428-430).
static_assert(PRALINE_KEEP_BH % 12 == 0, "checkpoint rows sit at static positions of the unrolled loop");
// checkpoint blocks per strip: the rounds compute rows up to rsteps <= max(max_l1 + 12, PRALINE_PIPE_MIN_STEPS); block 0 is never written
__host__ __device__ constexpr int pipe_keep_blocks(int max_l1)
{
    return ((max_l1 + 12 > PRALINE_PIPE_MIN_STEPS) ? max_l1 + 12 : PRALINE_PIPE_MIN_STEPS) / PRALINE_KEEP_BH + 1;
}
#ifndef PRALINE_PIPE_KEEP_ABLATE
#define PRALINE_PIPE_KEEP_ABLATE 0   // timing experiments only: 1 no checkpoint stores, 2 no kept columns
#endif
struct PipeKeep {
    char *st = nullptr;        // kept column being written: this lane's float4 of the row this step stores
    f4n *ckpt = nullptr;       // this strip's checkpoint blocks (+ lane)
    float snap_m = 0.0f, snap_u = 0.0f;
};
#ifndef PRALINE_PIPE_ABLATE
#define PRALINE_PIPE_ABLATE 0   // timing experiments only (results invalid): 1 no half select, 2 four more MFMAs per step
#endif

// the analytic column 0 of a task - (o[y,0,1], -inf) for every pair - as float2 [rows][32]: what a task's FIRST strip
// reads as its boundary column (praline/component/align.py:371-376)
__global__ void k_pipe_analytic(float2 *col, int rows, RunParams rp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = i >> 5;
    if (y >= rows) return;
    col[i] = make_float2(y >= 1 ? boundary_value(y, rp.go1, rp.ge1, mode_free_one(rp.mode)) : 0.0f, PRALINE_NEG_INF);
}
// the same column as three states, float4 (-inf, o[y,0,1], -inf, 0) [rows][32]: what k_trace_recompute reads as the boundary
// column of every task's first strip behind the KEEP forward fill
__global__ void k_pipe_analytic4(float4 *col, int rows, RunParams rp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = i >> 5;
    if (y >= rows) return;
    col[i] = make_float4(PRALINE_NEG_INF, y >= 1 ? boundary_value(y, rp.go1, rp.ge1, mode_free_one(rp.mode)) : 0.0f, PRALINE_NEG_INF, 0.0f);
}

struct PipeDma {
    unsigned long long src;    // wave-uniform: P16 + (row of the next position to fetch) * row bytes
    unsigned dst;              // LDS byte address of this wave's piece in the ring slot of that position
    unsigned left;             // positions left in the round before the row cursor wraps
    unsigned rsteps;
    unsigned ring_lo, ring_hi; // ring bounds for this wave's piece (dst wraps from ring_hi to ring_lo)
    unsigned gofs;             // per-lane byte offset of the piece (VGPR)
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
// four boundary rows (1 KiB: float2 [4][32]) from memory into a boundary ring
__device__ __forceinline__ void pipe_issue_block(unsigned long long src, unsigned dst, unsigned lane16)
{
    unsigned keep;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 : "=&s"(keep)
                 : "v"(lane16), "s"(src), "s"(dst)
                 : "memory");
}

// Boundary traffic of a wave, per strip (wave-uniform).
struct PipeBnd {
    unsigned long long in_src;   // next block of the incoming column to fetch (analytic column or the wrap-around column)
    bool in_dma;                 // this strip's boundary column comes from memory (wave 0, or the task's first strip)
    unsigned long long out_dst;  // wave 3: next block of the wrap-around column to store
    bool out_mem;                // wave 3: rows go to memory (four at a time)
    unsigned ring_addr;          // LDS address of this wave's incoming ring
    const char *out_stage;       // wave 3: its outgoing rows in LDS (+ lane * 16)
};

// One step.  K = u % 12.  Register roles as in split16_step (BSRC = 2 without DM):
//   CUR row u + 1's scores, PREV row u's (receives row u + 2's); BOPS operands of row u + 2, BFILL receives row u + 3.
// KEEP: 0 scores only; 1 kept column (+ corner snapshot); 2 the same with the checkpoint stores of a row u + 1 - h = 0 (mod BH)
template <int NR, int NTERM, bool LOCAL, bool SEMI, int K, bool SNAP, int KEEP = 0>
__device__ __forceinline__ void pipe_step(int u, int L1, bool have_pair, int h, const f32x16 &CUR, f32x16 &PREV,
                                          const float4 (&BOPS)[4], float4 (&BFILL)[4], const float4 (&aop)[4],
                                          const char *ring, const unsigned (&stage_rd)[4], const char *bnd_in, char *bnd_out,
                                          bool wr_lane, PipeDma &dma, PipeBnd &pb, unsigned lane16,
                                          float (&Hs)[17], float (&Uc)[16], float &dH, float &hd_x, float &l_x,
                                          float &best_run, float &col_run, float &out_best, float &out_rowmax, float &out_colmax,
                                          float &out_corner, float go, float ge, int cidx, bool last_owner, int xb, int L2,
                                          PipeKeep *ks = nullptr, bool ck_near = false)
{
    static_assert(NR == 2 && (NTERM == 2 || NTERM == 3), "k_dp_pipe is built for the 128-byte operand rows of float-profile arenas");
    static_assert(KEEP == 0 || (!LOCAL && !SEMI), "the kept-state forward fill serves global alignments");
    constexpr int NM = NTERM * NR;
    // every memory operation of this wave but the three youngest has completed: its pieces of the rows read below (issued
    // >= 4 steps ago; a block DMA / store or the A-tile fetch among the youngest only makes the wait stricter), then
    // the barrier: the other waves' pieces too, and the previous step's boundary hand-off
    // KEEP: every step - idle steps included - issues exactly one kept-column store in front of its operand DMA, so six
    // operations are younger than the piece issued four steps ago (vmcnt counts loads and stores together, in order: with
    // the scores kernel's vmcnt(3) every step would wait for the stores of the last two steps to reach memory)
    // The 12 checkpoint stores of a step (KEEP == 2, at K = 11 and at K = 0 of the next iteration - the same wave-uniform
    // condition ck_near: row u0 = 0 mod BH) are also younger than the pieces the next steps wait for: waiting for all but 6
    // would park the wave until those stores have reached memory (measured: +0.7 ms on C2).  Operations younger than the
    // piece of step u - 4 when ck_near: K = 0: 2 + 2 + 14; K = 1, 2: 2 + 14 + 14 (the order varies); K = 3: 14 + 2 + 2.
    if constexpr (KEEP != 0) {
        constexpr int WBIG = (K == 0 || K == 3) ? 18 : 30;
        if constexpr (K <= 3) {
            if (ck_near) PRALINE_VMCNT(WBIG); else PRALINE_VMCNT(6);
        } else PRALINE_VMCNT(6);
    } else PRALINE_VMCNT(3);
    PRALINE_PIPE_BARRIER();
    // boundary column of row u + 1: (H[y][x0], L[y][x0 + 1]); u = K (mod 12), so every ring slot is static
    const float2 bv = *reinterpret_cast<const float2 *>(bnd_in + ((K + 1) % 12) * 256);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        BFILL[q] = *reinterpret_cast<const float4 *>(ring + ((K + 2) % PRALINE_PIPE_RING) * 4096 + stage_rd[q]);
    // the half select does not depend on the boundary value: it covers the LDS latency of the read above
    f2 m2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#if PRALINE_PIPE_ABLATE & 1
        m2[c].x = CUR[2 * c];
        m2[c].y = CUR[2 * c + 1];
#else
        m2[c].x = h ? PREV[2 * c] : CUR[2 * c];
        m2[c].y = h ? PREV[2 * c + 1] : CUR[2 * c + 1];
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    Hs[0] = h ? hd_x : dH;
    float lrun = h ? l_x : bv.y;
    const float hd_out = Hs[16];
    const f2 go2 = {go, go}, ge2 = {ge, ge};
    f2 hs = {Hs[0], Hs[1]};
    bool ck_mine = false;
    f4n ckM, ckU, ckL;         // KEEP == 2: the states of four columns, stored once complete
    float kM = 0.0f, kU = 0.0f, kL = 0.0f;   // KEEP: states of this lane's last column
    if constexpr (KEEP != 0) {
        const int yy = u + 1 - h;
        if constexpr (KEEP == 2) ck_mine = yy >= PRALINE_KEEP_BH && (yy % PRALINE_KEEP_BH) == 0;
        if constexpr (SNAP) {
            if (have_pair && yy == L1 && last_owner) {   // the corner cell is in this row: its M and U (split16_step, KEEP)
                float hw[16], mw[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) { hw[c] = Hs[c]; mw[c] = (c & 1) ? m2[c >> 1].y : m2[c >> 1].x; }
                ks->snap_m = select16(hw, cidx) + select16(mw, cidx);
                ks->snap_u = select16(Uc, cidx);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        const int term = k / NR;
        const int r = k % NR;
        const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
        const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(BOPS[ib]), acc, 0, 0, 0);
#if PRALINE_PIPE_ABLATE & 2
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ib]), as_half8(BOPS[ia]), acc, 0, 0, 0);
#endif
#pragma unroll
        for (int cp = (8 * k) / NM; cp < (8 * (k + 1)) / NM; ++cp) {
            f2 M = pk_add(hs, m2[cp]);                            // max_k o[y-1,x-1,k] + m      (cext.c:192-222)
            if (LOCAL) { M.x = __builtin_fmaxf(M.x, 0.0f); M.y = __builtin_fmaxf(M.y, 0.0f); }  // cext.c:208-209
            const f2 Mo = pk_add(M, go2);
            const f2 U = {Uc[2 * cp], Uc[2 * cp + 1]};
            const f2 Ug = pk_add(U, ge2);
            const float H0 = max3f(M.x, U.x, lrun);
            const float lin0 = lrun;
            lrun = __builtin_fmaxf(Mo.x, lrun + ge);      // L[y][x+1]   (cext.c:169-183,276-283)
            const float H1 = max3f(M.y, U.y, lrun);
            if constexpr (KEEP != 0) {
                if (cp == 7) { kM = M.y; kU = U.y; kL = lrun; }
            }
            if constexpr (KEEP == 2) {
                if ((cp & 1) == 0) { ckM.x = M.x; ckM.y = M.y; ckU.x = U.x; ckU.y = U.y; ckL.x = lin0; ckL.y = lrun; }
                else {
                    ckM.z = M.x; ckM.w = M.y; ckU.z = U.x; ckU.w = U.y; ckL.z = lin0; ckL.w = lrun;
                    if (ck_mine && !(PRALINE_PIPE_KEEP_ABLATE & 1)) {
                        f4n *q = ks->ckpt + (int64_t)((u + 1 - h) / PRALINE_KEEP_BH) * PRALINE_CKPT_BLOCK_F4 + (cp >> 1) * 64;
                        __builtin_nontemporal_store(ckM, q);
                        __builtin_nontemporal_store(ckU, q + 4 * 64);
                        __builtin_nontemporal_store(ckL, q + 8 * 64);
                    }
                }
            }
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
    // hand row yy = u of the last column to the next strip (ring slot u % 12 of the next wave; wave 3: its own staging
    // ring; lanes of the lower half and strips that end a task write nothing)
    if (wr_lane) *reinterpret_cast<float2 *>(bnd_out + (K % 12) * 256) = make_float2(Hs[16], lrun);
    if constexpr (KEEP != 0) {
        // (M, U, L) of the cell (u, last column): what the NEXT strip's recompute reads as its boundary column
        // (the upper half of EVERY strip stores - a task's last strip into the spare column behind it - so that the count of
        // memory operations per step is the same for all waves: see the vmcnt above)
        if (h) {
            const f4n kv = {kM, kU, kL, 0.0f};
            __builtin_nontemporal_store(kv, reinterpret_cast<f4n *>(ks->st));
        }
        ks->st += 32 * sizeof(float4);
    }
    pipe_issue(dma);
    if constexpr ((K & 3) == 3) {
        // every fourth step: the next four rows of a boundary column that lives in memory -
        //   incoming (wave 0, or a task's first strip): rows u + 5 .. u + 8 -> ring slots (K + 5) % 12 ..
        //   outgoing (wave 3): rows u - 3 .. u, just completed in the staging ring
        if (pb.in_dma) {
            pipe_issue_block(pb.in_src, pb.ring_addr + ((K + 5) % 12) * 256, lane16);
            pb.in_src += 1024;
        }
        if (pb.out_mem) {
            const f4n v = *reinterpret_cast<const f4n *>(pb.out_stage + (K - 3) * 256);
            __builtin_nontemporal_store(v, reinterpret_cast<f4n *>(pb.out_dst + lane16));
            pb.out_dst += 1024;
        }
    }
    dH = bv.x;
    hd_x = from_lower_half(hd_out);
    l_x = from_lower_half(lrun);
    if constexpr (SEMI) {
        // (a branch over 16 instructions: the strips that own a last column are one in ~13)
        if (last_owner) col_run = __builtin_fmaxf(col_run, select16s(Hs, cidx));
    }
    if constexpr (SNAP) {
        if (have_pair && u + 1 - h == L1) {
            if constexpr (LOCAL) asm volatile("");   // keep this a branch (see split16_step, SNAPBR)
            if (LOCAL) out_best = best_run;
            if (SEMI) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    out_rowmax = __builtin_fmaxf(out_rowmax, (xb + c + 1 <= L2) ? Hs[c + 1] : PRALINE_NEG_INF);
                out_colmax = col_run;
            }
            if (last_owner) out_corner = select16s(Hs, cidx);
        }
    }
}

// a step of a wave that has no strip (lead-in, the last round's spare waves): its share of the operand stream only
template <bool KEEP = false>
__device__ __forceinline__ void pipe_idle_step(PipeDma &dma, char *dummy = nullptr, int h = 0)
{
    if constexpr (KEEP) PRALINE_VMCNT(6); else PRALINE_VMCNT(3);
    PRALINE_PIPE_BARRIER();
    if constexpr (KEEP) {
        // the store a real step issues in front of its DMA (pipe_step): same operation count per step for every wave
        if (h) {
            const f4n z = {0.0f, 0.0f, 0.0f, 0.0f};
            __builtin_nontemporal_store(z, reinterpret_cast<f4n *>(dummy));
        }
    }
    pipe_issue(dma);
}

template <int NR, int NTERM, bool LOCAL, bool SEMI, bool KEEP = false>
__global__ __launch_bounds__(256, 2) void k_dp_pipe(Arena16Dev ar, const PipeItem *__restrict__ items, const WaveTask *__restrict__ tasks,
                                                    const int32_t *__restrict__ set_one, const int32_t *__restrict__ lane_pair,
                                                    float2 *bnd, const float2 *__restrict__ analytic, float *__restrict__ scores,
                                                    RunParams rp, float4 *keep_bnd = nullptr, float *ckpt = nullptr,
                                                    int32_t *__restrict__ end_cells = nullptr)
{
    static_assert(!(LOCAL && SEMI), "one mode at a time");
    static_assert(!KEEP || (!LOCAL && !SEMI), "the kept-state forward fill serves global alignments");
    __shared__ __attribute__((aligned(16))) char lds[pipe_lds_bytes()];
    char *ring = lds;
    const int rank = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform (SGPR)
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    char *atile = lds + pipe_ring_bytes() + rank * 4096;
    char *bring_all = lds + pipe_ring_bytes() + 4 * 4096;          // boundary rings [wave 0..3][12 rows][32] float2, then wave 3's outgoing rows
    float *res = reinterpret_cast<float *>(bring_all + 5 * pipe_bring_bytes());   // [task][2][32]
    const PipeItem it = items[blockIdx.x];
    const int rsteps = it.rsteps, nrounds = it.nrounds, nstrips_all = it.nstrips;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const float go = rp.go1, ge = rp.ge1;
    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);
    const float h00 = max3f(0.0f, o001, o002);

    // defined LDS contents before the first DMA / hand-off; results start at -inf
    for (int i = threadIdx.x * 16; i < pipe_ring_bytes() + 4 * 4096 + 5 * pipe_bring_bytes(); i += 256 * 16)
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

    auto uniform64 = [](unsigned long long v) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    };
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
        dma.src = uniform64(reinterpret_cast<unsigned long long>(ar.P16));
        const unsigned ring_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)ring);
        dma.ring_lo = ring_addr + (unsigned)rank * 1024u;
        dma.ring_hi = dma.ring_lo + (unsigned)pipe_ring_bytes();
        dma.dst = dma.ring_lo;
        dma.left = (unsigned)rsteps;
        dma.rsteps = (unsigned)rsteps;
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned long long wrap_col = uniform64(reinterpret_cast<unsigned long long>(bnd + it.bnd_off));      // float2 [rsteps + 16][32]
    const unsigned long long analytic_col = uniform64(reinterpret_cast<unsigned long long>(analytic));
    PipeBnd pb;
    pb.ring_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(bring_all + rank * pipe_bring_bytes()));
    pb.out_mem = rank == 3;
    pb.out_stage = bring_all + 4 * pipe_bring_bytes() + lane * 16;
    pb.out_dst = wrap_col;
    pb.in_dma = false;
    pb.in_src = 0;
    const char *bnd_in = bring_all + rank * pipe_bring_bytes() + j * 8;           // rows handed to this wave
    char *bnd_out = bring_all + (rank + 1) * pipe_bring_bytes() + j * 8;          // rows this wave hands on (wave 3: its staging ring)

    // ---- this wave's strips ----
    int ti = 0, s = rank;          // task (inside the item) and strip of this wave's current strip
    const bool any = rank < nstrips_all;
    if (any) {
        while (s >= tasks[it.task0 + ti].nstrips) { s -= tasks[it.task0 + ti].nstrips; ++ti; }
    }
    // positions 0 .. AHEAD - 1 of the stream, and the first two boundary blocks (rows 0 .. 7) of the wave's first strip
    // when they come from memory: the analytic column - a wave's first strip can only be fed from memory if it is a
    // task's first strip (wave 0's predecessor, the wrap-around column, does not exist yet)
#pragma unroll
    for (int i = 0; i < PRALINE_PIPE_AHEAD; ++i) pipe_issue(dma);
    if (any && s == 0) {
        pipe_issue_block(analytic_col, pb.ring_addr, lane16);
        pipe_issue_block(analytic_col + 1024, pb.ring_addr + 1024, lane16);
        pb.in_dma = true;
        pb.in_src = analytic_col + 2048;
    }
    PRALINE_VMCNT(0);
    PRALINE_PIPE_BARRIER();
    // KEEP: where idle steps put their store - row 0 of column 0 of the item's first task (column 0 is never read: the
    // recompute kernel takes the analytic column; row 0 of any kept column is the dummy row of step 0)
    char *idle_st = KEEP ? reinterpret_cast<char *>(keep_bnd + tasks[it.task0].aux_off + j) : nullptr;
    for (int i = 0; i < PRALINE_PIPE_LAG * rank; ++i) pipe_idle_step<KEEP>(dma, idle_st, h);

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
            for (int u = 0; u < rsteps; ++u) pipe_idle_step<KEEP>(dma, idle_st, h);
            continue;
        }
        const WaveTask tk = tasks[it.task0 + ti];
        const int two = tk.two[0];
        const int L2 = ar.len[two];
        const int nstrips = tk.nstrips;
        const int clast = (L2 - 1) & 31;
        const bool own_last = (clast >> 4) == h;
        const int x0 = s * 32;
        const bool last_owner = (s == nstrips - 1) && own_last;
        const int xb = x0 + 16 * h;
        cidx = clast & 15;
        asm volatile("" : "+v"(cidx));
        const bool have_pair = have_one && lane_pair[(it.task0 + ti) * 32 + j] >= 0;
        // the upper half hands its rows on - unless this strip ends its task: the next strip then starts from the
        // analytic column, which its wave fetches into the same ring
        const bool wr_lane = h == 1 && (s < nstrips - 1 || rank == 3);
        pb.out_dst = wrap_col;
        PipeKeep ks;
        if constexpr (KEEP) {
            // this strip's kept column (index s + 1 of the task, row 0 = the dummy row of step 0's upper half) and checkpoints
            const int64_t keep_col = (int64_t)(tk.max_l1 + PRALINE_TB2_PAD) * 32 * (int64_t)sizeof(float4);
            ks.st = reinterpret_cast<char *>(keep_bnd + tk.aux_off + j) + (int64_t)(s + 1) * keep_col;
            ks.ckpt = reinterpret_cast<f4n *>(ckpt + tk.tb_off) + (int64_t)s * pipe_keep_blocks(tk.max_l1) * PRALINE_CKPT_BLOCK_F4 + lane;
        }

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
        // b0 row 2's operands - and its first two boundary blocks were requested at steps rsteps - 5 and rsteps - 1)

        float Hs[17], Uc[16];
        Hs[0] = PRALINE_NEG_INF;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Hs[c + 1] = boundary_value(xb + c + 1, go, ge, free_two);   // H[0][x] = o[0,x,2]
            Uc[c] = PRALINE_NEG_INF;                                     // U[1][x]
        }
        float dH = (s == 0) ? h00 : boundary_value(x0, go, ge, free_two);
        float hd_x = PRALINE_NEG_INF, l_x = PRALINE_NEG_INF;
        // per-strip results (folded into the task's table at the end of the strip); the column-0 / row-0 cells and
        // o[0,0] enter at finalisation
        float out_best = PRALINE_NEG_INF, out_rowmax = PRALINE_NEG_INF, out_colmax = PRALINE_NEG_INF, out_corner = PRALINE_NEG_INF;
        float best_run = LOCAL ? h00 : PRALINE_NEG_INF;
        float col_run = PRALINE_NEG_INF;

        // the next strip of this wave (strip q + 4): its A tile is fetched into LDS during the last iteration, its
        // boundary blocks (if they come from memory) from step rsteps - 5 on
        int ti_n = ti, s_n = s + 4;
        const bool more = q + 4 < nstrips_all;
        if (more) {
            while (s_n >= tasks[it.task0 + ti_n].nstrips) { s_n -= tasks[it.task0 + ti_n].nstrips; ++ti_n; }
        }
        const bool in_dma_next = more && (rank == 0 || s_n == 0);
        const unsigned long long in_src_next = s_n == 0 ? analytic_col : wrap_col;

#define PRALINE_PIPE_STEPK(KK, SN, KP, CURA, PREVA, BUSE, BFIL)                                                        \
        pipe_step<NR, NTERM, LOCAL, SEMI, KK, SN, KP>(u0 + KK, L1, have_pair, h, CURA, PREVA, BUSE, BFIL, aop, ring, stage_rd, bnd_in, \
                                                  bnd_out, wr_lane, dma, pb, lane16, Hs, Uc, dH, hd_x, l_x, best_run, col_run, \
                                                  out_best, out_rowmax, out_colmax, out_corner, go, ge, cidx, last_owner, xb, L2, &ks, ck0)
#define PRALINE_PIPE_STEP(KK, SN, CURA, PREVA, BUSE, BFIL) PRALINE_PIPE_STEPK(KK, SN, (KEEP ? 1 : 0), CURA, PREVA, BUSE, BFIL)
        // KEEP: the step at which a half reaches a checkpoint row (wave-uniform test, see PipeKeep)
#define PRALINE_PIPE_STEPC(KK, SN, CK, CURA, PREVA, BUSE, BFIL)                                                        \
        do {                                                                                                           \
            if constexpr (KEEP) {                                                                                      \
                if (CK) PRALINE_PIPE_STEPK(KK, SN, 2, CURA, PREVA, BUSE, BFIL);                                        \
                else PRALINE_PIPE_STEPK(KK, SN, 1, CURA, PREVA, BUSE, BFIL);                                           \
            } else PRALINE_PIPE_STEPK(KK, SN, 0, CURA, PREVA, BUSE, BFIL);                                             \
        } while (0)
#define PRALINE_PIPE_TAIL(SN)                                                                                          \
            PRALINE_PIPE_STEP(1, SN, accB, accA, b1, b0);                                                              \
            PRALINE_PIPE_STEP(2, SN, accA, accB, b0, b1);                                                              \
            PRALINE_PIPE_STEP(3, SN, accB, accA, b1, b0);                                                              \
            PRALINE_PIPE_STEP(4, SN, accA, accB, b0, b1);                                                              \
            PRALINE_PIPE_STEP(5, SN, accB, accA, b1, b0);                                                              \
            PRALINE_PIPE_STEP(6, SN, accA, accB, b0, b1);                                                              \
            if (last_it) {   /* the block requested at step 7 (rows rsteps .. rsteps + 3) is the next strip's rows 0 .. 3 */ \
                pb.in_dma = in_dma_next;                                                                               \
                pb.in_src = in_src_next;                                                                               \
            }                                                                                                          \
            PRALINE_PIPE_STEP(7, SN, accB, accA, b1, b0);                                                              \
            PRALINE_PIPE_STEP(8, SN, accA, accB, b0, b1);                                                              \
            PRALINE_PIPE_STEP(9, SN, accB, accA, b1, b0);                                                              \
            PRALINE_PIPE_STEP(10, SN, accA, accB, b0, b1);                                                             \
            if (a_fetch) {                                                                                             \
                /* the last step's MFMAs compute row 1 of the NEXT strip: switch to its A tile (fetched 10 steps ago) */ \
                _Pragma("unroll") for (int qq = 0; qq < 4; ++qq)                                                       \
                    aop[qq] = *reinterpret_cast<const float4 *>(atile + 1024 * qq + lane * 16);                        \
            }                                                                                                          \
            PRALINE_PIPE_STEPC(11, SN, ck11, accB, accA, b1, b0)
        // one 12-step iteration; SN: with the snapshot test (the iterations that can contain a sequence's last row)
#define PRALINE_PIPE_ITER(SN)                                                                                          \
        {                                                                                                              \
            const bool last_it = u0 + 12 >= rsteps;                                                                    \
            const bool a_fetch = last_it && more;                                                                      \
            /* KEEP: step 0's upper half is at row u0, step 11's lower half at row u0 + 12 */                           \
            const bool ck0 = KEEP && u0 > 0 && (u0 % PRALINE_KEEP_BH) == 0;                                             \
            const bool ck11 = KEEP && ((u0 + 12) % PRALINE_KEEP_BH) == 0;                                               \
            if (u0 == 0) {                                                                                             \
                /* step 0: only the lower half has a row (row 1); the upper half's garbage is undone right after */    \
                float Hsave[17];                                                                                       \
                _Pragma("unroll") for (int c = 0; c < 17; ++c) Hsave[c] = Hs[c];                                       \
                const float best_s = best_run, col_s = col_run;                                                        \
                PRALINE_PIPE_STEP(0, SN, accA, accB, b0, b1);                                                          \
                if (h) {                                                                                               \
                    _Pragma("unroll") for (int c = 0; c < 17; ++c) Hs[c] = Hsave[c];                                   \
                    _Pragma("unroll") for (int c = 0; c < 16; ++c) Uc[c] = PRALINE_NEG_INF;                            \
                    best_run = best_s;                                                                                 \
                    col_run = col_s;                                                                                   \
                }                                                                                                      \
            } else {                                                                                                   \
                PRALINE_PIPE_STEPC(0, SN, ck0, accA, accB, b0, b1);                                                    \
            }                                                                                                          \
            if (a_fetch) {                                                                                             \
                /* A tile of strip q + 4 (rows x0' .. x0' + 31 of its sequence two, this lane's 64 bytes) -> LDS */    \
                const unsigned long long qs = uniform64(reinterpret_cast<unsigned long long>(ar.Q16) +                 \
                                              ((unsigned long long)ar.row_off[tasks[it.task0 + ti_n].two[0]] + (unsigned long long)(s_n * 32)) * 128ull); \
                _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                                                     \
                    unsigned keep;                                                                                     \
                    const unsigned go_ = a_gofs + 16u * qq, dst = a_lds + 1024u * qq;                                  \
                    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"                                                    \
                                 "global_load_lds_dwordx4 %1, %2\n\t"                                                  \
                                 : "=&s"(keep)                                                                         \
                                 : "v"(go_), "s"(qs), "s"(dst)                                                         \
                                 : "memory");                                                                          \
                }                                                                                                      \
            }                                                                                                          \
            PRALINE_PIPE_TAIL(SN);                                                                                     \
        }
        // two loops rather than a test per iteration: the compiler spills when both bodies hang off one branch
        int u0 = 0;
        // first iteration whose rows u0 .. u0 + 12 reach min_l1.  (LOCAL: every iteration keeps the test - without any branch
        // the twelve steps fuse into one basic block and the compiler spills ~110 registers: see split16_step, SNAPBR.)
        const int u_snap = LOCAL ? 0 : min(rsteps, max(0, (min_l1 - 1) / 12 * 12));
        for (; u0 < u_snap; u0 += 12) PRALINE_PIPE_ITER(false)
        for (; u0 < rsteps; u0 += 12) PRALINE_PIPE_ITER(true)
#undef PRALINE_PIPE_ITER
#undef PRALINE_PIPE_TAIL
#undef PRALINE_PIPE_STEP
#undef PRALINE_PIPE_STEPC
#undef PRALINE_PIPE_STEPK

        // ---- fold this strip's share into the task's results ----
        {
            const float corner_all = __builtin_fmaxf(out_corner, partner_value(out_corner, h));
            const float rowmax_all = __builtin_fmaxf(out_rowmax, partner_value(out_rowmax, h));
            const float colmax_all = __builtin_fmaxf(out_colmax, partner_value(out_colmax, h));
            const float best_all = __builtin_fmaxf(out_best, partner_value(out_best, h));
            if (h == 0) {
                float *r0 = res + ti * 64 + j;
                const float v0 = LOCAL ? best_all : (SEMI ? rowmax_all : corner_all);
                if (v0 != PRALINE_NEG_INF) atomicMax(r0, v0);
                if (SEMI && colmax_all != PRALINE_NEG_INF) atomicMax(r0 + 32, colmax_all);
            }
            if constexpr (KEEP) {
                // global end state: the first of (M, U, L) of the corner cell that equals its maximum (np.argmax, align.py:428-430);
                // only the half of the strip that owns the corner holds a finite out_corner
                if (out_corner != PRALINE_NEG_INF) {
                    const float kf = (ks.snap_m == out_corner) ? 0.0f : ((ks.snap_u == out_corner) ? 1.0f : 2.0f);
                    atomicMax(res + ti * 64 + 32 + j, kf);
                }
            }
        }
        ti = ti_n;
        s = s_n;
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
        else if (SEMI) {
            // o[L1,0,1] and o[0,L2,2] are the column-0 / row-0 members of the last row / last column (align.py:406-424)
            const float rowmax = __builtin_fmaxf(v0, boundary_value(L1, go, ge, free_one));
            const float colmax = __builtin_fmaxf(v1, boundary_value(L2, go, ge, free_two));
            score = (rowmax > colmax && free_two) ? rowmax : colmax;
        } else score = v0;
        scores[pair] = score;
        if constexpr (KEEP) {
            int32_t *ec = end_cells + (int64_t)pair * 4;
            ec[0] = L1; ec[1] = L2; ec[2] = (int)v1; ec[3] = 0;
        }
    }
}
