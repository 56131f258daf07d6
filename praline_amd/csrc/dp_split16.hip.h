// dp_split16.hip.h -- k_dp_split16: the split-strip scores kernel with the match scores on the
// MATRIX pipe (v_mfma_f32_32x32x16_f16) instead of the fp32 MFMA.
//
// Why: measured on MI355X (scripts/micro/mfma_valu.hip) the fp32-input MFMA occupies the vector
// FP32 datapath - an fp32-MFMA wave and a VALU wave on one SIMD take the SUM of their times -
// while f16/bf16 MFMAs run on the separate matrix pipe and overlap with the VALU recurrence.
//
// Arithmetic: every fp32 operand x is split exactly into two halves hi = f16(x), lo = f16(x - hi)
// (22 significant bits), and the contraction m = sum_k Q2[x][k] * P1[y][k] is evaluated as
//     sum_k ( qlo*phi + qhi*plo ) + sum_k qhi*phi          (the lo*lo term, <= 2^-22, is dropped)
// in ONE fp32 accumulator, small terms first.  Products of f16 values are exact in fp32; the
// result differs from the reference's fp32 evaluation by <~ 1e-6 relative (north_star allows
// 1e-5).  EXACT mode (NTERM = 1): when every P and Q2 entry is exactly representable in f16 -
// one-hot profiles x integer matrices, the reference's "integer scoring" - only the hi*hi term
// is issued and every intermediate is an exactly representable integer: bit-identical results.
//
// Layout per arena row and k-parity half hh (lane >> 5): piece-major 16-byte slots
//     [hi r0][hi r1]...[lo r0][lo r1]...   slot (piece, r) holds k = 16 r + 8 hh + 0..7
// (lane l supplies A[row l&31][k = 8 (l>>5) + j] / B[k = 8 (l>>5) + j][col l&31], j = 0..7).
//
// Everything else (split-strip layout, two-lane wavefront, software pipeline, snapshots) is
// dp_split.hip.h's; see there.
#pragma once
#include "dp_split.hip.h"
#include "dp_arena16.h"

#ifndef PRALINE_S16_ABLATE
#define PRALINE_S16_ABLATE 0   // experiments only (results invalid): 1 no operand refills, 2 no boundary column traffic, 4 no MFMAs;
                               // staged stream: 8 no in-loop DMA, 16 no LDS operand / boundary reads, 32 no half select,
                               // 64 L chain cut (every column's L from its U), 256 no vmcnt wait
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half8 as_half8(const float4 &v) { return __builtin_bit_cast(half8, v); }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4n __attribute__((ext_vector_type(4)));

// A lane's 64-byte line of a dense tile (BSRC = 4) arrives as four 16-byte loads.  Plain loads: the first one brings the
// line into the CU's L1 and the other three hit there; streaming (nt) loads fetched every line four times from L2
// (measured on C2: 25 GB of tiles in 8.7 ms = 11.5 TB/s of L2 reads, the L2's ceiling).
#ifndef PRALINE_DENSE_NT
#define PRALINE_DENSE_NT 0
#endif
#if PRALINE_DENSE_NT
#define PRALINE_DENSE_LOAD(p) __builtin_nontemporal_load(p)
#else
#define PRALINE_DENSE_LOAD(p) (*(p))
#endif

// One-hot operand table (LDS): row `sym` holds the B operand slots of a profile row with all its mass
// on active symbol sym, [hh][r][8 halves]; row 16 NR is the all-zero row (padding rows, symbols that
// cannot score).  The odd 16-byte row stride spreads the rows over the LDS banks.
__host__ __device__ constexpr int onehot_stride(int NR) { return 32 * NR + 16; }
__host__ __device__ constexpr int onehot_bytes(int NR) { return (16 * NR + 1) * onehot_stride(NR); }

// Match-score LOOKUP (BSRC = 3; one-hot arenas in exact mode): with a one-hot sequence ONE the match score of a cell is
// no contraction at all - m[y][x] = Q2[x][symbol of row y] - so the strip's 32 pre-multiplied rows are transposed into
// an LDS table once per strip, lookup_tab[symbol][32 strip columns] (fp32; row 16 NR: zeros, for padding rows), and a
// step reads its lane's 16 values with four ds_read_b128 instead of issuing two to four MFMAs: no MFMA, no accumulator
// tiles, no operand registers.  The odd 16-byte row stride spreads the symbols' rows over the LDS banks.
__host__ __device__ constexpr int lookup_stride() { return 128 + 16; }
__host__ __device__ constexpr int lookup_bytes(int NR) { return (16 * NR + 1) * lookup_stride(); }

// Measured: forcing v_pk_add_f32 for the three per-column adds made the kernel 13 % SLOWER (packed fp32
// VALU beside MFMAs is an anti-lever on gfx950); plain scalar adds are used.
__device__ __forceinline__ f2 pk_add(f2 a, f2 b)
{
    f2 d;
    d.x = a.x + b.x;
    d.y = a.y + b.y;
    return d;
}

// v[idx + 1] of the shifted H array (Hs[c + 1] = H of this lane's column c)
__device__ __forceinline__ float select16s(const float (&v)[17], int idx)
{
    float w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) w[c] = v[c + 1];
    return select16(w, idx);
}


// ---- LDS-staged operand stream (BSRC = 2) ------------------------------------------------------
// Measured (scripts/exp_step.py): a wave-wide load whose 64 lanes touch 64 different rows costs the CU's
// L1 about one tag cycle per lane; with 4 such loads per step and wave the four SIMDs of a CU saturate
// that path (0.49 us per step whatever the occupancy).  The staged stream fetches the same bytes in
// pieces of whole rows - C = row_bytes / 16 consecutive lanes read one row (128 contiguous bytes in the
// 3-term, <= 32 symbol case), one piece = 64 / C pairs - by LDS-DMA (global_load_lds_dwordx4: no VGPR
// destination, so the look-ahead costs LDS instead of registers), and every lane picks its operand
// slots up from LDS with ds_read_b128.  The DMA destination is lane-linear, so the swizzle that makes
// those reads bank-conflict free is applied to the SOURCE chunk and to the read address (the same XOR).
// The strip-boundary column takes the same road (global_load_lds_dword, one 256-byte row per step) so
// that no compiler-counted load is in flight inside the loop: the waits are counted by hand.
//
// LDS per wave: [4 x 256 B boundary rows][4 x SLOT operand rows], SLOT = 32 pairs x row_bytes.
// Row r of either kind lives in ring slot r % 4.
__host__ __device__ constexpr int stage_slot_bytes(int NOPB) { return 32 * 32 * NOPB; }  // 32 pairs x (2 halves x NOPB x 16 B)
__host__ __device__ constexpr int stage_lds_bytes(int NOPB) { return 1024 + 4 * stage_slot_bytes(NOPB); }
// swizzle of pair p's row: chunk c is kept at chunk position c ^ stage_swz(p)
template <int C> __device__ __forceinline__ unsigned stage_swz(unsigned p) { return (p / (16 / C)) & (C - 1); }

// (M0, the DMA's LDS base, is written in the statement that uses it and not restored: nothing else in these
// kernels reads M0 - LDS instructions do not need it on gfx9 - and the compiler never assumes it survives asm.
// %0 is an unused scratch SGPR kept so that the operand numbers stay put.)
// One step's DMA: NOPB pieces of operand rows (wave-uniform row cursor curB, per-lane byte offsets gofs,
// LDS slot address ldsB) and one boundary row (cursor curN, lane offset gofs_n, LDS address ldsN).
// The instruction offset moves source AND destination (checked: scripts/micro/glds_test.hip); gofs[i]
// carries the matching -1024 i (and a +4096 bias that the cursor takes back) so only the LDS side moves.
template <int NOPB>
__device__ __forceinline__ void stage_issue(unsigned long long curB, const unsigned (&gofs)[4], unsigned ldsB,
                                            unsigned long long curN, unsigned gofs_n, unsigned ldsN)
{
    unsigned keep;
    if constexpr (NOPB == 4)
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %5\n\t"
                     "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                     "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\t"
                     "global_load_lds_dword %7, %8\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "v"(gofs[1]), "v"(gofs[2]), "v"(gofs[3]), "s"(curB), "s"(ldsB), "v"(gofs_n), "s"(curN), "s"(ldsN)
                     : "memory");
    else if constexpr (NOPB == 2)
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
                     "global_load_lds_dword %5, %6\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "v"(gofs[1]), "s"(curB), "s"(ldsB), "v"(gofs_n), "s"(curN), "s"(ldsN)
                     : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "global_load_lds_dword %4, %5\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "s"(curB), "s"(ldsB), "v"(gofs_n), "s"(curN), "s"(ldsN)
                     : "memory");
}
// operand rows only (prologue)
template <int NOPB>
__device__ __forceinline__ void stage_issue_rows(unsigned long long curB, const unsigned (&gofs)[4], unsigned ldsB)
{
    unsigned keep;
    if constexpr (NOPB == 4)
        asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %5\n\t"
                     "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                     "global_load_lds_dwordx4 %4, %5 offset:3072\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "v"(gofs[1]), "v"(gofs[2]), "v"(gofs[3]), "s"(curB), "s"(ldsB)
                     : "memory");
    else if constexpr (NOPB == 2)
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "v"(gofs[1]), "s"(curB), "s"(ldsB)
                     : "memory");
    else
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     : "=&s"(keep)
                     : "v"(gofs[0]), "s"(curB), "s"(ldsB)
                     : "memory");
}
// boundary row only (prologue)
__device__ __forceinline__ void stage_issue_bnd(unsigned long long curN, unsigned gofs_n, unsigned ldsN)
{
    unsigned keep;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, %2\n\t"
                 : "=&s"(keep)
                 : "v"(gofs_n), "s"(curN), "s"(ldsN)
                 : "memory");
}
#define PRALINE_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// NR: 16-wide k ranges (1: <= 16 active symbols, 2: <= 32); NTERM: 1 exact / 3 split.
// BSRC: where the B operands (profile rows of the sequences ONE) come from.
//   0  per-lane global loads, 3-deep register ring (BOPS: row t+1 on entry, refilled with row t+4)
//   1  one-hot arenas: looked up in the one-hot operand table in LDS by the row's symbol (byte SB of symw)
//   2  LDS-staged stream (see above): SB = t % 4 selects the ring slots; BOPS holds row t+1, BFILL
//      receives row t+2 from LDS; the boundary value of the next step is read from LDS into bnd_pref
//
// DM ("double MFMA", staged stream only): the two halves of the wave need the tiles of DIFFERENT rows (lower
// half row t, upper half row t-1).  Without DM both tiles are kept and every lane selects its 16 values
// (16 v_cndmask per step, 11 % of the VALU work).  With DM the tile is built for both at once: the A operand
// is split by output row into aop (rows delivered to the lower half, zero elsewhere) and aopH (the complement),
// and the accumulator receives mfma(aop, row t+1) + mfma(aopH, row t): each lane's 16 values are its own row's,
// the extra products are exact zeros (x + 0 = x), the order of the real terms is unchanged -> same bits.
// Used in exact mode only (NTERM = 1: two MFMAs per step instead of one, +4 % on C2 one-hot); with the three-term
// split the six dependent MFMAs per step outweigh the selects (measured: 2224 -> 2050 GCUPS on C2 float profiles).
// KEEP (the forward fill of the two-pass alignments-with-paths scheme on THIS kernel, see dp_trace2.hip.h): besides
// its own (H, L) boundary hand-off the step writes what k_trace_recompute starts from -
//   * every strip's boundary column as three states: float4 (M, U, L, 0) of the cell (yy, last column of the strip),
//     [strip + 1][row][32 pairs] (the recurrence itself only carries H = max(M, U, L));
//   * KEEP == 2 (the two steps in 32 in which a half is at a row yy = 32 i): the (M, U, L) states of that row,
//     float4 [yy / 32][3][4][64] per strip (state, group of four columns, lane);
//   * global mode: M and U of the corner cell (the end state k is the first of M, U, L that equals the score).
// The states are the recurrence's own intermediates (M before the maximum, U and L on entry to the cell): no extra
// arithmetic, and fl(max3(Mp, Up, Lp) + m) = max3 of the three rounded sums, so they are bit for bit the values the
// three-state kernels carry.
struct KeepState {
    char *st = nullptr;       // kept boundary column being written: this lane's float4 of the row this step stores
    f4n *ckpt = nullptr;      // this strip's checkpoint blocks (+ lane)
    float snap_m = 0.0f, snap_u = 0.0f;
};
#define PRALINE_CKPT_BLOCK_F4 (3 * 4 * 64)   // float4 elements per checkpoint block

template <int NR, int NTERM, bool LOCAL, int BSRC = 0, int SB = 0, bool DM = false, bool SNAPBR = false, int KEEP = 0>
__device__ __forceinline__ void split16_step(int yy, int L1, bool have_pair, int h, const f32x16 &CUR, f32x16 &PREV,
                                             float4 (&BOPS)[(NTERM == 1 ? 1 : 2) * NR],
                                             const float4 (&aop)[(NTERM == 1 ? 1 : 2) * NR],
                                             const float4 (&aopH)[(NTERM == 1 ? 1 : 2) * NR],    // DM only
                                             const float4 (&BPREV)[(NTERM == 1 ? 1 : 2) * NR],   // DM only: row t
                                             float4 (&BFILL)[(NTERM == 1 ? 1 : 2) * NR],         // staged stream: receives row t+2;
                                                                                                 // one-hot table with DM: row t+4
                                             const char *&b_next,
                                             int b_stride, const char *&bnd_ld, char *&bnd_st, float2 &bnd_pref,
                                             float (&Hs)[17], float (&Uc)[16], float &dH, float &hd_x, float &l_x,
                                             float &best_run, float &col_run, float &out_best, float &out_rowmax, float &out_colmax,
                                             float &out_corner, float go, float ge, bool semiglobal, bool last_owner, int cidx,
                                             int xb, int L2, const char *onehot_lane = nullptr, unsigned symw = 0,
                                             const char *stage_lds = nullptr,
                                             const unsigned *stage_rd = nullptr, unsigned stage_rd_bnd = 0,
                                             const unsigned (*stage_gofs)[4] = nullptr, unsigned long long *stage_cur = nullptr,
                                             unsigned stage_lds_addr = 0, unsigned stage_gofs_n = 0, bool may_snap = true,
                                             KeepState *ks = nullptr)
{
    static_assert(!DM || BSRC == 2 || BSRC == 1, "the double-MFMA tile is wired for the staged stream and the one-hot table");
    static_assert(KEEP == 0 || !DM, "the kept states are taken from the non-DM tile");
    constexpr bool ONEHOT = BSRC == 1;
    constexpr bool DENSE = BSRC == 4;    // the next row's scores are read from a dense tile (dp_reftile.hip.h) at onehot_lane
    constexpr bool LOOKUP = BSRC == 3 || DENSE;   // CUR: this lane's 16 match scores of its row; PREV receives the next row's (symbol symw)
    // bnd_pref: this step's boundary value on entry; refilled with the value 3 rows ahead.
    // BOPS: B operands of row t+1 on entry; refilled with row t+4 (3-deep rings, the caller rotates
    // the register names through a 6x unrolled loop).
    constexpr int NP = (NTERM == 1) ? 1 : 2;   // pieces held per operand
    constexpr int NM = LOOKUP ? 1 : NTERM * NR;   // MFMAs per step
    // ---- match scores of this lane's row: lower half row t (CUR), upper half row t-1 (PREV) ----
    f2 m2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if constexpr (DM || LOOKUP || (PRALINE_S16_ABLATE & 32) != 0) { m2[c].x = CUR[2 * c]; m2[c].y = CUR[2 * c + 1]; }
        else { m2[c].x = h ? PREV[2 * c] : CUR[2 * c]; m2[c].y = h ? PREV[2 * c + 1] : CUR[2 * c + 1]; }
    }

    const float2 bv = bnd_pref;
    if constexpr (BSRC == 2) {
        constexpr int NOPB = ((NTERM == 1) ? 1 : 2) * NR;
        // the DMA of operand row t+2 and boundary row t+1 was issued three steps ago: everything but the
        // last two steps' pieces has landed
#if !(PRALINE_S16_ABLATE & 256)
        PRALINE_VMCNT(2 * (NOPB + 1));
#endif
#if !(PRALINE_S16_ABLATE & 16)
        bnd_pref = *reinterpret_cast<const float2 *>(stage_lds + ((SB + 1) & 3) * 256 + stage_rd_bnd);
#pragma unroll
        for (int q = 0; q < NOPB; ++q)
            BFILL[q] = *reinterpret_cast<const float4 *>(stage_lds + 1024 + ((SB + 2) & 3) * stage_slot_bytes(NOPB) + stage_rd[q]);
#endif
    } else {
#if !(PRALINE_S16_ABLATE & 2)
        bnd_pref = *reinterpret_cast<const float2 *>(bnd_ld);
#endif
        bnd_ld += 32 * sizeof(float2);
    }
    // Hs[c] = H[y-1] of the column LEFT of column c (Hs[0]: handed in), so the diagonal inputs of two
    // adjacent columns sit in one aligned register pair and the adds below are v_pk_add_f32.
    Hs[0] = h ? hd_x : dH;
    float lrun = h ? l_x : bv.y;
    const float hd_out = Hs[16];  // H[y-1] of this lane's last column: diagonal input of the next lane / strip
    const f2 go2 = {go, go}, ge2 = {ge, ge};
    f2 hs = {Hs[0], Hs[1]};  // previous-row H left of the next column pair (read BEFORE that pair's slots are rewritten)
    bool ck_mine = false;
    f4n ckM, ckU, ckL;         // KEEP == 2: the states of four columns, stored once complete
    float kM = 0.0f, kU = 0.0f, kL = 0.0f;   // KEEP: states of this lane's last column
    if constexpr (KEEP != 0) {
        ck_mine = yy >= 32 && (yy & 31) == 0;
        if (!LOCAL && may_snap && have_pair && yy == L1 && last_owner) {   // the corner cell is in this row
            float hw[16], mw[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { hw[c] = Hs[c]; mw[c] = (c & 1) ? m2[c >> 1].y : m2[c >> 1].x; }
            ks->snap_m = select16(hw, cidx) + select16(mw, cidx);
            ks->snap_u = select16(Uc, cidx);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (DENSE) {
        // the next row's match scores: one 64-byte line of the task's dense tile, read once (streaming loads)
        const f4n *q = reinterpret_cast<const f4n *>(onehot_lane);
        const f4n a0 = PRALINE_DENSE_LOAD(q), a1 = PRALINE_DENSE_LOAD(q + 1), a2 = PRALINE_DENSE_LOAD(q + 2), a3 = PRALINE_DENSE_LOAD(q + 3);
        PREV[0] = a0.x; PREV[1] = a0.y; PREV[2] = a0.z; PREV[3] = a0.w; PREV[4] = a1.x; PREV[5] = a1.y; PREV[6] = a1.z; PREV[7] = a1.w;
        PREV[8] = a2.x; PREV[9] = a2.y; PREV[10] = a2.z; PREV[11] = a2.w; PREV[12] = a3.x; PREV[13] = a3.y; PREV[14] = a3.z; PREV[15] = a3.w;
    } else if constexpr (LOOKUP) {
        // the next row's match scores: this lane's 16 strip columns of the table row of that row's symbol
        const float4 *q = reinterpret_cast<const float4 *>(onehot_lane + symw * lookup_stride());
        const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
        PREV[0] = a0.x; PREV[1] = a0.y; PREV[2] = a0.z; PREV[3] = a0.w; PREV[4] = a1.x; PREV[5] = a1.y; PREV[6] = a1.z; PREV[7] = a1.w;
        PREV[8] = a2.x; PREV[9] = a2.y; PREV[10] = a2.z; PREV[11] = a2.w; PREV[12] = a3.x; PREV[13] = a3.y; PREV[14] = a3.z; PREV[15] = a3.w;
    }
    // ---- MFMAs of row t+1 on the matrix pipe, interleaved with the recurrence of this row ----
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        // order: (qlo, phi) r.., (qhi, plo) r.., then (qhi, phi) r..   [small terms first]
        const int term = (NTERM == 1) ? 2 : k / NR;
        const int r = k % NR;
        const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);          // A piece: lo for term 0, hi otherwise
        const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);          // B piece: lo for term 1, hi otherwise
#if !(PRALINE_S16_ABLATE & 4)
        if constexpr (!LOOKUP) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(BOPS[ib]), acc, 0, 0, 0);
        if constexpr (DM) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aopH[ia]), as_half8(BPREV[ib]), acc, 0, 0, 0);
#else
        acc[k] += BOPS[ib].x * aop[ia].x;
#endif
        // this MFMA's share of the 8 column pairs
#pragma unroll
        for (int cp = (8 * k) / NM; cp < (8 * (k + 1)) / NM; ++cp) {
            f2 M = pk_add(hs, m2[cp]);                            // max_k o[y-1,x-1,k] + m      (cext.c:192-222)
            if (LOCAL) { M.x = __builtin_fmaxf(M.x, 0.0f); M.y = __builtin_fmaxf(M.y, 0.0f); }  // cext.c:208-209
            const f2 Mo = pk_add(M, go2);                       // gap opened from these cells
            const f2 U = {Uc[2 * cp], Uc[2 * cp + 1]};
            const f2 Ug = pk_add(U, ge2);
            // column 2cp
            const float H0 = max3f(M.x, U.x, lrun);
            const float lin0 = lrun;
#if PRALINE_S16_ABLATE & 64
            lrun = __builtin_fmaxf(Mo.x, Ug.x + ge);
#else
            lrun = __builtin_fmaxf(Mo.x, lrun + ge);      // L[y][x+1]   (cext.c:169-183,276-283)
#endif
            // column 2cp + 1
            const float H1 = max3f(M.y, U.y, lrun);
            if constexpr (KEEP != 0) {
                if (cp == 7) { kM = M.y; kU = U.y; kL = lrun; }
            }
            if constexpr (KEEP == 2) {
                if ((cp & 1) == 0) { ckM.x = M.x; ckM.y = M.y; ckU.x = U.x; ckU.y = U.y; ckL.x = lin0; ckL.y = lrun; }
                else {
                    ckM.z = M.x; ckM.w = M.y; ckU.z = U.x; ckU.w = U.y; ckL.z = lin0; ckL.w = lrun;
#ifndef PRALINE_KEEP_ABLATE
#define PRALINE_KEEP_ABLATE 0   // timing experiments only: 1 no checkpoint stores, 2 no kept boundary columns
#endif
                    if (ck_mine && !(PRALINE_KEEP_ABLATE & 1)) {
                        f4n *q = ks->ckpt + (int64_t)(yy >> 5) * PRALINE_CKPT_BLOCK_F4 + (cp >> 1) * 64;
                        __builtin_nontemporal_store(ckM, q);
                        __builtin_nontemporal_store(ckU, q + 4 * 64);
                        __builtin_nontemporal_store(ckL, q + 8 * 64);
                    }
                }
            }
#if PRALINE_S16_ABLATE & 64
            lrun = __builtin_fmaxf(Mo.y, Ug.y + ge);
#else
            lrun = __builtin_fmaxf(Mo.y, lrun + ge);
#endif
            if (LOCAL) best_run = max3f(best_run, H0, H1);
            Uc[2 * cp] = __builtin_fmaxf(Mo.x, Ug.x);        // U[y+1][x]   (cext.c:152-166,247-254)
            Uc[2 * cp + 1] = __builtin_fmaxf(Mo.y, Ug.y);
            if (cp < 7) { hs.x = Hs[2 * cp + 2]; hs.y = Hs[2 * cp + 3]; }  // still the previous row's values
            Hs[2 * cp + 1] = H0;
            Hs[2 * cp + 2] = H1;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!LOOKUP) PREV = acc;
    if constexpr (LOOKUP) {
        // (nothing to refill: the table serves every row of the strip)
    } else if constexpr (BSRC == 2) {
        // rows t+5 / t+4 go where rows t+1 / t lived (both consumed: their reads were waited for)
        constexpr int NOPB = ((NTERM == 1) ? 1 : 2) * NR;
#if !(PRALINE_S16_ABLATE & 8)
        stage_issue<NOPB>(stage_cur[0], *stage_gofs, stage_lds_addr + 1024 + ((SB + 1) & 3) * stage_slot_bytes(NOPB),
                          stage_cur[1], stage_gofs_n, stage_lds_addr + (SB & 3) * 256);
#endif
        stage_cur[0] += 64 * NR;  // one arena row
        stage_cur[1] += 256;
    } else if constexpr (ONEHOT) {
        const unsigned sym = (symw >> (8 * SB)) & 0xffu;
        const float4 *bsrc = reinterpret_cast<const float4 *>(onehot_lane + sym * onehot_stride(NR));
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if constexpr (DM) BFILL[q] = bsrc[q];   // into the registers of row t (consumed by this step's MFMAs)
            else BOPS[q] = bsrc[q];
        }
    } else {
#if !(PRALINE_S16_ABLATE & 1)
        const float4 *bsrc = reinterpret_cast<const float4 *>(b_next);
#pragma unroll
        for (int q = 0; q < NP * NR; ++q) BOPS[q] = bsrc[q];
#endif
        b_next += b_stride;
    }
    dH = bv.x;
    hd_x = from_lower_half(hd_out);
    l_x = from_lower_half(lrun);

#if !(PRALINE_S16_ABLATE & 2)
#if PRALINE_S16_NT_BND
    // streaming store: the column is read once, by the next strip, long after it has left this XCD's L2 anyway -
    // it should not push operand rows out on its way (+1 %)
    if (h) {   // H[yy][x0+32], L[yy][x0+33]
        const unsigned long long v = (unsigned long long)__float_as_uint(Hs[16]) | ((unsigned long long)__float_as_uint(lrun) << 32);
        __builtin_nontemporal_store(v, reinterpret_cast<unsigned long long *>(bnd_st));
    }
#else
    if (h) *reinterpret_cast<float2 *>(bnd_st) = make_float2(Hs[16], lrun);  // H[yy][x0+32], L[yy][x0+33]
#endif
#endif
    bnd_st += 32 * sizeof(float2);
    if constexpr (KEEP != 0) {
        if (h && !(PRALINE_KEEP_ABLATE & 2)) {
            const f4n kv = {kM, kU, kL, 0.0f};
            __builtin_nontemporal_store(kv, reinterpret_cast<f4n *>(ks->st));
        }
        ks->st += 32 * sizeof(float4);
    }
    if (semiglobal && last_owner) col_run = __builtin_fmaxf(col_run, select16s(Hs, cidx));
    // may_snap (wave-uniform): this step can be some lane's last row (the task's pairs are sorted by length, so
    // for most of a strip it cannot, and a scalar branch replaces the per-lane test)
    if (may_snap && have_pair && yy == L1) {
        // SNAPBR: keep this a real branch.  In the one-wave LOCAL kernels only out_best is live here; the compiler
        // turns the block into a select, the twelve unrolled steps fuse into one basic block and the scheduler
        // takes 256 VGPRs + ~110 AGPRs for it (one wave per SIMD); with the branch kept: ~190 VGPRs, two waves.
        if constexpr (SNAPBR) asm volatile("");
        if (LOCAL) out_best = best_run;
        if (semiglobal) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                out_rowmax = __builtin_fmaxf(out_rowmax, (xb + c + 1 <= L2) ? Hs[c + 1] : PRALINE_NEG_INF);
            out_colmax = col_run;
        }
        if (last_owner) out_corner = select16s(Hs, cidx);
    }
}


// ONEHOT (exact mode only): every sequence ONE of the arena is one-hot (an ordinary sequence).  Its
// operand rows are then not streamed from HBM - 64 lanes reading 64 different rows per step is what
// saturates the CU's texture-address path (measured: scripts/exp_step.py, 0.47 -> 0.32 us per step and
// SIMD without these loads) - but looked up in a one-hot table in LDS by the row's symbol; the symbols
// arrive as one dword per lane and four rows.
#ifndef PRALINE_S16_DM
#define PRALINE_S16_DM 1
#endif
#ifndef PRALINE_S16_NT_BND
#define PRALINE_S16_NT_BND 1
#endif
#ifdef PRALINE_TRACE
// experiments only (scripts/exp_trace.py): per wave {block, wave | share << 8, HW_ID, XCC_ID, start, end} (s_memtime)
__device__ unsigned long long *praline_trace_buf = nullptr;
#endif

// BSRC = 2: the LDS-staged operand stream described above (one wave per workgroup).
//
// WPG = 4 (staged stream only): workgroups of FOUR waves described by wg[blockIdx.x] (WgDesc): `share`
// consecutive waves work on ONE task - the wave of rank r takes the strips r, r + share, ... and runs
// two 12-row iterations behind rank r - 1.  A task's strips form a chain (strip s+1 needs the boundary column
// of strip s row by row), so the waves pipeline it and the task's critical path shrinks by ~share.  That
// is what small batches need: with about one task per SIMD (BASELINE C2: 1144 tasks, 1024 SIMDs) a launch
// lasts as long as its longest task while most SIMDs idle.  The waves of a workgroup stay in lock step
// through one s_barrier per 12 rows; the hand-off is the ordinary boundary buffer in global memory
// (producer and consumer are >= 2 iterations = 24 rows apart in every direction, and the per-step
// counted vmcnt retires every store older than three steps).  share == 1: four
// independent tasks, no barriers.
// KEEP: the forward fill of the two-pass alignments with paths (see KeepState): kept boundary columns at
// keep_bnd + tk.aux_off (float4 [nstrips + 1][max_l1 + PRALINE_TB2_PAD][32]), row checkpoints at ckpt + tk.tb_off
// (floats), end cells (global: the corner and its first maximal state) to end_cells.
#ifndef PRALINE_LOOKUP_WAVES
#define PRALINE_LOOKUP_WAVES 3   // waves per SIMD the lookup instances are compiled for (168 VGPRs)
#endif
#ifndef PRALINE_DENSE_WAVES
#define PRALINE_DENSE_WAVES 2    // waves per SIMD of the dense-tile instances (four 16-register sets of match scores in flight)
#endif
template <int NR, int NTERM, bool LOCAL, int BSRC = 0, int WPG = 1, bool KEEP = false>
__global__ __launch_bounds__(256, KEEP ? 2 : (BSRC == 3 ? PRALINE_LOOKUP_WAVES : (BSRC == 4 ? PRALINE_DENSE_WAVES : 1))) void k_dp_split16(Arena16Dev ar, const WaveTask *__restrict__ tasks,
                                                    const int32_t *__restrict__ lane_one,
                                                    const int32_t *__restrict__ lane_pair, float2 *bnd,
                                                    float *__restrict__ scores, RunParams rp, int n_tasks,
                                                    const WgDesc *__restrict__ wg = nullptr, float4 *keep_bnd = nullptr,
                                                    float *ckpt = nullptr, int32_t *__restrict__ end_cells = nullptr)
{
    static_assert(!KEEP || (BSRC == 2 && !LOCAL), "the kept-state forward fill runs on the staged stream (global / semiglobal recurrences)");
    constexpr int NP = (NTERM == 1) ? 1 : 2;
    constexpr int NOP = NP * NR;  // 16-byte operand slots held per lane
    static_assert(WPG == 1 || (WPG == 4 && (BSRC == 2 || BSRC == 3 || BSRC == 4)), "four-wave workgroups use the staged stream, the match-score lookup or dense tiles");
    constexpr bool LOOKUP = BSRC == 3;
    constexpr bool DENSE = BSRC == 4;   // match scores from the task's dense tile (ar.dense, dp_reftile.hip.h): no operands at all
    static_assert(!LOOKUP || (NTERM == 1 && !KEEP), "the match-score lookup is an exact-mode path");
    static_assert(!DENSE || (NTERM == 1 && !KEEP), "the dense-tile instances take no operands");
    constexpr bool MW = WPG > 1;
    constexpr bool SNAPBR = LOCAL && !MW;   // see split16_step
#ifdef PRALINE_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr bool ONEHOT = BSRC == 1;
    constexpr bool STAGED = BSRC == 2;
    constexpr bool DM = (STAGED || ONEHOT) && NTERM == 1 && (PRALINE_S16_DM != 0);   // one tile for both halves (see split16_step)
    static_assert(!ONEHOT || NTERM == 1, "the one-hot table path is an exact-mode path");
    __shared__ __attribute__((aligned(16))) char onehot_tab[ONEHOT ? onehot_bytes(NR) : 16];
    __shared__ __attribute__((aligned(16))) char lookup_all[LOOKUP ? WPG * lookup_bytes(NR) : 16];   // one table per wave
    __shared__ __attribute__((aligned(16))) char stage_lds_all[STAGED ? WPG * stage_lds_bytes(NP * NR) : 16];
    __shared__ float mw_out[MW ? WPG * 4 * 32 : 1];  // partial results of the waves sharing a task
    const int wv = MW ? (int)(threadIdx.x >> 6) : 0;
    char *stage_lds = stage_lds_all + (STAGED ? wv * stage_lds_bytes(NP * NR) : 0);
    if constexpr (STAGED) {
        // defined contents before the first DMA (the compiler does not see the DMA's writes)
        for (int i = threadIdx.x * 16; i < WPG * stage_lds_bytes(NP * NR); i += blockDim.x * 16)
            *reinterpret_cast<float4 *>(stage_lds_all + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    if constexpr (ONEHOT) {
        _Float16 *tab = reinterpret_cast<_Float16 *>(onehot_tab);
        constexpr int per_row = onehot_stride(NR) / 2;  // halves per table row (the last 8 are padding)
        for (int i = threadIdx.x; i < (16 * NR + 1) * per_row; i += blockDim.x) {
            const int sym = i / per_row, e = i % per_row;
            const int hh = e / (8 * NR), r = (e / 8) % NR, jj = e % 8;
            tab[i] = (e < 16 * NR && 16 * r + 8 * hh + jj == sym) ? (_Float16)1.0f : (_Float16)0.0f;
        }
        __syncthreads();
    }
    int task;
    int share = 1, rank = 0;   // MW: `share` waves pipeline this task, this wave is number `rank` of them
    int mw_left = 0;           // MW: barriers this wave still owes its workgroup
    if constexpr (MW) {
        const WgDesc *d = wg + blockIdx.x;
        share = d->share;
        rank = wv & (share - 1);
        task = d->task[wv - rank];
        mw_left = d->barriers;
        if (task < 0) {  // idle wave: keep the workgroup's barrier count
            for (; mw_left > 0; --mw_left) __builtin_amdgcn_s_barrier();
            if (share > 1) __syncthreads();
            return;
        }
    } else {
        task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (task >= n_tasks) return;
    }
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    const int j = lane & 31;
    const WaveTask tk = tasks[task];
    const int base = task * 32;
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const float go = rp.go1, ge = rp.ge1;

    const int my_one = lane_one[base + j];
    const int two = tk.two[0];
    const bool have_pair = my_one >= 0;
    const int L1 = have_pair ? ar.len[my_one] : 0;
    const int L2 = ar.len[two];
    const int nstrips = (L2 + 31) >> 5;
    const int clast = (L2 - 1) & 31;
    const bool own_last = (clast >> 4) == h;
    const int max_l1 = tk.max_l1;

    const char *pB = ar.P16 + (int64_t)(have_pair ? ar.row_off[my_one] : 0) * ar.row_bytes + h * ar.half_bytes;
    const int b_stride = ar.row_bytes;
    const unsigned *pSym = reinterpret_cast<const unsigned *>(ar.sym8 + (have_pair ? ar.row_off[my_one] : 0));
    char *lookup_tab = lookup_all + (LOOKUP ? wv * lookup_bytes(NR) : 0);
    const char *onehot_lane = LOOKUP ? lookup_tab + h * 64 : onehot_tab + h * (16 * NR);
    // DENSE: this lane's 64-byte line of row y of strip s is at dense_task + s * dense_strip + y * 4096 (the upper half's
    // pointer is one row back: at step T both halves fetch "row T + 1")
    const int64_t dense_strip = (int64_t)(max_l1 + PRALINE_DENSE_PAD) * 4096;
    const char *dense_task = DENSE ? reinterpret_cast<const char *>(ar.dense + ar.dense_off[task]) + (int64_t)lane * 64 - (int64_t)h * 4096
                                   : nullptr;
    const char *dense_lane = dense_task;
    // STAGED: per-lane source offsets of the DMA pieces and LDS read addresses (see the comment above)
    unsigned stage_gofs[4] = {0, 0, 0, 0}, stage_rd[4] = {0, 0, 0, 0};
    const unsigned stage_rd_bnd = (unsigned)j * 8u;
    const unsigned stage_gofs_n = (unsigned)lane * 4u;
    const unsigned stage_lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)stage_lds);  // wave-uniform (SGPR)
    unsigned long long stage_cur[2] = {0, 0};
    unsigned long long stage_bnd_base = 0;
    if constexpr (STAGED) {
        constexpr int C = 2 * NOP;          // 16-byte chunks per row
        constexpr int PP = 64 / C;          // pairs per DMA piece
#pragma unroll
        for (int i = 0; i < NOP; ++i) {
            const int p = i * PP + lane / C;
            const int one_p = lane_one[base + p];
            const unsigned row0 = one_p >= 0 ? (unsigned)ar.row_off[one_p] : 0u;
            const unsigned chunk = ((unsigned)lane % C) ^ stage_swz<C>((unsigned)p);  // = hh * NOP + slot
            // arena rows always hold hi and lo pieces ([hh][2 NR slots]); exact mode fetches the hi slots only
            const unsigned mem_chunk = (chunk / NOP) * (2 * NR) + chunk % NOP;
            stage_gofs[i] = row0 * (unsigned)(64 * NR) + mem_chunk * 16u + 4096u - 1024u * i;
        }
#pragma unroll
        for (int q = 0; q < NOP; ++q)
            stage_rd[q] = (unsigned)j * (unsigned)(32 * NOP) + (((unsigned)(h * NOP + q)) ^ stage_swz<C>((unsigned)j)) * 16u;
        const unsigned long long bb = reinterpret_cast<unsigned long long>(bnd) + (unsigned long long)tk.bnd_off * sizeof(float2);
        // (readfirstlane returns int: widen through unsigned)
        stage_bnd_base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bb >> 32)) << 32) |
                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bb);
    }
    const int acol = 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3);
    const char *qA = ar.Q16 + ((int64_t)ar.row_off[two] + acol) * ar.row_bytes + h * ar.half_bytes;

    char *my_bnd = reinterpret_cast<char *>(bnd + tk.bnd_off + j);  // float2 [y][32]
    constexpr int BROW = 32 * (int)sizeof(float2);

    const float o001 = free_one ? 0.0f : (go - ge);
    const float o002 = free_two ? 0.0f : (go - ge);
    const float h00 = max3f(0.0f, o001, o002);

    if (h == 0 && rank == 0)
        for (int y = 1; y <= max_l1 + 2; ++y)
            *reinterpret_cast<float2 *>(my_bnd + (int64_t)y * BROW) = make_float2(boundary_value(y, go, ge, free_one), PRALINE_NEG_INF);
    KeepState ks;
    const int64_t keep_col = (int64_t)(max_l1 + PRALINE_TB2_PAD) * 32 * (int64_t)sizeof(float4);   // bytes per kept column
    char *keep_base = nullptr;
    if constexpr (KEEP) {
        keep_base = reinterpret_cast<char *>(keep_bnd + tk.aux_off + j);
        // strip 0's column: states of (y, 0) = (-inf, o[y,0,1], -inf)
        if (h == 0 && rank == 0)
            for (int y = 1; y <= max_l1 + 4; ++y)
                *reinterpret_cast<float4 *>(keep_base + (int64_t)y * 32 * sizeof(float4)) =
                    make_float4(PRALINE_NEG_INF, boundary_value(y, go, ge, free_one), PRALINE_NEG_INF, 0.0f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);

    // shared task: rank r starts PRALINE_MW_LAG 12-row iterations behind rank r - 1.  Two iterations are
    // enough: the waves meet at every iteration start, so a consumer is at most at step 12 i + 13 while its
    // producer is at least at step 12 (i + 2) + 2; it prefetches boundary row t + 4, stored at the
    // producer's step t + 5 and retired by the counted vmcnt three steps later.
    if constexpr (MW) {
        if (share > 1) {
            const int delay = rank * PRALINE_MW_LAG;
            for (int i = 0; i < delay; ++i) { __builtin_amdgcn_s_barrier(); --mw_left; }
        }
    }

    // shortest sequence one of the task (wave-uniform): no snapshot before row min_l1
    int min_l1 = have_pair ? L1 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) min_l1 = min(min_l1, __shfl_xor(min_l1, off));
    min_l1 = __builtin_amdgcn_readfirstlane(min_l1);

    const bool semiglobal = rp.mode >= 2;
    int cidx = clast & 15;
    asm volatile("" : "+v"(cidx));  // keep it a per-lane value so select16 stays a register select tree

    // per-pair results, snapshotted when the lane is at its last row
    float out_best = LOCAL ? h00 : 0.0f;
    float out_rowmax = (have_pair && h == 0) ? boundary_value(L1, go, ge, free_one) : PRALINE_NEG_INF;
    float out_colmax = (have_pair && own_last) ? boundary_value(L2, go, ge, free_two) : PRALINE_NEG_INF;
    float out_corner = PRALINE_NEG_INF;

    for (int s = rank; s < nstrips; s += share) {
        const int x0 = s * 32;
        const int xb = x0 + 16 * h;
        const bool last_owner = (s == nstrips - 1) && own_last;

        float4 aop[NOP];
        if constexpr (!LOOKUP && !DENSE) {
            const float4 *sa = reinterpret_cast<const float4 *>(qA + (int64_t)x0 * ar.row_bytes);
#pragma unroll
            for (int q = 0; q < NOP; ++q) aop[q] = sa[q];
        }
        float4 aopH[NOP];
        if constexpr (DM) {
            // A row j feeds output row j; rows with (j >> 2) & 1 are the ones the upper half receives
            // (bit masks, not vector selects: the compiler turns a select between float4 values into an indexed
            // stack array)
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
        float Hs[17], Uc[16];  // Hs[c + 1] = H[y-1] of this lane's column c; Hs[0] is handed in per row
        Hs[0] = PRALINE_NEG_INF;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            Hs[c + 1] = boundary_value(xb + c + 1, go, ge, free_two);   // H[0][x] = o[0,x,2]
            Uc[c] = PRALINE_NEG_INF;                                       // U[1][x]
        }
        float dH = (s == 0) ? h00 : boundary_value(x0, go, ge, free_two);
        float hd_x = PRALINE_NEG_INF, l_x = PRALINE_NEG_INF;
        float best_run = out_best;
        float col_run = out_colmax;
        if constexpr (KEEP) {
            ks.st = keep_base + (int64_t)(s + 1) * keep_col;   // upper half stores row yy = t - 1 (row 0: dummy)
            ks.ckpt = reinterpret_cast<f4n *>(ckpt + tk.tb_off) + (int64_t)s * PRALINE_TB2_CKPT_BLOCKS(max_l1) * PRALINE_CKPT_BLOCK_F4 + lane;
        }

        // pipeline prologue: B operands of rows 1..4, MFMAs of row 1, boundary column of rows 1..3
        float4 b0[NOP], b1[NOP], b2[NOP], b3[NOP];   // b3: one-hot table with DM (four sets: rows t .. t+3)
        f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 accB = accA;
        f32x16 accC = accA, accD = accA;   // DENSE: four sets
        unsigned d1 = 0, d2 = 0, d3 = 0, d4 = 0;  // ONEHOT: symbols of the rows the next 12 steps refill
        unsigned lw0 = 0, lw1 = 0, lw2 = 0, lw3 = 0;   // LOOKUP: the 16-byte symbol window
        float2 p0, p1, p2;
        const char *b_next = pB + 4 * b_stride;                   // first refill: row 5
        const char *bnd_ld = my_bnd + 4 * BROW;                   // first prefetch inside a step: row 4
        char *bnd_st = my_bnd;                                    // upper half stores row yy = t - 1 (row 0: dummy)
        if constexpr (STAGED) {
            constexpr int SLOT = stage_slot_bytes(NOP);
            PRALINE_VMCNT(0);  // the previous strip's look-ahead
            // operand row r is arena row r - 1 of each sequence; rows 1..4 -> ring slots 1, 2, 3, 0
            unsigned long long cb = reinterpret_cast<unsigned long long>(ar.P16) - 4096ull;
            unsigned long long cn = stage_bnd_base + 256;
#pragma unroll
            for (int r = 1; r <= 4; ++r) {
                stage_issue_rows<NOP>(cb, stage_gofs, stage_lds_addr + 1024 + (r & 3) * SLOT);
                stage_issue_bnd(cn, stage_gofs_n, stage_lds_addr + (r & 3) * 256);
                cb += 64 * NR;
                cn += 256;
            }
            PRALINE_VMCNT(0);
            float4 br1[NOP];
#pragma unroll
            for (int q = 0; q < NOP; ++q) {
                br1[q] = *reinterpret_cast<const float4 *>(stage_lds + 1024 + 1 * SLOT + stage_rd[q]);
                b0[q] = *reinterpret_cast<const float4 *>(stage_lds + 1024 + 2 * SLOT + stage_rd[q]);
            }
            p0 = *reinterpret_cast<const float2 *>(stage_lds + 1 * 256 + stage_rd_bnd);
#pragma unroll
            for (int k = 0; k < NTERM * NR; ++k) {
                const int term = (NTERM == 1) ? 2 : k / NR;
                const int r = k % NR;
                const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
                const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(br1[ib]), accA, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DM) {
#pragma unroll
                for (int q = 0; q < NOP; ++q) b2[q] = br1[q];   // row 1 again: the upper half's share of the next tile
            }
            // row 5 takes row 1's slot (its operands are in the accumulator now)
            stage_issue_rows<NOP>(cb, stage_gofs, stage_lds_addr + 1024 + 1 * SLOT);
            cb += 64 * NR;
            stage_cur[0] = cb;  // next: operand row 6 (step 1)
            stage_cur[1] = cn;  // next: boundary row 5 (step 1)
        } else if constexpr (DENSE) {
            // rows 1, 2, 3 of the strip (the upper half takes rows 0, 1, 2: its first step is undone below); four register
            // sets rotate, the step at T fetches row T + 3: a row's loads have three steps (~1 us) to arrive
            dense_lane = dense_task + (int64_t)s * dense_strip;
#pragma unroll
            for (int r = 1; r <= 3; ++r) {
                const f4n *q = reinterpret_cast<const f4n *>(dense_lane + (int64_t)r * 4096);
                const f4n a0 = PRALINE_DENSE_LOAD(q), a1 = PRALINE_DENSE_LOAD(q + 1), a2 = PRALINE_DENSE_LOAD(q + 2), a3 = PRALINE_DENSE_LOAD(q + 3);
                f32x16 &d = r == 1 ? accA : (r == 2 ? accB : accC);
                d[0] = a0.x; d[1] = a0.y; d[2] = a0.z; d[3] = a0.w; d[4] = a1.x; d[5] = a1.y; d[6] = a1.z; d[7] = a1.w;
                d[8] = a2.x; d[9] = a2.y; d[10] = a2.z; d[11] = a2.w; d[12] = a3.x; d[13] = a3.y; d[14] = a3.z; d[15] = a3.w;
            }
            p0 = *reinterpret_cast<const float2 *>(my_bnd + BROW);      // row 1
            p1 = *reinterpret_cast<const float2 *>(my_bnd + 2 * BROW);  // row 2
            p2 = *reinterpret_cast<const float2 *>(my_bnd + 3 * BROW);  // row 3
        } else if constexpr (LOOKUP) {
            // this strip's table: lane (c = lane & 31, half hh) transposes the hi pieces of half hh of the pre-multiplied
            // row x0 + c (exact mode: Q2 = hi exactly), k = 16 r + 8 hh + jj  ->  lookup_tab[k][c]
            {
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
            }
            // symbols: byte r - 1 of the stream is the symbol of row r.  At step T the lower half fetches the scores of row
            // T + 1 (byte T), the upper half those of row T (byte T - 1): the upper half works on the stream shifted by one
            // byte, so both take byte T - 12 i of their 16-byte window (the window starts at byte 12 i).
            lw0 = pSym[0]; lw1 = pSym[1]; lw2 = pSym[2]; lw3 = pSym[3];
            {
                const unsigned first = h ? (lw0 << 8) : lw0;
                const float4 *q = reinterpret_cast<const float4 *>(onehot_lane + (first & 0xffu) * lookup_stride());
                const float4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                accA[0] = a0.x; accA[1] = a0.y; accA[2] = a0.z; accA[3] = a0.w; accA[4] = a1.x; accA[5] = a1.y; accA[6] = a1.z; accA[7] = a1.w;
                accA[8] = a2.x; accA[9] = a2.y; accA[10] = a2.z; accA[11] = a2.w; accA[12] = a3.x; accA[13] = a3.y; accA[14] = a3.z; accA[15] = a3.w;
            }
            p0 = *reinterpret_cast<const float2 *>(my_bnd + BROW);      // row 1
            p1 = *reinterpret_cast<const float2 *>(my_bnd + 2 * BROW);  // row 2
            p2 = *reinterpret_cast<const float2 *>(my_bnd + 3 * BROW);  // row 3
        } else {
            float4 br1[NOP];
            const float4 *s1, *s2, *s3, *s4;
            if constexpr (ONEHOT) {
                const unsigned w0 = pSym[0];
                d1 = pSym[1]; d2 = pSym[2]; d3 = pSym[3]; d4 = pSym[4];
                s1 = reinterpret_cast<const float4 *>(onehot_lane + (w0 & 0xffu) * onehot_stride(NR));
                s2 = reinterpret_cast<const float4 *>(onehot_lane + ((w0 >> 8) & 0xffu) * onehot_stride(NR));
                s3 = reinterpret_cast<const float4 *>(onehot_lane + ((w0 >> 16) & 0xffu) * onehot_stride(NR));
                s4 = reinterpret_cast<const float4 *>(onehot_lane + (w0 >> 24) * onehot_stride(NR));
            } else {
                s1 = reinterpret_cast<const float4 *>(pB);
                s2 = reinterpret_cast<const float4 *>(pB + b_stride);
                s3 = reinterpret_cast<const float4 *>(pB + 2 * b_stride);
                s4 = reinterpret_cast<const float4 *>(pB + 3 * b_stride);
            }
#pragma unroll
            for (int q = 0; q < NOP; ++q) { br1[q] = s1[q]; b0[q] = s2[q]; b1[q] = s3[q]; b2[q] = s4[q]; b3[q] = s1[q]; }
#pragma unroll
            for (int k = 0; k < NTERM * NR; ++k) {
                const int term = (NTERM == 1) ? 2 : k / NR;
                const int r = k % NR;
                const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
                const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(aop[ia]), as_half8(br1[ib]), accA, 0, 0, 0);
            }
            p0 = *reinterpret_cast<const float2 *>(my_bnd + BROW);      // row 1
            p1 = *reinterpret_cast<const float2 *>(my_bnd + 2 * BROW);  // row 2
            p2 = *reinterpret_cast<const float2 *>(my_bnd + 3 * BROW);  // row 3
        }

#define PRALINE_STEP16X(T, CUR, PREV, BSET, PSLOT, SYMW, SB)                                                          \
        split16_step<NR, NTERM, LOCAL, BSRC, SB, false, SNAPBR>((T) - h, L1, have_pair, h, CUR, PREV, BSET, aop, aop, BSET, BSET, b_next, b_stride, bnd_ld, \
                                       bnd_st, PSLOT, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,  \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2, onehot_lane, SYMW, \
                                       nullptr, nullptr, 0u, nullptr, nullptr, 0u, 0u, (T) >= min_l1)
#define PRALINE_STEP16(T, CUR, PREV, BSET, PSLOT) PRALINE_STEP16X(T, CUR, PREV, BSET, PSLOT, d1, 0)
        // match-score lookup: CUR holds this row's scores, PREV receives the next row's (symbol SYM)
#define PRALINE_STEP16L(T, CUR, PREV, PSLOT, SYM)                                                                     \
        split16_step<NR, NTERM, LOCAL, 3, 0, false, SNAPBR>((T) - h, L1, have_pair, h, CUR, PREV, b0, aop, aop, b0, b0, b_next, b_stride, bnd_ld, \
                                       bnd_st, PSLOT, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,  \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2, onehot_lane, SYM, \
                                       nullptr, nullptr, 0u, nullptr, nullptr, 0u, 0u, (T) >= min_l1)
        // dense tile: CUR holds this row's scores, PREV receives those of the row three steps on (lower half: row T + 3)
#define PRALINE_STEP16M(T, CUR, PREV, PSLOT)                                                                          \
        split16_step<NR, NTERM, LOCAL, 4, 0, false, SNAPBR>((T) - h, L1, have_pair, h, CUR, PREV, b0, aop, aop, b0, b0, b_next, b_stride, bnd_ld, \
                                       bnd_st, PSLOT, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,  \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2,           \
                                       dense_lane + (int64_t)((T) + 3) * 4096, 0u,                                     \
                                       nullptr, nullptr, 0u, nullptr, nullptr, 0u, 0u, (T) >= min_l1)
        // one-hot table with DM: BUSE holds row T+1, BOLD row T (refilled with row T+4 once its MFMAs are issued)
#define PRALINE_STEP16XD(T, CUR, PREV, BUSE, BOLD, PSLOT, SYMW, SB)                                                   \
        split16_step<NR, NTERM, LOCAL, BSRC, SB, true, SNAPBR>((T) - h, L1, have_pair, h, CUR, PREV, BUSE, aop, aopH, BOLD, BOLD, b_next, b_stride, bnd_ld, \
                                       bnd_st, PSLOT, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,  \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2, onehot_lane, SYMW, \
                                       nullptr, nullptr, 0u, nullptr, nullptr, 0u, 0u, (T) >= min_l1)
        // staged stream: BUSE holds row T+1, BFILL receives row T+2, PH = T % 4
#define PRALINE_STEP16D(T, CUR, PREV, BUSE, BFILL, PH, BOLD) PRALINE_STEP16Y(T, CUR, PREV, BUSE, BFILL, PH, BOLD, DM)
#define PRALINE_STEP16Y(T, CUR, PREV, BUSE, BFILL, PH, BOLD, DMF)                                                     \
        split16_step<NR, NTERM, LOCAL, 2, PH, DMF, SNAPBR>((T) - h, L1, have_pair, h, CUR, PREV, BUSE, aop, aopH, BOLD, BFILL, b_next, b_stride, bnd_ld,  \
                                       bnd_st, p0, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,     \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2, onehot_lane, 0u, \
                                       stage_lds, stage_rd, stage_rd_bnd, &stage_gofs, stage_cur, stage_lds_addr, stage_gofs_n, \
                                       (T) >= min_l1)
#define PRALINE_STEP16K(T, CUR, PREV, BUSE, BFILL, PH, KP)                                                            \
        split16_step<NR, NTERM, LOCAL, 2, PH, false, SNAPBR, KP>((T) - h, L1, have_pair, h, CUR, PREV, BUSE, aop, aopH, BUSE, BFILL, b_next, b_stride, bnd_ld,  \
                                       bnd_st, p0, Hs, Uc, dH, hd_x, l_x, best_run, col_run, out_best, out_rowmax,     \
                                       out_colmax, out_corner, go, ge, semiglobal, last_owner, cidx, xb, L2, onehot_lane, 0u, \
                                       stage_lds, stage_rd, stage_rd_bnd, &stage_gofs, stage_cur, stage_lds_addr, stage_gofs_n, \
                                       (T) >= min_l1, &ks)
        // KEEP: the two steps in 32 whose rows (lower half T, upper half T - 1) include a checkpoint row take the
        // instance with the checkpoint stores (wave-uniform test)
#define PRALINE_STEP16S(T, CUR, PREV, BUSE, BFILL, PH)                                                                \
        do {                                                                                                          \
            if constexpr (KEEP) {                                                                                     \
                if ((T) >= 32 && ((T) & 31) <= 1) PRALINE_STEP16K(T, CUR, PREV, BUSE, BFILL, PH, 2);                   \
                else PRALINE_STEP16K(T, CUR, PREV, BUSE, BFILL, PH, 1);                                               \
            } else PRALINE_STEP16Y(T, CUR, PREV, BUSE, BFILL, PH, BUSE, false);                                       \
        } while (0)
        // step 1: only the lower half has a row; the upper half's garbage is undone right after
        {
            float Hsave[17];
#pragma unroll
            for (int c = 0; c < 17; ++c) Hsave[c] = Hs[c];
            const float best_s = best_run, col_s = col_run;
            if constexpr (DENSE) PRALINE_STEP16M(1, accA, accD, p0);
            else if constexpr (LOOKUP) PRALINE_STEP16L(1, accA, accB, p0, ((h ? (lw0 << 8) : lw0) >> 8) & 0xffu);
            else if constexpr (DM && ONEHOT) PRALINE_STEP16XD(1, accA, accB, b0, b3, p0, d1, 0);
            else if constexpr (DM) PRALINE_STEP16D(1, accA, accB, b0, b1, 1, b2);
            else if constexpr (STAGED) PRALINE_STEP16S(1, accA, accB, b0, b1, 1);
            else PRALINE_STEP16(1, accA, accB, b0, p0);
            if (h) {
#pragma unroll
                for (int c = 0; c < 17; ++c) Hs[c] = Hsave[c];
#pragma unroll
                for (int c = 0; c < 16; ++c) Uc[c] = PRALINE_NEG_INF;
                best_run = best_s;
                col_run = col_s;
            }
        }
        // steps 2 .. max_l1 + 1: the accumulators ping-pong (period 2), the B operand sets and the boundary
        // prefetch slots rotate (period 3; staged stream: operand sets period 2, ring slots period 4);
        // steps past max_l1 + 1 compute rows that nobody reports
        if constexpr (STAGED) {
            for (int t = 2; t <= max_l1 + 1; t += 12) {
                if constexpr (MW) {
                    if (share > 1) { __builtin_amdgcn_s_barrier(); --mw_left; }
                }
                if constexpr (DM) {
                    // operand sets: (row t, row t+1, row t+2 arriving) rotate with period 3
                    PRALINE_STEP16D(t, accB, accA, b1, b2, 2, b0);
                    PRALINE_STEP16D(t + 1, accA, accB, b2, b0, 3, b1);
                    PRALINE_STEP16D(t + 2, accB, accA, b0, b1, 0, b2);
                    PRALINE_STEP16D(t + 3, accA, accB, b1, b2, 1, b0);
                    PRALINE_STEP16D(t + 4, accB, accA, b2, b0, 2, b1);
                    PRALINE_STEP16D(t + 5, accA, accB, b0, b1, 3, b2);
                    PRALINE_STEP16D(t + 6, accB, accA, b1, b2, 0, b0);
                    PRALINE_STEP16D(t + 7, accA, accB, b2, b0, 1, b1);
                    PRALINE_STEP16D(t + 8, accB, accA, b0, b1, 2, b2);
                    PRALINE_STEP16D(t + 9, accA, accB, b1, b2, 3, b0);
                    PRALINE_STEP16D(t + 10, accB, accA, b2, b0, 0, b1);
                    PRALINE_STEP16D(t + 11, accA, accB, b0, b1, 1, b2);
                } else {
                PRALINE_STEP16S(t, accB, accA, b1, b0, 2);
                PRALINE_STEP16S(t + 1, accA, accB, b0, b1, 3);
                PRALINE_STEP16S(t + 2, accB, accA, b1, b0, 0);
                PRALINE_STEP16S(t + 3, accA, accB, b0, b1, 1);
                PRALINE_STEP16S(t + 4, accB, accA, b1, b0, 2);
                PRALINE_STEP16S(t + 5, accA, accB, b0, b1, 3);
                PRALINE_STEP16S(t + 6, accB, accA, b1, b0, 0);
                PRALINE_STEP16S(t + 7, accA, accB, b0, b1, 1);
                PRALINE_STEP16S(t + 8, accB, accA, b1, b0, 2);
                PRALINE_STEP16S(t + 9, accA, accB, b0, b1, 3);
                PRALINE_STEP16S(t + 10, accB, accA, b1, b0, 0);
                PRALINE_STEP16S(t + 11, accA, accB, b0, b1, 1);
                }
            }
        } else if constexpr (DENSE) {
            for (int t = 2; t <= max_l1 + 1; t += 12) {
                if constexpr (MW) {
                    // shared task: the ranks meet every 12 rows; this wave's boundary rows of more than three steps ago are
                    // complete (a step issues six memory operations, they retire in order) - the next rank, two iterations
                    // behind, asks for nothing younger; the tile rows in flight stay in flight
                    if (share > 1) { PRALINE_VMCNT(18); __builtin_amdgcn_s_barrier(); --mw_left; }
                }
                PRALINE_STEP16M(t, accB, accA, p1);
                PRALINE_STEP16M(t + 1, accC, accB, p2);
                PRALINE_STEP16M(t + 2, accD, accC, p0);
                PRALINE_STEP16M(t + 3, accA, accD, p1);
                PRALINE_STEP16M(t + 4, accB, accA, p2);
                PRALINE_STEP16M(t + 5, accC, accB, p0);
                PRALINE_STEP16M(t + 6, accD, accC, p1);
                PRALINE_STEP16M(t + 7, accA, accD, p2);
                PRALINE_STEP16M(t + 8, accB, accA, p0);
                PRALINE_STEP16M(t + 9, accC, accB, p1);
                PRALINE_STEP16M(t + 10, accD, accC, p2);
                PRALINE_STEP16M(t + 11, accA, accD, p0);
            }
        } else if constexpr (LOOKUP) {
            const unsigned *pn = pSym + 4;
            for (int t = 2; t <= max_l1 + 1; t += 12) {
                if constexpr (MW) {
                    // shared task: the ranks meet here every 12 rows; the boundary rows this wave has stored are complete
                    // (in L2) before the next rank, two iterations behind, can ask for them
                    if (share > 1) { PRALINE_VMCNT(0); __builtin_amdgcn_s_barrier(); --mw_left; }
                }
                const unsigned n1 = pn[0], n2 = pn[1], n3 = pn[2];   // the next window's new dwords
                pn += 3;
                // the upper half's window is the stream shifted by one byte (see above)
                const unsigned W0 = h ? (lw0 << 8) : lw0;
                const unsigned W1 = h ? __builtin_amdgcn_alignbit(lw1, lw0, 24) : lw1;
                const unsigned W2 = h ? __builtin_amdgcn_alignbit(lw2, lw1, 24) : lw2;
                const unsigned W3 = h ? __builtin_amdgcn_alignbit(lw3, lw2, 24) : lw3;
                PRALINE_STEP16L(t, accB, accA, p1, (W0 >> 16) & 0xffu);
                PRALINE_STEP16L(t + 1, accA, accB, p2, W0 >> 24);
                PRALINE_STEP16L(t + 2, accB, accA, p0, W1 & 0xffu);
                PRALINE_STEP16L(t + 3, accA, accB, p1, (W1 >> 8) & 0xffu);
                PRALINE_STEP16L(t + 4, accB, accA, p2, (W1 >> 16) & 0xffu);
                PRALINE_STEP16L(t + 5, accA, accB, p0, W1 >> 24);
                PRALINE_STEP16L(t + 6, accB, accA, p1, W2 & 0xffu);
                PRALINE_STEP16L(t + 7, accA, accB, p2, (W2 >> 8) & 0xffu);
                PRALINE_STEP16L(t + 8, accB, accA, p0, (W2 >> 16) & 0xffu);
                PRALINE_STEP16L(t + 9, accA, accB, p1, W2 >> 24);
                PRALINE_STEP16L(t + 10, accB, accA, p2, W3 & 0xffu);
                PRALINE_STEP16L(t + 11, accA, accB, p0, (W3 >> 8) & 0xffu);
                lw0 = lw3; lw1 = n1; lw2 = n2; lw3 = n3;
            }
        } else if constexpr (ONEHOT) {
            // twelve per iteration: the step at t refills the operands of row t + 4, i.e. symbol t + 3 of
            // the sequence; t = 2 (mod 12), so the twelve symbols are bytes 1..3 of d1, d2, d3 and byte 0
            // of d4 - d4 becomes the next iteration's d1, three new dwords arrive per iteration.
            const unsigned *pn = pSym + 5;
            for (int t = 2; t <= max_l1 + 1; t += 12) {
                const unsigned n2 = pn[0], n3 = pn[1], n4 = pn[2];
                pn += 3;
                if constexpr (DM) {
                    PRALINE_STEP16XD(t, accB, accA, b1, b0, p1, d1, 1);
                    PRALINE_STEP16XD(t + 1, accA, accB, b2, b1, p2, d1, 2);
                    PRALINE_STEP16XD(t + 2, accB, accA, b3, b2, p0, d1, 3);
                    PRALINE_STEP16XD(t + 3, accA, accB, b0, b3, p1, d2, 0);
                    PRALINE_STEP16XD(t + 4, accB, accA, b1, b0, p2, d2, 1);
                    PRALINE_STEP16XD(t + 5, accA, accB, b2, b1, p0, d2, 2);
                    PRALINE_STEP16XD(t + 6, accB, accA, b3, b2, p1, d2, 3);
                    PRALINE_STEP16XD(t + 7, accA, accB, b0, b3, p2, d3, 0);
                    PRALINE_STEP16XD(t + 8, accB, accA, b1, b0, p0, d3, 1);
                    PRALINE_STEP16XD(t + 9, accA, accB, b2, b1, p1, d3, 2);
                    PRALINE_STEP16XD(t + 10, accB, accA, b3, b2, p2, d3, 3);
                    PRALINE_STEP16XD(t + 11, accA, accB, b0, b3, p0, d4, 0);
                } else {
                PRALINE_STEP16X(t, accB, accA, b1, p1, d1, 1);
                PRALINE_STEP16X(t + 1, accA, accB, b2, p2, d1, 2);
                PRALINE_STEP16X(t + 2, accB, accA, b0, p0, d1, 3);
                PRALINE_STEP16X(t + 3, accA, accB, b1, p1, d2, 0);
                PRALINE_STEP16X(t + 4, accB, accA, b2, p2, d2, 1);
                PRALINE_STEP16X(t + 5, accA, accB, b0, p0, d2, 2);
                PRALINE_STEP16X(t + 6, accB, accA, b1, p1, d2, 3);
                PRALINE_STEP16X(t + 7, accA, accB, b2, p2, d3, 0);
                PRALINE_STEP16X(t + 8, accB, accA, b0, p0, d3, 1);
                PRALINE_STEP16X(t + 9, accA, accB, b1, p1, d3, 2);
                PRALINE_STEP16X(t + 10, accB, accA, b2, p2, d3, 3);
                PRALINE_STEP16X(t + 11, accA, accB, b0, p0, d4, 0);
                }
                d1 = d4; d2 = n2; d3 = n3; d4 = n4;
            }
        } else {
            for (int t = 2; t <= max_l1 + 1; t += 6) {
                PRALINE_STEP16(t, accB, accA, b1, p1);
                PRALINE_STEP16(t + 1, accA, accB, b2, p2);
                PRALINE_STEP16(t + 2, accB, accA, b0, p0);
                PRALINE_STEP16(t + 3, accA, accB, b1, p1);
                PRALINE_STEP16(t + 4, accB, accA, b2, p2);
                PRALINE_STEP16(t + 5, accA, accB, b0, p0);
            }
        }
#undef PRALINE_STEP16
#undef PRALINE_STEP16L
#undef PRALINE_STEP16M
#undef PRALINE_STEP16S
#undef PRALINE_STEP16K
#undef PRALINE_STEP16D
#undef PRALINE_STEP16Y
#undef PRALINE_STEP16X
#undef PRALINE_STEP16XD
    }

    if constexpr (STAGED) PRALINE_VMCNT(0);  // no DMA may be in flight when the wave ends
#ifdef PRALINE_TRACE
    if (praline_trace_buf != nullptr && (threadIdx.x & 63) == 0) {
        unsigned hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long *rec = praline_trace_buf + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 6;
        rec[0] = blockIdx.x; rec[1] = (threadIdx.x >> 6) | ((unsigned)share << 8) | ((unsigned)task << 16);
        rec[2] = hw_id; rec[3] = xcc_id; rec[4] = trace_t0; rec[5] = __builtin_amdgcn_s_memtime();
    }
#endif

    if constexpr (KEEP) {
        // global end state: the first of (M, U, L) of the corner cell that equals its maximum (np.argmax, praline/component/align.py:428-430);
        // only the half (and rank) that owns the corner holds a finite out_corner - the others report 0
        float kf = 0.0f;
        if (out_corner != PRALINE_NEG_INF) kf = (ks.snap_m == out_corner) ? 0.0f : ((ks.snap_u == out_corner) ? 1.0f : 2.0f);
        out_best = kf;   // (not LOCAL: the slot is free; folded with max like the other results)
    }
    float corner_all = __builtin_fmaxf(out_corner, partner_value(out_corner, h));
    float rowmax_all = __builtin_fmaxf(out_rowmax, partner_value(out_rowmax, h));
    float colmax_all = __builtin_fmaxf(out_colmax, partner_value(out_colmax, h));
    float best_all = __builtin_fmaxf(out_best, partner_value(out_best, h));
    if constexpr (MW) {
        if (share > 1) {
            for (; mw_left > 0; --mw_left) __builtin_amdgcn_s_barrier();
            // every result is a maximum over strips: rank 0 folds the other ranks' shares in
            if (rank > 0 && h == 0) {
                float *o = mw_out + wv * 128;
                o[j] = corner_all; o[32 + j] = rowmax_all; o[64 + j] = colmax_all; o[96 + j] = best_all;
            }
            __syncthreads();
            if (rank > 0) return;
            for (int r = 1; r < share; ++r) {
                const float *o = mw_out + (wv + r) * 128;
                corner_all = __builtin_fmaxf(corner_all, o[j]);
                rowmax_all = __builtin_fmaxf(rowmax_all, o[32 + j]);
                colmax_all = __builtin_fmaxf(colmax_all, o[64 + j]);
                best_all = __builtin_fmaxf(best_all, o[96 + j]);
            }
        }
    }
    if (have_pair && h == 0) {
        float score;
        if (LOCAL) score = best_all;
        else if (semiglobal) score = (rowmax_all > colmax_all && free_two) ? rowmax_all : colmax_all;
        else score = corner_all;
        scores[lane_pair[base + j]] = score;
        if constexpr (KEEP) {
            int32_t *ec = end_cells + (int64_t)lane_pair[base + j] * 4;
            ec[0] = L1; ec[1] = L2; ec[2] = (int)best_all; ec[3] = 0;
        }
    }
}

// ---- arena side: split the fp32 parity-layout operands into f16 pieces ----------------------
// src: P or Q in the fp32 layout [rowp][hh][KS] (k = 2 s + hh).  dst: [rowp][hh][piece][r][8 halves],
// slot (piece, r) element jj holds k = 16 r + 8 hh + jj.  flag[0] is set when any value needs a
// non-zero lo piece (i.e. the exact single-term mode is not applicable).
#ifdef PRALINE_SPLIT16_AUX   // non-template kernel: defined in dp_split16_instance.hip's translation unit only
__global__ void k_split_f16(const float *__restrict__ src, int KP, int KS, int n_active, int NR, int64_t rows_pad,
                            _Float16 *__restrict__ dst, int *__restrict__ flag)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = 2 * NR * 8;  // (hh, r, jj) triples per row
    if (idx >= rows_pad * per_row) return;  // (total is a multiple of 64: whole waves leave together)
    const bool inexact = split_f16_entry(src, KP, KS, n_active, NR, idx, dst);
    // flag == nullptr: the representability check was already done when the arena was created
    if (flag != nullptr) {
        // (one atomic per arena, not one per wave: with float profiles every wave has something to report, and 50 000
        // atomics on one word took 0.55 ms per call - a quarter of C2's host-to-host time)
        if (__ballot(inexact) != 0ull && (threadIdx.x & 63) == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            atomicOr(flag, 1);
    }
}
#endif

// Dense match-score matrix of one pair with EXACTLY the arithmetic of k_dp_split16 (diagnostics /
// tests: the DP is verified bit-for-bit on these values).  One wave per 32x32 tile.
template <int NR, int NTERM>
__global__ __launch_bounds__(64) void k_scores_tile16(Arena16Dev ar, int one, int two, float *__restrict__ m)
{
    constexpr int NP = (NTERM == 1) ? 1 : 2;
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int L1 = ar.len[one], L2 = ar.len[two];
    const int y0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    // B operand = profile rows of `one` (columns of the tile = rows y), A operand = Q rows of `two`
    const float4 *pb = reinterpret_cast<const float4 *>(ar.P16 + ((int64_t)ar.row_off[one] + y0 + j) * ar.row_bytes + h * ar.half_bytes);
    const float4 *qa = reinterpret_cast<const float4 *>(ar.Q16 + ((int64_t)ar.row_off[two] + x0 + j) * ar.row_bytes + h * ar.half_bytes);
    float4 a[NP * NR], b[NP * NR];
#pragma unroll
    for (int q = 0; q < NP * NR; ++q) { a[q] = qa[q]; b[q] = pb[q]; }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NTERM * NR; ++k) {
        const int term = (NTERM == 1) ? 2 : k / NR;
        const int r = k % NR;
        const int ia = (NTERM == 2) ? k : ((term == 0) ? NR + r : r);
        const int ib = (NTERM == 2) ? k : ((term == 1) ? NR + r : r);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(a[ia]), as_half8(b[ib]), acc, 0, 0, 0);
    }
    // D[i][jcol]: lane holds column jcol = j (row y0 + j of the DP), i = x within the strip
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int x = x0 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        const int y = y0 + j;
        if (y < L1 && x < L2) m[(int64_t)y * L2 + x] = acc[rr];
    }
}
