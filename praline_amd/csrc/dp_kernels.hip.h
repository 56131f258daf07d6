// dp_kernels.hip.h -- gfx950 (MI355X / CDNA4) device code of libpraline_dp.so.
//
// Hot path restated for the device (reference lines relative to /root/reference):
//   * match scores  m = sum_sets P1 . S . P2^T        praline/util/cext.c:308-455 (inner 33-97)
//   * 3-state affine fill M/U/L + traceback flags     praline/util/cext.c:99-306
//   * boundary init / end cell / traceback            praline/component/align.py:357-431,
//                                                     praline/util/align.py:144-185,268-297
//
// This header holds what every translation unit shares: the device-side views, the boundary-cell formulas, the f16 hi/lo
// operand split, and - under PRALINE_AUX_KERNELS, for praline_dp.hip alone - the non-template kernels around the fill:
// profile packing and the S pre-multiply, the fp32-MFMA match-score tiles, the end cells of the semiglobal modes, the
// traceback walk over the packed flag planes, the mask words of many-rectangle plans, the one-cell-per-thread match scores
// in the reference's summation order, the merge of aligned profiles.  The batched fills themselves are in dp_split16.hip.h
// (scores), dp_split16_tb.hip.h (with flags), dp_pipe.hip.h, dp_quad.hip.h and dp_trace2.hip.h.
//
// Raw kernels (k_raw_*): the reference's own buffer layout (m, g1, g2, o, t, z), one pair, a
// single wavefront walking anti-diagonals: lane l owns column x0+l, computes row t-l at step t
// and receives its left/diagonal neighbours from lane l-1 with __shfl_up.
#pragma once
#include "dp_types.h"
#include "dp_arena16.h"   // PRALINE_DENSE_PAD
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PRALINE_NEG_INF (-__builtin_inff())

typedef float f32x16 __attribute__((ext_vector_type(16)));

// --------------------------------------------------------------------------------------------
// Device-side views
// --------------------------------------------------------------------------------------------
struct ArenaDev {
    const float *P;          // [rows_pad][KP]  parity-split profiles (B operand, sequence one)
    const float *Q;          // [rows_pad][KP]  parity-split Q2 = P . S^T (A operand, sequence two)
    const int32_t *row_off;  // [n_seqs] first padded row of each sequence
    const int32_t *len;      // [n_seqs]
    int KP;                  // floats per row  (2 * KS)
    int KS;                  // floats per parity half (multiple of 4)
};


struct RunParams {
    int mode;
    float go1, ge1;  // gap open / extend of gap_score_model_one (state U, cext.c:155-158)
    float go2, ge2;  // gap open / extend of gap_score_model_two (state L, cext.c:172-175)
    // per-position gap scores (GapScoreModel, praline/container/score.py:45-68; cext.c:155-158,172-175 read g1[y-1],
    // g2[x-1]): float [rows_pad][2] = (open, extend) of every arena row, indexed like the padded profile rows
    // (ArenaDev::row_off[seq] + position); nullptr: the constant scores above
    const float *gaps = nullptr;
};

struct RectList {
    const int32_t *rect_off;  // [n_pairs + 1] or nullptr
    const int32_t *rects;     // [.][4] y0,y1,x0,x1 inclusive DP coordinates
    // plans in which some pair has more than PRALINE_MAX_RECTS rectangles (dense-tile instances, k_dp_quad_tb): per pair, strip and
    // DP row the 32-bit mask of the strip's zeroed columns, prepared by k_build_zmask
    const unsigned *zmask = nullptr;
    const int64_t *zm_off = nullptr;   // [n_pairs]
};

__device__ __forceinline__ bool mode_free_one(int mode) { return mode == 2 || mode == 3; }
__device__ __forceinline__ bool mode_free_two(int mode) { return mode == 2 || mode == 4; }

// Boundary value o[y,0,1] / o[0,x,2] for idx >= 1 (praline/component/align.py:375,383):
// float64 arithmetic, one rounding to float32.
__device__ __forceinline__ float boundary_value(int idx, float go, float ge, bool is_free)
{
    return is_free ? 0.0f : (float)((double)(idx - 1) * (double)ge + (double)go);
}

// The same with per-position gap scores g = (open, extend) rows of the sequence (align.py:371-385):
// o[0,0,k] = open[0] - extend[0];  o[idx,0,1] / o[0,idx,2] = (idx - 1) * extend[idx - 1] + open[0]  (float64, one rounding).
__device__ __forceinline__ float boundary_value_pp(int idx, const float *g, bool is_free)
{
    if (is_free) return 0.0f;
    const int k = idx > 0 ? idx - 1 : 0;
    return (float)((double)(idx - 1) * (double)g[2 * k + 1] + (double)g[0]);
}

// Where the per-cell match-score kernels (k_match_ref, k_match_reft, k_scores_tile_batch) write: loc == nullptr - pair p's own
// [L1][L2] matrix at m + m_off[p]; otherwise the dense tile of the task the pair runs in (split-strip layout, dp_split16.hip.h:
// float [strip - strip_lo][row 1 .. max_l1 (+ PRALINE_DENSE_PAD)][half][lane 0..31][16 columns] at m + dense_off[task]), just the
// strips [strip_lo, strip_lo + strip_cnt) of it.  The columns between the end of sequence two and the end of its last strip
// receive zeros (a local alignment must not see stale scores there); rows past a lane's own sequence are never reported.
struct TileOut {
    const PairLoc *loc = nullptr;      // [n_pairs]
    const WaveTask *tasks = nullptr;   // the plan's task list
    const int64_t *dense_off = nullptr;   // [n_tasks], float offsets
    int strip_lo = 0, strip_cnt = 0x3fffffff;
};
struct TileDst {
    float *base;    // the task's tile, at this lane's 16-column group of row 0 / strip strip_lo / half 0
    int64_t strip_floats;
    int x_lo, x_hi;   // the columns (0-based positions of sequence two) this launch covers: whole strips
    __device__ __forceinline__ float *at(int y, int x) const   // y, x: 0-based positions
    {
        const int xr = x - x_lo;
        return base + (int64_t)(xr >> 5) * strip_floats + (int64_t)(y + 1) * 1024 + ((xr >> 4) & 1) * 512 + (xr & 15);
    }
};
__device__ __forceinline__ TileDst tile_dst(const TileOut &to, float *m, int p, int L2)
{
    const PairLoc pl = to.loc[p];
    const WaveTask tk = to.tasks[pl.task];
    TileDst d;
    d.strip_floats = (int64_t)(tk.max_l1 + PRALINE_DENSE_PAD) * 1024;
    d.base = m + to.dense_off[pl.task] + (int64_t)pl.lane * 16;
    const int nstrips = (L2 + 31) >> 5;
    const int s1 = to.strip_cnt < nstrips - to.strip_lo ? to.strip_lo + to.strip_cnt : nstrips;
    d.x_lo = to.strip_lo * 32;
    d.x_hi = s1 * 32;
    if (d.x_hi < d.x_lo) d.x_hi = d.x_lo;
    return d;
}

__device__ __forceinline__ float max3f(float a, float b, float c)
{
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);  // folds to v_max3_f32
}

// One entry of the f16 hi/lo operand layout (dp_split16.hip.h): src is P or Q in the fp32 parity layout
// [rowp][hh][KS] (k = 2 s + hh); dst: [rowp][hh][piece][r][8 halves], slot (piece, r) element jj holds
// k = 16 r + 8 hh + jj.  Returns whether the value needs a non-zero lo piece (exact single-term mode not applicable).
__device__ __forceinline__ bool split_f16_entry(const float *__restrict__ src, int KP, int KS, int n_active, int NR,
                                                int64_t idx, _Float16 *__restrict__ dst)
{
    const int per_row = 2 * NR * 8;  // (hh, r, jj) triples per row
    const int64_t rowp = idx / per_row;
    const int rem = (int)(idx % per_row);
    const int hh = rem / (NR * 8), r = (rem / 8) % NR, jj = rem % 8;
    const int k = 16 * r + 8 * hh + jj;
    float v = 0.0f;
    if (k < n_active) v = src[rowp * KP + (k & 1) * KS + (k >> 1)];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int64_t half_elems = 2 * NR * 8;  // halves per (rowp, hh)
    _Float16 *o = dst + (rowp * 2 + hh) * half_elems;
    o[(0 * NR + r) * 8 + jj] = hi;
    o[(1 * NR + r) * 8 + jj] = lo;
    return (float)lo != 0.0f || (float)hi + (float)lo != v;
}

// The K-PACKED form of the three-term split (arenas with at most 21 active symbols, NR = 2): the three partial products
//     q_lo . p_hi  +  q_hi . p_lo  +  q_hi . p_hi          (dp_split16.hip.h)
// are laid out one after the other along the MFMA's k axis - 3 n <= 63 of the 64 k slots of FOUR 32x32x16 MFMAs
// instead of six (two 16-wide ranges per term, 12 of every 32 slots padding).  Measured on C2: the six MFMAs of a step
// cost 21 % of the scores kernel (an ablation build without them: 2.22 -> 1.76 ms) although the matrix pipe is only a
// third busy - fewer of them is the one lever left on a kernel that sits on its VALU floor.
// Packed slot q (0..3) of half hh, element jj holds packed index kp = 16 q + 8 hh + jj: term kp / n, symbol kp % n.
//   side 0 (A operand, Q = P . S^T): terms take lo, hi, hi;   side 1 (B operand, P): hi, lo, hi.
__device__ __forceinline__ void split_f16_entry_packed(const float *__restrict__ src, int KP, int KS, int n_active, int side,
                                                       int64_t idx, _Float16 *__restrict__ dst)
{
    const int64_t rowp = idx / 64;
    const int rem = (int)(idx % 64);
    const int hh = rem / 32, q = (rem / 8) % 4, jj = rem % 8;
    const int kp = 16 * q + 8 * hh + jj;
    _Float16 out = (_Float16)0.0f;
    if (kp < 3 * n_active) {
        const int term = kp / n_active, k = kp % n_active;
        const float v = src[rowp * KP + (k & 1) * KS + (k >> 1)];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        const bool want_lo = side == 0 ? term == 0 : term == 1;
        out = want_lo ? lo : hi;
    }
    dst[(rowp * 2 + hh) * 32 + q * 8 + jj] = out;
}

#ifdef PRALINE_AUX_KERNELS  // non-template kernels: defined in praline_dp.hip's translation unit only
// --------------------------------------------------------------------------------------------
// Arena packing + the profile x matrix pre-multiply
// --------------------------------------------------------------------------------------------
// raw: [rows][A] fp32 (host layout, sequences concatenated); P: parity-split, compacted to the
// active symbols: P[rowp][h][s] = raw[row][active[2s+h]], zero in padding rows / columns.
__device__ __forceinline__ void pack_entry(int64_t idx, const float *__restrict__ raw, const int32_t *__restrict__ seq_of_rowp,
                                           const int32_t *__restrict__ row_off_pad, const int32_t *__restrict__ row_off_raw,
                                           const int32_t *__restrict__ len, const int32_t *__restrict__ active, int n_active,
                                           int A, int KP, int KS, float *__restrict__ P)
{
    const int64_t rowp = idx / KP;
    const int c = (int)(idx % KP);
    const int h = c / KS, s = c % KS;
    const int k = 2 * s + h;
    float v = 0.0f;
    const int seq = seq_of_rowp[rowp];
    if (seq >= 0 && k < n_active) {
        const int r = (int)(rowp - row_off_pad[seq]);
        if (r < len[seq]) v = raw[(int64_t)(row_off_raw[seq] + r) * A + active[k]];
    }
    P[idx] = v;
}

// What arena creation needs to know about the raw profiles (praline_arena_create): per row its one-hot symbol (255: the
// row is not one-hot), and over all rows which symbols carry mass (flags[i], i < A) and how many rows are not one-hot
// (flags[A]).  One thread per row; the column flags are gathered per workgroup in LDS.
__global__ __launch_bounds__(256) void k_scan_profiles(const float *__restrict__ raw, int64_t rows, int A, unsigned char *__restrict__ sym_raw,
                                                       int *__restrict__ flags)
{
    __shared__ int col[256];
    __shared__ int not_hot;
    col[threadIdx.x] = 0;
    if (threadIdx.x == 0) not_hot = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        const float *row = raw + r * A;
        int nz = 0, hot = 0, ones = 0;
        for (int i = 0; i < A; ++i) {
            const float v = row[i];
            if (v != 0.0f) { ++nz; hot = i; if (!col[i]) col[i] = 1; }   // (benign race: every writer stores 1)
            ones += v == 1.0f;
        }
        const bool onehot = nz == 1 && ones == 1;
        sym_raw[r] = onehot ? (unsigned char)hot : (unsigned char)255;
        if (!onehot) atomicAdd(&not_hot, 1);
    }
    __syncthreads();
    if ((int)threadIdx.x < A && col[threadIdx.x]) atomicOr(&flags[threadIdx.x], 1);
    if (threadIdx.x == 0 && not_hot) atomicAdd(&flags[A], not_hot);
}

// one-hot arenas: sym8[padded row] = active-symbol slot of the row's symbol (slot_of[255] = "none" for padding rows)
__global__ void k_build_sym8(const unsigned char *__restrict__ sym_raw, const int32_t *__restrict__ seq_of_rowp,
                             const int32_t *__restrict__ row_off_pad, const int32_t *__restrict__ row_off_raw,
                             const int32_t *__restrict__ len, const unsigned char *__restrict__ slot_of, int64_t rows_pad,
                             int64_t rows_out, unsigned char *__restrict__ sym8)
{
    const int64_t rp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rp >= rows_out) return;
    unsigned char v = slot_of[255];
    if (rp < rows_pad) {
        const int s = seq_of_rowp[rp];
        if (s >= 0) {
            const int y = (int)(rp - row_off_pad[s]);
            if (y < len[s]) v = slot_of[sym_raw[row_off_raw[s] + y]];
        }
    }
    sym8[rp] = v;
}

__global__ void k_pack_profiles(const float *__restrict__ raw, const int32_t *__restrict__ seq_of_rowp,
                                const int32_t *__restrict__ row_off_pad,
                                const int32_t *__restrict__ row_off_raw,
                                const int32_t *__restrict__ len, const int32_t *__restrict__ active,
                                int n_active, int A, int KP, int KS, int64_t rows_pad, float *__restrict__ P)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows_pad * KP) return;
    pack_entry(idx, raw, seq_of_rowp, row_off_pad, row_off_raw, len, active, n_active, A, KP, KS, P);
}

// Q2[x][i] = sum_j S[i][j] * P2[x][j], an fp32 fmaf chain over j ascending, evaluated with
// v_mfma_f32_32x32x2_f32: one wave computes 32 rows x 32 (compacted) symbols.
//   A operand (lane l: row l&31, k = l>>5)  = raw[row0 + (l&31)][2s + (l>>5)]
//   B operand (lane l: k = l>>5, col l&31)  = S[active[c0 + (l&31)]][2s + (l>>5)]
// D[i][j]: lane holds column j = l&31 (symbol), rows i = (r&3) + 8(r>>2) + 4(l>>5).
__device__ __forceinline__ void premultiply_block(int64_t rowp0, int c0, const float *__restrict__ raw,
                                                     const float *__restrict__ S,
                                                     const int32_t *__restrict__ seq_of_rowp,
                                                     const int32_t *__restrict__ row_off_pad,
                                                     const int32_t *__restrict__ row_off_raw,
                                                     const int32_t *__restrict__ len,
                                                     const int32_t *__restrict__ active, int n_active,
                                                     int A, int KP, int KS, int64_t rows_pad,
                                                     float *__restrict__ Q)
{
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t rowp = rowp0 + j;
    const float *src = nullptr;
    if (rowp < rows_pad) {
        const int seq = seq_of_rowp[rowp];
        if (seq >= 0) {
            const int r = (int)(rowp - row_off_pad[seq]);
            if (r < len[seq]) src = raw + (int64_t)(row_off_raw[seq] + r) * A;
        }
    }
    const int sym = (c0 + j < n_active) ? active[c0 + j] : -1;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; 2 * s < A; ++s) {
        const int k = 2 * s + h;
        const float a = (src != nullptr && k < A) ? src[k] : 0.0f;
        const float b = (sym >= 0 && k < A) ? S[(int64_t)sym * A + k] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const int c = c0 + j;
    if (c < KP) {
        const int hh = c & 1, ss = c >> 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int64_t rp = rowp0 + i;
            if (rp < rows_pad) Q[rp * KP + hh * KS + ss] = (c < n_active) ? acc[r] : 0.0f;
        }
    }
}

__global__ __launch_bounds__(64) void k_premultiply(const float *__restrict__ raw, const float *__restrict__ S,
                                                     const int32_t *__restrict__ seq_of_rowp,
                                                     const int32_t *__restrict__ row_off_pad,
                                                     const int32_t *__restrict__ row_off_raw,
                                                     const int32_t *__restrict__ len,
                                                     const int32_t *__restrict__ active, int n_active, int A, int KP, int KS,
                                                     int64_t rows_pad, float *__restrict__ Q)
{
    premultiply_block((int64_t)blockIdx.x * 32, blockIdx.y * 32, raw, S, seq_of_rowp, row_off_pad, row_off_raw, len, active,
                      n_active, A, KP, KS, rows_pad, Q);
}

// Everything praline_arena_premultiply does, for 32 rows per wave in ONE launch: pack P, Q = P . S^T (MFMA), and the
// f16 hi/lo pieces of both (the four separate launches cost ~0.1 ms of a 2.3 ms bench step in launch gaps).
__global__ __launch_bounds__(256) void k_prepare_rows(const float *__restrict__ raw, const float *__restrict__ S,
                                                      const int32_t *__restrict__ seq_of_rowp,
                                                      const int32_t *__restrict__ row_off_pad,
                                                      const int32_t *__restrict__ row_off_raw,
                                                      const int32_t *__restrict__ len,
                                                      const int32_t *__restrict__ active, int n_active, int A, int KP, int KS,
                                                      int64_t rows_pad, float *__restrict__ P, float *__restrict__ Q, int NR,
                                                      _Float16 *__restrict__ P16, _Float16 *__restrict__ Q16, int64_t block0 = 0,
                                                      int packed = 0)
{
    // (four waves per 32 rows: the element-wise passes are chains of dependent loads - their length, not the traffic, is
    // what the launch lasts; the MFMA pre-multiply of a 32-column block is one wave's work)
    const int64_t rowp0 = ((int64_t)blockIdx.x + block0) * 32;   // block0: first 32-row block (appended sequences only)
    const int nthr = (int)blockDim.x, wave = (int)(threadIdx.x >> 6);
    {
        // pack_entry for the block's 32 rows: they belong to ONE sequence (sequences are padded to multiples of 32 rows),
        // so the row bookkeeping is looked up once
        const int seq = seq_of_rowp[rowp0];
        const int r0 = seq >= 0 ? (int)(rowp0 - row_off_pad[seq]) : 0;
        const int n_rows = seq >= 0 ? len[seq] - r0 : 0;   // rows of the block that hold profile rows
        const float *src = raw + (seq >= 0 ? (int64_t)(row_off_raw[seq] + r0) * A : 0);
        for (int i = threadIdx.x; i < 32 * KP; i += nthr) {
            const int r = i / KP, c = i % KP;
            const int k = 2 * (c % KS) + c / KS;
            P[rowp0 * KP + i] = (r < n_rows && k < n_active) ? src[(int64_t)r * A + active[k]] : 0.0f;
        }
    }
    for (int c0 = 32 * wave; c0 < KP; c0 += 32 * (nthr >> 6))
        premultiply_block(rowp0, c0, raw, S, seq_of_rowp, row_off_pad, row_off_raw, len, active, n_active, A, KP, KS, rows_pad, Q);
    if (NR > 0) {
        __syncthreads();   // the block's P and Q rows are visible to all its lanes
        if (packed) {   // K-packed three-term layout (NR = 2, n_active <= 21): 64 halves per row and side
            // one 16-byte slot (eight consecutive entries of split_f16_entry_packed) per thread and side: 256 slots per
            // side in a block, written with one store each instead of eight 2-byte ones
            for (int c = threadIdx.x; c < 32 * 8; c += nthr) {
                const int64_t idx0 = rowp0 * 64 + (int64_t)c * 8;
                const int64_t rowp = idx0 / 64;
                const int rem = (int)(idx0 % 64);
                const int hh = rem / 32, q = (rem / 8) % 4;
                _Float16 op[8], oq[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int kp = 16 * q + 8 * hh + jj;
                    _Float16 vp = (_Float16)0.0f, vq = (_Float16)0.0f;
                    if (kp < 3 * n_active) {
                        const int term = kp / n_active, k = kp % n_active;
                        const float fp = P[rowp * KP + (k & 1) * KS + (k >> 1)], fq = Q[rowp * KP + (k & 1) * KS + (k >> 1)];
                        const _Float16 hp = (_Float16)fp, hq = (_Float16)fq;
                        const _Float16 lp = (_Float16)(fp - (float)hp), lq = (_Float16)(fq - (float)hq);
                        vp = term == 1 ? lp : hp;   // side 1 (B operand, P): hi, lo, hi
                        vq = term == 0 ? lq : hq;   // side 0 (A operand, Q): lo, hi, hi
                    }
                    op[jj] = vp; oq[jj] = vq;
                }
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                const h8 wp = {op[0], op[1], op[2], op[3], op[4], op[5], op[6], op[7]};
                const h8 wq = {oq[0], oq[1], oq[2], oq[3], oq[4], oq[5], oq[6], oq[7]};
                *reinterpret_cast<h8 *>(P16 + (rowp * 2 + hh) * 32 + q * 8) = wp;
                *reinterpret_cast<h8 *>(Q16 + (rowp * 2 + hh) * 32 + q * 8) = wq;
            }
            return;
        }
        // hi / lo layout (split_f16_entry): one (row, half, 16-symbol range) per thread - eight hi and eight lo halves of
        // each side, four 16-byte stores instead of thirty-two 2-byte ones
        for (int c = threadIdx.x; c < 32 * 2 * NR; c += nthr) {
            const int64_t rowp = rowp0 + c / (2 * NR);
            const int hh = (c / NR) % 2, r = c % NR;
            _Float16 ph[8], pl[8], qh[8], ql[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int k = 16 * r + 8 * hh + jj;
                float fp = 0.0f, fq = 0.0f;
                if (k < n_active) { fp = P[rowp * KP + (k & 1) * KS + (k >> 1)]; fq = Q[rowp * KP + (k & 1) * KS + (k >> 1)]; }
                ph[jj] = (_Float16)fp; pl[jj] = (_Float16)(fp - (float)ph[jj]);
                qh[jj] = (_Float16)fq; ql[jj] = (_Float16)(fq - (float)qh[jj]);
            }
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const int64_t half_elems = 2 * NR * 8;  // halves per (rowp, hh): [piece][r][8]
            _Float16 *op = P16 + (rowp * 2 + hh) * half_elems, *oq = Q16 + (rowp * 2 + hh) * half_elems;
            *reinterpret_cast<h8 *>(op + (0 * NR + r) * 8) = h8{ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]};
            *reinterpret_cast<h8 *>(op + (1 * NR + r) * 8) = h8{pl[0], pl[1], pl[2], pl[3], pl[4], pl[5], pl[6], pl[7]};
            *reinterpret_cast<h8 *>(oq + (0 * NR + r) * 8) = h8{qh[0], qh[1], qh[2], qh[3], qh[4], qh[5], qh[6], qh[7]};
            *reinterpret_cast<h8 *>(oq + (1 * NR + r) * 8) = h8{ql[0], ql[1], ql[2], ql[3], ql[4], ql[5], ql[6], ql[7]};
        }
    }
}

// m[y][x] = sum_k P1[y][k] * Q2[x][k] as a dense matrix in HBM (the cext_build_scores twin).
// One wave per 32x32 tile: A operand = P rows of sequence `one`, B operand = Q rows of `two`;
// lane holds column x0 + (l&31): coalesced 128-byte row segments on the store.
__global__ __launch_bounds__(64) void k_scores_tile(ArenaDev ar, int one, int two, int nstep,
                                                     float *__restrict__ m)
{
    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    const int L1 = ar.len[one], L2 = ar.len[two];
    const int y0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    const float *pa = ar.P + ((int64_t)ar.row_off[one] + y0 + j) * ar.KP + h * ar.KS;
    const float *qb = ar.Q + ((int64_t)ar.row_off[two] + x0 + j) * ar.KP + h * ar.KS;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < nstep; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], qb[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int x = x0 + j;
        if (y < L1 && x < L2) m[(int64_t)y * L2 + x] = acc[r];
    }
}

// The same for the pairs chunk_pairs[0 .. gridDim.x) of a plan, pair p's matrix at mref + m_off[p] (plans that run with
// per-position gap scores take their match scores from dense matrices, like the reference-order plans).
__global__ __launch_bounds__(64) void k_scores_tile_batch(ArenaDev ar, const int32_t *__restrict__ pairs,
                                                           const int32_t *__restrict__ chunk_pairs,
                                                           const int64_t *__restrict__ m_off, int nstep, int tiles_x,
                                                           float *__restrict__ mref, TileOut to)
{
    // grid (pairs of the chunk, 32-row blocks): the wave keeps its 32 rows of sequence one and walks the 32-column blocks
    // of sequence two (one workgroup per 32 x 32 block spent its time being dispatched: 5 M workgroups on C2)
    // (giving every XCD a contiguous run of the chunk's pairs - the 32 pairs of a task write neighbouring 64-byte pieces of
    // the same tile rows - was slower: 15.5 against 13.4 ms on C2; the kernel writes 21 GB at 1.6 TB/s either way)
    const int p = chunk_pairs[blockIdx.x];
    const int one = pairs[2 * p], two = pairs[2 * p + 1];
    const int lane = threadIdx.x;
    const int j = lane & 31, h = lane >> 5;
    const int L1 = ar.len[one], L2 = ar.len[two];
    const int y0 = (int)blockIdx.y * 32;
    if (y0 >= L1) return;
    const bool tiled = to.loc != nullptr;
    TileDst d = {nullptr, 0, 0, L2};
    if (tiled) d = tile_dst(to, mref, p, L2);
    float *m = tiled ? nullptr : mref + m_off[p];
    const float *pa = ar.P + ((int64_t)ar.row_off[one] + y0 + j) * ar.KP + h * ar.KS;
    for (int xt = 0; xt < tiles_x; ++xt) {
        const int x0 = d.x_lo + xt * 32;
        if (x0 >= L2 || x0 >= d.x_hi) break;
        const float *qb = ar.Q + ((int64_t)ar.row_off[two] + x0 + j) * ar.KP + h * ar.KS;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < nstep; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], qb[s], acc, 0, 0, 0);
        const int x = x0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (tiled) {
                // (x0 is a strip's first column: the strip's 32 columns, zeros past the sequence)
                if (y < L1) *d.at(y, x) = x < L2 ? acc[r] : 0.0f;
            } else if (y < L1 && x < L2) {
                m[(int64_t)y * L2 + x] = acc[r];
            }
        }
    }
}

#endif  // PRALINE_AUX_KERNELS

__device__ __forceinline__ void swap_halves(float &a, float &b)
{
    // v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b.
    // NOTE: extract both results into scalars before any bit cast.  hipcc (ROCm 7.2) folds
    // __builtin_bit_cast(float, r[1]) on the builtin's result vector into r[0] (checked in the
    // ISA: both stores used the vdst register); this form lowers to one swap + two live outputs.
    unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    ua = r[0];
    ub = r[1];
    a = __builtin_bit_cast(float, ua);
    b = __builtin_bit_cast(float, ub);
}

#ifdef PRALINE_AUX_KERNELS
// --------------------------------------------------------------------------------------------
// Chain mode, local alignments: the first flat argmax of o (align.py:402) over the candidates the strips of a
// task reported - largest value, then smallest y, then smallest x.  Thread = (task, lane).
__global__ void k_chain_local_end(const WaveTask *__restrict__ tasks, const int32_t *__restrict__ lane_pair,
                                  const float4 *__restrict__ cand, int n_tasks, int stride, int32_t *__restrict__ end_cells,
                                  float *__restrict__ scores)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = (int)(gid / 32), lane = (int)(gid % 32);
    if (t >= n_tasks) return;
    const int64_t p = lane_pair[(int64_t)t * 32 + lane];
    if (p < 0) return;
    const int nstrips = tasks[t].nstrips;
    float best = PRALINE_NEG_INF;
    int by = 0, bx = 0, bk = 0;
    for (int s = 0; s < nstrips; ++s) {
        const float4 c = cand[((int64_t)t * stride + s) * 32 + lane];
        const int y = __builtin_bit_cast(int, c.y), x = __builtin_bit_cast(int, c.z), k = __builtin_bit_cast(int, c.w);
        if (c.x > best || (c.x == best && (y < by || (y == by && x < bx)))) { best = c.x; by = y; bx = x; bk = k; }
    }
    end_cells[p * 4 + 0] = by;
    end_cells[p * 4 + 1] = bx;
    end_cells[p * 4 + 2] = bk;
    end_cells[p * 4 + 3] = 0;
    scores[p] = best;
}

// Semiglobal end cell (praline/component/align.py:406-424): the maxima of the last row o[L1, x, k] and the last
// column o[y, L2, k] (boundary cells included) and, for the side that wins, the cell the reference finds scanning
// from the far end (x from L2 down / y from L1 down) and k = 0, 1, 2: the largest coordinate holding the maximum,
// smallest k there.  The scratch rows interleave the pairs of a task (one 128-byte row per (coordinate, state)),
// so the scan runs per TASK - thread = (task, lane), every lane walks its own pair but all lanes of a task read the
// same row together.  (Scanning per pair touched a whole line per value: k_traceback took 4.7 ms instead of 1.2.)
__global__ __launch_bounds__(64) void k_semiglobal_end(ArenaDev ar, const WaveTask *__restrict__ tasks,
                                                        const int32_t *__restrict__ lane_one,
                                                        const int32_t *__restrict__ lane_pair, const int32_t *__restrict__ pairs,
                                                        const float *__restrict__ aux, int32_t *__restrict__ end_cells,
                                                        float *__restrict__ scores, RunParams rp, int32_t task_lo,
                                                        int32_t task_hi, int layout)
{
    const int ls = layout == 2 ? 16 : (layout ? 32 : 64);   // lanes (pairs) per task = lane stride of the scratch
    const int t = task_lo + (int)((blockIdx.x * 64 + threadIdx.x) / ls);
    const int lane = (int)(threadIdx.x % ls);
    if (t >= task_hi) return;
    const int64_t p = lane_pair[(int64_t)t * ls + lane];
    if (p < 0) return;
    const WaveTask tk = tasks[t];
    const int L1 = ar.len[pairs[2 * p]], L2 = ar.len[pairs[2 * p + 1]];
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const float *lastcol = aux + tk.aux_off + lane;
    const float *lastrow = aux + tk.aux_off + (int64_t)(tk.max_l1 + 1) * 3 * ls + lane;
    // scanning upwards with >= keeps the LARGEST coordinate; k upwards with > keeps the smallest k there
    float rmax = PRALINE_NEG_INF, cmax = PRALINE_NEG_INF;
    int rx = 0, rk = 0, cy = 0, ck = 0;
    {
        const float *g1p = rp.gaps ? rp.gaps + (int64_t)ar.row_off[pairs[2 * p]] * 2 : nullptr;
        const float *g2p = rp.gaps ? rp.gaps + (int64_t)ar.row_off[pairs[2 * p + 1]] * 2 : nullptr;
        const float b1 = g1p ? boundary_value_pp(L1, g1p, free_one) : boundary_value(L1, rp.go1, rp.ge1, free_one);   // o[L1, 0, :] = (-inf, b1, -inf)
        rmax = b1; rx = 0; rk = 1;
        if (!(b1 > PRALINE_NEG_INF)) rk = 0;
        const float b2 = g2p ? boundary_value_pp(L2, g2p, free_two) : boundary_value(L2, rp.go2, rp.ge2, free_two);   // o[0, L2, :] = (-inf, -inf, b2)
        cmax = b2; cy = 0; ck = 2;
        if (!(b2 > PRALINE_NEG_INF)) ck = 0;
    }
    // eight coordinates per round, all their loads issued before the first compare: one memory latency per round
    // instead of per coordinate (a single alignment spent 0.36 ms here, as long as a 147 072-pair batch).  The tail
    // repeats the last coordinate, which the >= rule absorbs.
    constexpr int SCAN = 8;
    for (int x0 = 1; x0 <= L2; x0 += SCAN) {
        float v[SCAN][3];
#pragma unroll
        for (int i = 0; i < SCAN; ++i) {
            const float *q = lastrow + (int64_t)(min(x0 + i, L2) - 1) * 3 * ls;
            v[i][0] = q[0]; v[i][1] = q[ls]; v[i][2] = q[2 * ls];
        }
#pragma unroll
        for (int i = 0; i < SCAN; ++i) {
            const float m = max3f(v[i][0], v[i][1], v[i][2]);
            if (m >= rmax) { rmax = m; rx = min(x0 + i, L2); rk = (v[i][0] == m) ? 0 : ((v[i][1] == m) ? 1 : 2); }
        }
    }
    for (int y0 = 1; y0 <= L1; y0 += SCAN) {
        float v[SCAN][3];
#pragma unroll
        for (int i = 0; i < SCAN; ++i) {
            const float *q = lastcol + (int64_t)min(y0 + i, L1) * 3 * ls;
            v[i][0] = q[0]; v[i][1] = q[ls]; v[i][2] = q[2 * ls];
        }
#pragma unroll
        for (int i = 0; i < SCAN; ++i) {
            const float m = max3f(v[i][0], v[i][1], v[i][2]);
            if (m >= cmax) { cmax = m; cy = min(y0 + i, L1); ck = (v[i][0] == m) ? 0 : ((v[i][1] == m) ? 1 : 2); }
        }
    }
    const bool from_row = rmax > cmax && free_two;   // align.py:411
    end_cells[p * 4 + 0] = from_row ? L1 : cy;
    end_cells[p * 4 + 1] = from_row ? rx : L2;
    end_cells[p * 4 + 2] = from_row ? rk : ck;
    scores[p] = from_row ? rmax : cmax;
}

// Device traceback over the packed planes: one lane per pair.
// get_paths (praline/util/align.py:144-185) + end-cell rules (praline/component/align.py:401-431)
// + extend_path_semiglobal (praline/util/align.py:268-297).  Paths are written backwards from the
// end of the pair's slot, so they come out in start->end order: rows [path_start, slot_end).
// --------------------------------------------------------------------------------------------
#ifndef PRALINE_TBW
#define PRALINE_TBW 8   // rows per lane in k_traceback's flag-word window
#endif
__global__ void k_traceback(ArenaDev ar, const WaveTask *__restrict__ tasks,
                            const PairLoc *__restrict__ loc, const int32_t *__restrict__ pairs,
                            const uint4 *__restrict__ tb, const float *__restrict__ aux, RectList rl,
                            const int32_t *__restrict__ end_cells, float *__restrict__ scores,
                            const int64_t *__restrict__ slot_off, int32_t *__restrict__ paths,
                            int64_t *__restrict__ path_start, int32_t *__restrict__ path_rows,
                            int64_t n_pairs, RunParams rp, int32_t task_lo, int32_t task_hi, int layout)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const PairLoc pl = loc[p];
    if (pl.task < task_lo || pl.task >= task_hi) return;  // pair belongs to another launch chunk
    const WaveTask tk = tasks[pl.task];
    const int L1 = ar.len[pairs[2 * p]], L2 = ar.len[pairs[2 * p + 1]];
    const bool free_one = mode_free_one(rp.mode), free_two = mode_free_two(rp.mode);
    const bool semiglobal = rp.mode >= 2;
    // layout 0: uint4 planes per 32 cells, rows max_l1 + 1 (the retired 64-lane task layout; no caller left); layout 1: k_dp_split16_tb planes
    // (uint2 per 16 cells, lane j = strip columns 1..16, lane j + 32 = columns 17..32, rows max_l1 + 8)
    const uint4 *my_tb = tb + tk.tb_off + pl.lane;
    const uint2 *my_tb2 = reinterpret_cast<const uint2 *>(tb) + tk.tb_off + pl.lane;

    int y = end_cells[p * 4 + 0], x = end_cells[p * 4 + 1], k = end_cells[p * 4 + 2];
    // (semiglobal: k_semiglobal_end has put the end cell into end_cells and the score into scores)

    int n_rects = 0, r0 = 0;
    if (rl.rect_off != nullptr) { r0 = rl.rect_off[p]; n_rects = rl.rect_off[p + 1] - r0; }

    const int64_t slot_end = slot_off[p] + (L1 + L2 + 2);
    int64_t w = slot_end;  // next row is written at w-1
    auto emit = [&](int yy, int xx) { --w; paths[2 * w] = yy; paths[2 * w + 1] = xx; };

    // suffix extension (align.py:284-295)
    if (semiglobal) {
        if (y != L1) { for (int yy = L1; yy > y; --yy) emit(yy, x); }
        else if (x != L2) { for (int xx = L2; xx > x; --xx) emit(y, xx); }
    }
    // traceback (praline/util/align.py:155-180)
    emit(y, x);
    if (layout == 2) {
        // k_dp_quad_tb planes (dp_quad.hip.h): uint2 [strip][step][64]; lane 16 q + p holds the columns 8 q + 1 .. 8 q + 8 of
        // pair p; step ((y + 1) >> 1) + q holds the rows y = 2 (t - q) - 1 (.x) and 2 (t - q) (.y); per row: match source
        // low | high << 8 | U-extend << 16 | L-extend << 24
        const int64_t nsteps = PRALINE_QUAD_STEPS(tk.max_l1);
        int guard = L1 + L2 + 2;
        bool stopped = false;
        while (y > 0 && x > 0 && guard-- > 0) {
            bool masked = false;
            for (int r = 0; r < n_rects; ++r) {
                const int32_t *q = rl.rects + (int64_t)(r0 + r) * 4;
                masked = masked || (y >= q[0] && y <= q[1] && x >= q[2] && x <= q[3]);
            }
            const int c = (x - 1) & 31, qq = c >> 3, bit = c & 7;
            const uint2 word = my_tb2[((int64_t)((x - 1) >> 5) * nsteps + ((y + 1) >> 1) + qq) * 64 + 16 * qq];
            const unsigned w = (y & 1) ? word.x : word.y;
            const int code = (int)(((w >> bit) & 1u) | (((w >> (8 + bit)) & 1u) << 1));
            const int ub = (int)((w >> (16 + bit)) & 1u), lb = (int)((w >> (24 + bit)) & 1u);
            if (masked || (k == 0 && code == 0)) { stopped = true; break; }   // t is 0 there (cext.c:141-149 / clamp)
            const int nk = k == 0 ? code - 1 : (k == 1 ? ub : 2 * lb);
            y -= (k != 2);
            x -= (k != 1);
            k = nk;
            emit(y, x);
        }
        while (!stopped && guard-- > 0) {
            if (x == 0 && y >= 1 && k == 1 && !free_one) --y;
            else if (y == 0 && x >= 1 && k == 2 && !free_two) --x;
            else break;
            emit(y, x);
        }
    } else if (layout == 3) {
        // k_dp_pk16_tb planes (dp_pk16.hip.h): uint4 [strip][step][64] at tk.tb_off counted in uint2; lane 16 q + (slot & 15)
        // holds the columns 8 q + 1 .. 8 q + 8 of the task's slots (slot & 15) and (slot & 15) + 16; step ((y + 1) >> 1) + q
        // holds the rows y = 2 (t - q) - 1 (.x, .y) and 2 (t - q) (.z, .w); per row: first word = match source low bits of slot
        // A | B << 8 | high bits A << 16 | B << 24, second word = U-extend A | B << 8 | L-extend A << 16 | B << 24
        const int64_t nsteps = PRALINE_QUAD_STEPS(tk.max_l1);
        const uint4 *my_tb4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint2 *>(tb) + tk.tb_off) + (pl.lane & 15);
        const int hsel = 8 * (pl.lane >> 4);
        int guard = L1 + L2 + 2;
        bool stopped = false;
        while (y > 0 && x > 0 && guard-- > 0) {
            bool masked = false;
            for (int r = 0; r < n_rects; ++r) {
                const int32_t *q = rl.rects + (int64_t)(r0 + r) * 4;
                masked = masked || (y >= q[0] && y <= q[1] && x >= q[2] && x <= q[3]);
            }
            const int c = (x - 1) & 31, qq = c >> 3, bit = (c & 7) + hsel;
            const uint4 word = my_tb4[((int64_t)((x - 1) >> 5) * nsteps + ((y + 1) >> 1) + qq) * 64 + 16 * qq];
            const unsigned w0 = (y & 1) ? word.x : word.z, w1 = (y & 1) ? word.y : word.w;
            const int code = (int)(((w0 >> bit) & 1u) | (((w0 >> (16 + bit)) & 1u) << 1));
            const int ub = (int)((w1 >> bit) & 1u), lb = (int)((w1 >> (16 + bit)) & 1u);
            if (masked || (k == 0 && code == 0)) { stopped = true; break; }   // t is 0 there (cext.c:141-149 / clamp)
            const int nk = k == 0 ? code - 1 : (k == 1 ? ub : 2 * lb);
            y -= (k != 2);
            x -= (k != 1);
            k = nk;
            emit(y, x);
        }
        while (!stopped && guard-- > 0) {
            if (x == 0 && y >= 1 && k == 1 && !free_one) --y;
            else if (y == 0 && x >= 1 && k == 2 && !free_two) --x;
            else break;
            emit(y, x);
        }
    } else if (layout == 1) {
        // k_dp_split16_tb planes: the interior walk as a short loop without a branch per state - the lanes of a wave
        // are at different cells and states, every divergent branch is paid by all of them (and a single alignment
        // pays the instruction count of one lane: ~120 instructions per step before)
        // Small plans (at most one wave of pairs - the merge steps of the progressive MSA): the walk is one dependent load
        // per path row, for a single alignment as long as a fifth of its chain-mode fill.  Every lane then keeps a WINDOW
        // of flag words in LDS - PRALINE_TBW rows upwards from its cell, for its 16-column group and the group to its
        // left (a step moves at most one row up and one column left) - refilled with 2 x PRALINE_TBW independent loads
        // when any lane of the wave has left its window: one memory latency per ~PRALINE_TBW rows.  Measured, single
        // alignments: 400 x 400 0.69 -> 0.61 ms, 3000 x 3000 5.1 -> 4.4 ms.  Not for large plans: with full waves some lane
        // leaves its window almost every step, and the 16 words per refill are more traffic than the direct loads (no gain
        // measured on C2 with paths).
        const int64_t rows = tk.max_l1 + 8;
        int guard = L1 + L2 + 2;
        bool stopped = false;
        __shared__ uint2 win_s[2 * PRALINE_TBW][64];
        const bool windowed = n_pairs <= 64;
        const int tid = threadIdx.x & 63;
        int wy = -1, wg = -(1 << 30);   // the window holds rows wy .. wy - PRALINE_TBW + 1 of the groups wg and wg - 1
        while (y > 0 && x > 0 && guard-- > 0) {
            bool masked = false;
            for (int r = 0; r < n_rects; ++r) {
                const int32_t *q = rl.rects + (int64_t)(r0 + r) * 4;
                masked = masked || (y >= q[0] && y <= q[1] && x >= q[2] && x <= q[3]);
            }
            const int c = (x - 1) & 31, bit = c & 15;
            uint2 word;
            if (windowed) {
                const int g = (x - 1) >> 4;   // 16-column group: strip g / 2, lane half g % 2
                const bool miss = !((g == wg || g == wg - 1) && y <= wy && y > wy - PRALINE_TBW);
                if (__ballot(miss) != 0ull) {
                    uint2 t0[PRALINE_TBW], t1[PRALINE_TBW];
                    const int gl = g > 0 ? g - 1 : 0;
                    const uint2 *b0 = my_tb2 + (int64_t)(g >> 1) * rows * 64 + 32 * (g & 1);
                    const uint2 *b1 = my_tb2 + (int64_t)(gl >> 1) * rows * 64 + 32 * (gl & 1);
#pragma unroll
                    for (int i = 0; i < PRALINE_TBW; ++i) {
                        const int64_t r = y - i > 0 ? y - i : 0;
                        t0[i] = b0[r * 64];
                        t1[i] = b1[r * 64];
                    }
#pragma unroll
                    for (int i = 0; i < PRALINE_TBW; ++i) { win_s[i][tid] = t0[i]; win_s[PRALINE_TBW + i][tid] = t1[i]; }
                    wy = y; wg = g;
                }
                word = win_s[(g == wg ? 0 : PRALINE_TBW) + (wy - y)][tid];
            } else {
                word = my_tb2[((int64_t)((x - 1) >> 5) * rows + y) * 64 + 32 * (c >> 4)];
            }
            // word.x: match source as two bit planes (low bits | high bits << 16); word.y: U-extend | L-extend << 16
            const int code = (int)(((word.x >> bit) & 1u) | (((word.x >> (16 + bit)) & 1u) << 1));
            const int ub = (int)((word.y >> bit) & 1u), lb = (int)((word.y >> (16 + bit)) & 1u);
            if (masked || (k == 0 && code == 0)) { stopped = true; break; }   // t is 0 there (cext.c:141-149 / clamp)
            // M: diagonal, state code - 1 (1 MM, 2 MU, 3 ML); U: up, 0 UO -> M / 1 UE -> U; L: left, 0 LO -> M / 1 LE -> L
            const int nk = k == 0 ? code - 1 : (k == 1 ? ub : 2 * lb);
            y -= (k != 2);
            x -= (k != 1);
            k = nk;
            emit(y, x);
        }
        // on a boundary cell: the pre-initialised flags (align.py:377,385): t[y>=1,0,1] = UE, t[0,x>=1,2] = LE
        while (!stopped && guard-- > 0) {
            if (x == 0 && y >= 1 && k == 1 && !free_one) --y;
            else if (y == 0 && x >= 1 && k == 2 && !free_two) --x;
            else break;
            emit(y, x);
        }
    } else
    for (int guard = 0; guard < L1 + L2 + 2; ++guard) {
        int ny, nx, nk;
        if (y == 0 || x == 0) {
            // pre-initialised boundary flags (align.py:377,385): t[y>=1,0,1] = UE, t[0,x>=1,2] = LE
            if (x == 0 && y >= 1 && k == 1 && !free_one) { ny = y - 1; nx = 0; nk = 1; }
            else if (y == 0 && x >= 1 && k == 2 && !free_two) { ny = 0; nx = x - 1; nk = 2; }
            else break;
        } else {
            bool masked = false;
            for (int r = 0; r < n_rects; ++r) {
                const int32_t *q = rl.rects + (int64_t)(r0 + r) * 4;
                masked = masked || (y >= q[0] && y <= q[1] && x >= q[2] && x <= q[3]);
            }
            if (masked) break;  // t stays 0 in masked cells (cext.c:141-149)
            const int s = (x - 1) >> 5, c = (x - 1) & 31;
            unsigned p_mlo, p_mhi, p_u, p_l;
            int bit = c;
            if (layout == 0) {
                const uint4 word = my_tb[((int64_t)s * (tk.max_l1 + 1) + y) * 64];
                p_mlo = word.x; p_mhi = word.y; p_u = word.z; p_l = word.w;
            } else {
                const uint2 word = my_tb2[((int64_t)s * (tk.max_l1 + 8) + y) * 64 + 32 * (c >> 4)];
                // word.x: match source as two bit planes (low bits | high bits << 16); word.y: U-extend bits |
                // L-extend bits << 16
                bit = c & 15;
                const unsigned code2 = ((word.x >> bit) & 1u) | (((word.x >> (16 + bit)) & 1u) << 1);
                p_mlo = (code2 & 1u) << bit; p_mhi = (code2 >> 1) << bit; p_u = word.y & 0xffffu; p_l = word.y >> 16;
            }
            if (k == 0) {
                const int code = ((p_mlo >> bit) & 1) | (((p_mhi >> bit) & 1) << 1);
                if (code == 0) break;
                ny = y - 1; nx = x - 1; nk = code - 1;     // 1 MM, 2 MU, 3 ML
            } else if (k == 1) {
                ny = y - 1; nx = x; nk = (p_u >> bit) & 1;                 // 0 UO -> M, 1 UE -> U
            } else {
                ny = y; nx = x - 1; nk = ((p_l >> bit) & 1) ? 2 : 0;       // 0 LO -> M, 1 LE -> L
            }
        }
        y = ny; x = nx; k = nk;
        emit(y, x);
    }
    // prefix extension (align.py:270-279): (y, x) is now the first path row
    if (semiglobal) {
        if (y != 0) { for (int yy = y - 1; yy >= 0; --yy) emit(yy, 0); }
        else if (x != 0) { for (int xx = x - 1; xx >= 0; --xx) emit(0, xx); }
    }
    path_start[p] = w;
    path_rows[p] = (int32_t)(slot_end - w);
}

// ---- preprofile stage on the device ------------------------------------------------------------
// Fuses, per pair (sequence one = master, two = slave) of a path plan, what the reference does on the host after a
// master-slave alignment: compress_path(path, 0) (praline/util/align.py:215-232: keep the rows in which the master
// advances), extend_path_local for local mode (align.py:234-266: -1 outside the local path), Alignment.merge
// (container/align.py:30-61: the slave column of the merged path IS that compressed path) and the slave's share
// of ProfileBuilder / get_frequencies (align.py:187-213): with X_r the slave index of row r = master index r,
//     X_{r+1} > X_r  ->  counts[master row r][ sym_slave[X_{r+1} - 1] ] += 1
// including the reference's own corner case: a local path that starts at slave index 0 comes after -1 padding,
// 0 - (-1) > 0 counts as an advance and values[0 - 1] wraps to the slave's LAST symbol.
// One lane per pair walks its path start -> end; the masters' own symbols are added by the host.
__global__ void k_path_counts(const int32_t *__restrict__ pairs, const float *__restrict__ scores,
                              const int32_t *__restrict__ paths, const int64_t *__restrict__ path_start,
                              const int32_t *__restrict__ path_rows, int64_t n_pairs, int use_threshold, float threshold,
                              int local, const int32_t *__restrict__ row_off_raw, const int32_t *__restrict__ len,
                              const unsigned char *__restrict__ sym_raw, int A, int32_t *__restrict__ counts)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    if (use_threshold && !(scores[p] >= threshold)) return;   // preprofile.py:145,258
    const int master = pairs[2 * p], slave = pairs[2 * p + 1];
    const int rows = path_rows[p];
    if (rows <= 0) return;
    const int32_t *path = paths + path_start[p] * 2;
    const unsigned char *ss = sym_raw + row_off_raw[slave];
    const int ls = len[slave];
    int32_t *mc = counts + (int64_t)row_off_raw[master] * A;
    int y = path[0], xk = path[1];   // last kept row: (master index, slave index)
    if (local && y > 0) {
        // row y - 1 of the extended path holds -1: xk - (-1) > 0 always
        const int sx = xk >= 1 ? xk - 1 : ls - 1;
        atomicAdd(mc + (int64_t)(y - 1) * A + ss[sx], 1);
    }
    for (int r = 1; r < rows; ++r) {
        const int y1 = path[2 * r], x1 = path[2 * r + 1];
        if (y1 > y) {   // the master advances: a kept row
            if (x1 > xk) atomicAdd(mc + (int64_t)(y1 - 1) * A + ss[x1 - 1], 1);
            xk = x1;
            y = y1;
        }
    }
}

// The same for pair lists in which the pairs of a master are contiguous (the preprofile stage's own order: master outer,
// slaves ascending): one workgroup per RUN of pairs with the same master keeps the master's count block [len][A] in LDS -
// the N - 1 slave paths of a master all hit those few thousand counters (262 M global atomics on C3, 7.5 ms) - and adds
// its nonzero entries to the arena once.  runs: int64 [n_runs][2] = first pair, one past the last.
__global__ __launch_bounds__(256) void k_path_counts_runs(const int32_t *__restrict__ pairs, const float *__restrict__ scores,
                                                          const int32_t *__restrict__ paths, const int64_t *__restrict__ path_start,
                                                          const int32_t *__restrict__ path_rows, const int64_t *__restrict__ runs,
                                                          int use_threshold, float threshold, int local,
                                                          const int32_t *__restrict__ row_off_raw, const int32_t *__restrict__ len,
                                                          const unsigned char *__restrict__ sym_raw, int A, int32_t *__restrict__ counts)
{
    extern __shared__ int hist[];   // [len[master]][A]
    const int64_t p0 = runs[2 * (int64_t)blockIdx.x], p1 = runs[2 * (int64_t)blockIdx.x + 1];
    const int master = pairs[2 * p0];
    const int n = len[master] * A;
    for (int i = threadIdx.x; i < n; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        if (use_threshold && !(scores[p] >= threshold)) continue;   // preprofile.py:145,258
        const int rows = path_rows[p];
        if (rows <= 0) continue;
        const int slave = pairs[2 * p + 1];
        const int32_t *path = paths + path_start[p] * 2;
        const unsigned char *ss = sym_raw + row_off_raw[slave];
        const int ls = len[slave];
        int y = path[0], xk = path[1];   // last kept row: (master index, slave index)
        if (local && y > 0) {
            const int sx = xk >= 1 ? xk - 1 : ls - 1;
            atomicAdd(&hist[(y - 1) * A + ss[sx]], 1);
        }
        for (int r = 1; r < rows; ++r) {
            const int y1 = path[2 * r], x1 = path[2 * r + 1];
            if (y1 > y) {   // the master advances: a kept row
                if (x1 > xk) atomicAdd(&hist[(y1 - 1) * A + ss[x1 - 1]], 1);
                xk = x1;
                y = y1;
            }
        }
    }
    __syncthreads();
    int32_t *mc = counts + (int64_t)row_off_raw[master] * A;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = hist[i];
        if (v != 0) atomicAdd(mc + i, v);
    }
}

// first and last row of every path = its bounding box (paths are monotone): (y0, y1, x0, x1), the rectangle the
// next Waterman-Eggert iteration masks (praline/component/preprofile.py:247-255)
__global__ void k_path_bounds(const int32_t *__restrict__ paths, const int64_t *__restrict__ path_start,
                              const int32_t *__restrict__ path_rows, int64_t n_pairs, int32_t *__restrict__ bounds)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int rows = path_rows[p];
    const int32_t *path = paths + path_start[p] * 2;
    bounds[4 * p + 0] = rows > 0 ? path[0] : 0;
    bounds[4 * p + 1] = rows > 0 ? path[2 * (rows - 1)] : -1;
    bounds[4 * p + 2] = rows > 0 ? path[1] : 0;
    bounds[4 * p + 3] = rows > 0 ? path[2 * (rows - 1) + 1] : -1;
}


// Waterman-Eggert on the device: the bounding box of every pair's path becomes that pair's zero rectangle number `slot`
// (rects: int32 [n_pairs][PRALINE_MAX_RECTS][4]; unused slots hold the empty rectangle) - no host round trip between
// the iterations (praline/component/preprofile.py:247-255).
__global__ void k_path_bounds_to_rects(const int32_t *__restrict__ paths, const int64_t *__restrict__ path_start,
                                       const int32_t *__restrict__ path_rows, int64_t n_pairs, int slot, int32_t *__restrict__ rects)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int rows = path_rows[p];
    const int32_t *path = paths + path_start[p] * 2;
    int32_t *q = rects + (p * PRALINE_MAX_RECTS + slot) * 4;
    q[0] = rows > 0 ? path[0] : (1 << 30);
    q[1] = rows > 0 ? path[2 * (rows - 1)] : -1;
    q[2] = rows > 0 ? path[1] : (1 << 30);
    q[3] = rows > 0 ? path[2 * (rows - 1) + 1] : -1;
}

// Resident progressive alignment: merge two clusters of the arena along the device path of their
// alignment into a NEW sequence at the arena's end - ProfileTrack.merge (praline/container/sequence.py:205-239) for
// every track set: alignment column c sums the integer counts of the positions that advance in it (the reference sums
// in float32 and truncates to int: the same integers), and the profile row PairwiseAligner will read is
// float32(float64(count) / float64(float32(rowsum))), the rowsum per track set (sequence.py:192-203, align.py:171-172).
// One 64-thread block per alignment column.
__global__ __launch_bounds__(64) void k_merge_clusters(const int32_t *__restrict__ path, int cols, int32_t *cnt, float *raw,
                                                        int A, int64_t row_one, int64_t row_two, int64_t row_new,
                                                        const int32_t *__restrict__ set_lo, int n_sets)
{
    __shared__ int32_t v[256];
    const int c = blockIdx.x;
    if (c >= cols) return;
    const int y0 = path[2 * c], x0 = path[2 * c + 1], y1 = path[2 * c + 2], x1 = path[2 * c + 3];
    const bool adv0 = y1 > y0, adv1 = x1 > x0;
    for (int a = threadIdx.x; a < A; a += 64) {
        int32_t t = 0;
        if (adv0) t += cnt[(row_one + y1 - 1) * A + a];
        if (adv1) t += cnt[(row_two + x1 - 1) * A + a];
        v[a] = t;
        cnt[(row_new + c) * A + a] = t;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += 64) {
        int s = 0;
        while (s + 1 < n_sets && a >= set_lo[s + 1]) ++s;
        long long sum = 0;
        for (int k = set_lo[s]; k < set_lo[s + 1]; ++k) sum += v[k];
        const float total = (float)sum;
        raw[(row_new + c) * A + a] = (float)((double)v[a] / (double)total);
    }
}

__global__ void k_fill_i32(int32_t *dst, int64_t n, int32_t value)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// Column masks of the zero rectangles of plans with more than PRALINE_MAX_RECTS per pair: one block per pair, entry (strip s, row y) = bit c set
// when cell (y, 32 s + c + 1) lies in one of the pair's rectangles (cext.c:141-149 skips those cells).
__global__ __launch_bounds__(256) void k_build_zmask(const int32_t *__restrict__ pairs, const int32_t *__restrict__ len,
                                                     const int32_t *__restrict__ rect_off, const int32_t *__restrict__ rects,
                                                     const int64_t *__restrict__ zm_off, unsigned *__restrict__ zmask)
{
    const int p = blockIdx.x;
    const int L1 = len[pairs[2 * p]], L2 = len[pairs[2 * p + 1]];
    const int nstrips = (L2 + 31) >> 5;
    const int r0 = rect_off[p], n = rect_off[p + 1] - r0;
    unsigned *out = zmask + zm_off[p];
    for (int e = threadIdx.x; e < nstrips * (L1 + 1); e += blockDim.x) {
        const int s = e / (L1 + 1), y = e % (L1 + 1);
        unsigned z = 0;
        for (int r = 0; r < n; ++r) {
            const int32_t *q = rects + (int64_t)(r0 + r) * 4;
            if (y >= q[0] && y <= q[1]) {
                const int lo = max(q[2] - (32 * s + 1), 0), hi = min(q[3] - (32 * s + 1), 31);
                if (lo <= hi) z |= (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo);
            }
        }
        out[e] = z;
    }
}

// --------------------------------------------------------------------------------------------
// Reference-order match scores (audit mode PRALINE_MATCH_REFERENCE).
// cext_build_scores / score_match_prof_prof (praline/util/cext.c:33-97,389-420): per track set, one float32 running sum
// over the nonzeros of row y of profile one (ascending, outer) and of row x of profile two (ascending, inner); the
// reference binary (built -ffast-math, setup.py:28) evaluates each term as (p2 * score) * p1 - see
// oracle/praline_oracle.c - and the per-set sums are added in list order to a float32 that starts at 0.
// Separately rounded multiplies and adds (__fmul_rn / __fadd_rn: no contraction), so the result is bit-identical
// to the reference's m for ANY profiles; the price is ~nnz1 * nnz2 dependent VALU operations per cell.
// --------------------------------------------------------------------------------------------
// per raw profile row: ascending indices of its nonzeros (build_nonzero_matrix, praline/component/align.py:449-458)
__global__ void k_build_nz(const float *__restrict__ raw, int64_t rows, int A, unsigned char *__restrict__ nzidx,
                           unsigned char *__restrict__ nzcnt)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *row = raw + r * A;
    unsigned char *out = nzidx + r * A;
    int k = 0;
    for (int a = 0; a < A; ++a)
        if (row[a] != 0.0f) out[k++] = (unsigned char)a;
    nzcnt[r] = (unsigned char)k;
}

// grid (pairs of the chunk, row groups); a block walks the cells of its rows with x fastest.
#define PRALINE_REF_ROWS 16
__global__ __launch_bounds__(256) void k_match_ref(const float *__restrict__ raw, const float *__restrict__ S, int A,
                                                   const int32_t *__restrict__ row_off_raw, const int32_t *__restrict__ len,
                                                   const unsigned char *__restrict__ nzidx,
                                                   const unsigned char *__restrict__ nzcnt,
                                                   const int32_t *__restrict__ set_lo, int n_sets,
                                                   const int32_t *__restrict__ pairs, const int32_t *__restrict__ chunk_pairs,
                                                   const int64_t *__restrict__ m_off, float *__restrict__ mref, TileOut to)
{
    const int p = chunk_pairs[blockIdx.x];
    const int one = pairs[2 * p], two = pairs[2 * p + 1];
    const int L1 = len[one], L2 = len[two];
    const int y0 = blockIdx.y * PRALINE_REF_ROWS;
    if (y0 >= L1) return;
    const int ny = min(PRALINE_REF_ROWS, L1 - y0);
    const int64_t r1 = row_off_raw[one], r2 = row_off_raw[two];
    const bool tiled = to.loc != nullptr;
    TileDst d = {nullptr, 0, 0, L2};
    if (tiled) d = tile_dst(to, mref, p, L2);
    float *out = tiled ? nullptr : mref + m_off[p];
    const int W = d.x_hi - d.x_lo;   // columns per row this launch writes (tiles: whole strips)
    for (int c = threadIdx.x; c < ny * W; c += blockDim.x) {
        const int y = y0 + c / W, x = d.x_lo + c % W;
        if (x >= L2) { *d.at(y, x) = 0.0f; continue; }
        const float *p1 = raw + (r1 + y) * A, *p2 = raw + (r2 + x) * A;
        const unsigned char *i1 = nzidx + (r1 + y) * A, *i2 = nzidx + (r2 + x) * A;
        const int n1 = nzcnt[r1 + y], n2 = nzcnt[r2 + x];
        float score = 0.0f;
        int a = 0;
        for (int s = 0; s < n_sets; ++s) {
            const int hi = set_lo[s + 1], lo = set_lo[s];
            // the nonzeros of row x that fall into this set: [b0, b1)
            int b0 = 0;
            while (b0 < n2 && i2[b0] < lo) ++b0;
            int b1 = b0;
            while (b1 < n2 && i2[b1] < hi) ++b1;
            float acc = 0.0f;
            for (; a < n1 && i1[a] < hi; ++a) {
                const int i = i1[a];
                const float v1 = p1[i];
                const float *srow = S + (int64_t)i * A;
                for (int b = b0; b < b1; ++b) {
                    const int j = i2[b];
                    acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(p2[j], srow[j]), v1));
                }
            }
            score = __fadd_rn(score, acc);
        }
        if (tiled) *d.at(y, x) = score;
        else out[(int64_t)y * L2 + x] = score;
    }
}

// The same sums with the y-independent half of every term prepared per arena row: for a row x used as sequence TWO,
//     T[x][i][b] = fl(p2[x][j_b] * S[i][j_b])        j_b = b-th nonzero of row x, zero past its end
// (zero too where j_b lies in another track set than i: S is block diagonal and the reference's inner loop of set s
// only visits set s).  A cell is then  acc = fl(acc + fl(T[x][i_a][b] * p1[y][i_a]))  over the nonzeros a of row y
// (outer, ascending) and b = 0 .. TB-1 (inner): exactly the reference's terms in the reference's order, the added
// zero terms leave the float32 sum unchanged (x + (+0) = x).  One multiply and one add per term, no index chasing in
// the inner loop; the T rows of a pair (L2 x A x TB floats) stay in L2 across the rows of sequence one.  Layout
// T[i][row][b]: the 64 lanes of a wave (consecutive x, the same y and hence the same i) read one contiguous run.
// The nonzero list of each of the block's rows y - wave-uniform - is staged in LDS once.
__global__ void k_build_reft(const float *__restrict__ raw, const float *__restrict__ S, int A, int64_t rows,
                             const unsigned char *__restrict__ nzidx, const unsigned char *__restrict__ nzcnt,
                             const int32_t *__restrict__ set_lo, int n_sets, int TB, float *__restrict__ T)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (row, i)
    if (e >= rows * A) return;
    const int64_t r = e / A;
    const int i = (int)(e % A);
    int s = 0;
    while (s + 1 < n_sets && i >= set_lo[s + 1]) ++s;
    const int lo = set_lo[s], hi = set_lo[s + 1];
    const int n = nzcnt[r];
    const unsigned char *idx = nzidx + r * A;
    const float *p2 = raw + r * A, *srow = S + (int64_t)i * A;
    float *out = T + ((int64_t)i * rows + r) * TB;   // symbol-major: for one symbol, consecutive rows are contiguous
    for (int b = 0; b < TB; ++b) {
        float t = 0.0f;
        if (b < n) {
            const int j = idx[b];
            if (j >= lo && j < hi) t = __fmul_rn(p2[j], srow[j]);
        }
        out[b] = t;
    }
}

template <int TB>
__global__ __launch_bounds__(256) void k_match_reft(const float *__restrict__ raw, int A, const float *__restrict__ T, int64_t rows,
                                                    const int32_t *__restrict__ row_off_raw, const int32_t *__restrict__ len,
                                                    const unsigned char *__restrict__ nzidx,
                                                    const unsigned char *__restrict__ nzcnt,
                                                    const int32_t *__restrict__ set_lo, int n_sets,
                                                    const int32_t *__restrict__ pairs, const int32_t *__restrict__ chunk_pairs,
                                                    const int64_t *__restrict__ m_off, float *__restrict__ mref, TileOut to)
{
    const int p = chunk_pairs[blockIdx.x];
    const int one = pairs[2 * p], two = pairs[2 * p + 1];
    const int L1 = len[one], L2 = len[two];
    const int y0 = blockIdx.y * PRALINE_REF_ROWS;
    if (y0 >= L1) return;
    const int ny = min(PRALINE_REF_ROWS, L1 - y0);
    const int64_t r1 = row_off_raw[one], r2 = row_off_raw[two];
    const bool tiled = to.loc != nullptr;
    TileDst d = {nullptr, 0, 0, L2};
    if (tiled) d = tile_dst(to, mref, p, L2);
    float *out = tiled ? nullptr : mref + m_off[p];
    const int W = d.x_hi - d.x_lo;   // columns per row this launch writes (tiles: whole strips)
    // nonzero lists of this block's rows: (symbol, value) pairs, at most 32 per row kept here (longer rows: global)
    __shared__ int s_n[PRALINE_REF_ROWS];
    __shared__ int s_i[PRALINE_REF_ROWS][32];
    __shared__ float s_v[PRALINE_REF_ROWS][32];
    for (int e = threadIdx.x; e < ny * 32; e += blockDim.x) {
        const int yy = e >> 5, a = e & 31;
        const int n1 = nzcnt[r1 + y0 + yy];
        if (a == 0) s_n[yy] = n1;
        if (a < n1) {
            const int i = nzidx[(r1 + y0 + yy) * A + a];
            s_i[yy][a] = i;
            s_v[yy][a] = raw[(r1 + y0 + yy) * A + i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ny * W; c += blockDim.x) {
        const int yy = c / W, y = y0 + yy, x = d.x_lo + c % W;
        if (x >= L2) { *d.at(y, x) = 0.0f; continue; }
        const float *p1 = raw + (r1 + y) * A;
        const unsigned char *i1 = nzidx + (r1 + y) * A;
        const int n1 = s_n[yy];
        const float *Tx = T + (r2 + x) * (int64_t)TB;
        float score = 0.0f, acc = 0.0f;
        int s = 0;
        for (int a = 0; a < n1; ++a) {
            const int i = a < 32 ? s_i[yy][a] : (int)i1[a];
            while (s + 1 < n_sets && i >= set_lo[s + 1]) { score = __fadd_rn(score, acc); acc = 0.0f; ++s; }
            const float v1 = a < 32 ? s_v[yy][a] : p1[i];
            const float4 *t4 = reinterpret_cast<const float4 *>(Tx + (int64_t)i * rows * TB);
#pragma unroll
            for (int q = 0; q < TB / 4; ++q) {
                const float4 t = t4[q];
                acc = __fadd_rn(acc, __fmul_rn(t.x, v1));
                acc = __fadd_rn(acc, __fmul_rn(t.y, v1));
                acc = __fadd_rn(acc, __fmul_rn(t.z, v1));
                acc = __fadd_rn(acc, __fmul_rn(t.w, v1));
            }
        }
        for (; s < n_sets; ++s) { score = __fadd_rn(score, acc); acc = 0.0f; }
        if (tiled) *d.at(y, x) = score;
        else out[(int64_t)y * L2 + x] = score;
    }
}

// --------------------------------------------------------------------------------------------
// Raw parity kernels: the reference's buffers (contiguous copies on the device).
// --------------------------------------------------------------------------------------------
// Boundary initialisation of RawPairwiseAligner (praline/component/align.py:357-385).
__global__ void k_raw_init(int mode, const float *__restrict__ g1, const float *__restrict__ g2,
                           float *__restrict__ o, uint8_t *__restrict__ t, int L1, int L2)
{
    const int64_t C = L2 + 1;
    const int64_t n = (int64_t)(L1 + 1) * C;
    const bool free_one = mode_free_one(mode), free_two = mode_free_two(mode);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = i / C, x = i % C;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        uint8_t t1 = 0, t2 = 0;
        if (y == 0 || x == 0) { v0 = PRALINE_NEG_INF; v1 = PRALINE_NEG_INF; v2 = PRALINE_NEG_INF; }
        if (y == 0 && x == 0) v0 = 0.0f;
        if (x == 0) {
            if (free_one) v1 = 0.0f;
            else if (y == 0) v1 = g1[0] - g1[1];
            else { v1 = (float)((double)(y - 1) * (double)g1[(y - 1) * 2 + 1] + (double)g1[0]); t1 = 32; }
        }
        if (y == 0) {
            if (free_two) v2 = 0.0f;
            else if (x == 0) v2 = g2[0] - g2[1];
            else { v2 = (float)((double)(x - 1) * (double)g2[(x - 1) * 2 + 1] + (double)g2[0]); t2 = 128; }
        }
        o[i * 3 + 0] = v0; o[i * 3 + 1] = v1; o[i * 3 + 2] = v2;
        t[i * 3 + 0] = 0; t[i * 3 + 1] = t1; t[i * 3 + 2] = t2;
    }
}

// cext_align (praline/util/cext.c:99-306) on the reference's own buffers.  Lane l of a wavefront owns column
// x0 + l + 1 of a 64-column strip and computes row (step - l): the cells of one step form an anti-diagonal.  Left /
// diagonal neighbours come from lane l-1 via __shfl_up; the up neighbour is the lane's own previous step.  All
// seven tie flags are reproduced.  One workgroup of up to 16 wavefronts: wave w takes strip g0 + w and runs 64
// steps behind wave w - 1, whose last column it receives through LDS (written in one step, read in the next, one
// barrier per step); groups of 16 strips follow each other and read the previous group's last column back from o.
#define PRALINE_RAW_WAVES 16
__global__ __launch_bounds__(64 * PRALINE_RAW_WAVES) void k_raw_align(int local_mode, const float *__restrict__ m,
                                                                       const float *__restrict__ g1,
                                                                       const float *__restrict__ g2, float *o,
                                                                       uint8_t *t, const uint8_t *__restrict__ z, int L1,
                                                                       int L2)
{
    __shared__ float hand[2][PRALINE_RAW_WAVES][3];   // (M, U, L) of a wave's last column, by step parity
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int64_t C = L2 + 1;
    const float base = local_mode ? 0.0f : PRALINE_NEG_INF;
    const int nstrips = (L2 + 63) / 64;
    for (int g0 = 0; g0 < nstrips; g0 += W) {
        const int wg = min(W, nstrips - g0);              // strips (= busy waves) of this group
        const int x0 = (g0 + wave) * 64;
        const int x = x0 + lane + 1;
        const bool col_ok = wave < wg && x <= L2;
        // up neighbour (y-1, x): starts at the boundary row
        float upM = PRALINE_NEG_INF, upU = PRALINE_NEG_INF, upL = PRALINE_NEG_INF;
        // diagonal neighbour (y-1, x-1)
        float dgM = PRALINE_NEG_INF, dgU = PRALINE_NEG_INF, dgL = PRALINE_NEG_INF;
        float go2 = 0.0f, ge2 = 0.0f;
        if (col_ok) {
            upM = o[(int64_t)x * 3 + 0]; upU = o[(int64_t)x * 3 + 1]; upL = o[(int64_t)x * 3 + 2];
            dgM = o[(int64_t)(x - 1) * 3 + 0]; dgU = o[(int64_t)(x - 1) * 3 + 1]; dgL = o[(int64_t)(x - 1) * 3 + 2];
            go2 = g2[(x - 1) * 2]; ge2 = g2[(x - 1) * 2 + 1];
        }
        float curM = 0.0f, curU = 0.0f, curL = 0.0f;
        // inputs of the NEXT step's cell are fetched one step ahead (the step is a chain of dependent loads otherwise)
        float n_ms = 0.0f, n_go1 = 0.0f, n_ge1 = 0.0f;
        uint8_t n_z = 0;
        {
            const int y1 = 1 - 64 * wave - lane;
            if (y1 >= 1 && y1 <= L1 && col_ok) {
                n_ms = m[(int64_t)(y1 - 1) * L2 + (x - 1)]; n_z = z[(int64_t)y1 * C + x];
                n_go1 = g1[(y1 - 1) * 2]; n_ge1 = g1[(y1 - 1) * 2 + 1];
            }
        }
        const int total = L1 + 63 + 64 * (wg - 1);
        for (int gs = 1; gs <= total; ++gs) {
            const int y = gs - 64 * wave - lane;          // this wave runs 64 steps behind the previous one
            const float ms = n_ms, go1 = n_go1, ge1 = n_ge1;
            const uint8_t zc = n_z;
            {
                const int yn = y + 1;
                if (yn >= 1 && yn <= L1 && col_ok) {
                    n_ms = m[(int64_t)(yn - 1) * L2 + (x - 1)]; n_z = z[(int64_t)yn * C + x];
                    n_go1 = g1[(yn - 1) * 2]; n_ge1 = g1[(yn - 1) * 2 + 1];
                }
            }
            // left neighbour (y, x-1): what lane-1 produced in the previous step
            float lfM = __shfl_up(curM, 1), lfU = __shfl_up(curU, 1), lfL = __shfl_up(curL, 1);
            const bool row_ok = y >= 1 && y <= L1;
            if (lane == 0 && row_ok) {
                if (wave == 0) {                          // boundary column / the previous group's last column
                    const int64_t i = ((int64_t)y * C + x0) * 3;
                    lfM = o[i]; lfU = o[i + 1]; lfL = o[i + 2];
                } else {                                  // the previous wave's lane 63, one step ago
                    lfM = hand[(gs - 1) & 1][wave - 1][0]; lfU = hand[(gs - 1) & 1][wave - 1][1]; lfL = hand[(gs - 1) & 1][wave - 1][2];
                }
            }
            if (row_ok && col_ok) {
                const int64_t cell = (int64_t)y * C + x;
                if (zc) {
                    // masked: the cell keeps whatever the caller pre-initialised (cext.c:147-149)
                    curM = o[cell * 3]; curU = o[cell * 3 + 1]; curL = o[cell * 3 + 2];
                } else {
                    const float up_open = upM + go1, up_ext = upU + ge1;
                    const float lf_open = lfM + go2, lf_ext = lfL + ge2;
                    const float mm = dgM + ms, mu = dgU + ms, ml = dgL + ms;
                    float mmax = base;
                    if (mm > mmax) mmax = mm;
                    if (mu > mmax) mmax = mu;
                    if (ml > mmax) mmax = ml;
                    uint8_t tm = 0;
                    if (mm == mmax) tm |= 2;
                    if (mu == mmax) tm |= 4;
                    if (ml == mmax) tm |= 8;
                    float umax = PRALINE_NEG_INF;
                    if (up_open > umax) umax = up_open;
                    if (up_ext > umax) umax = up_ext;
                    uint8_t tu = 0;
                    if (up_open == umax) tu |= 16;
                    if (up_ext == umax) tu |= 32;
                    float lmax = PRALINE_NEG_INF;
                    if (lf_open > lmax) lmax = lf_open;
                    if (lf_ext > lmax) lmax = lf_ext;
                    uint8_t tl = 0;
                    if (lf_open == lmax) tl |= 64;
                    if (lf_ext == lmax) tl |= 128;
                    o[cell * 3] = mmax; o[cell * 3 + 1] = umax; o[cell * 3 + 2] = lmax;
                    t[cell * 3] = tm; t[cell * 3 + 1] = tu; t[cell * 3 + 2] = tl;
                    curM = mmax; curU = umax; curL = lmax;
                }
                upM = curM; upU = curU; upL = curL;
            }
            if (y >= 1) { dgM = lfM; dgU = lfU; dgL = lfL; }  // (y, x-1) is the diagonal of (y+1, x)
            if (lane == 63) { hand[gs & 1][wave][0] = curM; hand[gs & 1][wave][1] = curU; hand[gs & 1][wave][2] = curL; }
            __syncthreads();
        }
        // the next group's wave 0 reads this group's last column back from o
        __threadfence();
        __syncthreads();
    }
}

// End cell + traceback on the raw o / t buffers (align.py:401-431, util/align.py:144-185,268-297).
// One workgroup: a parallel first-argmax / row-col maxima, then thread 0 walks the path.
__global__ __launch_bounds__(256) void k_raw_trace(int mode, const float *__restrict__ o,
                                                    const uint8_t *__restrict__ t, int L1, int L2,
                                                    float *__restrict__ score_out,
                                                    int32_t *__restrict__ path,
                                                    int64_t *__restrict__ path_info)
{
    __shared__ float s_val[256];
    __shared__ long long s_idx[256];
    const int tid = threadIdx.x;
    const int64_t C = L2 + 1, R = L1 + 1;
    const bool semiglobal = mode >= 2;
    const bool trace_from_row = mode_free_two(mode);
    int64_t cy = L1, cx = L2, ck = 0;
    if (mode == 1) {
        // first flat argmax (align.py:402): per-thread contiguous chunks keep index order
        const int64_t n = R * C * 3;
        const int64_t chunk = (n + 255) / 256;
        const int64_t lo = tid * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
        float bv = PRALINE_NEG_INF; long long bi = -1;
        for (int64_t i = lo; i < hi; ++i) { const float v = o[i]; if (bi < 0 || v > bv) { bv = v; bi = i; } }
        s_val[tid] = bv; s_idx[tid] = bi;
        __syncthreads();
        if (tid == 0) {
            float v = s_val[0]; long long b = s_idx[0];
            for (int q = 1; q < 256; ++q) if (s_idx[q] >= 0 && (b < 0 || s_val[q] > v)) { v = s_val[q]; b = s_idx[q]; }
            cy = b / (C * 3); cx = (b / 3) % C; ck = b % 3;
        }
    }
    if (tid != 0) return;
    if (mode == 0) {
        const float *q = o + ((int64_t)L1 * C + L2) * 3;
        ck = 0;
        if (q[1] > q[ck]) ck = 1;
        if (q[2] > q[ck]) ck = 2;
    } else if (semiglobal) {
        float rmax = PRALINE_NEG_INF, cmax = PRALINE_NEG_INF;
        for (int64_t x = 0; x < C; ++x) for (int k = 0; k < 3; ++k) rmax = __builtin_fmaxf(rmax, o[((int64_t)L1 * C + x) * 3 + k]);
        for (int64_t y = 0; y < R; ++y) for (int k = 0; k < 3; ++k) cmax = __builtin_fmaxf(cmax, o[(y * C + L2) * 3 + k]);
        bool found = false;
        if (rmax > cmax && trace_from_row) {
            for (int64_t x = C - 1; x >= 0 && !found; --x)
                for (int k = 0; k < 3; ++k)
                    if (o[((int64_t)L1 * C + x) * 3 + k] == rmax) { cy = L1; cx = x; ck = k; found = true; break; }
        } else {
            for (int64_t y = R - 1; y >= 0 && !found; --y)
                for (int k = 0; k < 3; ++k)
                    if (o[(y * C + L2) * 3 + k] == cmax) { cy = y; cx = L2; ck = k; found = true; break; }
        }
    }
    *score_out = o[(cy * C + cx) * 3 + ck];
    const int64_t cap = L1 + L2 + 2;
    int64_t w = cap;
    auto emit = [&](int64_t yy, int64_t xx) { --w; path[2 * w] = (int32_t)yy; path[2 * w + 1] = (int32_t)xx; };
    int64_t y = cy, x = cx, k = ck;
    if (semiglobal) {
        if (y != L1) { for (int64_t yy = L1; yy > y; --yy) emit(yy, x); }
        else if (x != L2) { for (int64_t xx = L2; xx > x; --xx) emit(y, xx); }
    }
    emit(y, x);
    for (int64_t guard = 0; guard < cap; ++guard) {
        const uint8_t f = t[(y * C + x) * 3 + k];
        if (f & 2) { --y; --x; k = 0; }
        else if (f & 4) { --y; --x; k = 1; }
        else if (f & 8) { --y; --x; k = 2; }
        else if (f & 16) { --y; k = 0; }
        else if (f & 32) { --y; k = 1; }
        else if (f & 64) { --x; k = 0; }
        else if (f & 128) { --x; k = 2; }
        else break;
        emit(y, x);
    }
    if (semiglobal) {
        if (y != 0) { for (int64_t yy = y - 1; yy >= 0; --yy) emit(yy, 0); }
        else if (x != 0) { for (int64_t xx = x - 1; xx >= 0; --xx) emit(0, xx); }
    }
    path_info[0] = w;
    path_info[1] = cap - w;
}
#endif  // PRALINE_AUX_KERNELS
