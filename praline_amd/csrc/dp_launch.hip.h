// dp_launch.hip.h -- launch glue: what praline_dp.hip calls into the kernel translation units (one per kernel family, so the
// instances compile in parallel).
#pragma once
#include "dp_kernels.hip.h"
#include "dp_arena16.h"
#include "dp_reftile.h"
#include "praline_dp.h"

struct LaunchArgs {
    ArenaDev ar;
    const WaveTask *tasks;
    const int32_t *lane_one, *lane_pair;
    void *bnd;
    uint4 *tb;
    float *aux;
    RectList rl;
    float *scores;
    int32_t *end_cells;
    RunParams rp;
    unsigned n_tasks;
    hipStream_t stream;
    const struct Arena16Dev *a16;  // non-null: run k_dp_split16 on these f16 operands
    int nr16, nterm16;
    const struct WgDesc *wg;  // non-null: four-wave workgroups (k_dp_split16 WPG = 4), n_wg of them
    unsigned n_wg;
    int split;  // 1: k_dp_split task layout (32 lane entries per task, float2 [max_l1+2][32] boundary)
};

// k_dp_split instances (dp_split_instance.hip, built with -mllvm -amdgpu-mfma-vgpr-form)
int praline_launch_split_2(const LaunchArgs &la, bool local);
int praline_launch_split_8(const LaunchArgs &la, bool local);
int praline_launch_split_10(const LaunchArgs &la, bool local);
int praline_launch_split_12(const LaunchArgs &la, bool local);
int praline_launch_split_14(const LaunchArgs &la, bool local);
int praline_launch_split_16(const LaunchArgs &la, bool local);
// k_dp_split16 and friends (dp_split16_instance.hip)
int praline_launch_split16(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local);
int praline_launch_split16_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask);
int praline_launch_split16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask,
                                    int max_strips, int *flags, void *cand, int every);
void praline_launch_split_f16(const float *src, int KP, int KS, int n_active, int NR, int64_t rows_pad, void *dst, int *flag,
                              hipStream_t stream);
int praline_launch_scores_tile16(const Arena16Dev &a16, int nr, int nterm, int one, int two, int L1, int L2, float *m,
                                 hipStream_t stream);
// k_dp_pipe (dp_pipe_instance.hip): pipeline workgroups over PipeItem lists (scores only)
struct PipeLaunch {
    const PipeItem *items;
    unsigned n_items;
    const WaveTask *tasks;
    const int32_t *set_one, *lane_pair;
    void *bnd;
    void *analytic;         // float2 [analytic_rows][32]: the analytic column 0 of this run's mode and gap scores
    int analytic_rows;
    bool analytic_valid;    // the buffer already holds it (the plan's previous run had the same mode and gap scores)
    float *scores;
    RunParams rp;
    hipStream_t stream;
};
bool praline_pipe_supported(int nr, int nterm);
int praline_pipe_attrs(int nr, int nterm, int mode, int *vgprs, int *lds_bytes);
int praline_launch_pipe(const PipeLaunch &pl, const Arena16Dev &a16, int nr, int nterm);
// k_dp_pipe<..., KEEP>: the forward fill of the two-pass alignments with paths on the pipeline (global mode); backward:
// praline_launch_tb2_backward with keep_in_aux = 2 (blocks of PRALINE_KEEP_BH rows, analytic column 0)
int praline_launch_pipe_keep(const PipeLaunch &pl, const Arena16Dev &a16, int nr, int nterm, void *keep_bnd, float *ckpt,
                             int32_t *end_cells, void *analytic4);
int praline_pipe_keep_attrs(int nr, int nterm, int *vgprs, int *lds_bytes);
// k_dp_quad_tb (dp_quad_instance.hip): fill with packed traceback for plain sequences, 16 pairs per task (la.lane_one /
// lane_pair [task][16], la.bnd float4 [row][16] per task, la.tb uint2 [strip][step][64]); mask: 0, 1 (rectangles), 2 (mask words)
int praline_launch_quad_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool ints, bool local, int mask);
// k_dp_pk16_tb (dp_pk16_instance.hip): the same for integer scoring within int16, two pairs per lane - the 32-pair task layout
// of the strip kernels (la.lane_one / lane_pair [task][32], la.bnd uint4 [row][16] per task, la.tb uint4 [strip][step][64] at
// tk.tb_off counted in uint2, la.aux as the strip kernels'); mask: rectangles in registers
int praline_launch_pk16_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, int mask, float scale);   // mask: rectangle slots per pair (0, 1, 2, <= PRALINE_MAX_RECTS)
// ... in chain mode (one wave per task and strip; la.bnd: one uint4 [row][16] column per strip boundary; flags, cand as
// praline_launch_split16_tb_chain's)
int praline_launch_pk16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, int mask, float scale, int max_strips,
                                 int *flags, void *cand, int every);
// two-pass alignments with paths (dp_tb2_instance.hip): flag-free forward fill, then block recompute + traceback
struct Trace2Args {
    const int64_t *slot_off;
    int32_t *paths;
    int64_t *path_start;
    int32_t *path_rows;
};
int praline_launch_tb2_forward(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask);
// keep_in_aux: 0 the forward was k_dp_split16_tb<..., TWOPASS>; 1 k_dp_split16<..., KEEP> (kept columns at tk.aux_off);
// 2 k_dp_pipe<..., KEEP> (the same, blocks of PRALINE_KEEP_BH rows, column 0 of every task = analytic4)
int praline_launch_tb2_backward(const LaunchArgs &la, const Arena16Dev &a16, const Trace2Args &ta, int nr, int nterm, bool local,
                                bool mask, int keep_in_aux = 0, const void *analytic4 = nullptr);
// chain mode without flags: scores only (score plans of a few long sequences)
int praline_launch_scores_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, int max_strips, int *flags,
                                void *cand, int every);
// the same forward fill on the staged scores kernel (k_dp_split16<..., KEEP>, la.wg workgroups): la.bnd its (H, L) hand-off
// columns, keep_bnd / ckpt the kept columns (at tk.aux_off) and row checkpoints (at tk.tb_off)
int praline_launch_keep_forward(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, void *keep_bnd, float *ckpt);
// reference-order match scores in the split-strip layout (dp_dense_instance.hip): the interleaved table, the tile kernel,
// and the dense-tile instances of the scores / path kernels (a16.dense, a16.dense_off set)
int praline_launch_build_reft2(const float *raw, const float *S, int A, const int32_t *row_off_raw, const int32_t *len,
                               const int64_t *pr_off, int64_t PR, const unsigned char *nzidx, const unsigned char *nzcnt,
                               const int32_t *set_lo, int n_sets, int TB, float *T2, int n_seqs, hipStream_t stream);
bool praline_match_tile_supported(int A, int TB);
int praline_launch_match_tile(const RefTileArgs &g, int TB, unsigned n_blocks, hipStream_t stream);
int praline_launch_dense(const LaunchArgs &la, const Arena16Dev &a16, bool local);
// fill with packed traceback on dense tiles.  mask: rectangles (la.rl.rects) or - la.rl.zmask set - per-row mask words; ppg:
// per-position gap scores (la.rp.gaps); noflags: the fill alone (plans without paths: no planes, la.tb unused); strip_lo /
// strip_cnt: the strips of every task this launch sweeps (the tile holds just those)
int praline_launch_dense_tb(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, bool ppg = false, bool noflags = false,
                            int strip_lo = 0, int strip_cnt = 0x3fffffff);
int praline_launch_dense_tb_ppg(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, bool noflags, int strip_lo,
                                int strip_cnt);   // (dp_dense2_instance.hip)
