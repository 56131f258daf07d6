/*
 * praline_dp.h -- C ABI of libpraline_dp.so, the MI355X (gfx950) pairwise-DP hot path of PRALINE 2.
 *
 * This is the drop-in boundary: plain pointers and sizes, no Python / numpy / torch types.
 * Section A replaces, one for one, the six functions the reference's native extension exports
 * (praline/util/cext.c:506-520) and that praline/component/align.py:18-20,27-31,209,388 binds.
 * Section B is the batched form of the same path (one submission per all-pairs stage instead of
 * one Python call per pair): the seam the reference offers for it is Manager.execute_many
 * (praline/core/manager.py:154-170), which receives the whole request list of
 * GuideTreeBuilder (praline/component/tree.py:105-147), the master-slave aligners
 * (praline/component/preprofile.py:127-154,227-267) and AdHoc re-scoring
 * (praline/component/msa.py:488-558).
 *
 * All entry points return 0 on success and a negative PRALINE_ERR_* code otherwise;
 * praline_last_error() gives the message of the calling thread's last failure.  The library
 * owns its device memory and streams.  There is NO CPU fallback: without a usable HIP device
 * every compute entry point fails with PRALINE_ERR_DEVICE.
 *
 * Threading: one host thread per device (praline_init binds the calling process to a device).  The one exception is
 * praline_sched_prepare / praline_sched_destroy: host-only, no shared state - they may run on any other thread.
 */
#ifndef PRALINE_DP_H
#define PRALINE_DP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRALINE_DP_ABI_VERSION 1

#define PRALINE_OK 0
#define PRALINE_ERR_ARG (-1)         /* bad argument (NULL, negative size, unknown mode ...) */
#define PRALINE_ERR_DEVICE (-2)      /* no HIP device / HIP runtime error */
#define PRALINE_ERR_NOMEM (-3)       /* host or device allocation failed */
#define PRALINE_ERR_UNSUPPORTED (-4) /* valid request the batched kernels do not cover */

/* Alignment modes: the five dispatch keys of _CEXT_ALIGN_FUNCTIONS
 * (praline/component/align.py:27-31; praline/util/cext.c:27-31). */
#define PRALINE_MODE_GLOBAL 0
#define PRALINE_MODE_LOCAL 1
#define PRALINE_MODE_SEMIGLOBAL_BOTH 2
#define PRALINE_MODE_SEMIGLOBAL_ONE 3
#define PRALINE_MODE_SEMIGLOBAL_TWO 4

/* Traceback flag bits written to t (praline/util/cext.c:9-15, praline/util/align.py:15-21). */
#define PRALINE_TB_MATCH_MATCH (1 << 1)
#define PRALINE_TB_MATCH_INSERT_UP (1 << 2)
#define PRALINE_TB_MATCH_INSERT_LEFT (1 << 3)
#define PRALINE_TB_INSERT_UP_OPEN (1 << 4)
#define PRALINE_TB_INSERT_UP_EXTEND (1 << 5)
#define PRALINE_TB_INSERT_LEFT_OPEN (1 << 6)
#define PRALINE_TB_INSERT_LEFT_EXTEND (1 << 7)

/* How the batched plans (section B) evaluate the match scores m = sum_sets P1 . S . P2^T (cext.c:33-97,308-455):
 *   FAST       matrix pipe: exact f16 hi/lo split of the fp32 operands, fp32 accumulation.  Bit-identical to the
 *              reference for integer scoring (one-hot profiles x integral matrix); within ~1e-6 relative otherwise.
 *   F32        fp32 MFMA chain (k-ordered fmaf chain); same exactness statement, fp32 operands throughout.
 *   REFERENCE  the reference's own summation order on the vector ALU: per track set one float32 running sum over the
 *              nonzeros of both profile rows, every product and sum rounded separately as the reference binary does.
 *              Bit-identical match scores - hence scores AND alignments identical to the reference's - for ANY
 *              profiles; ~10-20x slower (an audit / strict-parity mode).
 * The mode is read when a plan is created.  Default: FAST, or what the environment variable PRALINE_MM
 * ("f32" / "ref") names; praline_set_match_mode(-1) returns to that default. */
#define PRALINE_MATCH_FAST 0
#define PRALINE_MATCH_F32 1
#define PRALINE_MATCH_REFERENCE 2

/* ---- runtime ------------------------------------------------------------------------------- */
int praline_abi_version(void);
int praline_device_count(int *count);         /* number of visible HIP devices */
int praline_init(int device);                 /* bind this process to `device` (lazy otherwise: 0) */
int praline_shutdown(void);                   /* release streams and cached device buffers */
int praline_synchronize(void);                /* wait for all work submitted by this library */
const char *praline_last_error(void);         /* message of the last failure on this thread */
/* Scratch buffers (strip boundaries, traceback planes, paths) are recycled through a pool of released device
 * blocks (cap: PRALINE_POOL_KEEP_MB, default 65536).  praline_pool_trim waits for the library's work and returns
 * every cached block to the driver - call it before handing the GPU's memory to another allocator. */
int praline_pool_trim(void);
int64_t praline_pool_cached_bytes(void);
void *praline_stream(void);                   /* the hipStream_t all kernels are launched on */
/* Page-locked host memory for the caller's staging of inputs: profiles handed to praline_arena_create from such a
 * buffer go up by DMA as they lie (11 MB of C2: 0.2 ms), without the runtime's copy through its own staging pages
 * that a pageable buffer costs.  Optional - every entry point takes any host pointer. */
int praline_host_alloc(size_t bytes, void **out);
int praline_host_free(void *p);
int praline_set_match_mode(int kind);         /* PRALINE_MATCH_*; -1 = back to the PRALINE_MM default */
int praline_get_match_mode(void);

/* ============================================================================================
 * A. Parity-layout entry points: the reference's native functions on raw host buffers.
 *    Ownership is the reference's: the caller allocates every buffer, outputs are written in
 *    place (praline/util/cext.c:303-305,452-454).  Unlike the reference (no checks, UB on misuse,
 *    praline/component/align.py:196-199) bad arguments return PRALINE_ERR_ARG.
 *    Arrays are described by pointer + dims + BYTE strides, honouring arbitrary numpy strides as
 *    the reference does (cext.c:109-129,356-377).
 * ========================================================================================== */
typedef struct praline_array {
    void *data;
    int64_t dim[3];     /* unused trailing dims = 1 */
    int64_t stride[3];  /* bytes; unused trailing strides = 0 */
} praline_array;

/* Replaces cext_build_scores(i1s, i2s, i1nzs, i2nzs, ss, m)  (praline/util/cext.c:308-455).
 *   i1s[n]: float32 [L1][A1_n] profile of sequence one for track set n; i2s[n]: [L2][A2_n];
 *   ss[n] : float32 [A1_n][A2_n] score matrix; m: float32 [L1][L2] output, written in place.
 *   i1nzs / i2nzs (the reference's -1 padded nonzero index lists, align.py:449-458) are accepted
 *   for signature compatibility and may be NULL: the device evaluates the dense contraction
 *   m = sum_n P1_n . S_n . P2_n^T with fp32 MFMA, which equals the reference's sparse loop
 *   exactly when all terms are exactly representable (one-hot / integer scoring) and to fp32
 *   rounding (<= 1e-5 relative) otherwise. */
int praline_build_scores(int num_sets, const praline_array *i1s, const praline_array *i2s,
                         const praline_array *i1nzs, const praline_array *i2nzs,
                         const praline_array *ss, const praline_array *m);

/* Replace cext_align_<mode>(m, g1, g2, o, t, z)  (praline/util/cext.c:99-306,457-485).
 *   m float32 [L1][L2]; g1 float32 [L1][2], g2 float32 [L2][2] (open, extend per position);
 *   o float32 [L1+1][L2+1][3] and t uint8 [L1+1][L2+1][3] in/out (caller pre-initialises the
 *   boundaries as RawPairwiseAligner does, align.py:357-385); z uint8 [L1+1][L2+1] mask.
 *   Results are bit-identical to the reference fill (all seven tie flags included). */
int praline_align_global(const praline_array *m, const praline_array *g1, const praline_array *g2,
                         const praline_array *o, const praline_array *t, const praline_array *z);
int praline_align_local(const praline_array *m, const praline_array *g1, const praline_array *g2,
                        const praline_array *o, const praline_array *t, const praline_array *z);
int praline_align_semiglobal_both(const praline_array *m, const praline_array *g1,
                                  const praline_array *g2, const praline_array *o,
                                  const praline_array *t, const praline_array *z);
int praline_align_semiglobal_one(const praline_array *m, const praline_array *g1,
                                 const praline_array *g2, const praline_array *o,
                                 const praline_array *t, const praline_array *z);
int praline_align_semiglobal_two(const praline_array *m, const praline_array *g1,
                                 const praline_array *g2, const praline_array *o,
                                 const praline_array *t, const praline_array *z);
/* Same, mode as an argument (PRALINE_MODE_*). */
int praline_align(int mode, const praline_array *m, const praline_array *g1,
                  const praline_array *g2, const praline_array *o, const praline_array *t,
                  const praline_array *z);

/* RawPairwiseAligner.execute in one call (praline/component/align.py:357-447): boundary init,
 * fill, end-cell selection (align.py:401-431), traceback (praline/util/align.py:144-185) and
 * semiglobal path extension (praline/util/align.py:268-297) on the device; only the score and the
 * path travel back.  path: int32 [L1+L2+2][2] caller buffer, *path_rows receives the row count.
 * z may be NULL. */
int praline_raw_align(int mode, const praline_array *m, const praline_array *g1,
                      const praline_array *g2, const praline_array *z, float *score,
                      int32_t *path, int64_t *path_rows);

/* ============================================================================================
 * B. Batched entry points: a profile arena resident in HBM + pair lists.
 * ========================================================================================== */
typedef struct praline_arena praline_arena; /* profiles of N sequences, packed for the kernels */
typedef struct praline_plan praline_plan;   /* a scheduled pair list (wave tasks) on the device */

/* Uploads N sequences.  profiles: float32 [sum(lens)][A], sequence s occupying rows
 * row_off(s) = lens[0]+...+lens[s-1]; this is PairwiseAligner's per-track fp32 profile
 * (one-hot for PlainTrack, counts/rowsum for ProfileTrack, praline/component/align.py:163-177).
 * Several track sets are passed concatenated along the alphabet axis (A = sum A_t) with
 * S = blockdiag(S_t) (cext.c:389-420 sums the sets).  S: float32 [A][A], row = symbol of
 * sequence one, column = symbol of sequence two (align.py:205).  Runs the profile x matrix
 * pre-multiply (MFMA) on the device.
 * Limits: A <= 254.  The MFMA operand layouts hold up to 32 ACTIVE symbols - symbols that have mass in some
 * profile and a non-zero row in S (BLOSUM62 on 20-residue data: 20 of 27); an arena with more keeps the raw
 * profiles only and its plans take the PRALINE_MATCH_REFERENCE path whatever the match mode (correct for any
 * alphabet, bit-identical to the reference, an order of magnitude slower). */
int praline_arena_create(int64_t n_seqs, const int32_t *lens, int32_t A, const float *profiles,
                         const float *S, praline_arena **out);
/* The same in three steps, for callers that assemble `profiles` piece by piece (a binding that concatenates per-sequence
 * arrays into page-locked staging): begin sizes the arena, put_rows uploads rows [row0, row0 + n_rows) of the
 * concatenation - asynchronously: the upload of the first part runs while the caller copies the second; `rows` must stay
 * valid until praline_arena_finish returns, and part 0 (row0 = 0) is where the library looks for non-float16 values -,
 * finish does the rest of praline_arena_create.  Between begin and finish only put_rows and destroy may be called;
 * a failing finish destroys the arena. */
int praline_arena_begin(int64_t n_seqs, const int32_t *lens, int32_t A, praline_arena **out);
int praline_arena_put_rows(praline_arena *arena, int64_t row0, int64_t n_rows, const float *rows);
int praline_arena_finish(praline_arena *arena, const float *S);
int praline_arena_destroy(praline_arena *arena);
/* Tells the arena how its alphabet axis is partitioned into track sets (sizes[0] + ... + sizes[n_sets-1] = A).
 * Only PRALINE_MATCH_REFERENCE needs it: the reference keeps one running sum per set and adds the sets in list
 * order (cext.c:389-420).  Default: one set of size A. */
int praline_arena_set_track_sets(praline_arena *arena, int32_t n_sets, const int32_t *sizes);

/* Resident progressive alignment (the N - 1 merge steps of TreeMultipleSequenceAligner / AdHocMultipleSequenceAligner,
 * praline/component/msa.py:124-237,250-558).  praline_arena_set_counts hands the arena the INTEGER counts behind its
 * profile rows (int32 [sum of lengths][A]: one-hot counts for plain tracks, ProfileTrack.counts otherwise,
 * msa.py:71-98) and reserves room for reserve_seqs further sequences of reserve_rows rows in total (it grows by itself
 * beyond that).  praline_arena_append_merged merges the two sequences of pair `pair_index` of a path plan that has been
 * run (global or semiglobal mode) along that pair's device path - ProfileTrack.merge for every track set
 * (praline/container/sequence.py:205-239), profile rows as ProfileTrack.profile forms them (sequence.py:192-203) - and
 * appends the result to the arena as a NEW sequence (packed operands included): the growing clusters never leave the
 * GPU; only the path itself is needed on the host, for the bookkeeping of Alignment.merge
 * (praline/container/align.py:30-61).  After the first append the arena is no longer one-hot. */
int praline_arena_set_counts(praline_arena *arena, const int32_t *counts, int64_t reserve_seqs, int64_t reserve_rows);
int praline_arena_append_merged(praline_arena *arena, praline_plan *plan, int64_t pair_index, int32_t *new_index,
                                int32_t *new_len);
/* The same for n pairs of one plan (pair_index[n] -> new_index[n], new_len[n]): merge steps of DIFFERENT subtrees of the
 * guide tree do not depend on each other, so a whole level of the tree is one path plan and one call here (one round
 * trip for the path locations, the merge kernels of all pairs, one packing launch over the new rows). */
int praline_arena_append_merged_many(praline_arena *arena, praline_plan *plan, int64_t n, const int64_t *pair_index,
                                     int32_t *new_index, int32_t *new_len);
/* Re-runs the device-side packing + pre-multiply from the resident raw profiles (the part of
 * cext_build_scores that is per sequence, not per pair); asynchronous on praline_stream(). */
int praline_arena_premultiply(praline_arena *arena);

/* Schedules n_pairs alignments (pairs: int32 [n][2] = (sequence_one, sequence_two) arena
 * indices, any order, duplicates allowed).  want_paths != 0 additionally reserves packed
 * traceback storage.  rect_off (int32 [n+1]) / rects (int32 [rect_off[n]][4] = y0,y1,x0,x1
 * inclusive DP coordinates) give per-pair zero rectangles (the Waterman-Eggert masks of
 * praline/component/preprofile.py:247-255); both may be NULL.  Any number of rectangles per pair: up to 4 per pair
 * run on the split-strip kernels (masks held in registers); plans in which some pair has more take the
 * PRALINE_MATCH_REFERENCE path with per-row column masks prepared on the device (correct, an order of magnitude
 * slower - more than five Waterman-Eggert iterations are rare). */
int praline_plan_create(praline_arena *arena, int64_t n_pairs, const int32_t *pairs,
                        int want_paths, const int32_t *rect_off, const int32_t *rects,
                        praline_plan **out);
int praline_plan_destroy(praline_plan *plan);
/* The host scheduling of a scores-only plan ahead of its arena: praline_sched_prepare needs the sequence lengths and the
 * pair list only and touches no device - it may run on another host thread while praline_arena_create uploads and
 * packs the same sequences (C2: 0.4 ms of scheduling beside 0.7 ms of arena creation).  praline_plan_create_prepared
 * is praline_plan_create(arena, n_pairs, pairs, 0, NULL, NULL, out) that takes the prepared schedule when it fits the
 * arena (same lengths, same pair list, a kernel that runs that kind of schedule) and schedules itself otherwise;
 * `sched` may be NULL.  The caller destroys the schedule object either way. */
typedef struct praline_sched praline_sched;
int praline_sched_prepare(int64_t n_seqs, const int32_t *lens, int64_t n_pairs, const int32_t *pairs, praline_sched **out);
int praline_sched_destroy(praline_sched *sched);
int praline_plan_create_prepared(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, praline_sched *sched,
                                 praline_plan **out);
int64_t praline_plan_cells(const praline_plan *plan);      /* sum L1*L2 over the pairs */
/* Diagnostics: wavefront steps one praline_plan_run executes (one step = one DP row of a 32-pair x 32-column
 * strip = 1024 cells incl. padding) and the number of 32-pair tasks; bench.py prices VALU issue with them. */
int64_t praline_plan_steps(const praline_plan *plan);
int64_t praline_plan_tasks(const praline_plan *plan);
int64_t praline_plan_path_capacity(const praline_plan *plan); /* rows: sum (L1+L2+2) */

/* Launches the fused match-score + affine DP fill for every pair of the plan (asynchronous on
 * praline_stream()).  gap_open/gap_extend: PairwiseAligner's gap_series [open, extend]
 * (align.py:182-189,212-217; both <= 0).  d_scores: DEVICE pointer to n_pairs float32 in pair
 * order, or NULL to use the plan's own buffer. */
int praline_plan_run(praline_plan *plan, int mode, float gap_open, float gap_extend,
                     void *d_scores);
/* Per-position gap scores.  The reference's fill honours one (open, extend) per position of each sequence
 * (GapScoreModel, praline/container/score.py:45-68; cext.c:155-158 reads g1[y-1] for U[y][x], cext.c:172-175 g2[x-1]
 * for L[y][x]; boundary cells align.py:371-385), PairwiseAligner fills the rows with one constant (align.py:212-217).
 * praline_arena_set_gap_scores stores g = float32 [sum of the lengths][2] (arena order, all values <= 0; NULL removes
 * them) on the device; plans created while they are set keep the k_dp_batch task layout, take their match scores from
 * dense matrices (the fp32 MFMA chain of praline_build_scores kind 0, or the reference order under
 * PRALINE_MATCH_REFERENCE) and run with them through praline_plan_run_gaps - scores-only or with paths, all five
 * modes, zero rectangles included.  praline_plan_run on such a plan still takes one constant pair.  Constant-gap
 * plans are not affected (the tuned split-strip kernels). */
int praline_arena_set_gap_scores(praline_arena *arena, const float *g);
int praline_plan_run_gaps(praline_plan *plan, int mode, void *d_scores);
/* Copies the scores of the last praline_plan_run (pair order) to the host - from the plan's own buffer or from
 * the d_scores that run was given; synchronises. */
int praline_plan_scores(praline_plan *plan, float *scores);
/* Device pointer of the plan's own score buffer. */
void *praline_plan_device_scores(praline_plan *plan);
/* want_paths plans: copies the alignment paths to the host; synchronises.  path_rows: int32 [n];
 * path_off: int64 [n] row offset of pair p's first row inside `paths`
 * (int32 [praline_plan_path_capacity][2], rows (y, x) in start->end order, semiglobal paths
 * already extended to the corners, praline/util/align.py:268-297). */
int praline_plan_paths(praline_plan *plan, int32_t *paths, int64_t *path_off, int32_t *path_rows);

/* Preprofile stage on the device.  For plans whose pairs are (master, slave) alignments with paths,
 * praline_plan_add_counts adds, for every pair whose score passes the threshold, the slave's share of the
 * master's profile counts - what the reference obtains on the host with compress_path + extend_path_local
 * (local != 0) + Alignment.merge + ProfileBuilder.get_frequencies (praline/component/preprofile.py:145-152,
 * 258-265, praline/util/align.py:187-266, praline/component/profile.py:41-74) - into the arena's count buffer
 * int32 [sum of lengths][A].  Requires one-hot profiles (plain sequences) for the slaves.  The masters' own
 * symbols are not added.  praline_plan_path_bounds returns (y0, y1, x0, x1) per pair: the bounding box the next
 * Waterman-Eggert iteration masks (preprofile.py:247-255). */
int praline_arena_counts_reset(praline_arena *arena);
/* Accumulate the counts in a caller-owned DEVICE buffer int32 [sum of lengths][A] instead of the arena's own (so that a
 * multi-GPU caller can all-reduce them in place); NULL returns to the arena's buffer.  praline_arena_counts_reset
 * zeroes whichever buffer is bound, on praline_stream(). */
int praline_arena_counts_bind(praline_arena *arena, void *d_counts);
int praline_plan_add_counts(praline_plan *plan, int use_threshold, float threshold, int local);
int praline_arena_counts_read(praline_arena *arena, int32_t *counts);
int praline_plan_path_bounds(praline_plan *plan, int32_t *bounds);
/* Waterman-Eggert without a host round trip: the bounding box of every pair's current path (of the last
 * praline_plan_run) becomes one more zero rectangle of that pair, on the device, and the SAME plan - its schedule and
 * scratch - runs the next iteration.  Up to 4 per pair this way (plans created without rectangle lists only). */
int praline_plan_mask_path_bounds(praline_plan *plan);

/* Convenience: arena-resident one-shot (plan + run + copy back). */
int praline_batch_scores(praline_arena *arena, int mode, float gap_open, float gap_extend,
                         int64_t n_pairs, const int32_t *pairs, float *scores);

/* A batch of RawPairwiseAligner requests (praline/component/align.py:254-447) in one submission - the operator the
 * reference runs once per caller-supplied MatchScoreModel / GapScoreModel pair (align.py:302-447 on cext_align_*,
 * praline/util/cext.c:99-306).  Request r aligns its own match scores m_r (float32 [l1[r]][l2[r]], row-major) under its own
 * gap scores g1_r (float32 [l1[r]][2]) and g2_r (float32 [l2[r]][2]: open, extend per position, align.py:346-348) with its
 * own zero cells (zero_idxs: (y, x) cells of the DP matrix fixed to zero, align.py:362-367).  m, g1, g2: the requests'
 * arrays one after the other, host or device memory; zero_off [n + 1] (host memory) delimits request r's (y, x) pairs in
 * zero_idx (host or device memory; both NULL: no zero cells; cells outside 1 .. l1[r] x 1 .. l2[r] have no effect, as in
 * cext.c:141-149, whose loops start at 1).  l1, l2: host memory.  The inputs are copied once; praline_raw_batch_run may be called any number of times (per-request
 * modes, or one mode for all with modes = NULL), asynchronously on the library stream; praline_raw_batch_results waits and
 * returns the scores and path lengths, praline_raw_batch_paths the paths (int32 (y, x) rows as get_paths +
 * extend_path_semiglobal produce them, praline/util/align.py:144-185, 268-297) one after the other in request order.
 * Scores, end cells and paths are those of the reference's RawPairwiseAligner bit for bit (same fp32 operations in the same
 * order, same tie rules). */
typedef struct praline_raw_batch praline_raw_batch;
int praline_raw_batch_create(int64_t n, const int32_t *l1, const int32_t *l2, const float *m, const float *g1, const float *g2,
                             const int64_t *zero_off, const int32_t *zero_idx, praline_raw_batch **out);
/* The same with one pointer per request: m[r], g1[r], g2[r] (host or device memory each) - no concatenation on the caller's side. */
int praline_raw_batch_create_v(int64_t n, const int32_t *l1, const int32_t *l2, const float *const *m, const float *const *g1,
                               const float *const *g2, const int64_t *zero_off, const int32_t *zero_idx, praline_raw_batch **out);
int praline_raw_batch_run(praline_raw_batch *batch, const int32_t *modes, int mode);
int praline_raw_batch_results(praline_raw_batch *batch, float *scores, int64_t *path_rows);
int praline_raw_batch_paths(praline_raw_batch *batch, int32_t *paths, int64_t cap_rows);
int64_t praline_raw_batch_cells(const praline_raw_batch *batch);
/* Device time of the last run (boundary rows + fill + end cells and paths), valid after praline_raw_batch_results. */
int praline_raw_batch_last_timing(const praline_raw_batch *batch, float *kernel_ms);
void praline_raw_batch_destroy(praline_raw_batch *batch);

/* Guide-tree clustering (host code, no device needed): the merge order of the agglomerative clustering of
 * praline/util/cluster.py:27-114 on an n x n float64 distance matrix (row-major; (i, j) and (j, i) are read separately
 * as in the reference) - repeatedly the first minimum of the cluster linkage table in cluster-id order, the merged
 * cluster keeping the id of the first.  linkage: 0 single, 1 complete, 2 average.  order: int32 [n - 1][2]. */
int praline_merge_order(int64_t n, const double *dist, int linkage, int32_t *order);

/* Diagnostics / audits.  praline_plan_match_kind: 0 fp32 MFMA chain, 1 f16 split, 2 reference order.
 * praline_arena_match_scores writes the dense match-score matrix
 * m (float32 [L1][L2], host) of the arena pair (one, two) exactly as the kernels evaluate it:
 * kind 0 = fp32 MFMA chain (k-ordered fmaf chain, used by the traceback plans and
 * praline_build_scores), kind 1 = f16 hi/lo split on the matrix pipe (used by scores-only plans;
 * identical to kind 0 whenever all operands are f16-representable, e.g. one-hot x integer matrix),
 * kind 2 = the reference's own summation order (PRALINE_MATCH_REFERENCE plans, arenas with > 32 active symbols).
 * praline_plan_match_kind tells which of the two a plan's praline_plan_run uses. */
int praline_arena_match_scores(praline_arena *arena, int32_t one, int32_t two, int kind, float *m);
/* f16_terms: 1 = every operand is exactly f16-representable (single term, bit-exact), 3 = hi/lo split in three terms
 * of f16_ranges MFMAs each, 2 = the same three terms K-packed into four MFMAs (at most 21 active symbols). */
int praline_arena_info(const praline_arena *arena, int32_t *n_active, int32_t *mfma_steps_f32,
                       int32_t *f16_ranges, int32_t *f16_terms);
int praline_plan_match_kind(const praline_plan *plan);
/* Who writes the match scores of this plan's runs: 0 - the fill itself (matrix pipe / lookup; no match-score matrix in
 * memory); plans whose fill reads dense tiles: 1 - k_match_tile (reference order, <= 32 symbols, <= 8 nonzeros per row),
 * 2 - one thread per cell (reference order, any alphabet and row density), 3 - the fp32 MFMA chain (plans made on an
 * arena with per-position gap scores).  -1: NULL plan. */
int praline_plan_tile_producer(const praline_plan *plan);

/* Timing of the last praline_plan_run on THIS plan, measured with the plan's own HIP events on the launch
 * stream: kernel_ms = the DP kernel alone (scores-only plans) / fill + end cells + traceback (path plans). */
int praline_plan_last_timing(praline_plan *plan, float *kernel_ms);
/* Registers per lane, LDS bytes per workgroup and the resulting resident waves per SIMD of the kernel instance the
 * last run launched (hipFuncGetAttributes); zeros for instances that are not reported (measurement aid, bench.py). */
int praline_plan_kernel_resources(const praline_plan *plan, int32_t *vgprs, int32_t *lds_bytes, int32_t *waves_per_simd);
/* The DP kernel instance the last praline_plan_run launched, spelled as rocprofv3 prints it without the leading
 * "void " (e.g. "k_dp_split16<2, 3, false, 2, 4>"); bench.py matches profile files against it. */
int praline_plan_kernel_name(const praline_plan *plan, char *buf, int64_t size);

#ifdef __cplusplus
}
#endif
#endif /* PRALINE_DP_H */
