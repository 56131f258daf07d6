// dp_types.h -- plain (HIP-free) descriptors shared by the host scheduler (sched.cpp) and the kernels.
#pragma once
#include <stdint.h>

struct WaveTask {
    int32_t two[2];   // shared sequence (arena index) of lanes 0-31 / 32-63, -1 = half unused
    int32_t max_l1;   // longest sequence one among the lanes
    int32_t nstrips;  // max over halves of ceil(len(two)/32)
    int64_t bnd_off;  // element offset of the strip-boundary scratch [max_l1+1][64]
    int64_t tb_off;   // uint4 offset of the packed traceback planes [nstrips][max_l1+1][64]
    int64_t aux_off;  // float offset of the end-cell scratch: lastcol [max_l1+1][3][64] then
                      // lastrow [nstrips*32][3][64]  (semiglobal paths only)
};

struct PairLoc { int32_t task; int32_t lane; };   // where pair p runs

// Workgroup descriptor of the four-wave launch of k_dp_split16: `share` consecutive waves pipeline one task
// (share = 1, 2 or 4; the task id sits in the slot of the group's first wave, -1 = idle waves); `barriers` =
// s_barriers every wave of the workgroup executes (0 when share == 1).
struct WgDesc {
    int32_t task[4];
    int32_t share;
    int32_t barriers;
    int32_t pad[2];
};

// Pipeline workgroups (k_dp_pipe, dp_pipe.hip.h): the four waves of a workgroup sweep the strips of a LIST of tasks that
// share one set of 32 sequences one - wave r takes the strips r, r + 4, ... of the concatenated strip list and runs
// PRALINE_PIPE_LAG steps behind wave r - 1 - so the set's operand rows are streamed into LDS once per workgroup and
// step, and a strip's boundary column reaches the next strip through LDS (every fourth hand-off through `bnd`).
struct PipeItem {
    int32_t set;        // index into set_one [n_sets][32]
    int32_t task0;      // first task of the list (tasks task0 .. task0 + ntasks - 1 share the set; PipeTask order)
    int32_t ntasks;
    int32_t nstrips;    // strips of the whole list
    int32_t rsteps;     // steps per round (a multiple of 12, >= max_l1 + 1, >= PRALINE_PIPE_MIN_STEPS)
    int32_t nrounds;    // ceil(nstrips / 4)
    int64_t bnd_off;    // float2 element offset of the workgroup's wrap-around boundary column [rsteps + 16][32]
};
#define PRALINE_PIPE_LAG 2          // steps between consecutive waves of a pipeline workgroup
#define PRALINE_PIPE_RING 12        // operand rows held by the workgroup's LDS ring
#define PRALINE_PIPE_MIN_STEPS 36   // shortest round: the wrap-around hand-off (wave 3 -> wave 0) goes through memory
#define PRALINE_PIPE_MAX_TASKS 4    // tasks per item (the per-task result table lives in LDS: 256 bytes per task)

// two-pass alignments with paths: rows per kept boundary column beyond max_l1, and checkpoint blocks per strip (one
// per 32 rows; the unrolled loops compute rows up to max_l1 + 12; block 0 is never written)
#define PRALINE_TB2_PAD 72
#define PRALINE_TB2_CKPT_BLOCKS(max_l1) (((max_l1) + 12) / 32 + 1)
#define PRALINE_TB2_CKPT_FLOATS (3 * 16 * 64)   // floats per checkpoint block: float4 [state][four-column group][64 lanes]

// the pipeline kernel as the forward fill of the two-pass scheme (k_dp_pipe<..., KEEP>, dp_pipe.hip.h): rows between two
// kept (M, U, L) rows = rows per block k_trace_recompute rebuilds; a multiple of 12 (the pipeline's unrolled loop)
#ifndef PRALINE_KEEP_BH
#define PRALINE_KEEP_BH 36
#endif

// k_dp_quad_tb (dp_quad.hip.h): steps per strip - two DP rows per step, quarter q of a pair's 32 strip columns runs q steps
// behind quarter q - 1, and the loop is unrolled four steps at a time
#define PRALINE_QUAD_STEPS(max_l1) ((((max_l1) + 1) / 2 + 4 + 3) / 4 * 4 + 4)

#define PRALINE_MAX_RECTS 4   // zero rectangles per pair carried by the batched kernels
#define PRALINE_MW_LAG 2      // 12-row iterations between consecutive ranks of a shared task
