// dp_reftile.h -- plain descriptors of the reference-order match-score tiles (kernels: dp_reftile.hip.h).
#pragma once
#include <stdint.h>
#include "dp_types.h"
#include "dp_arena16.h"   // PRALINE_DENSE_PAD

#define PRALINE_REFTILE_ROWS 16     // rows y per work item of k_match_tile
#define PRALINE_REFTILE_COLS 128    // columns x per workgroup

// Workgroup -> (group of tasks that share their 32 sequences one, 128-column chunk of the group's sequences two laid end
// to end, each padded to whole 32-column strips).  The group's record in RefTileArgs::grp at `base`: count + 1 cumulative
// pair-row counts (16 per strip), then the count task
// indices (relative to RefTileArgs::tasks).
struct RefTileBlock { int32_t base, count, chunk, pad; };

struct RefTileArgs {
    const float *raw;               // raw profiles [rows_raw][A]
    int A;
    const float *T2;                // [A][PR][TB][2]
    int64_t PR;                     // pair rows of the whole arena
    const int32_t *row_off_raw, *len;
    const int64_t *pr_off;          // first pair row of each sequence
    const int32_t *set_lo;          // track-set boundaries on the symbol axis (n_sets + 1 entries)
    int n_sets;
    const WaveTask *tasks;          // split-layout tasks of this launch
    const int32_t *lane_one;        // [task][32]
    const int64_t *dense_off;       // float offset of each task's tile in m
    float *m;
    const RefTileBlock *blocks;
    const int32_t *grp;             // group records (see RefTileBlock)
    int waves;                      // waves per workgroup (blockDim.x / 64)
};

