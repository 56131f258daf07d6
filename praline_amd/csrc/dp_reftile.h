// dp_reftile.h -- plain descriptors of the reference-order match-score tiles (kernels: dp_reftile.hip.h).
#pragma once
#include <stdint.h>
#include "dp_types.h"
#include "dp_arena16.h"   // PRALINE_DENSE_PAD

#define PRALINE_REFTILE_ROWS 16     // rows y per work item of k_match_tile
#define PRALINE_REFTILE_COLS 128    // columns x per workgroup

struct RefTileBlock { int32_t task, chunk; };   // workgroup -> (task of the launch, 128-column chunk of its sequence two)

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
    int waves;                      // waves per workgroup (blockDim.x / 64)
};

