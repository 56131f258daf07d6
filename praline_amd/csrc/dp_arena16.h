// dp_arena16.h -- device view of the f16 hi/lo operand arrays (see dp_split16.hip.h).
#pragma once
#include <stdint.h>
#include "dp_types.h"
struct Arena16Dev {
    const char *P16;        // [rows_pad][2][slots][16 bytes]
    const char *Q16;
    const int32_t *row_off;
    const int32_t *len;
    const unsigned char *sym8;  // [rows_pad + 64] active-symbol index of each (one-hot) profile row, 16 NR = none;
                                // NULL unless every profile of the arena is one-hot
    int stage;              // 1: use the LDS-staged operand stream (k_dp_split16 BSRC = 2)
    int row_bytes;          // 2 * half_bytes
    int half_bytes;         // 2 (pieces) * NR * 16
    // dense-tile instances (BSRC = 4, reference-order match scores, dp_reftile.hip.h): task t's tile at dense + dense_off[t]
    const float *dense = nullptr;
    const int64_t *dense_off = nullptr;
};
#ifndef PRALINE_DENSE_PAD
#define PRALINE_DENSE_PAD 24   // rows per strip of a dense tile beyond max_l1 (row 0 + the DP kernels' look-ahead)
#endif

