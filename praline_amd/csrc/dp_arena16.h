// dp_arena16.h -- device view of the f16 hi/lo operand arrays (see dp_split16.hip.h).
#pragma once
#include <stdint.h>
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
};

// Workgroup descriptor of the four-wave launch of k_dp_split16 (small batches): `share` consecutive waves
// pipeline one task (share = 1, 2 or 4; the task id sits in the slot of the group's first wave, -1 = idle
// waves); `barriers` = s_barriers every wave of the workgroup executes (0 when share == 1).
struct WgDesc {
    int32_t task[4];
    int32_t share;
    int32_t barriers;
    int32_t pad[2];
};
