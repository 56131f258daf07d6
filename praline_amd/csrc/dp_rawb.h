// dp_rawb.h -- descriptors of a batch of RawPairwiseAligner requests (dp_rawb.hip.h holds the kernels, praline_rawb.hip.h the
// host side).  Every request brings its own match scores m [L1][L2], gap scores g1 [L1][2] and g2 [L2][2] and zero cells:
// nothing is shared between requests, so the lanes of a wave are the ROWS of one request (see k_rawb_fill).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define PRALINE_RAWB_WAVES 8      // waves of a workgroup = row strips of one request in flight
#define PRALINE_RAWB_GROUPS 512   // workgroups of a fill launch (each takes every 512th request of the list)
#define PRALINE_RAWB_RING 256     // columns of a wave's hand-off ring (a power of two)
#define PRALINE_RAWB_ROW_PAD 160  // entries behind a request's g2 and boundary rows that the prefetches may read
#define PRALINE_RAWB_M_PAD 128    // floats in front of and behind every request's m (the skewed 16-float loads of the first and last rows)

struct RawReq {
    int32_t L1, L2, mode, nstrips;   // nstrips = ceil(L1 / 64)
    int32_t ts, zs;                  // bytes per row of flags (a multiple of 16); 16-bit words per row of the zero mask (= ts / 16)
    int32_t index, pad;              // the request's place in the caller's list
    int64_t m_off;                   // floats: m[0][0]
    int64_t g1_off, g2_off;          // float2 elements
    int64_t t_off;                   // bytes (a multiple of 16): flags, rows 0 .. 64 * nstrips
    int64_t z_off;                   // 16-bit words (even): zero mask, rows 0 .. L1
    int64_t top_off;                 // float4 elements: the boundary row o[0][x], x = 0 .. L2 (+ PRALINE_RAWB_ROW_PAD); the same offset in `wrap`
    int64_t edge_off;                // float4 elements: o[L1][x], x = 0 .. L2, then o[y][L2], y = 0 .. L1
    int64_t best_off;                // float4 elements [PRALINE_RAWB_WAVES * 64]: every lane's first maximum (local mode)
    int64_t path_off;                // path rows (int32 pairs): a slot of L1 + L2 + 2 rows
};

struct RawBatchDev {
    const RawReq *reqs;              // in launch order (largest first)
    int n;
    const float *m;
    const float2 *g1, *g2;
    uint16_t *z;
    float4 *top, *wrap, *edge, *best;
    uint8_t *t;
    int32_t *paths;
    int64_t *path_info;              // [n][2] by request index: first row of the path in its slot, rows
    float *scores;                   // [n] by request index
    int32_t *error;                  // set when a wave gave up waiting for its neighbour (never, unless the kernel is broken)
};

void praline_launch_rawb_init(const RawBatchDev &d, hipStream_t st);
void praline_launch_rawb_zero(const RawBatchDev &d, const int32_t *zero_req, const int32_t *zero_idx, int64_t n_zero, hipStream_t st);
void praline_launch_rawb_fill(const RawBatchDev &d, int waves, bool mask, bool global_like, bool local, hipStream_t st);
void praline_launch_rawb_trace(const RawBatchDev &d, hipStream_t st);
