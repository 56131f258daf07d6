// dp_rawb.h -- descriptors of a batch of RawPairwiseAligner requests (dp_rawb.hip.h holds the kernels, praline_rawb.hip.h the
// host side).  Every request brings its own match scores m [L1][L2], gap scores g1 [L1][2] and g2 [L2][2] and zero cells:
// nothing is shared between requests, so the lanes of a wave are the ROWS of one request (see k_rawb_fill).
// Everything the fill touches per 16-step chunk lies in the order the wave touches it - [strip][chunk][lane] blocks: m as four
// float4 per lane and chunk (the 16 columns lane l needs in chunk c of its strip: 16 c - l ...; k_rawb_stage writes them once
// from the caller's row-major m), the flag bytes (uint4 per lane and chunk), the zero-cell bits (16 per lane and chunk).  One
// wave-level load or store is then one contiguous, aligned block (read row-major, with every lane on its own row and cache
// line, the vector memory pipeline was 69 % busy and the kernel bound by it: 64 lines per instruction).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define PRALINE_RAWB_WAVES 8      // waves of a workgroup = row strips of one request in flight
#define PRALINE_RAWB_GROUPS 512   // workgroups of a fill launch (each takes every 512th request of the list)
#define PRALINE_RAWB_RING 256     // columns of a wave's hand-off ring (a power of two)
#define PRALINE_RAWB_ROW_PAD 160  // entries behind a request's g2 and boundary rows that the prefetches may read

struct RawReq {
    int32_t L1, L2, mode, nstrips;   // nstrips = ceil(L1 / 64)
    int32_t ncs, pad0;               // chunk blocks per strip in the m / flag / zero-bit arenas: ceil((L2 + 63) / 16) + 1 (the last one is only ever prefetched)
    int32_t index, pad1;             // the request's place in the caller's list
    int64_t m_off;                   // float4 elements: block (strip k, chunk c) = 4 x 64 float4 at m_off + (k * ncs + c) * 256, piece q of lane l at + q * 64 + l
    int64_t src_off;                 // floats: the request's m in the caller's list (k_rawb_stage)
    int64_t g1_off, g2_off;          // float2 elements
    int64_t t_off;                   // uint4 elements: flags of (strip k, chunk c), lane l at t_off + (k * ncs + c) * 64 + l; byte i = step 16 c + 1 + i
    int64_t z_off;                   // 16-bit words (even): zero-cell bits of (strip k, chunk c), lane l at z_off + (k * ncs + c) * 64 + l
    int64_t top_off;                 // float4 elements: the boundary row o[0][x], x = 0 .. L2 (+ PRALINE_RAWB_ROW_PAD); the same offset in `wrap`
    int64_t edge_off;                // float4 elements: o[L1][x], x = 0 .. L2, then o[y][L2], y = 0 .. L1
    int64_t best_off;                // float4 elements [min(nstrips, PRALINE_RAWB_WAVES) * 64]: every lane's first maximum (local mode)
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
void praline_launch_rawb_stage(const RawBatchDev &d, const float *src, float *dst, const int64_t *block0, int64_t n_blocks, hipStream_t st);
void praline_launch_rawb_zero(const RawBatchDev &d, const int32_t *zero_req, const int32_t *zero_idx, int64_t n_zero, hipStream_t st);
void praline_launch_rawb_fill(const RawBatchDev &d, int waves, bool mask, bool global_like, bool local, hipStream_t st);
void praline_launch_rawb_trace(const RawBatchDev &d, hipStream_t st);
