// dp_dense_instance.hip -- reference-order match scores in the split-strip layout: k_build_reft2 / k_match_tile
// (dp_reftile.hip.h) and the dense-tile instances (BSRC = 4) of k_dp_split16 and k_dp_split16_tb, which read a task's
// match scores from the tile k_match_tile wrote instead of computing them on the matrix pipe.
#include "dp_launch.hip.h"
#include "dp_split16.hip.h"
#include "dp_split16_tb.hip.h"
#include "dp_reftile.hip.h"

#include <algorithm>

int praline_launch_build_reft2(const float *raw, const float *S, int A, const int32_t *row_off_raw, const int32_t *len,
                               const int64_t *pr_off, int64_t PR, const unsigned char *nzidx, const unsigned char *nzcnt,
                               const int32_t *set_lo, int n_sets, int TB, float *T2, int n_seqs, hipStream_t stream)
{
    if (n_seqs <= 0) return PRALINE_OK;
    hipLaunchKernelGGL(k_build_reft2, dim3((unsigned)n_seqs), dim3(256), 0, stream, raw, S, A, row_off_raw, len, pr_off, PR, nzidx,
                       nzcnt, set_lo, n_sets, TB, T2);
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

// LDS of one k_match_tile workgroup of `waves` waves
static size_t match_tile_lds(int A, int TB, int waves) { return (size_t)A * (TB / 2) * 1024 + (size_t)waves * ((size_t)A * 64 + 128) + 256; }

bool praline_match_tile_supported(int A, int TB)
{
    return (TB == 4 || TB == 8) && A >= 1 && A <= 32 && match_tile_lds(A, TB, 8) <= (size_t)160 * 1024;
}

template <int TB, bool MULTI> static int launch_tile(RefTileArgs g, unsigned n_blocks, hipStream_t stream)
{
    // as many waves per workgroup as the LDS beside the table rows allows (one workgroup per CU: 16 = four per SIMD)
    int waves = MULTI ? 12 : 16;   // (the instances with per-set sums hold 32 more registers: launch bounds 768)
    while (waves > 4 && match_tile_lds(g.A, TB, waves) > (size_t)160 * 1024) --waves;
    const size_t lds = match_tile_lds(g.A, TB, waves);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_match_tile<TB, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRALINE_ERR_DEVICE;
        lds_set = lds;
    }
    g.waves = waves;
    hipLaunchKernelGGL((k_match_tile<TB, MULTI>), dim3(n_blocks), dim3(64 * waves), lds, stream, g);
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

int praline_launch_match_tile(const RefTileArgs &g, int TB, unsigned n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return PRALINE_OK;
    if (!praline_match_tile_supported(g.A, TB)) return PRALINE_ERR_UNSUPPORTED;
    const bool multi = g.n_sets > 1;
    if (TB == 4) return multi ? launch_tile<4, true>(g, n_blocks, stream) : launch_tile<4, false>(g, n_blocks, stream);
    return multi ? launch_tile<8, true>(g, n_blocks, stream) : launch_tile<8, false>(g, n_blocks, stream);
}

// scores only: one wave (= one task) per workgroup
int praline_launch_dense(const LaunchArgs &la, const Arena16Dev &a16, bool local)
{
    if (a16.dense == nullptr || a16.dense_off == nullptr) return PRALINE_ERR_ARG;
    const dim3 grid((unsigned)la.n_tasks), block(64);
    if (local)
        hipLaunchKernelGGL((k_dp_split16<1, 1, true, 4>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,
                           (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks);
    else
        hipLaunchKernelGGL((k_dp_split16<1, 1, false, 4>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,
                           (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks);
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

// fill with packed traceback (task mode): the tie flags compare the candidate sums (the NTERM = 3 flavour of the step)
int praline_launch_dense_tb(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask)
{
    if (a16.dense == nullptr || a16.dense_off == nullptr) return PRALINE_ERR_ARG;
    const dim3 grid((unsigned)la.n_tasks), block(64);
#define PRALINE_DENSE_TB(LOC, MSK)                                                                                       \
    hipLaunchKernelGGL((k_dp_split16_tb<1, 3, LOC, MSK, false, false, 4>), grid, block, 0, la.stream, a16, la.tasks,       \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks)
    if (local) { if (mask) PRALINE_DENSE_TB(true, true); else PRALINE_DENSE_TB(true, false); }
    else { if (mask) PRALINE_DENSE_TB(false, true); else PRALINE_DENSE_TB(false, false); }
#undef PRALINE_DENSE_TB
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
