// dp_dense_instance.hip -- reference-order match scores in the split-strip layout: k_build_reft2 / k_match_tile
// (dp_reftile.hip.h) and the dense-tile instances (BSRC = 4) of k_dp_split16 and k_dp_split16_tb, which read a task's
// match scores from the tile k_match_tile wrote instead of computing them on the matrix pipe.
#include "dp_launch.hip.h"
#include "dp_split16.hip.h"
#include "dp_split16_tb.hip.h"
#include "dp_reftile.hip.h"

#include <algorithm>
#include <cstdlib>

int praline_launch_build_reft2(const float *raw, const float *S, int A, const int32_t *row_off_raw, const int32_t *len,
                               const int64_t *pr_off, int64_t PR, const unsigned char *nzidx, const unsigned char *nzcnt,
                               const int32_t *set_lo, int n_sets, int TB, float *T2, int n_seqs, hipStream_t stream)
{
    if (n_seqs <= 0) return PRALINE_OK;
    hipLaunchKernelGGL(k_build_reft2, dim3((unsigned)n_seqs), dim3(256), 0, stream, raw, S, A, row_off_raw, len, pr_off, PR, nzidx,
                       nzcnt, set_lo, n_sets, TB, T2);
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

// LDS of one k_match_tile workgroup of `waves` waves with G rows per work item
static size_t match_tile_lds(int A, int TB, int waves, int G) { return (size_t)A * (TB / 2) * 1024 + (size_t)waves * ((size_t)A * G * 4 + 128) + 256 + 512; }
static const size_t kLdsPerCu = (size_t)160 * 1024;

bool praline_match_tile_supported(int A, int TB)
{
    return (TB == 4 || TB == 8) && A >= 1 && A <= 32 && match_tile_lds(A, TB, 8, 16) <= kLdsPerCu;
}

template <int TB, bool MULTI, int G> static int launch_tile(RefTileArgs g, unsigned n_blocks, int waves, hipStream_t stream)
{
    const size_t lds = match_tile_lds(g.A, TB, waves, G);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_match_tile<TB, MULTI, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRALINE_ERR_DEVICE;
        lds_set = lds;
    }
    g.waves = waves;
    hipLaunchKernelGGL((k_match_tile<TB, MULTI, G>), dim3(n_blocks), dim3(64 * waves), lds, stream, g);
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

template <int TB, bool MULTI> static int launch_tile_g(const RefTileArgs &g, unsigned n_blocks, hipStream_t stream)
{
    // 16 rows per work item and as many waves as the LDS beside the table rows allows (one workgroup per CU: 16 waves =
    // four per SIMD).  Measured on C2 (k_match_tile alone): 16 rows x 16 waves 12.8 ms, 16 x 12 13.8, 32 rows x 12 waves
    // 13.7, 32 x 8 16.4 - the kernel wants waves, not fewer table-row reads.  PRALINE_REFTILE_G=32 selects the 32-row
    // instances (12 waves, no per-set sums).
    int G = 16, waves = 12;
    if (const char *env = getenv("PRALINE_REFTILE_G")) { if (atoi(env) == 32 && !MULTI) G = 32; }
    if (G == 32 && match_tile_lds(g.A, TB, waves, 32) > kLdsPerCu) G = 16;
    if (G == 16) {
        waves = MULTI ? 12 : 16;
        while (waves > 4 && match_tile_lds(g.A, TB, waves, 16) > kLdsPerCu) --waves;
    }
    if (const char *env = getenv("PRALINE_REFTILE_WAVES")) { const int w = atoi(env); if (w >= 1 && w <= waves) waves = w; }
    if constexpr (!MULTI) {
        if (G == 32) return launch_tile<TB, MULTI, 32>(g, n_blocks, waves, stream);
    }
    return launch_tile<TB, MULTI, 16>(g, n_blocks, waves, stream);
}

int praline_launch_match_tile(const RefTileArgs &g, int TB, unsigned n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return PRALINE_OK;
    if (!praline_match_tile_supported(g.A, TB)) return PRALINE_ERR_UNSUPPORTED;
    const bool multi = g.n_sets > 1;
    if (TB == 4) return multi ? launch_tile_g<4, true>(g, n_blocks, stream) : launch_tile_g<4, false>(g, n_blocks, stream);
    return multi ? launch_tile_g<8, true>(g, n_blocks, stream) : launch_tile_g<8, false>(g, n_blocks, stream);
}

// scores only: one wave (= one task) per workgroup
int praline_launch_dense(const LaunchArgs &la, const Arena16Dev &a16, bool local)
{
    if (a16.dense == nullptr || a16.dense_off == nullptr) return PRALINE_ERR_ARG;
    if (la.wg != nullptr) {   // small batch: four-wave workgroups whose waves share tasks (more tile rows in flight per task)
        const dim3 g4(la.n_wg), b4(256);
        if (local)
            hipLaunchKernelGGL((k_dp_split16<1, 1, true, 4, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,
                               (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
        else
            hipLaunchKernelGGL((k_dp_split16<1, 1, false, 4, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,
                               (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
        return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
    }
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
int praline_launch_dense_tb(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, bool ppg, bool noflags, int strip_lo,
                            int strip_cnt)
{
    if (a16.dense == nullptr || a16.dense_off == nullptr) return PRALINE_ERR_ARG;
    if (ppg) return praline_launch_dense_tb_ppg(la, a16, local, mask, noflags, strip_lo, strip_cnt);
    if (noflags && mask) return PRALINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)la.n_tasks), block(64);
#define PRALINE_DENSE_TB(LOC, MSK, NOF)                                                                                  \
    hipLaunchKernelGGL((k_dp_split16_tb<1, 3, LOC, MSK, false, false, 4, false, NOF>), grid, block, 0, la.stream, a16, la.tasks, \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks, nullptr, 0, nullptr, 6, strip_lo, strip_cnt)
    if (noflags) { if (local) PRALINE_DENSE_TB(true, false, true); else PRALINE_DENSE_TB(false, false, true); }
    else if (local) { if (mask) PRALINE_DENSE_TB(true, true, false); else PRALINE_DENSE_TB(true, false, false); }
    else { if (mask) PRALINE_DENSE_TB(false, true, false); else PRALINE_DENSE_TB(false, false, false); }
#undef PRALINE_DENSE_TB
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
