// dp_dense2_instance.hip -- the dense-tile instances of k_dp_split16_tb with per-position gap scores (GapScoreModel arrays,
// praline_arena_set_gap_scores / praline_plan_run_gaps): with flags (alignments with paths, with or without zero rectangles /
// mask words) and without (scores only).
#include "dp_launch.hip.h"
#include "dp_split16.hip.h"
#include "dp_split16_tb.hip.h"

int praline_launch_dense_tb_ppg(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, bool noflags, int strip_lo,
                                int strip_cnt)
{
    if (a16.dense == nullptr || a16.dense_off == nullptr || la.rp.gaps == nullptr) return PRALINE_ERR_ARG;
    if (noflags && mask) return PRALINE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)la.n_tasks), block(64);
#define PRALINE_DENSE_TB(LOC, MSK, NOF)                                                                                  \
    hipLaunchKernelGGL((k_dp_split16_tb<1, 3, LOC, MSK, false, false, 4, true, NOF>), grid, block, 0, la.stream, a16, la.tasks, \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks, nullptr, 0, nullptr, 6, strip_lo, strip_cnt)
    if (noflags) { if (local) PRALINE_DENSE_TB(true, false, true); else PRALINE_DENSE_TB(false, false, true); }
    else if (local) { if (mask) PRALINE_DENSE_TB(true, true, false); else PRALINE_DENSE_TB(true, false, false); }
    else { if (mask) PRALINE_DENSE_TB(false, true, false); else PRALINE_DENSE_TB(false, false, false); }
#undef PRALINE_DENSE_TB
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
