// dp_pk16_instance.hip -- k_dp_pk16_tb instances (dp_pk16.hip.h): fill with packed traceback for plain sequences under integer
// scoring, two pairs per lane in packed int16.
#include "dp_launch.hip.h"
#include "dp_pk16.hip.h"

template <int NR> static void launch_pk16(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, float scale)
{
    const dim3 grid((la.n_tasks + 3) / 4), block(256);
#define PRALINE_PK16(LOC, MSK)                                                                                           \
    hipLaunchKernelGGL((k_dp_pk16_tb<NR, LOC, MSK>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,  \
                       (uint4 *)la.bnd, (uint4 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, (int)la.n_tasks, scale)
    if (local) { if (mask) PRALINE_PK16(true, true); else PRALINE_PK16(true, false); }
    else { if (mask) PRALINE_PK16(false, true); else PRALINE_PK16(false, false); }
#undef PRALINE_PK16
}

// chain mode: one wave per (task, strip), grid strip-major (see k_dp_split16_tb's)
template <int NR> static void launch_pk16_chain(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, float scale, int max_strips,
                                                int *flags, void *cand, int every)
{
    const dim3 grid((unsigned)la.n_tasks * (unsigned)max_strips), block(64);
    const int stride = max_strips + 1;
#define PRALINE_PK16C(LOC, MSK)                                                                                          \
    hipLaunchKernelGGL((k_dp_pk16_tb<NR, LOC, MSK, true>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair, \
                       (uint4 *)la.bnd, (uint4 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, (int)la.n_tasks, scale, flags, \
                       stride, (float4 *)cand, every)
    if (local) { if (mask) PRALINE_PK16C(true, true); else PRALINE_PK16C(true, false); }
    else { if (mask) PRALINE_PK16C(false, true); else PRALINE_PK16C(false, false); }
#undef PRALINE_PK16C
}

int praline_launch_pk16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, bool mask, float scale, int max_strips,
                                 int *flags, void *cand, int every)
{
    if (a16.sym8 == nullptr || max_strips < 1) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1) launch_pk16_chain<1>(la, a16, local, mask, scale, max_strips, flags, cand, every);
    else if (nr == 2) launch_pk16_chain<2>(la, a16, local, mask, scale, max_strips, flags, cand, every);
    else return PRALINE_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}

// nr: 16-wide symbol ranges of the arena (1 or 2); scale: 2^k, the DP runs on value * scale (integers below 32 000 in
// magnitude: checked by the caller)
int praline_launch_pk16_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, bool mask, float scale)
{
    if (a16.sym8 == nullptr) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1) launch_pk16<1>(la, a16, local, mask, scale);
    else if (nr == 2) launch_pk16<2>(la, a16, local, mask, scale);
    else return PRALINE_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
