// dp_pk16_instance.hip -- k_dp_pk16_tb instances (dp_pk16.hip.h): fill with packed traceback for plain sequences under integer
// scoring, two pairs per lane in packed int16.
#include "dp_launch.hip.h"
#include "dp_pk16.hip.h"

// mask: rectangle slots per pair the kernel holds (0: none; rounded up to 1, 2 or PRALINE_MAX_RECTS)
template <int NR> static void launch_pk16(const LaunchArgs &la, const Arena16Dev &a16, bool local, int mask, float scale)
{
    const dim3 grid((la.n_tasks + 3) / 4), block(256);
#define PRALINE_PK16(LOC, MSK)                                                                                           \
    hipLaunchKernelGGL((k_dp_pk16_tb<NR, LOC, MSK>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair,  \
                       (uint4 *)la.bnd, (uint4 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, (int)la.n_tasks, scale)
#define PRALINE_PK16M(LOC)                                                                                               \
    { if (mask <= 0) PRALINE_PK16(LOC, 0); else if (mask == 1) PRALINE_PK16(LOC, 1); else if (mask == 2) PRALINE_PK16(LOC, 2);   \
      else PRALINE_PK16(LOC, PRALINE_MAX_RECTS); }
    if (local) PRALINE_PK16M(true) else PRALINE_PK16M(false)
#undef PRALINE_PK16M
#undef PRALINE_PK16
}

// chain mode: one wave per (task, strip), grid strip-major (see k_dp_split16_tb's)
template <int NR> static void launch_pk16_chain(const LaunchArgs &la, const Arena16Dev &a16, bool local, int mask, float scale, int max_strips,
                                                int *flags, void *cand, int every)
{
    const dim3 grid((unsigned)la.n_tasks * (unsigned)max_strips), block(64);
    const int stride = max_strips + 1;
#define PRALINE_PK16C(LOC, MSK)                                                                                          \
    hipLaunchKernelGGL((k_dp_pk16_tb<NR, LOC, MSK, true>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair, \
                       (uint4 *)la.bnd, (uint4 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, (int)la.n_tasks, scale, flags, \
                       stride, (float4 *)cand, every)
#define PRALINE_PK16CM(LOC)                                                                                              \
    { if (mask <= 0) PRALINE_PK16C(LOC, 0); else if (mask == 1) PRALINE_PK16C(LOC, 1); else if (mask == 2) PRALINE_PK16C(LOC, 2); \
      else PRALINE_PK16C(LOC, PRALINE_MAX_RECTS); }
    if (local) PRALINE_PK16CM(true) else PRALINE_PK16CM(false)
#undef PRALINE_PK16CM
#undef PRALINE_PK16C
}

int praline_launch_pk16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, int mask, float scale, int max_strips,
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
int praline_launch_pk16_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool local, int mask, float scale)
{
    if (a16.sym8 == nullptr) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1) launch_pk16<1>(la, a16, local, mask, scale);
    else if (nr == 2) launch_pk16<2>(la, a16, local, mask, scale);
    else return PRALINE_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
