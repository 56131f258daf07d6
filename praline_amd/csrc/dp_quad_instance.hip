// dp_quad_instance.hip -- k_dp_quad_tb instances (dp_quad.hip.h): fill with packed traceback for plain sequences, 16 pairs
// per wave.
#include "dp_launch.hip.h"
#include "dp_quad.hip.h"

template <int NR, bool INTS> static void launch_quad(const LaunchArgs &la, const Arena16Dev &a16, bool local, int mask)
{
    const dim3 grid((la.n_tasks + 3) / 4), block(256);
#define PRALINE_QUAD(LOC, MSK)                                                                                           \
    hipLaunchKernelGGL((k_dp_quad_tb<NR, INTS, LOC, MSK>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, la.lane_pair, \
                       (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, (int)la.n_tasks)
    if (local) { if (mask == 2) PRALINE_QUAD(true, 2); else if (mask) PRALINE_QUAD(true, 1); else PRALINE_QUAD(true, 0); }
    else { if (mask == 2) PRALINE_QUAD(false, 2); else if (mask) PRALINE_QUAD(false, 1); else PRALINE_QUAD(false, 0); }
#undef PRALINE_QUAD
}

// nr: 16-wide symbol ranges of the arena (1 or 2); ints: integer scoring (tie flags from the predecessor states)
int praline_launch_quad_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, bool ints, bool local, int mask)
{
    if (a16.sym8 == nullptr) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1) { if (ints) launch_quad<1, true>(la, a16, local, mask); else launch_quad<1, false>(la, a16, local, mask); }
    else if (nr == 2) { if (ints) launch_quad<2, true>(la, a16, local, mask); else launch_quad<2, false>(la, a16, local, mask); }
    else return PRALINE_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
