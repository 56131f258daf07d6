// dp_pipe_instance.hip -- k_dp_pipe instances (pipeline workgroups); compiled with -mllvm -amdgpu-mfma-vgpr-form.
#include "dp_launch.hip.h"
#include "dp_pipe.hip.h"

template <int NR, int NTERM> static void launch_pipe(const PipeLaunch &pl, const Arena16Dev &a16, bool local)
{
    const dim3 grid(pl.n_items), block(256);
    if (local)
        hipLaunchKernelGGL((k_dp_pipe<NR, NTERM, true>), grid, block, 0, pl.stream, a16, pl.items, pl.tasks, pl.set_one, pl.lane_pair,
                           (float2 *)pl.bnd, pl.scores, pl.rp);
    else
        hipLaunchKernelGGL((k_dp_pipe<NR, NTERM, false>), grid, block, 0, pl.stream, a16, pl.items, pl.tasks, pl.set_one, pl.lane_pair,
                           (float2 *)pl.bnd, pl.scores, pl.rp);
}

bool praline_pipe_supported(int nr, int nterm) { return nr == 2 && (nterm == 2 || nterm == 3); }

int praline_launch_pipe(const PipeLaunch &pl, const Arena16Dev &a16, int nr, int nterm, bool local)
{
    if (nr == 2 && nterm == 2) launch_pipe<2, 2>(pl, a16, local);
    else if (nr == 2 && nterm == 3) launch_pipe<2, 3>(pl, a16, local);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}
