// dp_ref_instance.hip -- k_dp_batch reading dense match scores (MSRC = 1): the reference-order audit mode
// PRALINE_MATCH_REFERENCE, and plans that run with per-position gap scores (PPG).  NSTEP is irrelevant here (no MFMAs are
// issued); TP = 1.
#include "dp_launch.hip.h"

template <bool LOCAL, int OUT, int MASK, bool PPG> static void launch_ref(const LaunchArgs &la)
{
    hipLaunchKernelGGL((k_dp_batch<2, 1, LOCAL, OUT, MASK, 1, PPG>), dim3(la.n_tasks), dim3(64), 0, la.stream, la.ar, la.tasks,
                       la.lane_one, la.lane_pair, la.bnd, la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp, la.mref,
                       la.m_off);
}

template <bool PPG> static int launch_ref_any(const LaunchArgs &la, bool local, int out, int mask)
{
    if (out == 0) {
        if (mask) return PRALINE_ERR_UNSUPPORTED;
        if (local) launch_ref<true, 0, 0, PPG>(la); else launch_ref<false, 0, 0, PPG>(la);
    } else if (local) {
        if (mask == 2) launch_ref<true, 1, 2, PPG>(la);
        else if (mask) launch_ref<true, 1, 1, PPG>(la);
        else launch_ref<true, 1, 0, PPG>(la);
    } else {
        if (mask == 2) launch_ref<false, 1, 2, PPG>(la);
        else if (mask) launch_ref<false, 1, 1, PPG>(la);
        else launch_ref<false, 1, 0, PPG>(la);
    }
    return PRALINE_OK;
}

int praline_launch_dp_ref(const LaunchArgs &la, bool local, int out, int mask)
{
    if (la.split || la.mref == nullptr || la.m_off == nullptr) return PRALINE_ERR_ARG;
    return la.rp.gaps != nullptr ? launch_ref_any<true>(la, local, out, mask) : launch_ref_any<false>(la, local, out, mask);
}

