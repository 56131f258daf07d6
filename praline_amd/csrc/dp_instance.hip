// dp_instance.hip -- one instantiation set of k_dp_batch; compiled with -DPRALINE_NSTEP_INST=N.
#include "dp_launch.hip.h"

#ifndef PRALINE_NSTEP_INST
#error "compile with -DPRALINE_NSTEP_INST=<MFMA steps per tile>"
#endif
#define PRALINE_CAT2(a, b) a##b
#define PRALINE_CAT(a, b) PRALINE_CAT2(a, b)

int PRALINE_CAT(praline_launch_dp_, PRALINE_NSTEP_INST)(const LaunchArgs &la, int tp, bool local, int out, int mask)
{
    return launch_nstep<PRALINE_NSTEP_INST>(la, tp, local, out, mask);
}

#ifdef PRALINE_EXP_BATCH_MASK2
// experiment builds: read and clear the debug words of this translation unit's k_dp_batch instances
extern "C" int PRALINE_CAT(praline_debug_read_, PRALINE_NSTEP_INST)(unsigned *out)
{
    unsigned zero[128] = {0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(praline_dbg), sizeof(zero)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(praline_dbg), zero, sizeof(zero)) != hipSuccess) return -1;
    return 0;
}
#endif
