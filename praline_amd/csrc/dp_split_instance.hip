// dp_split_instance.hip -- k_dp_split for one MFMA step count; compiled with
// -DPRALINE_NSTEP_INST=N -mllvm -amdgpu-mfma-vgpr-form (MFMA results straight into VGPRs: the
// recurrence reads them without v_accvgpr_read copies).
#include "dp_launch.hip.h"
#include "dp_split.hip.h"
#include <cstdlib>

#ifndef PRALINE_NSTEP_INST
#error "compile with -DPRALINE_NSTEP_INST=<MFMA steps per tile>"
#endif
#define PRALINE_CAT2(a, b) a##b
#define PRALINE_CAT(a, b) PRALINE_CAT2(a, b)

int PRALINE_CAT(praline_launch_split_, PRALINE_NSTEP_INST)(const LaunchArgs &la, bool local)
{
    unsigned wpb = 4;  // wavefronts (= tasks) per workgroup
    if (const char *env = getenv("PRALINE_WPB")) { const int v = atoi(env); if (v >= 1 && v <= 4) wpb = (unsigned)v; }
#ifdef PRALINE_EXPERIMENT
    if (const char *env = getenv("PRALINE_EXP")) {
        const int e = atoi(env);
#define PRALINE_EXP_CASE(E) case E: hipLaunchKernelGGL((k_dp_split<PRALINE_NSTEP_INST, false, E>), dim3((la.n_tasks + wpb - 1) / wpb), dim3(64 * wpb), 0, la.stream, la.ar, la.tasks, la.lane_one, la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks); return PRALINE_OK;
        switch (e) { PRALINE_EXP_CASE(1) PRALINE_EXP_CASE(2) PRALINE_EXP_CASE(3) PRALINE_EXP_CASE(4) PRALINE_EXP_CASE(8) PRALINE_EXP_CASE(7) PRALINE_EXP_CASE(11) PRALINE_EXP_CASE(12) PRALINE_EXP_CASE(16) PRALINE_EXP_CASE(19) }
    }
#endif
    if (local)
        hipLaunchKernelGGL((k_dp_split<PRALINE_NSTEP_INST, true>), dim3((la.n_tasks + wpb - 1) / wpb), dim3(64 * wpb), 0, la.stream, la.ar,
                           la.tasks, la.lane_one, la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks);
    else
        hipLaunchKernelGGL((k_dp_split<PRALINE_NSTEP_INST, false>), dim3((la.n_tasks + wpb - 1) / wpb), dim3(64 * wpb), 0, la.stream, la.ar,
                           la.tasks, la.lane_one, la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks);
    return PRALINE_OK;
}
