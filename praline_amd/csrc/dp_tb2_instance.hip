// dp_tb2_instance.hip -- the two-pass alignments-with-paths kernels: k_dp_split16_tb<..., TWOPASS> (flag-free forward
// fill with kept boundary columns and row checkpoints) and k_trace_recompute (dp_trace2.hip.h); compiled with
// -mllvm -amdgpu-mfma-vgpr-form like the other split16 kernels.
#include "dp_launch.hip.h"
#include "dp_trace2.hip.h"

template <int NR, int NTERM> static void launch_fwd(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask)
{
    const dim3 grid(la.n_tasks), block(64);
#define PRALINE_FWD(LOC, MSK)                                                                                            \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK, false, true>), grid, block, 0, la.stream, a16, la.tasks,      \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks)
    if constexpr (NTERM == 1) {
        if (a16.sym8 != nullptr) {   // one-hot arena: operand rows from the one-hot table in LDS
#define PRALINE_FWD_OH(LOC, MSK)                                                                                         \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK, false, true, 1>), grid, block, 0, la.stream, a16, la.tasks,   \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks)
            if (local) { if (mask) PRALINE_FWD_OH(true, true); else PRALINE_FWD_OH(true, false); }
            else { if (mask) PRALINE_FWD_OH(false, true); else PRALINE_FWD_OH(false, false); }
#undef PRALINE_FWD_OH
            return;
        }
    }
    if (local) { if (mask) PRALINE_FWD(true, true); else PRALINE_FWD(true, false); }
    else { if (mask) PRALINE_FWD(false, true); else PRALINE_FWD(false, false); }
#undef PRALINE_FWD
}

template <int NR, int NTERM> static void launch_bwd(const LaunchArgs &la, const Arena16Dev &a16, const Trace2Args &ta, bool local,
                                                    bool mask, int keep_in_aux, const void *analytic4)
{
    const dim3 grid(la.n_tasks), block(64);
    if (keep_in_aux == 2) {
        // behind the pipeline forward fill (global mode, no rectangles): blocks of PRALINE_KEEP_BH rows
        if constexpr (NR == 2 && NTERM != 1)
            hipLaunchKernelGGL((k_trace_recompute<NR, NTERM, false, false, PRALINE_KEEP_BH>), grid, block, 0, la.stream, a16, la.tasks,
                               la.lane_one, la.lane_pair, (const float4 *)la.bnd, (const float *)la.tb, la.rl, la.end_cells,
                               ta.slot_off, ta.paths, ta.path_start, ta.path_rows, la.rp, (int)la.n_tasks, 1,
                               (const float4 *)analytic4, 1);
        return;
    }
#define PRALINE_BWD(LOC, MSK)                                                                                            \
    hipLaunchKernelGGL((k_trace_recompute<NR, NTERM, LOC, MSK>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one,    \
                       la.lane_pair, (const float4 *)la.bnd, (const float *)la.tb, la.rl, la.end_cells, ta.slot_off,       \
                       ta.paths, ta.path_start, ta.path_rows, la.rp, (int)la.n_tasks, keep_in_aux)
    if (local) { if (mask) PRALINE_BWD(true, true); else PRALINE_BWD(true, false); }
    else { if (mask) PRALINE_BWD(false, true); else PRALINE_BWD(false, false); }
#undef PRALINE_BWD
}

int praline_launch_tb2_forward(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask)
{
    if (nr == 1 && nterm == 1) launch_fwd<1, 1>(la, a16, local, mask);
    else if (nr == 1 && nterm == 3) launch_fwd<1, 3>(la, a16, local, mask);
    else if (nr == 2 && nterm == 1) launch_fwd<2, 1>(la, a16, local, mask);
    else if (nr == 2 && nterm == 3) launch_fwd<2, 3>(la, a16, local, mask);
    else if (nr == 2 && nterm == 2) launch_fwd<2, 2>(la, a16, local, mask);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

int praline_launch_tb2_backward(const LaunchArgs &la, const Arena16Dev &a16, const Trace2Args &ta, int nr, int nterm, bool local,
                                bool mask, int keep_in_aux, const void *analytic4)
{
    if (keep_in_aux == 2 && (nr != 2 || nterm == 1 || local || mask || analytic4 == nullptr)) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1 && nterm == 1) launch_bwd<1, 1>(la, a16, ta, local, mask, keep_in_aux, analytic4);
    else if (nr == 1 && nterm == 3) launch_bwd<1, 3>(la, a16, ta, local, mask, keep_in_aux, analytic4);
    else if (nr == 2 && nterm == 1) launch_bwd<2, 1>(la, a16, ta, local, mask, keep_in_aux, analytic4);
    else if (nr == 2 && nterm == 3) launch_bwd<2, 3>(la, a16, ta, local, mask, keep_in_aux, analytic4);
    else if (nr == 2 && nterm == 2) launch_bwd<2, 2>(la, a16, ta, local, mask, keep_in_aux, analytic4);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

template <int NR, int NTERM> static void launch_keep(const LaunchArgs &la, const Arena16Dev &a16, void *keep_bnd, float *ckpt)
{
    hipLaunchKernelGGL((k_dp_split16<NR, NTERM, false, 2, 4, true>), dim3(la.n_wg), dim3(256), 0, la.stream, a16, la.tasks,
                       la.lane_one, la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg, (float4 *)keep_bnd,
                       ckpt, la.end_cells);
}

int praline_launch_keep_forward(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, void *keep_bnd, float *ckpt)
{
    if (!a16.stage || la.wg == nullptr) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 1 && nterm == 3) launch_keep<1, 3>(la, a16, keep_bnd, ckpt);
    else if (nr == 2 && nterm == 3) launch_keep<2, 3>(la, a16, keep_bnd, ckpt);
    else if (nr == 2 && nterm == 2) launch_keep<2, 2>(la, a16, keep_bnd, ckpt);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

// chain mode without flags (k_dp_split16_tb<..., CHAIN, TWOPASS>): the scores-only fill of plans of a few long sequences
template <int NR, int NTERM> static void launch_scores_chain(const LaunchArgs &la, const Arena16Dev &a16, bool local, int max_strips,
                                                             int *flags, void *cand, int every)
{
    const dim3 grid(la.n_tasks * (unsigned)max_strips), block(64);
    if (local)
        hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, true, false, true, true>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one,
                           la.lane_pair, (float4 *)la.bnd, (uint2 *)nullptr, la.aux, la.rl, la.scores, la.end_cells, la.rp,
                           (int)la.n_tasks, flags, max_strips + 1, (float4 *)cand, every);
    else
        hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, false, false, true, true>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one,
                           la.lane_pair, (float4 *)la.bnd, (uint2 *)nullptr, la.aux, la.rl, la.scores, la.end_cells, la.rp,
                           (int)la.n_tasks, flags, max_strips + 1, (float4 *)cand, every);
}

int praline_launch_scores_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, int max_strips, int *flags,
                                void *cand, int every)
{
    if (nr == 1 && nterm == 1) launch_scores_chain<1, 1>(la, a16, local, max_strips, flags, cand, every);
    else if (nr == 1 && nterm == 3) launch_scores_chain<1, 3>(la, a16, local, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 1) launch_scores_chain<2, 1>(la, a16, local, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 3) launch_scores_chain<2, 3>(la, a16, local, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 2) launch_scores_chain<2, 2>(la, a16, local, max_strips, flags, cand, every);
    else return PRALINE_ERR_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? PRALINE_OK : PRALINE_ERR_DEVICE;
}
