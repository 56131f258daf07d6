// dp_pipe_instance.hip -- k_dp_pipe instances (pipeline workgroups); compiled with -mllvm -amdgpu-mfma-vgpr-form.
#include "dp_launch.hip.h"
#include "dp_pipe.hip.h"

template <int NR, int NTERM> static void launch_pipe(const PipeLaunch &pl, const Arena16Dev &a16)
{
    const dim3 grid(pl.n_items), block(256);
    // the analytic column 0 (gap scores and mode of this run) that every task's first strip reads
    if (!pl.analytic_valid)
        hipLaunchKernelGGL(k_pipe_analytic, dim3((unsigned)((pl.analytic_rows * 32 + 255) / 256)), dim3(256), 0, pl.stream,
                           (float2 *)pl.analytic, pl.analytic_rows, pl.rp);
#define PRALINE_PIPE_LAUNCH(LOC, SEMI)                                                                                  \
    hipLaunchKernelGGL((k_dp_pipe<NR, NTERM, LOC, SEMI>), grid, block, 0, pl.stream, a16, pl.items, pl.tasks, pl.set_one, \
                       pl.lane_pair, (float2 *)pl.bnd, (const float2 *)pl.analytic, pl.scores, pl.rp)
    if (pl.rp.mode == PRALINE_MODE_LOCAL) PRALINE_PIPE_LAUNCH(true, false);
    else if (pl.rp.mode >= 2) PRALINE_PIPE_LAUNCH(false, true);
    else PRALINE_PIPE_LAUNCH(false, false);
#undef PRALINE_PIPE_LAUNCH
}

// registers / LDS of the instance a run launches (hipFuncGetAttributes): bench.py reports them beside the roofline
template <int NR, int NTERM> static int pipe_attrs(int mode, int *vgprs, int *lds_bytes)
{
    hipFuncAttributes fa;
    hipError_t e;
    if (mode == PRALINE_MODE_LOCAL) e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_dp_pipe<NR, NTERM, true, false>));
    else if (mode >= 2) e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_dp_pipe<NR, NTERM, false, true>));
    else e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_dp_pipe<NR, NTERM, false, false>));
    if (e != hipSuccess) return PRALINE_ERR_DEVICE;
    *vgprs = fa.numRegs;
    *lds_bytes = (int)fa.sharedSizeBytes;
    return PRALINE_OK;
}

int praline_pipe_attrs(int nr, int nterm, int mode, int *vgprs, int *lds_bytes)
{
    if (nr == 2 && nterm == 2) return pipe_attrs<2, 2>(mode, vgprs, lds_bytes);
    if (nr == 2 && nterm == 3) return pipe_attrs<2, 3>(mode, vgprs, lds_bytes);
    return PRALINE_ERR_UNSUPPORTED;
}

// the KEEP forward fill of the two-pass alignments with paths (global mode): kept columns at tk.aux_off of keep_bnd, row
// checkpoints at tk.tb_off of ckpt, end cells and scores per pair; analytic4: the float4 column 0 k_trace_recompute reads
template <int NR, int NTERM> static void launch_pipe_keep(const PipeLaunch &pl, const Arena16Dev &a16, void *keep_bnd, float *ckpt,
                                                          int32_t *end_cells, void *analytic4)
{
    const dim3 grid(pl.n_items), block(256);
    if (!pl.analytic_valid) {
        hipLaunchKernelGGL(k_pipe_analytic, dim3((unsigned)((pl.analytic_rows * 32 + 255) / 256)), dim3(256), 0, pl.stream,
                           (float2 *)pl.analytic, pl.analytic_rows, pl.rp);
        hipLaunchKernelGGL(k_pipe_analytic4, dim3((unsigned)((pl.analytic_rows * 32 + 255) / 256)), dim3(256), 0, pl.stream,
                           (float4 *)analytic4, pl.analytic_rows, pl.rp);
    }
    hipLaunchKernelGGL((k_dp_pipe<NR, NTERM, false, false, true>), grid, block, 0, pl.stream, a16, pl.items, pl.tasks, pl.set_one,
                       pl.lane_pair, (float2 *)pl.bnd, (const float2 *)pl.analytic, pl.scores, pl.rp, (float4 *)keep_bnd, ckpt,
                       end_cells);
}

int praline_launch_pipe_keep(const PipeLaunch &pl, const Arena16Dev &a16, int nr, int nterm, void *keep_bnd, float *ckpt,
                             int32_t *end_cells, void *analytic4)
{
    if (pl.rp.mode != PRALINE_MODE_GLOBAL) return PRALINE_ERR_UNSUPPORTED;
    if (nr == 2 && nterm == 2) launch_pipe_keep<2, 2>(pl, a16, keep_bnd, ckpt, end_cells, analytic4);
    else if (nr == 2 && nterm == 3) launch_pipe_keep<2, 3>(pl, a16, keep_bnd, ckpt, end_cells, analytic4);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

int praline_pipe_keep_attrs(int nr, int nterm, int *vgprs, int *lds_bytes)
{
    hipFuncAttributes fa;
    hipError_t e;
    if (nr == 2 && nterm == 2) e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_dp_pipe<2, 2, false, false, true>));
    else if (nr == 2 && nterm == 3) e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_dp_pipe<2, 3, false, false, true>));
    else return PRALINE_ERR_UNSUPPORTED;
    if (e != hipSuccess) return PRALINE_ERR_DEVICE;
    *vgprs = fa.numRegs;
    *lds_bytes = (int)fa.sharedSizeBytes;
    return PRALINE_OK;
}

bool praline_pipe_supported(int nr, int nterm) { return nr == 2 && (nterm == 2 || nterm == 3); }

int praline_launch_pipe(const PipeLaunch &pl, const Arena16Dev &a16, int nr, int nterm)
{
    if (nr == 2 && nterm == 2) launch_pipe<2, 2>(pl, a16);
    else if (nr == 2 && nterm == 3) launch_pipe<2, 3>(pl, a16);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}
