// dp_rawb_instance.hip -- the kernels of a batch of RawPairwiseAligner requests (dp_rawb.hip.h) and their launches.
#include "dp_rawb.hip.h"
#include <algorithm>

void praline_launch_rawb_init(const RawBatchDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_rawb_init, dim3((unsigned)d.n), dim3(256), 0, st, d);
}

void praline_launch_rawb_stage(const RawBatchDev &d, const float *src, float *dst, const int64_t *block0, int64_t n_blocks, hipStream_t st)
{
    hipLaunchKernelGGL(k_rawb_stage, dim3((unsigned)n_blocks), dim3(256), 0, st, d, src, reinterpret_cast<float4 *>(dst), block0);
}

void praline_launch_rawb_zero(const RawBatchDev &d, const int32_t *zero_req, const int32_t *zero_idx, int64_t n_zero, hipStream_t st)
{
    if (n_zero <= 0) return;
    hipLaunchKernelGGL(k_rawb_zero, dim3((unsigned)((n_zero + 255) / 256)), dim3(256), 0, st, d, zero_req, zero_idx, n_zero);
}

// global_like / local: whether the batch holds requests of that kind (each instance skips the other's)
void praline_launch_rawb_fill(const RawBatchDev &d, int waves, bool mask, bool global_like, bool local, hipStream_t st)
{
    // two workgroups of up to eight waves per CU (128 registers per lane): no more workgroups than are resident at once
    const dim3 grid((unsigned)std::min(d.n, PRALINE_RAWB_GROUPS)), block(64u * (unsigned)waves);
    if (global_like) {
        if (mask) hipLaunchKernelGGL((k_rawb_fill<true, false>), grid, block, 0, st, d);
        else hipLaunchKernelGGL((k_rawb_fill<false, false>), grid, block, 0, st, d);
    }
    if (local) {
        if (mask) hipLaunchKernelGGL((k_rawb_fill<true, true>), grid, block, 0, st, d);
        else hipLaunchKernelGGL((k_rawb_fill<false, true>), grid, block, 0, st, d);
    }
}

void praline_launch_rawb_trace(const RawBatchDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_rawb_trace, dim3((unsigned)((d.n + 3) / 4)), dim3(256), 0, st, d);
}
