// dp_rawb_instance.hip -- the kernels of a batch of RawPairwiseAligner requests (dp_rawb.hip.h) and their launches.
#include "dp_rawb.hip.h"

void praline_launch_rawb_init(const RawBatchDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_rawb_init, dim3((unsigned)d.n), dim3(256), 0, st, d);
}

void praline_launch_rawb_zero(const RawBatchDev &d, const int32_t *zero_req, const int32_t *zero_idx, int64_t n_zero, hipStream_t st)
{
    if (n_zero <= 0) return;
    hipLaunchKernelGGL(k_rawb_zero, dim3((unsigned)((n_zero + 255) / 256)), dim3(256), 0, st, d, zero_req, zero_idx, n_zero);
}

void praline_launch_rawb_fill(const RawBatchDev &d, int waves, bool mask, hipStream_t st)
{
    const dim3 grid((unsigned)d.n), block(64u * (unsigned)waves);
    if (mask) hipLaunchKernelGGL(k_rawb_fill<true>, grid, block, 0, st, d);
    else hipLaunchKernelGGL(k_rawb_fill<false>, grid, block, 0, st, d);
}

void praline_launch_rawb_trace(const RawBatchDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_rawb_trace, dim3((unsigned)((d.n + 63) / 64)), dim3(64), 0, st, d);
}
