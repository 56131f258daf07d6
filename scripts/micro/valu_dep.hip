// Microbenchmark: VALU issue rate of a lone wave vs. dependency distance (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ILP, int KIND>
__global__ void k(float* out, int iters, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / ILP; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) x[i] = x[i] + a;                       // v_add chain
                else if (KIND == 1) x[i] = __builtin_fmaxf(x[i] + a, b);  // add + max chain (2 dependent ops)
                else x[i] = __builtin_fmaxf(__builtin_fmaxf(x[i], a), x[(i + 1) % ILP] + b);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 16] = t1 - t0;
}
template <int ILP, int KIND> void run(const char* name, int waves_per_block, int blocks) {
    float* d; hipMalloc(&d, (1 << 20) * 4);
    int iters = 2000;
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, d, iters, 1.5f, 0.25f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, d, iters, 1.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, ((long long*)d) + (1 << 16), 8, hipMemcpyDeviceToHost);
    int ops_per_iter = 64 * (KIND == 0 ? 1 : KIND == 1 ? 2 : 3);
    printf("%-28s ILP=%d waves/blk=%d blocks=%d: %.2f memtime-ticks/VALU-op (wave0), kernel %.3f ms\n", name, ILP, waves_per_block, blocks,
           (double)cyc / ((double)iters * ops_per_iter), ms);
    hipFree(d);
}
int main() {
    // 256 CUs: 256 blocks of 4 waves = 1 wave per SIMD; 8 waves = 2 per SIMD
    run<1, 0>("add chain", 4, 256); run<2, 0>("add chain", 4, 256); run<4, 0>("add chain", 4, 256); run<8, 0>("add chain", 4, 256);
    run<1, 0>("add chain", 8, 256); run<2, 0>("add chain", 8, 256); run<8, 0>("add chain", 8, 256);
    run<1, 1>("add+max chain", 4, 256); run<2, 1>("add+max chain", 4, 256); run<4, 1>("add+max chain", 4, 256);
    run<1, 1>("add+max chain", 8, 256); run<4, 1>("add+max chain", 8, 256);
    run<1, 0>("add chain 1 wave/CU", 1, 256); run<4, 0>("add chain 1 wave/CU", 1, 256);
    return 0;
}
