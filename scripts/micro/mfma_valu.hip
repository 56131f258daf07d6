// Does an MFMA wave overlap with a VALU wave on the same SIMD?  (gfx950)
// Block = 8 waves (2 per SIMD).  Roles by wave id: waves 0-3 do role A, waves 4-7 role B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
enum { NONE = 0, MFMA_F32 = 1, VALU = 2, MFMA_BF16 = 3, MFMA_F32_16 = 4 };
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ float do_role(int role, int iters, float seed) {
    if (role == MFMA_F32) {
        f32x16 acc = {0}; float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        return acc[0] + acc[7];
    } else if (role == MFMA_F32_16) {
        f32x4 acc = {0}; float a = seed, b = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
        return acc[0] + acc[3];
    } else if (role == MFMA_BF16) {
        f32x16 acc = {0}; bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + i); b[i] = (short)(0x3f00 + i); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        return acc[0] + acc[7];
    } else if (role == VALU) {
        float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {   // 256 VALU ops per iteration, 4 independent chains
                x0 = __builtin_fmaxf(x0 + 1.5f, 0.25f); x1 = __builtin_fmaxf(x1 + 1.5f, 0.25f);
                x2 = __builtin_fmaxf(x2 + 1.5f, 0.25f); x3 = __builtin_fmaxf(x3 + 1.5f, 0.25f);
            }
        }
        return x0 + x1 + x2 + x3;
    }
    return 0.f;
}
__global__ void k(float* out, int roleA, int roleB, int itA, int itB) {
    const int wave = threadIdx.x >> 6;
    float r = (wave < 4) ? do_role(roleA, itA, threadIdx.x * 1e-3f) : do_role(roleB, itB, threadIdx.x * 1e-3f);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
float run(int roleA, int roleB, int itA, int itB) {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, roleA, roleB, itA, itB);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, roleA, roleB, itA, itB);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipFree(d); return ms;
}
int main() {
    const int IT = 4000;
    float m32 = run(MFMA_F32, NONE, IT, 0), v = run(NONE, VALU, 0, IT), both32 = run(MFMA_F32, VALU, IT, IT);
    printf("f32 32x32x2 MFMA alone (16/iter): %.3f ms -> %.1f cycles/MFMA @2.3GHz\n", m32, m32 * 1e-3 * 2.3e9 / (IT * 16));
    printf("VALU alone (256 ops/iter)        : %.3f ms -> %.2f cycles/op\n", v, v * 1e-3 * 2.3e9 / (IT * 256));
    printf("f32 MFMA wave + VALU wave / SIMD : %.3f ms  (sum %.3f, max %.3f)\n", both32, m32 + v, m32 > v ? m32 : v);
    float m16 = run(MFMA_F32_16, NONE, IT, 0), both16 = run(MFMA_F32_16, VALU, IT, IT);
    printf("f32 16x16x4 MFMA alone           : %.3f ms -> %.1f cycles/MFMA; with VALU wave %.3f ms (sum %.3f)\n", m16, m16 * 1e-3 * 2.3e9 / (IT * 16), both16, m16 + v);
    float mb = run(MFMA_BF16, NONE, IT, 0), bothb = run(MFMA_BF16, VALU, IT, IT);
    printf("bf16 32x32x16 MFMA alone         : %.3f ms -> %.1f cycles/MFMA; with VALU wave %.3f ms (sum %.3f, max %.3f)\n", mb, mb * 1e-3 * 2.3e9 / (IT * 16), bothb, mb + v, mb > v ? mb : v);
    float vv = run(VALU, VALU, IT, IT);
    printf("VALU + VALU waves on a SIMD      : %.3f ms (2x alone = %.3f)\n", vv, 2 * v);
    return 0;
}
