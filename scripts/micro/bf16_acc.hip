// Probe the accumulation semantics of v_mfma_f32_32x32x16_bf16 (gfx950): dump A, B, C, D.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
    // A: [32][16] row-major (row i, k), B: [16][32] (k, col j), C/D: [32][32]
    const int l = threadIdx.x, r = l & 31, hh = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[r * 16 + 8 * hh + j]; b[j] = (short)B[(8 * hh + j) * 32 + r]; }
    f32x16 c;
    for (int q = 0; q < 16; ++q) { const int row = (q & 3) + 8 * (q >> 2) + 4 * hh; c[q] = C[row * 32 + r]; }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int q = 0; q < 16; ++q) { const int row = (q & 3) + 8 * (q >> 2) + 4 * hh; D[row * 32 + r] = c[q]; }
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main() {
    std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> C(1024), D(1024);
    srand(7);
    FILE* f = fopen("gpurun_out/bf16_probe.bin", "wb");
    for (int trial = 0; trial < 8; ++trial) {
        for (auto& x : A) x = f2bf(((rand() % 20001) - 10000) / (trial < 4 ? 3000.0f : 7.0f) * ((rand() & 7) == 0 ? 1e-3f : 1.0f));
        for (auto& x : B) x = f2bf(((rand() % 20001) - 10000) / 3000.0f * ((rand() & 7) == 0 ? 1e3f : 1.0f));
        for (auto& x : C) x = (trial & 1) ? 0.0f : ((rand() % 20001) - 10000) / 100.0f;
        unsigned short *dA, *dB; float *dC, *dD;
        (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
        (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, 1024, f); fwrite(D.data(), 4, 1024, f);
    }
    fclose(f); printf("wrote probe\n"); return 0;
}
