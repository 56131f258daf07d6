// Microbenchmark (gfx950): rate of the reference-order match-score inner body - 8 independent multiplies followed by a
// chain of 8 dependent adds into one accumulator - written with plain fp32 VALU (v_mul_f32 / v_add_f32) and with packed
// fp32 (v_pk_mul_f32 / v_pk_add_f32, two cells per lane), at 1 .. 6 waves per SIMD.  Reported: cell-terms per second
// of the whole chip and cycles per VALU instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    const float l = (float)threadIdx.x * 1e-3f + seed;
    if constexpr (PK) {
        f2 acc[16], t[8];
        for (int i = 0; i < 16; ++i) acc[i] = f2{l + i, l - i};
        for (int i = 0; i < 8; ++i) t[i] = f2{1.0f + l * i, 1.0f - l * i};
        f2 vv = {1.0001f + l, 0.9999f - l};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int yy = 0; yy < 16; ++yy) {
                f2 p[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (yy & 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(p[b]) : "v"(t[b]), "v"(vv));
                    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p[b]) : "v"(t[b]), "v"(vv));
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[yy]) : "v"(p[b]));
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        float acc[16], t[8];
        for (int i = 0; i < 16; ++i) acc[i] = l + i;
        for (int i = 0; i < 8; ++i) t[i] = 1.0f + l * i;
        float v = 1.0001f + l;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int yy = 0; yy < 16; ++yy) {
                float p[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[b]) : "v"(t[b]), "v"(v));
#pragma unroll
                for (int b = 0; b < 8; ++b) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[yy]) : "v"(p[b]));
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <bool PK> void run(int waves)
{
    float *d;
    (void)hipMalloc(&d, (size_t)256 * 8 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<PK>), dim3(256 * waves), dim3(256), 0, 0, d, iters, 0.5f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr_per_wave = (double)iters * 16 * 16;
    const double waves_total = 256.0 * waves * 4;
    const double terms = instr_per_wave / 2 * waves_total * 64 * (PK ? 2 : 1);   // (mul, add) pairs x lanes x cells per lane
    printf("%s %d waves/SIMD: %8.3f ms  %7.2f Tterm/s  (C2 match scores at 64 terms per cell: %6.2f ms)   %5.2f ns per instr per SIMD\n",
           PK ? "packed" : "plain ", waves, ms, terms / ms / 1e9, 5.2e9 * 64 / (terms / ms) , ms * 1e6 / (instr_per_wave * waves));
    fflush(stdout);
    (void)hipFree(d);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int w = 1; w <= 6; ++w) { run<false>(w); run<true>(w); }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
