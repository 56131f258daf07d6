// Microbenchmark (gfx950): cycles per instruction and SIMD of the cross-lane moves k_rawb_fill could use for its
// "previous lane" neighbours - v_mov_b32_dpp wave_shr:1 (one instruction), row_shr:1 (within 16 lanes), row_bcast:15 +
// row_shr:1 (the two-instruction whole-wave shift) - beside v_add_f32, v_cmp + v_addc (a tie flag) and v_readlane + v_mov,
// at 1 .. 4 waves per SIMD, eight independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
    float b = seed * 0.5f;
    unsigned f = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 1) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (KIND == 2) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (KIND == 3) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xe bank_mask:0x1\n\ts_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_cmp_eq_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(f) : "v"(a[i]), "v"(b) : "vcc");
                if (KIND == 5) asm volatile("v_readlane_b32 s20, %0, 3\n\tv_mov_b32 %0, s20" : "+v"(a[i]) : : "s20");
                if (KIND == 6) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_cmp_eq_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, 0, 2, s[20:21]" : "+v"(f) : "v"(a[i]), "v"(b) : "s20", "s21");
            }
        }
    }
    float s = (float)f;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char *name, int per)
{
    float *d;
    (void)hipMalloc(&d, (size_t)256 * 4 * 256 * 4 * 4);
    const int iters = 2000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const dim3 grid(256 * waves), block(256);   // one wave per SIMD and block: `waves` blocks per CU
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 10, 1.0f);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 64 * per * waves;   // per SIMD
        printf("%-28s %d waves/SIMD: %.2f cycles per instruction and SIMD (2.4 GHz)\n", name, waves, ms * 1e-3 * 2.4e9 / insts);
    }
    (void)hipFree(d);
}

int main()
{
    run<0>("v_add_f32", 1);
    run<6>("v_max3_f32", 1);
    run<1>("dpp wave_shr:1", 1);
    run<2>("dpp row_shr:1", 1);
    run<3>("dpp row_bcast:15 + row_shr:1", 2);
    run<4>("v_cmp + v_addc", 2);
    run<7>("v_cmp(sgpr) + v_cndmask", 2);
    run<5>("v_readlane + v_mov", 2);
    return 0;
}
