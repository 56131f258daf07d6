// Microbenchmark (gfx950): what ONE wave pays per instruction of each kind beside the DP recurrence's VALU mix.
// Loop body = 96 VALU (16 cells x 6) + N extra instructions of kind X; reported: (cycles(N) - cycles(0)) / N per wave,
// at 2 waves per SIMD (two 256-thread workgroups per CU).  Kinds: SALU adds, SALU compare+cselect, a not-taken and a
// taken scalar branch, v_cndmask with an SGPR-pair mask, v_readlane/v_writelane (SGPR spills), s_waitcnt on empty
// counters, s_barrier (4 waves), ds_read_b64 + wait, an empty saveexec/branch/restore triple.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float *out, int iters, float go, float ge, int zero)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float Hs[17], U[16], m[16];
    for (int i = 0; i < 17; ++i) Hs[i] = threadIdx.x * 3 + i;
    for (int i = 0; i < 16; ++i) { U[i] = threadIdx.x + 7 * i; m[i] = (float)((threadIdx.x ^ i) & 15); }
    float lrun = threadIdx.x;
    unsigned s0 = (unsigned)zero, s1 = 1u + (unsigned)zero;
    float vx = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float hs = Hs[0];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float M = hs + m[c];
            const float Mo = M + go;
            const float Ug = U[c] + ge;
            const float H = __builtin_fmaxf(__builtin_fmaxf(M, U[c]), lrun);
            lrun = __builtin_fmaxf(Mo, lrun + ge);
            U[c] = __builtin_fmaxf(Mo, Ug);
            hs = Hs[c + 1];
            Hs[c + 1] = H;
            // extras spread over the 16 cells
#pragma unroll
            for (int e = (N * c) / 16; e < (N * (c + 1)) / 16; ++e) {
                if constexpr (KIND == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
                else if constexpr (KIND == 2) asm volatile("s_cmp_eq_u32 %0, %1\n\ts_cselect_b32 %0, %1, %0" : "+s"(s0) : "s"(s1) : "scc");
                else if constexpr (KIND == 3) asm volatile("s_cmp_eq_u32 %0, 12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(s1) : "scc");   // compare + not-taken branch + nop
                else if constexpr (KIND == 4) asm volatile("s_cmp_lg_u32 %0, 12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(s1) : "scc");   // compare + taken branch
                else if constexpr (KIND == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(vx) : "v"(lrun));
                else if constexpr (KIND == 6) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(vx));
                else if constexpr (KIND == 7) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(vx) : "s"(s1));
                else if constexpr (KIND == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (KIND == 9) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if constexpr (KIND == 10) { float t; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(threadIdx.x * 4)) : "memory"); vx += t; }
                else if constexpr (KIND == 11) asm volatile("s_and_saveexec_b64 s[4:5], s[2:3]\n\ts_cbranch_execz 1f\n\ts_nop 0\n1:\n\ts_or_b64 exec, exec, s[4:5]" ::: "s4", "s5", "scc");
                else if constexpr (KIND == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(vx) : "v"(ge));
                else if constexpr (KIND == 13) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(vx), "+v"(lrun));
                else if constexpr (KIND == 14) asm volatile("s_nop 0");
                else if constexpr (KIND == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(vx) : "v"(lrun));
            }
        }
        Hs[0] = lrun;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = lrun + vx + (float)s0;
    for (int i = 0; i < 17; ++i) s += Hs[i];
    for (int i = 0; i < 16; ++i) s += U[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 18] = t1 - t0;
}

static double base_cycles = 0;
template <int KIND, int N> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, (1 << 21) * 4);
    const int iters = 2000;
    double cyc = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, N>), dim3(512), dim3(256), 0, 0, d, iters, -11.0f, -1.0f, 0);
        (void)hipDeviceSynchronize();
        unsigned long long c; (void)hipMemcpy(&c, ((unsigned long long *)d) + (1 << 18), 8, hipMemcpyDeviceToHost);
        cyc = (double)c / iters;
    }
    if (KIND == 0) base_cycles = cyc;
    if (hipGetLastError() != hipSuccess) printf("HIP error after %s\n", name);
    printf("%-44s N=%2d: %7.1f cyc/iter  -> %5.2f cyc per extra\n", name, N, cyc, N ? (cyc - base_cycles) / N : 0.0);
    (void)hipFree(d);
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<0, 0>("base: 96 VALU");
    run<12, 16>("v_add_f32"); run<12, 48>("v_add_f32");
    run<15, 16>("v_mov_b32");
    run<1, 16>("s_add_u32"); run<1, 48>("s_add_u32");
    run<2, 16>("s_cmp + s_cselect (2 instr)");
    run<3, 16>("s_cmp + not-taken branch + s_nop (3)");
    run<4, 16>("s_cmp + taken branch (2)");
    run<5, 16>("v_cndmask with SGPR mask");
    run<6, 16>("v_readlane"); run<7, 16>("v_writelane");
    run<8, 16>("s_waitcnt vmcnt(0), nothing outstanding");
    run<10, 4>("ds_read_b32 + wait"); run<10, 16>("ds_read_b32 + wait");
    run<14, 16>("s_nop 0"); run<14, 48>("s_nop 0");
    run<13, 16>("v_permlane32_swap");
    run<9, 4>("s_waitcnt lgkmcnt(0) + s_barrier (2)"); run<9, 16>("s_waitcnt lgkmcnt(0) + s_barrier (2)");
    run<11, 16>("saveexec + execz branch + nop + restore (4)");
    return 0;
}
