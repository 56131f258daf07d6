// Microbenchmark: issue rate of the packed 16-bit integer VALU ops a 16-bit DP cell update would use
// (v_pk_add_i16 with clamp, v_pk_max_i16) against the f32 ops of the current kernels, at one and two waves per
// SIMD, and the LDS gather that would replace the MFMA tile (ds_read_u16_d16 / _d16_hi).  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned pk_add_sat(unsigned a, unsigned b)
{
    unsigned d;
    asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    unsigned d;
    asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// KIND 0: the packed cell update of 16 columns (M add, Mo add, Ug add, H = max(max(M, U), L), L chain add + max,
//         U max: 8 ops per column, two pairs per lane);  KIND 1: the f32 update (7 ops per column, v_max3)
template <int KIND>
__global__ void k(unsigned *out, int iters, unsigned go, unsigned ge, float gof, float gef)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (KIND == 0) {
        unsigned Hs[17], U[16], m[16];
        for (int i = 0; i < 17; ++i) Hs[i] = threadIdx.x * 3 + i;
        for (int i = 0; i < 16; ++i) { U[i] = threadIdx.x + 7 * i; m[i] = (threadIdx.x ^ i) & 0x000f000f; }
        unsigned lrun = threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            unsigned hs = Hs[0];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const unsigned M = pk_add_sat(hs, m[c]);
                const unsigned Mo = pk_add_sat(M, go);
                const unsigned Ug = pk_add_sat(U[c], ge);
                const unsigned H = pk_max(pk_max(M, U[c]), lrun);
                lrun = pk_max(Mo, pk_add_sat(lrun, ge));
                U[c] = pk_max(Mo, Ug);
                hs = Hs[c + 1];
                Hs[c + 1] = H;
            }
            Hs[0] = lrun;
        }
        unsigned s = lrun;
        for (int i = 0; i < 17; ++i) s ^= Hs[i];
        for (int i = 0; i < 16; ++i) s ^= U[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        float Hs[17], U[16], m[16];
        for (int i = 0; i < 17; ++i) Hs[i] = threadIdx.x * 3 + i;
        for (int i = 0; i < 16; ++i) { U[i] = threadIdx.x + 7 * i; m[i] = (float)((threadIdx.x ^ i) & 15); }
        float lrun = threadIdx.x;
        for (int it = 0; it < iters; ++it) {
            float hs = Hs[0];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float M = hs + m[c];
                const float Mo = M + gof;
                const float Ug = U[c] + gef;
                const float H = __builtin_fmaxf(__builtin_fmaxf(M, U[c]), lrun);   // folded into v_max3_f32
                lrun = __builtin_fmaxf(Mo, lrun + gef);
                U[c] = __builtin_fmaxf(Mo, Ug);
                hs = Hs[c + 1];
                Hs[c + 1] = H;
            }
            Hs[0] = lrun;
        }
        float s = lrun;
        for (int i = 0; i < 17; ++i) s += Hs[i];
        for (int i = 0; i < 16; ++i) s += U[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 16] = t1 - t0;
}

// LDS gather: 32 x ds_read_u16_d16(_hi) per row (16 columns, two pairs per lane) from a 33-row query profile
__global__ void k_gather(unsigned *out, int iters, const unsigned char *sym)
{
    __shared__ unsigned short tab[33 * 34];
    for (int i = threadIdx.x; i < 33 * 34; i += blockDim.x) tab[i] = (unsigned short)(i * 7);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    unsigned acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned sa = sym[(it * 64 + lane) & 4095] % 33, sb = sym[(it * 64 + lane + 17) & 4095] % 33;
        const unsigned aa = (unsigned)(uintptr_t)(tab + sa * 34), ab = (unsigned)(uintptr_t)(tab + sb * 34);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            unsigned v = 0;
            asm volatile("ds_read_u16_d16 %0, %1 offset:%2" : "+v"(v) : "v"(aa), "n"(2 * c));
            asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(v) : "v"(ab), "n"(2 * c));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc ^= v;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 16] = t1 - t0;
}

template <int KIND> void run(const char *name, int waves_per_block)
{
    unsigned *d; hipMalloc(&d, (1 << 20) * 4);
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, d, iters, 0xfff5fff5u, 0xffffffffu, -11.0f, -1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
            const double cells = (double)blocks * waves_per_block * 64 * iters * 16 * (KIND == 0 ? 2 : 1);
            printf("%-22s %d wave(s)/SIMD: %.3f ms, %.2f Tcell-updates/s on the whole chip\n", name, waves_per_block / 4, ms, cells / ms / 1e9);
        }
    }
    hipFree(d);
}

int main()
{
    run<1>("f32 update (7 ops)", 4); run<1>("f32 update (7 ops)", 8);
    run<0>("pk i16 update (8 ops)", 4); run<0>("pk i16 update (8 ops)", 8);
    unsigned *d; hipMalloc(&d, (1 << 20) * 4);
    unsigned char *s; hipMalloc(&s, 4096);
    std::vector<unsigned char> hs(4096);
    for (int i = 0; i < 4096; ++i) hs[i] = (unsigned char)((i * 2654435761u) >> 24);
    hipMemcpy(s, hs.data(), 4096, hipMemcpyHostToDevice);
    for (int wpb : {4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_gather, dim3(256), dim3(64 * wpb), 0, 0, d, 2000, s);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(256), dim3(64 * wpb), 0, 0, d, 2000, s);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("LDS gather (32 d16 reads per row, waited one by one) %d wave(s)/SIMD: %.3f ms = %.3f us per row and wave\n", wpb / 4, ms, ms * 1e3 / 2000);
    }
    return 0;
}
