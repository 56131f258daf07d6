// Microbenchmark (gfx950): what one DP step of k_dp_split16 is made of, piece by piece.
//   (1) VALU: the step's add/max mix as NCH independent 32-deep chains per wave (NCH = 1: today's kernel, 2: two tasks
//       per wave, 4) at 1..4 waves per SIMD: cycles per wave-instruction on one SIMD.
//   (2) LDS-DMA: K global_load_lds_dwordx4 (1 KiB each, 8 lanes per 128-byte row, rows scattered over a 14 MB arena)
//       + one global_load_lds_dword per iteration beside a VALU filler: cycles per iteration vs K at 2 waves per SIMD.
// Prints shader cycles (s_memtime of wave 0) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// (dynamic LDS only caps the residency: 160 KiB / wps per 256-thread block = wps blocks per CU)
template <int NCH>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float go, float ge)
{
    float Hs[NCH][17], U[NCH][16], m[NCH][16], lrun[NCH];
    for (int n = 0; n < NCH; ++n) {
        for (int i = 0; i < 17; ++i) Hs[n][i] = threadIdx.x * 3 + i + n;
        for (int i = 0; i < 16; ++i) { U[n][i] = threadIdx.x + 7 * i + n; m[n][i] = (float)((threadIdx.x ^ i) & 15); }
        lrun[n] = threadIdx.x + n;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float hs[NCH];
        for (int n = 0; n < NCH; ++n) hs[n] = Hs[n][0];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
#pragma unroll
            for (int n = 0; n < NCH; ++n) {
                const float M = hs[n] + m[n][c];
                const float Mo = M + go;
                const float Ug = U[n][c] + ge;
                const float H = __builtin_fmaxf(__builtin_fmaxf(M, U[n][c]), lrun[n]);
                lrun[n] = __builtin_fmaxf(Mo, lrun[n] + ge);
                U[n][c] = __builtin_fmaxf(Mo, Ug);
                hs[n] = Hs[n][c + 1];
                Hs[n][c + 1] = H;
            }
        }
        for (int n = 0; n < NCH; ++n) Hs[n][0] = lrun[n];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NCH; ++n) {
        s += lrun[n];
        for (int i = 0; i < 17; ++i) s += Hs[n][i];
        for (int i = 0; i < 16; ++i) s += U[n][i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 18] = t1 - t0;
}

template <int NCH> void run_valu(int wps)
{
    float *d; hipMalloc(&d, (1 << 21) * 4);
    const int iters = 3000, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_valu<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((k_valu<NCH>), dim3(blocks * wps), dim3(256), (size_t)(160 * 1024 / wps) & ~1023u, 0, d, iters, -11.0f, -1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
            unsigned long long cyc; hipMemcpy(&cyc, ((unsigned long long *)d) + (1 << 18), 8, hipMemcpyDeviceToHost);
            const double instr = (double)iters * 16 * 6 * NCH;   // per wave: 3 add, 2 max, 1 max3 per cell
            const double cells = (double)blocks * 4 * wps * 64 * iters * 16 * NCH;
            printf("valu  chains/wave=%d waves/SIMD=%d: %.2f cyc per wave-instr per wave, %.2f cyc per instr per SIMD, %.2f Tcells/s, %.3f ms, clk %.2f GHz\n",
                   NCH, wps, (double)cyc / instr, (double)cyc / instr / wps, cells / ms / 1e9, ms, (double)cyc / ms / 1e6);
        }
    }
    hipFree(d);
}

// ---- LDS-DMA issue cost ----
template <int K, int NV>
__global__ __launch_bounds__(256) void k_dma(const char *arena, const unsigned *rows, float *out, int iters, int nrows, float a)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * 4 * 4096];   // per wave: a ring of four 4 KiB slots
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char *mine = lds + wv * (4 * 4096);
    for (int i = lane * 16; i < 4 * 4096; i += 64 * 16) *reinterpret_cast<float4 *>(mine + i) = make_float4(0, 0, 0, 0);
    __syncthreads();
    const unsigned lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)mine);
    unsigned gofs[4];
    const unsigned widx = (blockIdx.x * 4 + wv) * 32;
    for (int i = 0; i < 4; ++i) gofs[i] = rows[(widx + i * 8 + lane / 8) % nrows] * 128u + (lane % 8) * 16u;
    unsigned long long cur = reinterpret_cast<unsigned long long>(arena);
    cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur >> 32)) << 32) |
          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.01f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = lds_addr + (it & 3) * 4096;
        unsigned keep;
        if constexpr (K >= 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory");
            const float4 v = *reinterpret_cast<const float4 *>(mine + ((it + 2) & 3) * 4096 + lane * 16);
            x[0] += v.x;
        }
#pragma unroll
        for (int r = 0; r < NV / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaxf(x[i] + a, x[(i + 3) & 7]);
        if constexpr (K == 4)
            asm volatile("s_mov_b32 m0, %6\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                         "global_load_lds_dwordx4 %3, %5\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                         : "=&s"(keep) : "v"(gofs[0]), "v"(gofs[1]), "v"(gofs[2]), "v"(gofs[3]), "s"(cur), "s"(slot) : "memory");
        else if constexpr (K == 2)
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                         : "=&s"(keep) : "v"(gofs[0]), "v"(gofs[1]), "s"(cur), "s"(slot) : "memory");
        else if constexpr (K == 1)
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, %2\n\t"
                         : "=&s"(keep) : "v"(gofs[0]), "s"(cur), "s"(slot) : "memory");
        cur += 128;   // next arena row of every sequence
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long *)out)[1 << 18] = t1 - t0;
}

template <int K, int NV> void run_dma(const char *arena, const unsigned *rows, int nrows, int blocks_per_cu)
{
    float *d; hipMalloc(&d, (1 << 21) * 4);
    const int iters = 400, blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_dma<K, NV>), dim3(blocks), dim3(256), 0, 0, arena, rows, d, iters, nrows, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) {
            unsigned long long cyc; hipMemcpy(&cyc, ((unsigned long long *)d) + (1 << 18), 8, hipMemcpyDeviceToHost);
            printf("dma   K=%d pieces/iter, %3d VALU/iter, waves/SIMD=%d: %.0f cyc per iter (wave 0), %.3f ms, %.2f TB/s chip\n", K, 2 * NV,
                   blocks_per_cu, (double)cyc / iters, ms, (double)blocks * 4 * K * 1024.0 * iters / ms / 1e9);
        }
    }
    hipFree(d);
}

int main()
{
    for (int wps = 1; wps <= 4; ++wps) run_valu<1>(wps);
    for (int wps = 1; wps <= 3; ++wps) run_valu<2>(wps);
    for (int wps = 1; wps <= 2; ++wps) run_valu<4>(wps);
    // 14 MB arena of 128-byte rows; 110 000 rows, 32-row groups start at random rows (like 32 sequences' cursors)
    const int nrows_total = 110000, nstart = 1 << 16;
    char *arena; hipMalloc(&arena, (size_t)nrows_total * 128 + (1 << 20));
    hipMemset(arena, 0, (size_t)nrows_total * 128 + (1 << 20));
    std::vector<unsigned> rows(nstart);
    srand(1);
    for (int i = 0; i < nstart; ++i) rows[i] = (unsigned)(rand() % (nrows_total - 1000));
    unsigned *drows; hipMalloc(&drows, nstart * 4);
    hipMemcpy(drows, rows.data(), nstart * 4, hipMemcpyHostToDevice);
    run_dma<0, 72>(arena, drows, nstart, 2);
    run_dma<1, 72>(arena, drows, nstart, 2);
    run_dma<2, 72>(arena, drows, nstart, 2);
    run_dma<4, 72>(arena, drows, nstart, 2);
    run_dma<0, 144>(arena, drows, nstart, 2);
    run_dma<1, 144>(arena, drows, nstart, 2);
    run_dma<2, 144>(arena, drows, nstart, 2);
    run_dma<4, 144>(arena, drows, nstart, 2);
    run_dma<4, 144>(arena, drows, nstart, 1);
    return 0;
}
