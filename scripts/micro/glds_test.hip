// Micro-test: addressing of global_load_lds_dwordx4 / _dword on gfx950 (does inst_offset move the LDS
// destination as well as the source?), run by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];  // 16 KB
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdead0000u + i;
    __syncthreads();
    const unsigned lane = threadIdx.x;
    const unsigned* p = src + lane * 4;                 // lane's 16 bytes: dwords 4*lane..
    const unsigned base = (unsigned)(uintptr_t)lds + 2048;  // byte offset 2048 inside the array
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p), "s"(base) : "memory");
    const unsigned* p4 = src + 8192 + lane;             // dword form
    const unsigned base4 = (unsigned)(uintptr_t)lds + 8192;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, off offset:-256\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p4), "s"(base4) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main()
{
    std::vector<unsigned> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 16384 * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(4096);
    hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i)
        if (r[i] != 0xdead0000u + i) { if (first < 0 || (i > 0 && r[i - 1] == 0xdead0000u + i - 1)) printf("changed run starts at dword %d (byte %d): value %u\n", i, 4 * i, r[i]); first = i; ++n; }
    printf("%d dwords changed\n", n);
    // expectations: x4: M0 byte 2048 -> dword 512 (if offset does not move LDS) or dword 768 (if it does);
    //               value 0 (offset not applied to source) or 256 (applied: +1024 bytes)
    return 0;
}
