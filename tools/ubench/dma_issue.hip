// Micro-benchmark (round 5): LDS-DMA ingest per CU vs the number of ISSUING waves with the address arithmetic of a real
// producer wave (precomputed per-lane source pointer + per-step scalar offset; no integer division in the loop, unlike
// dma_depth.hip whose 64-bit `%` per issue turned out to be what capped a wave at 1 KiB per ~300 cycles).
// Source: an L2-resident 2 MiB region shared by every CU (what weight tiles are), 16 rows x 64 B per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BURST>
__global__ void k(const char *src, int iters, int stride, int *sink, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, wave = t >> 6, nw = blockDim.x >> 6, lane = t & 63;
    const char *p = src + (size_t)((blockIdx.x & 7) * 65536) + (size_t)(wave * 16 + (lane >> 2)) * stride + ((lane & 3) << 4);
    char *dst = smem + wave * (BURST * 1024);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const char *q = p + (size_t)((it & 15) * 64);
#pragma unroll
        for (int b = 0; b < BURST; ++b)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(q + (size_t)b * (nw * 16) * stride),
                                             (__attribute__((address_space(3))) void *)(dst + b * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (t == 0 && smem[5] == 77) sink[0] = 1;
    if (t == 0 && blockIdx.x == 8) cyc[0] = t1 - t0;
}
template <int BURST>
void run(const char *src, int *sink, long long *cyc, int waves, int stride) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 512, grid = 256;
    hipFuncSetAttribute((const void *)k<BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, BURST * waves * 1024);
    hipLaunchKernelGGL(k<BURST>, dim3(grid), dim3(waves * 64), BURST * waves * 1024, 0, src, iters, stride, sink, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<BURST>, dim3(grid), dim3(waves * 64), BURST * waves * 1024, 0, src, iters, stride, sink, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double moved = (double)grid * iters * waves * BURST * 1024;
    printf("waves/CU %2d burst %2d stride %4d: %.3f ms %6.2f TB/s  %6.1f GB/s per CU  %5.1f GB/s per wave  %6.1f cycles per KiB per wave  %5.1f B/clk/CU\n", waves, BURST, stride,
           ms, moved / ms / 1e9, moved / ms / 1e6 / 256, moved / ms / 1e6 / 256 / waves, (double)c / (iters * BURST), (double)iters * BURST * waves * 1024 / c);
}
int main() {
    const size_t bytes = 8u << 20;
    char *src; int *sink; long long *cyc;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4); hipMalloc(&cyc, 8); hipMemset(src, 1, bytes);
    for (int stride : {64, 2304})
        for (int waves : {1, 2, 4, 8, 16}) {
            run<2>(src, sink, cyc, waves, stride); run<4>(src, sink, cyc, waves, stride); run<8>(src, sink, cyc, waves, stride);
            if (waves <= 8) run<16>(src, sink, cyc, waves, stride);
        }
    return 0;
}
