// Micro-benchmark: LDS-DMA throughput of ONE workgroup per CU vs the number of DMA instructions each wave keeps
// in flight (L2-resident source, 16 rows x 64 B per wave instruction) and vs waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH>
__global__ void k(const char *src, size_t bytes, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, wave = t >> 6, nw = blockDim.x >> 6;
    const int lrow = (t & 63) >> 2, lslot = t & 3;
    size_t base = ((size_t)blockIdx.x * 7919 * 64) % (bytes / 2);
    for (int it = 0; it < iters; ++it) {
        const size_t off = (base + (size_t)(wave * 16 + lrow) * 512 + (size_t)it * 64 + lslot * 16) % (bytes - 64);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (off & ~size_t(15))),
                                         (__attribute__((address_space(3))) void *)(smem + ((it % DEPTH) * nw + wave) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0 && smem[5] == 77) sink[0] = 1;
}
template <int DEPTH>
void run(const char *src, size_t bytes, int *sink, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, grid = 256;
    hipFuncSetAttribute((const void *)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * waves * 1024);
    hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(waves * 64), DEPTH * waves * 1024, 0, src, bytes, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(waves * 64), DEPTH * waves * 1024, 0, src, bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)grid * iters * waves * 1024;
    printf("waves/CU %2d depth %2d: %.3f ms %6.2f TB/s  %5.1f GB/s per CU  in flight %3d KB -> implied latency %.2f us\n", waves, DEPTH,
           ms, moved / ms / 1e9, moved / ms / 1e6 / 256, DEPTH * waves, DEPTH * waves * 1024.0 / (moved / ms / 1e3 / 256) );
}
int main() {
    const size_t bytes = 3u << 20;
    char *src; int *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4); hipMemset(src, 1, bytes);
    for (int waves : {4, 8, 16}) {
        run<1>(src, bytes, sink, waves); run<2>(src, bytes, sink, waves); run<4>(src, bytes, sink, waves);
        run<8>(src, bytes, sink, waves);
        if (waves <= 8) run<16>(src, bytes, sink, waves);
        if (waves <= 4) run<32>(src, bytes, sink, waves);
    }
    return 0;
}
