// Micro-benchmark: global_load_lds (LDS-DMA) throughput from an L2-resident buffer as a function of the
// row stride of the 64-B-per-row tile pattern used by the conv kernel (16 rows x 64 B per wave instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const char *src, size_t bytes, int row_stride, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, wave = t >> 6;
    const int lrow = t >> 2, lslot = t & 3;
    // each WG walks its own region; rows = 128 per chunk (2 passes of 64 rows), like BM = 128
    size_t base = ((size_t)blockIdx.x * 7919 * 64) % (bytes / 2);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 4 x (64 rows x 64 B) = 16 KB per iteration per WG
            const size_t off = (base + (size_t)(lrow + 64 * i) * row_stride + (size_t)it * 64 + lslot * 16) % (bytes - 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (off & ~size_t(15))),
                                             (__attribute__((address_space(3))) void *)(smem + (it & 1) * 16384 + i * 4096 + wave * 1024), 16, 0, 0);
        }
        if ((it & 3) == 3) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0 && smem[5] == 77) sink[0] = 1;
}
int main() {
    const size_t bytes = 3u << 20;  // 3 MB: L2-resident per XCD
    char *src; int *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4); hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu : {1, 2, 4}) {
        for (int stride : {64, 128, 256, 512, 1024, 2048, 4608}) {
            const int iters = 512, grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 32768, 0, src, bytes, stride, iters, sink);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 32768, 0, src, bytes, stride, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double moved = (double)grid * iters * 16384;
            printf("WG/CU %d row stride %5d B: %.3f ms  %.2f TB/s aggregate  (%.1f B/clk/CU @2.4GHz)\n", wgs_per_cu, stride, ms,
                   moved / ms / 1e9, moved / ms / 1e6 / 256 / 2.4e3 * 1e-3 * 1e3);
        }
    }
    return 0;
}
