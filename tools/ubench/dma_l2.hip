// Micro-benchmark (round 5): L2 -> LDS supply.  Every workgroup streams contiguous KiB pieces (full 128-byte lines, buffer_load ... lds)
// through a footprint of F bytes; SHARE = 1: all workgroups read the same F bytes in the same order at the same time (what the
// weight stream of a conv launch does), SHARE = 0: each workgroup has its own F bytes.  F from L1-sized to L2-sized to beyond.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BURST, int GLOBAL>
__global__ void k(const char *src, unsigned fp, int share, int iters, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, wave = t >> 6, nw = blockDim.x >> 6, lane = t & 63;
    const unsigned base = share ? 0u : (unsigned)blockIdx.x * fp;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
    char *dst = smem + wave * (BURST * 1024);
    unsigned pos = (unsigned)wave * BURST * 1024u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < BURST; ++b)
            if (GLOBAL)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(base + pos + b * 1024) + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(dst + b * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(dst + b * 1024), 16, (unsigned)(lane * 16), base + pos + b * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pos += (unsigned)nw * BURST * 1024u;
        if (pos >= fp) pos -= fp;
    }
    __syncthreads();
    if (t == 0 && smem[5] == 77) sink[0] = 1;
}
int main() {
    const size_t bytes = 1u << 30;
    char *src; int *sink;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&sink, 4); (void)hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256, iters = 256;
    constexpr int BURST = 8;
    for (int glob : {0, 1})
      for (int waves : {2, 4, 8})
        for (int share : {1, 0})
            for (unsigned fp : {16u << 10, 128u << 10, 2u << 20}) {
                if (!share && (size_t)fp * grid > bytes) continue;
                auto fn = glob ? k<BURST, 1> : k<BURST, 0>;
                (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BURST * waves * 1024);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), BURST * waves * 1024, 0, src, fp, share, iters, sink);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), BURST * waves * 1024, 0, src, fp, share, iters, sink);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double moved = (double)grid * iters * waves * BURST * 1024;
                printf("%s waves %d %s footprint %5u KiB per %s: %6.2f TB/s  %6.1f GB/s per CU  %5.1f B/clk/CU @2.1GHz\n", glob ? "global_load_lds" : "buffer_load_lds", waves,
                       share ? "shared " : "private", fp >> 10, share ? "chip" : "workgroup", moved / ms / 1e9, moved / ms / 1e6 / 256, moved / ms / 1e6 / 256 / 2.1);
            }
    return 0;
}
