// Micro-benchmark (round 4): what does ONE workgroup barrier per loop iteration cost as a function of the workgroup's wave count, when
// the iteration is otherwise (almost) empty?  The pair kernels' slope / intercept sweep left ~1.5 k cycles per 64-channel slice that no
// pipe accounts for (profiles/r04_pair_kernels.md); the per-slice barrier of a 12-16-wave workgroup is the suspect.
// Body per iteration: `work` dependent v_add per wave (skewed: wave w does work * (1 + w % 4) / 2), optionally 8 LDS writes, then s_barrier.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/barrier_waves.hip -o /tmp/barrier_waves && /tmp/barrier_waves
#include <hip/hip_runtime.h>
#include <stdio.h>

template <bool LDSW, bool BAR>
__global__ void k(long long *out, int iters, int work, int seed) {
    extern __shared__ int lds[];
    const int wave = threadIdx.x >> 6;
    int x = seed + threadIdx.x;
    const int mine = work * (1 + (wave & 3)) / 2;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < mine; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(seed));
        if (LDSW) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lds[(threadIdx.x * 4 + i * 4096 + it) & 16383] = x + i;
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (x == 0x7fffffff) out[1] = lds[threadIdx.x];
}

int main() {
    long long *d, h[1];
    (void)hipMalloc(&d, 4096);
    const int iters = 512;
    for (int grid = 1; grid <= 256; grid *= 256)
        for (int nt = 256; nt <= 1024; nt += 256)
            for (int work = 0; work <= 64; work += 64)
                for (int mode = 0; mode < 3; ++mode) {
                    for (int rep = 0; rep < 2; ++rep) {
                        if (mode == 0) hipLaunchKernelGGL((k<false, false>), dim3(grid), dim3(nt), 65536, 0, d, iters, work, rep);
                        if (mode == 1) hipLaunchKernelGGL((k<false, true>), dim3(grid), dim3(nt), 65536, 0, d, iters, work, rep);
                        if (mode == 2) hipLaunchKernelGGL((k<true, true>), dim3(grid), dim3(nt), 65536, 0, d, iters, work, rep);
                        (void)hipDeviceSynchronize();
                    }
                    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
                    printf("grid %3d  %2d waves  work %2d  %s : %.0f cycles per iteration (wave 0)\n", grid, nt / 64, work,
                           mode == 0 ? "no barrier          " : mode == 1 ? "s_barrier           " : "8 LDS writes+barrier", (double)h[0] / iters);
                }
    return 0;
}
