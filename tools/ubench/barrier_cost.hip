// Cost of one s_barrier per "step" of 48 independent MFMAs (8 accumulators) for 4 MFMA waves (one per SIMD),
// with and without 4 extra waves that only take part in the barrier.  cycles per step, s_memtime, one CU / all CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: no barrier, 1: s_barrier per step, 2: barrier + setprio
__global__ __launch_bounds__(512, 1) void k(long long *out, int steps, int seed, int nmfma_waves) {
    const int wave = threadIdx.x >> 6;
    v4i a[2] = {{seed, seed + 1, seed + 2, seed + 3}, {seed + 4, 5, 6, 7}}, b[4];
    for (int i = 0; i < 4; ++i) b[i] = v4i{seed * 3 + i, seed * 5, seed * 7, seed * 9};
    v16i acc[2][4];
    for (int c = 0; c < 2; ++c)
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 16; ++r) acc[c][q][r] = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (wave < nmfma_waves) {
        if (MODE == 2) __builtin_amdgcn_s_setprio(2);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int bt = 0; bt < 6; ++bt)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[c][q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[c], b[q], acc[c][q], 0, 0, 0);
            if (MODE) __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int s = 0; s < steps; ++s)
            if (MODE) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    int sum = 0;
    for (int c = 0; c < 2; ++c)
        for (int q = 0; q < 4; ++q) sum += acc[c][q][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0, out[1] = sum;
}

int main() {
    long long *d, h[2];
    (void)hipMalloc(&d, 4096);
    const int steps = 256;
    for (int grid = 1; grid <= 256; grid *= 256)
        for (int nt = 256; nt <= 512; nt *= 2)
            for (int mode = 0; mode < 3; ++mode) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(nt), 0, 0, d, steps, rep, 4);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(nt), 0, 0, d, steps, rep, 4);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(nt), 0, 0, d, steps, rep, 4);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("grid %3d  waves %d (4 MFMA)  %s : %.0f cycles per step of 48 MFMAs (1536 = pipe-bound)\n", grid, nt / 64,
                       mode == 0 ? "no barrier      " : mode == 1 ? "s_barrier/step  " : "barrier+setprio ", (double)h[0] / steps);
            }
    return 0;
}
