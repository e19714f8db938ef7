// MFMA issue rate on one CU: cycles per v_mfma (s_memtime) for the i8 shapes, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4o __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void k(long long *out, int iters, int seed) {
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 9};
    v16i acc[4];
    v4o acs[4];
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int r = 0; r < 4; ++r) acs[i][r] = 0;
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
            else acs[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acs[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acs[i][0];
    if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * 2] = t1 - t0, out[(threadIdx.x >> 6) * 2 + 1] = s;
}

int main() {
    long long *d, h[32];
    hipMalloc(&d, 4096);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int grid = 1; grid <= 256; grid *= 256)
    for (int shape = 0; shape < 2; ++shape)
        for (int nt = 256; nt <= 1024; nt *= 2) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(nt), 0, 0, d, iters, rep);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(nt), 0, 0, d, iters, rep);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("grid %3d  wall %.1f us (%.2f ns per MFMA per SIMD)  ", grid, ms * 1e3, ms * 1e6 / (iters * 4.0 * (nt / 256)));
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            long long lo = h[0], hi = h[0];
            for (int w = 0; w < nt / 64; ++w) lo = h[2 * w] < lo ? h[2 * w] : lo, hi = h[2 * w] > hi ? h[2 * w] : hi;
            printf("%s  waves/SIMD %d : fastest wave %.1f, slowest wave %.1f cycles per own MFMA -> %.1f cycles per MFMA per SIMD\n",
                   shape == 0 ? "i32_32x32x32_i8" : "i32_16x16x64_i8", nt / 256, lo / (iters * 4.0), hi / (iters * 4.0),
                   hi / (iters * 4.0) / (nt / 256));
        }
    return 0;
}
