// Micro-benchmark: achievable HBM read / write / copy rates with 16 B/lane streaming kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k_read(const v4i *a, v4i *sink, size_t n) {
    v4i s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4i v = a[i];
        s.x ^= v.x, s.y ^= v.y, s.z ^= v.z, s.w ^= v.w;
    }
    if (s.x == 0x12345678) sink[0] = s;
}
__global__ void k_write(v4i *a, size_t n, int val) {
    v4i s = {val, val, val, val};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = s;
}
__global__ void k_copy(const v4i *a, v4i *b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 2 reads + 1 write per element-group, like a residual epilogue (read res, read small, write res + q)
__global__ void k_rw2(const v4i *a, v4i *b, v4i *c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4i v = a[i];
        b[i] = v;
        if ((i & 1) == 0) c[i >> 1] = v;
    }
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 16;
    v4i *a, *b, *c;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n);
                if (mode == 1) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, rep);
                if (mode == 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                if (mode == 3) hipLaunchKernelGGL(k_rw2, dim3(grid), dim3(256), 0, 0, a, b, c, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = mode == 2 ? 2.0 * bytes : (mode == 3 ? 2.5 * bytes : 1.0 * bytes);
            const char *nm[] = {"read ", "write", "copy ", "r+1.5w"};
            printf("grid %6d %s: %.3f ms  %.2f TB/s\n", grid, nm[mode], best, moved / best / 1e9);
        }
    }
    return 0;
}
