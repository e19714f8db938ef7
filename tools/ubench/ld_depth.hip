// Micro-benchmark: plain global_load_dwordx4 (to VGPRs) throughput from an L2-resident buffer, one workgroup per CU,
// vs loads in flight per wave and waves per CU - the alternative to LDS-DMA for operand staging.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ void k(const char *src, size_t bytes, int iters, int *sink) {
    const int t = threadIdx.x, wave = t >> 6;
    const int lrow = (t & 63) >> 2, lslot = t & 3;
    size_t base = ((size_t)blockIdx.x * 7919 * 64) % (bytes / 2);
    v4i acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += DEPTH) {
        v4i v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const size_t off = (base + (size_t)(wave * 16 + lrow) * 512 + (size_t)(it + d) * 64 + lslot * 16) % (bytes - 64);
            v[d] = *reinterpret_cast<const v4i *>(src + (off & ~size_t(15)));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if (acc.x == 0x1234567 && acc.y == 77) sink[0] = acc.z;
}
template <int DEPTH>
void run(const char *src, size_t bytes, int *sink, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, grid = 256;
    hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(waves * 64), 0, 0, src, bytes, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(waves * 64), 0, 0, src, bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)grid * iters * waves * 1024;
    printf("plain loads: waves/CU %2d depth %2d: %.3f ms %6.2f TB/s  %5.1f GB/s per CU  %5.1f GB/s per wave\n", waves, DEPTH, ms,
           moved / ms / 1e9, moved / ms / 1e6 / 256, moved / ms / 1e6 / 256 / waves);
}
int main() {
    const size_t bytes = 3u << 20;
    char *src; int *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4); hipMemset(src, 1, bytes);
    for (int waves : {4, 8, 16}) { run<1>(src, bytes, sink, waves); run<2>(src, bytes, sink, waves); run<4>(src, bytes, sink, waves); run<8>(src, bytes, sink, waves); run<16>(src, bytes, sink, waves); }
    return 0;
}
