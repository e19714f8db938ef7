// Micro-benchmark: per-CU operand ingest rate (L2 -> LDS with global_load_lds, or L2 -> VGPR with plain loads) as a
// function of the SHAPE of one wave instruction's 1 KiB: SEG contiguous bytes per row (SEG/16 lanes per row), rows
// STRIDE bytes apart.  SEG = 64 / stride >= 128 is what the conv kernels' [row][64 B] operand tiles use today; SEG =
// 16 is the band kernel's plane fill (64 pixels x 16 B); SEG = 1024 is a fully contiguous piece (a pre-tiled weight
// image).  L2-resident source (3 MiB), one workgroup per CU, DEPTH instructions in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH, int LDS>  // LDS: 0 = loads into VGPRs, 1 = global_load_lds, 2 = buffer_load ... lds (SGPR resource + 32-bit lane offset)
__global__ void k(const char *src, size_t bytes, int iters, int seg, int stride, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
    const int lpr = seg >> 4;                       // lanes per row
    const int row = lane / lpr, col = lane % lpr;   // 64 / lpr rows per instruction
    const int rows = 64 / lpr;
    const size_t span = (size_t)rows * stride;      // bytes one instruction's rows cover
    size_t base = ((size_t)blockIdx.x * 7919 * 1024 + (size_t)wave * span * 5) % (bytes / 2);
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i acc = {0, 0, 0, 0};
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)bytes, 0x00020000);
#pragma unroll 8
    for (int it = 0; it < iters; ++it) {
        size_t off = (base + (size_t)row * stride + (size_t)col * 16 + (size_t)it * span) % (bytes - 1024);
        off &= ~size_t(15);
        if (LDS == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + ((it % DEPTH) * nw + wave) * 1024), 16,
                                                     (int)off, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        } else if (LDS == 1) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                             (__attribute__((address_space(3))) void *)(smem + ((it % DEPTH) * nw + wave) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        } else {
            v4i v = *reinterpret_cast<const v4i *>(src + off);
            acc += v;  // the compiler keeps DEPTH loads in flight only if unrolled: see the unroll pragma below
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (LDS == 0 && acc.x + acc.y + acc.z + acc.w == 12345) sink[1] = 1;
    if (t == 0 && smem[5] == 77) sink[0] = 1;
}
template <int DEPTH, int LDS>
void run(const char *src, size_t bytes, int *sink, int waves, int seg, int stride) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1024, grid = 256;
    const int lds = DEPTH * waves * 1024;
    (void)hipFuncSetAttribute((const void *)k<DEPTH, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<DEPTH, LDS>), dim3(grid), dim3(waves * 64), lds, 0, src, bytes, iters, seg, stride, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<DEPTH, LDS>), dim3(grid), dim3(waves * 64), lds, 0, src, bytes, iters, seg, stride, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)grid * iters * waves * 1024;
    printf("%s waves/CU %2d seg %4d B stride %4d B: %6.2f TB/s  %6.1f GB/s per CU  %5.1f GB/s per wave\n", LDS == 2 ? "buf-lds" : LDS == 1 ? "lds-dma" : "vgpr   ",
           waves, seg, stride, moved / ms / 1e9, moved / ms / 1e6 / 256, moved / ms / 1e6 / 256 / waves);
}
int main() {
    const size_t bytes = 3u << 20;
    char *src; int *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 8); hipMemset(src, 1, bytes);
    for (int waves : {4, 8, 16})
        for (int seg : {16, 64, 128, 256, 1024}) {
            for (int stride : {seg, 2304}) {
                if (seg == 1024 && stride != seg) continue;
                run<4, 1>(src, bytes, sink, waves, seg, stride);
            }
        }
    for (int waves : {4, 8})
        for (int seg : {16, 64, 128, 1024}) run<4, 0>(src, bytes, sink, waves, seg, seg == 1024 ? 1024 : 2304);
    for (int waves : {1, 2, 4, 8, 16})
        for (int seg : {64, 1024}) {
            run<8, 2>(src, bytes, sink, waves, seg, seg == 1024 ? 1024 : 2304);
            run<8, 1>(src, bytes, sink, waves, seg, seg == 1024 ? 1024 : 2304);
        }
    return 0;
}
