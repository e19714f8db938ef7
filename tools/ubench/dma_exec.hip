// Does an LDS-DMA (`buffer_load ... lds`) write LDS bytes for lanes that EXEC has switched off, and what does an out-of-range
// lane write?  (band_v2.hip keeps its 64 bytes of synchronisation words inside a band plane and masks the lanes that would hit them.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const char *src, int nbytes, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned smem[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    const unsigned voff = (lane & 1) ? 0x80000000u : (unsigned)lane * 16u;   // odd lanes out of range
    if (lane < 60) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = smem[i];
}
int main() {
    char *src; unsigned *out; unsigned h[256];
    (void)hipMalloc(&src, 4096); (void)hipMalloc(&out, 1024); (void)hipMemset(src, 0x11, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 4096, out);
    (void)hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 2, 3, 58, 59, 60, 61, 62, 63}) printf("lane %2d: %08x %08x %08x %08x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
